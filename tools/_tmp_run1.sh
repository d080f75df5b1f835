L=tfp-causalimpact_amd/lib/libcausalimpact_amd.so
for v in lib_head lib_new lib_head lib_new; do
  cp tools/build/$v.so $L; echo $v; python tools/exp_cfg4_routes.py 2>/dev/null | head -3
done
cp tools/build/lib_new.so $L
python -m pytest tests/test_gpu_gibbs.py -m gpu -x -q -k "cfg4 or clusters" 2>&1 | tail -2
