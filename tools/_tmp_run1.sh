python -m pytest tests/test_gpu_gibbs.py -m gpu -x -q -k "cfg4 or clusters or concurrent_cluster or contention or more_than_52 or general_seasonal or shortest" 2>&1 | tail -5
python tools/exp_cfg4_routes.py 2>&1 | head -5
python tools/debug_tp.py 2>&1 | grep -E "P=101|P= 61|P=61" -A1 | cut -c1-330
for shape in cfg4_c8_s1000 cfg4_c32_s1000; do
  rm -rf /tmp/pf /tmp/pw
  TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o cfg -- python tools/run_configs.py $shape > /dev/null 2>&1
  TMPDIR=/tmp rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o cfg -- python tools/run_configs.py $shape > /dev/null 2>&1
  python tools/pmc_summary.py hbm $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) --kernel gibbs_wide --out /tmp/x.json | grep -E "fetch_bytes|write_bytes"
done
