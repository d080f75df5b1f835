#!/bin/bash
set -u
OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gibbs.py tests/test_gpu_hmc.py tests/test_distributed_gloo.py tests/test_batch.py --maxfail=8 -q -m gpu -k "streamed or hmc or two_ranks or batch" > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json
timeout 300 python bench.py --no-cpu-baseline --sampler hmc --steps 2 > $OUT/bench_hmc.json 2> $OUT/bench_hmc.err; tail -2 $OUT/bench_hmc.err; cat $OUT/bench_hmc.json
