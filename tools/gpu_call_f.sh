#!/bin/bash
set -u
OUT=gpurun_out/r02f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gibbs.py --maxfail=5 -q -m gpu -k "five_wave or first_iterations or streamed or chain_ids" > $OUT/pytest.log 2>&1
tail -30 $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json
