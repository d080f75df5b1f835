#!/bin/bash
set -u
OUT=gpurun_out/r02e; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_gibbs.py tests/test_gpu_parity_fullsize.py tests/test_batch.py tests/test_reference_stat_pins.py --maxfail=8 -q -m gpu -k "season or cfg4 or 5000 or 9000 or 40000 or weekly or streamed" > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
timeout 600 python tools/run_configs.py cfg4 > $OUT/cfg4.jsonl 2> $OUT/cfg4.err; tail -2 $OUT/cfg4.err; cat $OUT/cfg4.jsonl
