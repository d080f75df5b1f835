#!/bin/bash
set -u
OUT=gpurun_out/r02g; mkdir -p $OUT
timeout 1500 python -m pytest tests --maxfail=8 -q -m gpu > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
