#!/bin/bash
# round-2 GPU call A: HMC parity tests + first cfg3 bench line
set -u
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests/test_gpu_hmc.py -x -q -m gpu > gpurun_out/r02a/pytest_hmc.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest_hmc.log
tail -30 gpurun_out/r02a/pytest_hmc.log
timeout 600 python bench.py --sampler hmc --steps 3 --warmup 1 > gpurun_out/r02a/bench_hmc.json 2> gpurun_out/r02a/bench_hmc.err
tail -3 gpurun_out/r02a/bench_hmc.err; cat gpurun_out/r02a/bench_hmc.json
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -3 gpurun_out/r02a/bench.err; cat gpurun_out/r02a/bench.json
