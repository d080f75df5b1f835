#!/bin/bash
# round-2 GPU call B: full GPU test suite, cfg3 (HMC) rocprof evidence, SQ counters for the
# Gibbs kernel, RCCL single-rank smoke.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02b
mkdir -p "$OUT"
timeout 1500 python -m pytest tests --maxfail=6 -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -15 "$OUT/pytest_gpu.log"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hmc_trace" -o hmc -- $B --sampler hmc --steps 2 --warmup 1 > "$OUT/hmc_trace.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/hmc_pmc_fetch" -o hmc -- $B --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/hmc_pmc_write" -o hmc -- $B --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/gibbs_pmc_sq" -o gibbs -- $B --steps 3 --warmup 1 > "$OUT/gibbs_pmc_sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/hmc_pmc_sq" -o hmc -- $B --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_sq.log" 2>&1
cd "$ROOT"
CI_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --no-cpu-baseline --steps 5 > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.err"
tail -2 "$OUT/bench_force_dist.err"; cat "$OUT/bench_force_dist.json"
timeout 200 python tools/profile_phases.py > "$OUT/phase_cycles.txt" 2>&1
find "$OUT" -name "*.csv" | head -40
