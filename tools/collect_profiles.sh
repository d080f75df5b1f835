#!/bin/bash
# Collects the evidence committed under profiles/ for one round (run on the GPU box through
# gpurun from the repo root):   tools/collect_profiles.sh <tag>      e.g. r02_f
# rocprofv3 runs from /tmp with TMPDIR=/tmp; PMC counters are collected in their own passes
# (never together with trace domains).
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# ---- cfg2 (bench line)
timeout 600 $B > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $B --no-cpu-baseline --no-pmc --steps 5 --warmup 1 > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $B --no-cpu-baseline --no-pmc --steps 3 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $B --no-cpu-baseline --no-pmc --steps 3 --warmup 1 > "$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/pmc_sq" -o bench -- $B --no-cpu-baseline --no-pmc --steps 3 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
# ---- cfg3 (HMC)
timeout 600 $B --sampler hmc > "$OUT/bench_hmc.json" 2> "$OUT/bench_hmc.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hmc_trace" -o hmc -- $B --no-cpu-baseline --no-pmc --sampler hmc --steps 2 --warmup 1 > "$OUT/hmc_trace.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/hmc_pmc_fetch" -o hmc -- $B --no-cpu-baseline --no-pmc --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/hmc_pmc_write" -o hmc -- $B --no-cpu-baseline --no-pmc --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/hmc_pmc_sq" -o hmc -- $B --no-cpu-baseline --no-pmc --sampler hmc --steps 1 --warmup 1 > "$OUT/hmc_pmc_sq.log" 2>&1
cd "$ROOT"
# ---- cfg4 / cfg5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg_trace" -o cfg -- python tools/run_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cfg_pmc_fetch" -o cfg -- python tools/run_configs.py cfg4 > "$OUT/cfg_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cfg_pmc_write" -o cfg -- python tools/run_configs.py cfg4 > "$OUT/cfg_pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/cfg_pmc_sq" -o cfg -- python tools/run_configs.py cfg4 > "$OUT/cfg_pmc_sq.log" 2>&1
# cfg4 at BASELINE's own size (S = 1000) under the counters: traffic against the 978 MB of algorithmic bytes
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cfg4full_pmc_fetch" -o cfg -- python tools/run_configs.py cfg4_full > "$OUT/cfg4full_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cfg4full_pmc_write" -o cfg -- python tools/run_configs.py cfg4_full > "$OUT/cfg4full_pmc_write.log" 2>&1
# ---- end-to-end fit_causalimpact, phase budgets, RCCL single-rank run
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_e2e" -o e2e -- python tools/time_end_to_end.py > "$OUT/e2e.log" 2>&1
timeout 200 python tools/profile_phases.py > "$OUT/phase_cycles.txt" 2>&1
timeout 200 python tools/profile_wide.py > "$OUT/cfg4_phase_cycles.txt" 2>&1
# the N > 1 code path with one rank: C-ABI communicator (librccl), barrier, max-reduce, gather from HBM
CI_BENCH_FORCE_DIST=1 NCCL_DEBUG=VERSION timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 5 > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.err"
# two ranks sharing GPU 0 through the self-launcher (host transport: RCCL refuses two ranks on one device)
timeout 300 python -m pytest tests/test_gpu_comm.py -q > "$OUT/comm_tests.txt" 2>&1
timeout 600 python tools/run_configs.py extras > "$OUT/extras.jsonl" 2> "$OUT/extras.err"
# scaling scans (covariates, series length) and the phase budget of the 17-52 column route
timeout 300 python tools/exp_p_scale.py > "$OUT/p_scale.txt" 2>&1
timeout 300 python tools/exp_t_scale.py > "$OUT/t_scale.txt" 2>&1
timeout 200 python tools/profile_phases.py 1000 51 1 8 > "$OUT/phase_cycles_p52.txt" 2>&1
CI_F64_PROF=1 timeout 300 python tools/run_configs.py extras 2>&1 | grep "phases" | tail -1 > "$OUT/f64_phase_cycles.txt"
# ---- round 5: the time-parallel general seasonal kernel (csrc/ci_seasonal_tp.h)
timeout 600 python tools/debug_tp.py > "$OUT/tp_debug.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tp_trace" -o tp -- python tools/run_configs.py general_gpu > "$OUT/general_gpu.jsonl" 2> "$OUT/general.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tp_pmc_fetch" -o tp -- python tools/run_configs.py general_gpu > "$OUT/tp_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tp_pmc_write" -o tp -- python tools/run_configs.py general_gpu > "$OUT/tp_pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/tp_pmc_sq" -o tp -- python tools/run_configs.py general_gpu > "$OUT/tp_pmc_sq.log" 2>&1
if [ -x tools/build/bench_tp_combine ]; then timeout 120 tools/build/bench_tp_combine > "$OUT/tp_combine_bench.txt" 2>&1; fi
# ---- round 5: phase cycles of an HMC leapfrog step (cfg3), host profile of fit_causalimpact
timeout 200 python tools/exp_hmc_phases.py > "$OUT/hmc_phase_cycles.txt" 2>&1
timeout 200 python tools/profile_e2e.py 10 2>&1 | head -70 | cut -c1-160 > "$OUT/e2e_host_profile.txt"
# ---- round 6: cfg4 routes (1 / 8 / 32 / 64 chains, LDS vs L2 workspace of the draw), large seasonal batches on both routes
timeout 300 python tools/exp_cfg4_routes.py > "$OUT/cfg4_routes.txt" 2>&1
CI_WIDE_DK_GLOBAL=1 timeout 300 python tools/exp_cfg4_routes.py 2>&1 | head -3 > "$OUT/cfg4_routes_dk_rows_in_l2.txt"
timeout 300 python tools/exp_seasonal_batch_routes.py > "$OUT/seasonal_batch_routes.txt" 2>&1
timeout 300 python tools/time_general_seasonal.py > "$OUT/general_seasonal_times.txt" 2>&1
timeout 300 python tools/time_bigp.py > "$OUT/bigp_phase_cycles.txt" 2>&1
python tools/kernel_resources.py > "$OUT/kernel_resources.txt" 2>&1
find "$OUT" -name "*.csv" | wc -l
tail -1 "$OUT/bench.json" | cut -c1-400
