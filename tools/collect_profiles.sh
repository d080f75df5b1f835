#!/bin/bash
# Collects the evidence committed under profiles/ for one round (run on the GPU box through
# gpurun from the repo root):   tools/collect_profiles.sh <tag>      e.g. r01_c
# rocprofv3 runs from /tmp with TMPDIR=/tmp; PMC counters are collected in their own passes.
set -u
TAG=${1:-r01_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --steps 5 --warmup 1 > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_e2e" -o e2e -- python tools/time_end_to_end.py > "$OUT/e2e.log" 2>&1
python tools/profile_phases.py > "$OUT/phase_cycles.txt" 2>&1
python tools/run_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
find "$OUT" -name "*.csv" | head -40
