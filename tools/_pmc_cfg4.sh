#!/bin/bash
# quick HBM-traffic check of cfg4 (8 chains, S = 200): separate FETCH_SIZE / WRITE_SIZE passes
set -u
ROOT=$(pwd); TAG=${1:-r06a}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cfg_pmc_fetch" -o cfg -- python tools/run_configs.py cfg4 > "$OUT/cfg_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cfg_pmc_write" -o cfg -- python tools/run_configs.py cfg4 > "$OUT/cfg_pmc_write.log" 2>&1
F=$(find $OUT/cfg_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/cfg_pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py hbm "$F" "$W" --kernel gibbs_wide --out $OUT/cfg4_pmc.json
cat $OUT/cfg4_pmc.json; tail -3 $OUT/cfg_pmc_fetch.log
