#!/bin/bash
# cfg4 / cfg5 evidence: kernel stats + HBM traffic counters (separate passes)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_trace -o cfg -- python tools/run_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
tail -3 $OUT/configs.err; cat $OUT/configs.jsonl
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cfg_pmc_fetch -o cfg -- python tools/run_configs.py cfg4 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cfg_pmc_write -o cfg -- python tools/run_configs.py cfg4 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/cfg_pmc_sq -o cfg -- python tools/run_configs.py cfg4 > $OUT/pmc_sq.log 2>&1
cat $OUT/cfg_trace/cfg_kernel_stats.csv
