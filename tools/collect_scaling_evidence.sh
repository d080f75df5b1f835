#!/bin/bash
# rocprofv3 kernel stats and SQ counters of the covariate scan (tools/exp_p_scale.py) and kernel
# stats of the capability routes (float64 kernel): run on the GPU box from the repo root.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pscale_trace -o pscale -- python $ROOT/tools/exp_p_scale.py > $OUT/pscale_trace.log 2>&1
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pscale_sq -o pscale -- python $ROOT/tools/exp_p_scale.py > $OUT/pscale_sq.log 2>&1
cd $ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/extras_trace -o extras -- python tools/run_configs.py extras > $OUT/extras_trace.log 2>&1
head -6 $OUT/pscale_trace/pscale_kernel_stats.csv | cut -c1-160
grep -h "gibbs64" $OUT/extras_trace/extras_kernel_stats.csv | cut -c1-200
