#!/bin/bash
# Copies the summaries the judge reads from gpurun_out/<tag>/ (tools/collect_profiles.sh) into
# profiles/ under the round's names:   tools/publish_profiles.sh r02_g
set -eu
TAG=${1:-r02_g}
SRC=gpurun_out/$TAG
P=profiles
rows() { head -1 "$1"; grep "$2" "$1" || true; }      # header + the kernel's rows
grep '^{"metric"' $SRC/bench.json | tail -1 > $P/r02_f_bench.json
grep '^{"metric"' $SRC/bench_hmc.json | tail -1 > $P/r02_f_bench_cfg3_hmc.json
cp $SRC/trace/bench_kernel_stats.csv $P/r02_f_kernel_stats.csv
rows $SRC/pmc_fetch/bench_counter_collection.csv gibbs_kernel > $P/r02_f_pmc_fetch_size.csv
rows $SRC/pmc_write/bench_counter_collection.csv gibbs_kernel > $P/r02_f_pmc_write_size.csv
python tools/pmc_summary.py hbm $SRC/pmc_fetch/bench_counter_collection.csv $SRC/pmc_write/bench_counter_collection.csv \
  --kernel gibbs_kernel --algorithmic-bytes 96840000 --out $P/r02_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/pmc_sq/bench_counter_collection.csv --kernel gibbs_kernel --out $P/r02_cfg2_sq_counters.json > /dev/null
cp $SRC/phase_cycles.txt $P/r02_f_phase_cycles.txt
# cfg3
cp $SRC/hmc_trace/hmc_kernel_stats.csv $P/r02_cfg3_kernel_stats.csv
rows $SRC/hmc_pmc_fetch/hmc_counter_collection.csv hmc_ > $P/r02_cfg3_pmc_fetch.csv
rows $SRC/hmc_pmc_write/hmc_counter_collection.csv hmc_ > $P/r02_cfg3_pmc_write.csv
python tools/pmc_summary.py hbm $SRC/hmc_pmc_fetch/hmc_counter_collection.csv $SRC/hmc_pmc_write/hmc_counter_collection.csv \
  --kernel hmc_kernel --also latents_kernel hmc_mean_kernel hmc_unpack_kernel --algorithmic-bytes 96840000 --out $P/r02_cfg3_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/hmc_pmc_sq/hmc_counter_collection.csv --kernel hmc_kernel --out $P/r02_cfg3_sq_counters.json > /dev/null
# cfg4 / cfg5
cp $SRC/configs.jsonl $P/r02_configs.jsonl
cp $SRC/cfg_trace/cfg_kernel_stats.csv $P/r02_cfg4_cfg5_kernel_stats.csv
rows $SRC/cfg_pmc_fetch/cfg_counter_collection.csv gibbs_wide > $P/r02_cfg4_pmc_fetch.csv
rows $SRC/cfg_pmc_write/cfg_counter_collection.csv gibbs_wide > $P/r02_cfg4_pmc_write.csv
python tools/pmc_summary.py hbm $SRC/cfg_pmc_fetch/cfg_counter_collection.csv $SRC/cfg_pmc_write/cfg_counter_collection.csv \
  --kernel gibbs_wide --last 3 --algorithmic-bytes 960000000 --out $P/r02_cfg4_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/cfg_pmc_sq/cfg_counter_collection.csv --kernel gibbs_wide --last 3 --out $P/r02_cfg4_sq_counters.json > /dev/null
cp $SRC/cfg4_phase_cycles.txt $P/r02_cfg4_phase_cycles.txt
# end to end, RCCL single rank
cp $SRC/trace_e2e/e2e_kernel_stats.csv $P/r02_f_fit_causalimpact_kernel_stats.csv
grep '^{"metric"' $SRC/bench_force_dist.json | tail -1 > $P/r02_bench_force_dist_rccl_1rank.json
grep -v '^{"metric"' $SRC/bench_force_dist.json > $P/r02_bench_force_dist_rccl_1rank.log || true
ls -la $P | grep r02 | wc -l
