#!/bin/bash
# Copies the summaries the judge reads from gpurun_out/<tag>/ (tools/collect_profiles.sh) into
# profiles/ under the round's names:   tools/publish_profiles.sh r03
set -eu
TAG=${1:-r06}
SRC=gpurun_out/$TAG
P=profiles
R=$TAG
rows() { head -1 "$1"; grep "$2" "$1" || true; }      # header + the kernel's rows
grep '^{"metric"' $SRC/bench.json | tail -1 > $P/${R}_bench.json
grep '^{"metric"' $SRC/bench_hmc.json | tail -1 > $P/${R}_bench_cfg3_hmc.json
cp $SRC/trace/bench_kernel_stats.csv $P/${R}_kernel_stats.csv
rows $SRC/pmc_fetch/bench_counter_collection.csv gibbs_kernel > $P/${R}_pmc_fetch_size.csv
rows $SRC/pmc_write/bench_counter_collection.csv gibbs_kernel > $P/${R}_pmc_write_size.csv
python tools/pmc_summary.py hbm $SRC/pmc_fetch/bench_counter_collection.csv $SRC/pmc_write/bench_counter_collection.csv \
  --kernel gibbs_kernel --algorithmic-bytes 96840000 --out $P/${R}_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/pmc_sq/bench_counter_collection.csv --kernel gibbs_kernel --out $P/${R}_cfg2_sq_counters.json > /dev/null
cp $SRC/phase_cycles.txt $P/${R}_phase_cycles.txt
# cfg3
cp $SRC/hmc_trace/hmc_kernel_stats.csv $P/${R}_cfg3_kernel_stats.csv
rows $SRC/hmc_pmc_fetch/hmc_counter_collection.csv hmc_ > $P/${R}_cfg3_pmc_fetch.csv
rows $SRC/hmc_pmc_write/hmc_counter_collection.csv hmc_ > $P/${R}_cfg3_pmc_write.csv
python tools/pmc_summary.py hbm $SRC/hmc_pmc_fetch/hmc_counter_collection.csv $SRC/hmc_pmc_write/hmc_counter_collection.csv \
  --kernel hmc_kernel --also latents_kernel hmc_mean_kernel hmc_unpack_kernel --algorithmic-bytes 96840000 --out $P/${R}_cfg3_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/hmc_pmc_sq/hmc_counter_collection.csv --kernel hmc_kernel --out $P/${R}_cfg3_sq_counters.json > /dev/null
# cfg4 / cfg5  (the counter passes run cfg4 with S = 200: 8 chains x (200 x 120,216 + 2,090,000) bytes)
cp $SRC/configs.jsonl $P/${R}_configs.jsonl
cp $SRC/extras.jsonl $P/${R}_extras.jsonl
cp $SRC/cfg_trace/cfg_kernel_stats.csv $P/${R}_cfg4_cfg5_kernel_stats.csv
rows $SRC/cfg_pmc_fetch/cfg_counter_collection.csv gibbs_wide > $P/${R}_cfg4_pmc_fetch.csv
rows $SRC/cfg_pmc_write/cfg_counter_collection.csv gibbs_wide > $P/${R}_cfg4_pmc_write.csv
python tools/pmc_summary.py hbm $SRC/cfg_pmc_fetch/cfg_counter_collection.csv $SRC/cfg_pmc_write/cfg_counter_collection.csv \
  --kernel gibbs_wide --last 3 --algorithmic-bytes 209065600 --out $P/${R}_cfg4_pmc.json > /dev/null
python tools/pmc_summary.py sq $SRC/cfg_pmc_sq/cfg_counter_collection.csv --kernel gibbs_wide --last 3 --out $P/${R}_cfg4_sq_counters.json > /dev/null
if [ -d $SRC/cfg4full_pmc_fetch ]; then
  python tools/pmc_summary.py hbm $SRC/cfg4full_pmc_fetch/cfg_counter_collection.csv $SRC/cfg4full_pmc_write/cfg_counter_collection.csv \
    --kernel gibbs_wide --algorithmic-bytes 978448000 --out $P/${R}_cfg4_pmc_s1000.json > /dev/null
fi
cp $SRC/cfg4_phase_cycles.txt $P/${R}_cfg4_phase_cycles.txt
# end to end, the multi-rank code path
cp $SRC/trace_e2e/e2e_kernel_stats.csv $P/${R}_fit_causalimpact_kernel_stats.csv
grep '^{"metric"' $SRC/bench_force_dist.json | tail -1 > $P/${R}_bench_force_dist_rccl_1rank.json
(grep -v '^{"metric"' $SRC/bench_force_dist.json; cat $SRC/bench_force_dist.err) > $P/${R}_bench_force_dist_rccl_1rank.log || true
cp $SRC/comm_tests.txt $P/${R}_two_ranks_on_gpu0_host_transport.txt
for f in p_scale t_scale phase_cycles_p52 f64_phase_cycles hmc_phase_cycles e2e_host_profile cfg4_routes cfg4_routes_dk_rows_in_l2 seasonal_batch_routes general_seasonal_times kernel_resources bigp_phase_cycles; do
  if [ -f $SRC/$f.txt ]; then cp $SRC/$f.txt $P/${R}_$f.txt; fi
done
# round 5: the time-parallel general seasonal kernel
if [ -f $SRC/tp_debug.txt ]; then
  cp $SRC/tp_debug.txt $P/${R}_tp_parity_and_phase_cycles.txt
  cp $SRC/tp_trace/tp_kernel_stats.csv $P/${R}_general_seasonal_kernel_stats.csv
  python tools/pmc_summary.py hbm $SRC/tp_pmc_fetch/tp_counter_collection.csv $SRC/tp_pmc_write/tp_counter_collection.csv \
    --kernel gibbs_seasonal_tp --out $P/${R}_general_seasonal_pmc.json > /dev/null
  python tools/pmc_summary.py sq $SRC/tp_pmc_sq/tp_counter_collection.csv --kernel gibbs_seasonal_tp --out $P/${R}_general_seasonal_sq_counters.json > /dev/null
  if [ -f $SRC/tp_combine_bench.txt ]; then cp $SRC/tp_combine_bench.txt $P/${R}_tp_combine_bench.txt; fi
fi
ls -la $P | grep ${R}_ | wc -l
