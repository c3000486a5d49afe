#!/bin/bash
# copy the judged artefacts of a measurement campaign (tools/r06_final.sh -> gpurun_out/) into profiles/ (tracked)
#   tools/collect_profiles.sh [r06]
R=${1:-r06}
cd $(dirname $0)/..
O=gpurun_out; P=profiles
for f in kernel_stats_c2 kernel_stats_c2_f32 kernel_stats_c5 kernel_stats_ens8 kernel_stats_train markers markers_train \
         pmc_traffic pmc_traffic_c5 pmc_traffic_train sq_counters sq_counters_train step_timeline step_timeline_ens8 \
         train_timeline c5_power_clock parity_report split_f16_sweep; do
  [ -s $O/${R}_$f.md ] && cp $O/${R}_$f.md $P/
done
for f in mfma_busy pmc_traffic pmc_traffic_c5 pmc_traffic_train bench_c2_1000 bench_c2_5000 bench_c3_unit bench_c5 \
         bench_driver_flags bench_ensemble8 bench_train bench_train_no_prefetch bench_train_torchrun_1range \
         bench_train_torchrun_3range; do
  [ -s $O/${R}_$f.json ] && cp $O/${R}_$f.json $P/
done
wrap() {  # log -> fenced markdown with a title
  [ -s $O/$1 ] || return
  { echo "# $2"; echo; echo '```'; cat $O/$1; echo '```'; } > $P/$3
}
wrap ${R}_ab_agg.log "stand-alone aggregation at configs[4] size: the windowed form against one wave per row (tools/ab_agg.py; tools/bin/lib_agg_old.so = -DTSD_AGW_MIN_ROWS=0)" ${R}_ab_agg.md
wrap ${R}_ab_train_prefetch.log "training loop at batch 200, prefetch forms in one process (tools/ab_train_prefetch.py)" ${R}_ab_train_prefetch.md
wrap ${R}_train_host_phases.log "host time of the training loop's phases (tools/train_host_phases.py; waits included)" ${R}_train_host_phases.md
wrap ${R}_ab_vs_r05.log "same-box A/B: this tree's library against the round-5 library (tools/ab_step.py; tools/bin/lib_r05.so built from commit 1c4256b)" ${R}_ab_vs_r05.md
[ -s $O/${R}_gpu_tests.log ] && { echo "# python -m pytest tests -q -m gpu on the campaign box"; echo; echo '```'; tail -25 $O/${R}_gpu_tests.log; echo '```'; } > $P/${R}_gpu_tests.md
ls $P | grep -c "^${R}_"
