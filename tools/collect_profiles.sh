#!/bin/bash
# copy the judged artefacts of a measurement campaign (tools/r05_final.sh -> gpurun_out/) into profiles/ (tracked)
#   tools/collect_profiles.sh [r05]
R=${1:-r05}
cd $(dirname $0)/..
O=gpurun_out; P=profiles
for f in kernel_stats_c2 kernel_stats_c2_f32 kernel_stats_c5 kernel_stats_ens8 kernel_stats_train markers markers_train \
         pmc_traffic pmc_traffic_c5 pmc_traffic_train sq_counters sq_counters_train step_timeline step_timeline_ens8 \
         train_timeline c5_power_clock energy_probe energy_probe_mfma2 energy_probe_wfake parity_report; do
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
wrap ${R}_filter_probe.log "filter-only launches (tools/filter_probe.py; default build vs 32-row filter tiles only)" ${R}_filter_probe.md
wrap ${R}_trace_unit.log "in-kernel phase trace of the fused per-unit encoder (tools/trace_unit.py, variant build -DTSD_UNIT_TRACE)" ${R}_trace_unit.md
wrap ${R}_trace_combo_g800.log "in-kernel trace of a block launch at 800 graphs (tools/trace_combo.py, variant build -DTSD_TRACE)" ${R}_trace_combo_g800.md
wrap ${R}_ab_vs_r04.log "same-box A/B: this tree's library against the round-4 library (tools/ab_step.py; tools/bin/lib_r04.so built from commit 46a9921)" ${R}_ab_vs_r04.md
wrap ${R}_ab_fused_ens8.log "8 checkpoints at batch 100: materialising forms against the fused per-unit encoder (tools/ab_step.py)" ${R}_ab_fused_ens8.md
[ -s $O/${R}_gpu_tests.log ] && { echo "# python -m pytest tests -q -m gpu on the campaign box"; echo; echo '```'; tail -25 $O/${R}_gpu_tests.log; echo '```'; } > $P/${R}_gpu_tests.md
ls $P | grep -c "^${R}_"
