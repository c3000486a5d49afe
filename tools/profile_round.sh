#!/bin/bash
# All profile artefacts of a round in one gpurun call:   gpurun -- 'bash tools/profile_round.sh r02'
# Writes gpurun_out/<tag>_*: kernel-trace stats (c2, c5, train), HBM traffic PMC passes (c2, c5: FETCH_SIZE and
# WRITE_SIZE in separate passes), SQ issue/stall/MFMA-busy counters (c2, c5; MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch time x clock)).  Copy what is to be judged to profiles/.
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
db() { ls $1/*/*results.db $1/*results.db 2>/dev/null | head -1; }
B="--no-cpu-baseline --no-extras --no-f32"
rm -rf /tmp/p_*
rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o c2 -- python3 bench.py --steps 200 --warmup 20 $B > gpurun_out/${TAG}_prof_c2.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_c2) > gpurun_out/${TAG}_kernel_stats_c2.md
TSDIFF_GEMM=f32 rocprofv3 --kernel-trace --stats -d /tmp/p_c2f -o c2f -- python3 bench.py --steps 200 --warmup 20 $B > gpurun_out/${TAG}_prof_c2_f32.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_c2f) > gpurun_out/${TAG}_kernel_stats_c2_f32.md
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python3 bench.py --workload c5 --steps 4 --warmup 1 $B > gpurun_out/${TAG}_prof_c5.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_c5) > gpurun_out/${TAG}_kernel_stats_c5.md
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python3 bench.py --workload train --steps 20 --warmup 5 > gpurun_out/${TAG}_prof_train.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_tr) > gpurun_out/${TAG}_kernel_stats_train.md
python3 tools/step_timeline.py $(db /tmp/p_tr) > gpurun_out/${TAG}_train_timeline.md
# the 8-checkpoint ensemble at batch 100 (configs[2]'s per-GPU unit): kernel stats and one step's timeline
rocprofv3 --kernel-trace --stats -d /tmp/p_e8 -o e8 -- python3 bench.py --models 8 --steps 50 --warmup 5 $B > gpurun_out/${TAG}_prof_ens8.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_e8) > gpurun_out/${TAG}_kernel_stats_ens8.md
python3 tools/step_timeline.py $(db /tmp/p_e8) step_tail > gpurun_out/${TAG}_step_timeline_ens8.md
python3 tools/step_timeline.py $(db /tmp/p_c2) step_tail > gpurun_out/${TAG}_step_timeline.md
# roctx ranges of the library's entry points (no counters in this pass)
rocprofv3 --kernel-trace --marker-trace -d /tmp/p_mk -o mk -- python3 bench.py --steps 20 --warmup 5 --no-graph $B > gpurun_out/${TAG}_prof_markers.log 2>&1
python3 tools/marker_summary.py $(db /tmp/p_mk) > gpurun_out/${TAG}_markers.md
rocprofv3 --kernel-trace --marker-trace -d /tmp/p_mt -o mt -- python3 bench.py --workload train --steps 10 --warmup 3 > gpurun_out/${TAG}_prof_markers_train.log 2>&1
python3 tools/marker_summary.py $(db /tmp/p_mt) > gpurun_out/${TAG}_markers_train.md
# HBM traffic: separate passes per counter (MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel trace only
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python3 bench.py --steps 30 --warmup 5 $B > gpurun_out/${TAG}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python3 bench.py --steps 30 --warmup 5 $B > gpurun_out/${TAG}_pmc_w.log 2>&1
python3 tools/pmc_summary.py $(db /tmp/p_f) $(db /tmp/p_w) gpurun_out/${TAG}_pmc_traffic > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f5 -o f -- python3 bench.py --workload c5 --steps 3 --warmup 1 $B > gpurun_out/${TAG}_pmc_f5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w5 -o w -- python3 bench.py --workload c5 --steps 3 --warmup 1 $B > gpurun_out/${TAG}_pmc_w5.log 2>&1
python3 tools/pmc_summary.py $(db /tmp/p_f5) $(db /tmp/p_w5) gpurun_out/${TAG}_pmc_traffic_c5 > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_ft -o f -- python3 bench.py --workload train --steps 8 --warmup 3 > gpurun_out/${TAG}_pmc_ft.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_wt -o w -- python3 bench.py --workload train --steps 8 --warmup 3 > gpurun_out/${TAG}_pmc_wt.log 2>&1
python3 tools/pmc_summary.py $(db /tmp/p_ft) $(db /tmp/p_wt) gpurun_out/${TAG}_pmc_traffic_train > /dev/null
# SQ counters of the hot kernels (one pass), plus the effective clock (GRBM)
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
rocprofv3 --kernel-trace --pmc $SQ -d /tmp/p_sq -o sq -- python3 bench.py --steps 30 --warmup 5 $B > gpurun_out/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d /tmp/p_sq5 -o sq -- python3 bench.py --workload c5 --steps 3 --warmup 1 $B > gpurun_out/${TAG}_pmc_sq5.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d /tmp/p_sqt -o sq -- python3 bench.py --workload train --steps 8 --warmup 2 > gpurun_out/${TAG}_pmc_sqt.log 2>&1
python3 tools/mfma_busy.py c2=$(db /tmp/p_sq) c5=$(db /tmp/p_sq5) train=$(db /tmp/p_sqt) > gpurun_out/${TAG}_mfma_busy.json
{
  echo "# SQ counters per launch of the training step (rocprofv3 --kernel-trace --pmc, one pass; MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch time x 2.4 GHz)), $TAG"
  echo; echo '```'
  for k in block_bwd layer_combo wgrad_h2 wgrad_batch wgrad_kernel embed_bwd edge_embed pair_output pair_bwd; do python3 tools/pmc_kernel_table.py $(db /tmp/p_sqt) $k; done
  echo '```'
} > gpurun_out/${TAG}_sq_counters_train.md
{
  echo "# SQ counters per launch (rocprofv3 --kernel-trace --pmc, one pass), $TAG"
  echo; echo "## configs[1] (batch 100)"; echo '```'
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq) forward_mega
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq) layer_combo
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq) typed_embed
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq) step_tail
  echo '```'; echo; echo "## configs[4] (1024 x 64 atoms)"; echo '```'
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq5) unit_encoder
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq5) layer_combo
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq5) pair_output
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq5) typed_embed
  python3 tools/pmc_kernel_table.py $(db /tmp/p_sq5) cfconv_aggregate
  echo '```'
} > gpurun_out/${TAG}_sq_counters.md
cat gpurun_out/${TAG}_markers.md gpurun_out/${TAG}_markers_train.md; head -14 gpurun_out/${TAG}_kernel_stats_c2.md; head -8 gpurun_out/${TAG}_kernel_stats_c5.md; cat gpurun_out/${TAG}_pmc_traffic.md | head -8; head -30 gpurun_out/${TAG}_sq_counters.md
