#!/bin/bash
# round 6: every measurement of the round at one HEAD in one gpurun call (tools/gpu.sh --timeout 3300 -- 'bash tools/r06_final.sh')
# Variant libraries it compares against (git-ignored, built here before the call):
#   tools/build_variant.sh agg_old "-DTSD_AGW_MIN_ROWS=0" kernels_misc.hip      (the one-wave-per-row aggregation)
#   tools/bin/lib_r05.so: `make -C tsdiff_amd/csrc ../libtsdiff_hip.so` in a worktree of commit 1c4256b (round 5), copied
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; R=r06
python3 -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2
( time timeout 1500 python3 -m pytest tests -q -m gpu ) > $O/${R}_gpu_tests.log 2>&1; tail -3 $O/${R}_gpu_tests.log
bash tools/profile_round.sh $R > $O/${R}_profile_round.log 2>&1
cp $O/${R}_pmc_traffic*.json $O/${R}_mfma_busy.json profiles/ 2>/dev/null  # (bench.py reads the counter files of THIS campaign)
bash tools/bench_round.sh $R > $O/${R}_bench_round.log 2>&1; tail -14 $O/${R}_bench_round.log
python3 tools/power_trace.py --hz 20 --seconds 6 --out $O/${R}_c5_power_clock.md c5 ens8 c2 > $O/${R}_power.log 2>&1; cat $O/${R}_c5_power_clock.md
# the windowed aggregation against the one-wave-per-row form (variant build -DTSD_AGW_MIN_ROWS=0), interleaved processes
python3 tools/ab_agg.py windowed=default one_wave_per_row=tools/bin/lib_agg_old.so 2>&1 | grep -v amdgpu > $O/${R}_ab_agg.log; cat $O/${R}_ab_agg.log
python3 tools/ab_train_prefetch.py 200 60 10 2>&1 | grep -v amdgpu > $O/${R}_ab_train_prefetch.log; cat $O/${R}_ab_train_prefetch.log
python3 tools/train_host_phases.py 200 pos 2>&1 | grep -v amdgpu > $O/${R}_train_host_phases.log; python3 tools/train_host_phases.py 2 pos 2>&1 | grep -v amdgpu >> $O/${R}_train_host_phases.log; TSDIFF_TRAIN_FLAT_GRAD=0 python3 tools/train_host_phases.py 2 pos 2>&1 | grep -v amdgpu | sed 's/^mode/per-parameter autograd form (TSDIFF_TRAIN_FLAT_GRAD=0): mode/' >> $O/${R}_train_host_phases.log; cat $O/${R}_train_host_phases.log
python3 tests/tools/split_f16_sweep.py > $O/${R}_split_f16_sweep.md 2> /dev/null; tail -4 $O/${R}_split_f16_sweep.md
for w in c2 g200 ens8 g300 c5 g300m8; do st=200; [ $w = c5 ] && st=8; [ $w = g300m8 ] && st=50; python3 tools/ab_step.py --workload $w --steps $st --rounds 2 r06=default r05=tools/bin/lib_r05.so; done 2>&1 | grep -v amdgpu > $O/${R}_ab_vs_r05.log; cat $O/${R}_ab_vs_r05.log
