#!/bin/bash
# round 5: every measurement of the round at one HEAD in one gpurun call (tools/gpu.sh --timeout 3300 -- 'bash tools/r05_final.sh')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python3 -m pytest tests -q -m gpu ) > $O/r05_gpu_tests.log 2>&1; tail -3 $O/r05_gpu_tests.log
bash tools/profile_round.sh r05 > $O/r05_profile_round.log 2>&1
cp $O/r05_pmc_traffic*.json $O/r05_mfma_busy.json profiles/ 2>/dev/null  # (bench.py reads the counter files of THIS campaign)
bash tools/bench_round.sh r05 > $O/r05_bench_round.log 2>&1; tail -12 $O/r05_bench_round.log
python3 tools/power_trace.py --hz 20 --seconds 6 --out $O/r05_c5_power_clock.md c5 d128 d256 ens8 c2 > $O/r05_power.log 2>&1; cat $O/r05_c5_power_clock.md
bash tools/energy_probe.sh > $O/r05_energy_probe.log 2>&1; cat $O/r05_energy_probe.log | tail -6
{ for g in 400 800; do python3 tools/filter_probe.py $g; TSDIFF_LIB=tools/bin/lib_fw0.so python3 tools/filter_probe.py $g; done; } 2>/dev/null > $O/r05_filter_probe.log; cat $O/r05_filter_probe.log
{ echo "== tools/trace_unit.py c5"; TSDIFF_LIB=tools/bin/lib_utrace.so python3 tools/trace_unit.py c5; echo "== tools/trace_unit.py ens8"; TSDIFF_LIB=tools/bin/lib_utrace.so python3 tools/trace_unit.py ens8; } 2>/dev/null > $O/r05_trace_unit.log; cat $O/r05_trace_unit.log
TSDIFF_LIB=$PWD/tools/bin/lib_trace.so python3 tools/trace_combo.py g800 3 h2 > $O/r05_trace_combo_g800.log 2>/dev/null; head -12 $O/r05_trace_combo_g800.log
for w in c2 g120 g200 ens2 ens8 g300 c5 g300m8; do st=200; [ $w = c5 ] && st=8; [ $w = g300m8 ] && st=50; python3 tools/ab_step.py --workload $w --steps $st --rounds 2 r05=default r04=tools/bin/lib_r04.so; done 2>&1 | grep -v amdgpu > $O/r05_ab_vs_r04.log; cat $O/r05_ab_vs_r04.log
python3 tools/ab_step.py --workload ens8 --rounds 2 materialised=default fused=default:fused1 2>&1 | grep -v amdgpu > $O/r05_ab_fused_ens8.log; cat $O/r05_ab_fused_ens8.log
