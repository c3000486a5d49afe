#!/bin/bash
# round-5 session 1: baselines on this round's box (power/clock trace, step times, kernel stats of the 8-checkpoint ensemble)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 tools/power_trace.py --hz 20 --seconds 6 --out $O/r05_c5_power_clock_base.md c5 d128 d256 ens8 c2 > $O/s1_power.log 2>&1
python3 tools/ab_step.py --workload ens8 --rounds 1 base=default fused=default:fused1 > $O/s1_ab_ens8.log 2>&1
python3 tools/ab_step.py --workload g300 --rounds 1 base=default fused=default:fused1 > $O/s1_ab_g300.log 2>&1
python3 tools/ab_step.py --workload g150 --rounds 1 base=default fused=default:fused1 > $O/s1_ab_g150.log 2>&1
python3 tools/ab_step.py --workload c2 --rounds 1 base=default > $O/s1_ab_c2.log 2>&1
python3 tools/ab_step.py --workload c5 --steps 8 --rounds 1 base=default > $O/s1_ab_c5.log 2>&1
db() { ls $1/*/*results.db $1/*results.db 2>/dev/null | head -1; }
B="--no-cpu-baseline --no-extras --no-f32"
rocprofv3 --kernel-trace --stats -d /tmp/p_e8 -o e8 -- python3 bench.py --models 8 --steps 50 --warmup 5 $B > $O/s1_prof_e8.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_e8) > $O/r05_kernel_stats_ens8_base.md
python3 tools/step_timeline.py $(db /tmp/p_e8) step_tail > $O/r05_step_timeline_ens8_base.md
cat $O/r05_c5_power_clock_base.md; cat $O/s1_ab_*.log; head -20 $O/r05_kernel_stats_ens8_base.md
