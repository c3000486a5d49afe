#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
db() { ls $1/*/*results.db $1/*results.db 2>/dev/null | head -1; }
B="--no-cpu-baseline --no-extras --no-f32"
rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o c2 -- python3 bench.py --steps 200 --warmup 20 $B > $O/s5_prof_c2.log 2>&1
python3 tools/rocpd_stats.py $(db /tmp/p_c2) > $O/s5_kernel_stats_c2.md; head -12 $O/s5_kernel_stats_c2.md
python3 tools/step_timeline.py $(db /tmp/p_c2) step_tail > $O/s5_step_timeline.md; tail -25 $O/s5_step_timeline.md
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -5
python3 tools/train_drift.py --steps 300 > $O/s5_drift.log 2>&1; tail -14 $O/s5_drift.log
