#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 1500 python3 -m pytest tests -q -m gpu -x ) > $O/s23_tests.log 2>&1; tail -3 $O/s23_tests.log
for w in c2 g50 g120 g150 g200; do python3 tools/ab_step.py --workload $w --steps 200 --rounds 3 new=default base=tools/bin/lib_base.so; done 2>&1 | grep -v amdgpu
