#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 1500 python3 -m pytest tests -q -m gpu -x ) > $O/s14_tests.log 2>&1; tail -5 $O/s14_tests.log
for w in c2 ens8 g300 c5; do st=200; [ $w = c5 ] && st=8; python3 tools/ab_step.py --workload $w --steps $st --rounds 2 new=default base=tools/bin/lib_base.so; done 2>&1 | grep -v amdgpu > $O/s14_ab.log; cat $O/s14_ab.log
