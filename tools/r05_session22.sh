#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 1500 python3 -m pytest tests -q -m gpu -x ) > $O/s22_tests.log 2>&1; tail -3 $O/s22_tests.log
for w in c2 c2 g50 g150 g20; do python3 tools/ab_step.py --workload $w --steps 200 --rounds 3 new=default base=tools/bin/lib_base.so; done 2>&1 | grep -v amdgpu
python3 tools/trace_mega.py 2>&1 | grep -v amdgpu | grep "filter layer\|block [0-9]:\|^pair\|^filter"
