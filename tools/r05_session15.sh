#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -q -m gpu -x ) > $O/s15_tests.log 2>&1; tail -3 $O/s15_tests.log
for w in c2 c2 g50; do python3 tools/ab_step.py --workload $w --steps 200 --rounds 3 new=default base=tools/bin/lib_base.so; done 2>&1 | grep -v amdgpu > $O/s15_ab.log; cat $O/s15_ab.log
