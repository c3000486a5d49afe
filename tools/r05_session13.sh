#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -q -m gpu -x ) > $O/s13_tests.log 2>&1; tail -5 $O/s13_tests.log
for w in c2 ens8 g300 c5; do st=200; [ $w = c5 ] && st=8; python3 tools/ab_step.py --workload $w --steps $st --rounds 2 tr32=default:tr32 tr64=default:tr64 ta1=tools/bin/lib_ta1.so:tr32; done 2>&1 | grep -v amdgpu > $O/s13_ab.log; cat $O/s13_ab.log
