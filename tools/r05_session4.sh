#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 tools/debug_trans2.py 2>&1 | grep "mega =="
( time timeout 2700 python3 -m pytest tests -q -m gpu ) > $O/s4_tests.log 2>&1; tail -12 $O/s4_tests.log
python3 tools/ab_step.py --workload c2 --rounds 2 trans=default old=tools/bin/lib_notrans.so > $O/s4_ab_c2.log 2>&1; cat $O/s4_ab_c2.log
