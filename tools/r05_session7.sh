#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 2400 python3 -m pytest tests -x -q -m gpu ) > $O/s7_tests.log 2>&1; tail -4 $O/s7_tests.log
python3 tools/ab_step.py --workload c5 --steps 8 --rounds 2 planes=default fp32rows=tools/bin/lib_fp32rows.so > $O/s7_ab_c5.log 2>&1; cat $O/s7_ab_c5.log
python3 tools/ab_step.py --workload ens8 --rounds 2 planes=default fp32rows=tools/bin/lib_fp32rows.so > $O/s7_ab_ens8.log 2>&1; cat $O/s7_ab_ens8.log
python3 tools/ab_step.py --workload c2 --rounds 2 planes=default fp32rows=tools/bin/lib_fp32rows.so > $O/s7_ab_c2.log 2>&1; cat $O/s7_ab_c2.log
python3 tools/ab_step.py --workload g300 --rounds 1 planes=default fp32rows=tools/bin/lib_fp32rows.so > $O/s7_ab_g300.log 2>&1; cat $O/s7_ab_g300.log
