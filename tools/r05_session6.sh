#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_split16.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -5
python3 tools/ab_step.py --workload ens8 --rounds 2 wide=default narrow=tools/bin/lib_nw0.so > $O/s6_ab_ens8.log 2>&1; cat $O/s6_ab_ens8.log
python3 tools/ab_step.py --workload g300 --rounds 2 wide=default narrow=tools/bin/lib_nw0.so > $O/s6_ab_g300.log 2>&1; cat $O/s6_ab_g300.log
python3 tools/ab_step.py --workload g600 --rounds 1 wide=default narrow=tools/bin/lib_nw0.so > $O/s6_ab_g600.log 2>&1; cat $O/s6_ab_g600.log
