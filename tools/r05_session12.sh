#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 tools/ab_step.py --workload c2 --rounds 3 r05=default r04=tools/bin/lib_r04.so > $O/r05_ab_vs_r04_c2.log 2>&1; cat $O/r05_ab_vs_r04_c2.log
python3 tools/ab_step.py --workload ens8 --rounds 2 r05=default r04=tools/bin/lib_r04.so > $O/r05_ab_vs_r04_ens8.log 2>&1; cat $O/r05_ab_vs_r04_ens8.log
python3 tools/ab_step.py --workload g300 --rounds 2 r05=default r04=tools/bin/lib_r04.so > $O/r05_ab_vs_r04_g300.log 2>&1; cat $O/r05_ab_vs_r04_g300.log
python3 tools/ab_step.py --workload c5 --steps 8 --rounds 2 r05=default r04=tools/bin/lib_r04.so > $O/r05_ab_vs_r04_c5.log 2>&1; cat $O/r05_ab_vs_r04_c5.log
python3 tools/ab_step.py --workload g300m8 --steps 50 --rounds 1 r05=default r04=tools/bin/lib_r04.so > $O/r05_ab_vs_r04_g300m8.log 2>&1; cat $O/r05_ab_vs_r04_g300m8.log
