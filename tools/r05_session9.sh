#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "fused or forms or ensemble" 2>&1 | tail -3
python3 tools/ab_step.py --workload ens8 --rounds 2 early=default late=tools/bin/lib_pe0.so > $O/s9_ab_ens8.log 2>&1; cat $O/s9_ab_ens8.log
python3 tools/ab_step.py --workload c5 --steps 8 --rounds 1 early=default late=tools/bin/lib_pe0.so > $O/s9_ab_c5.log 2>&1; cat $O/s9_ab_c5.log
TSDIFF_LIB=$PWD/tools/bin/lib_trace.so python3 tools/trace_combo.py g800 3 h2 > $O/s9_trace_g800.log 2>&1; cat $O/s9_trace_g800.log
