#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "fused or forms or unit" 2>&1 | tail -3
python3 tools/ab_step.py --workload ens8 --rounds 2 mat=default fused=default:fused1 > $O/s10_ab_ens8.log 2>&1; cat $O/s10_ab_ens8.log
python3 tools/ab_step.py --workload g300 --rounds 1 mat=default fused=default:fused1 > $O/s10_ab_g300.log 2>&1; cat $O/s10_ab_g300.log
python3 tools/ab_step.py --workload g150 --rounds 1 mat=default fused=default:fused1 > $O/s10_ab_g150.log 2>&1; cat $O/s10_ab_g150.log
echo "== trace ens8"; TSDIFF_LIB=tools/bin/lib_utrace.so python3 tools/trace_unit.py ens8 2>&1 | tail -10
