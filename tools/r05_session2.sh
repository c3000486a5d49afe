#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "fused_encoder or unit" > $O/s2_tests.log 2>&1; tail -5 $O/s2_tests.log
python3 tools/ab_step.py --workload c5 --steps 8 --rounds 2 trans=default old=tools/bin/lib_utr0.so > $O/s2_ab_c5.log 2>&1; cat $O/s2_ab_c5.log
python3 tools/ab_step.py --workload ens8 --rounds 2 trans=default:fused1 old=tools/bin/lib_utr0.so:fused1 > $O/s2_ab_ens8.log 2>&1; cat $O/s2_ab_ens8.log
for w in c5 ens8; do
echo "== trace $w trans"; TSDIFF_LIB=tools/bin/lib_utrace.so python3 tools/trace_unit.py $w 2>&1 | tail -12
echo "== trace $w old"; TSDIFF_LIB=tools/bin/lib_utrace0.so python3 tools/trace_unit.py $w 2>&1 | tail -12
done > $O/s2_trace.log 2>&1; cat $O/s2_trace.log
