#!/bin/bash
# The bound on what ANY weight-stationary tile GEMM could buy: a variant build in which every k-step
# of every split-f16 tile GEMM re-reads its first weight fragment (-DTSD_RING_WFAKE: weights from L1, WRONG results) against
# the default build, over the regimes the review names.   tools/build_variant.sh wfake_all "-DTSD_RING_WFAKE" kernels_combo.hip kernels_typed.hip kernels_unit.hip kernels_mlp.hip
cd $GRAFT_REPO_ROOT
W=tools/bin/lib_wfake_all.so
{
echo "# Weights from L1 in every split-f16 tile GEMM (TSD_RING_WFAKE, wrong results) against the default build"; echo
echo '```'
for wl in c2 ens8 g300m8; do python3 tools/ab_step.py --workload $wl --steps 100 default=default wfake=$W 2>&1 | tail -3; done
python3 tools/ab_step.py --workload c5 --steps 10 --rounds 2 default=default wfake=$W 2>&1 | tail -3
python3 tools/filter_probe.py 800 2>&1 | tail -1
TSDIFF_LIB=$W python3 tools/filter_probe.py 800 2>&1 | tail -1
echo '```'
} > gpurun_out/r06_weight_stationary_bound.md 2>&1
cat gpurun_out/r06_weight_stationary_bound.md
