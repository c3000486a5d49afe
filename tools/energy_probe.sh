#!/bin/bash
# Where does the time of the fused encoder go at BASELINE configs[4] (a launch that runs at the socket power limit)?  Timing
# experiments with WRONG results (variant builds): two of the three MFMAs per product; weights from L1 instead of L2.
#   tools/build_variant.sh mfma2 "-DTSD_MFMA2" kernels_unit.hip; tools/build_variant.sh wfake "-DTSD_UNIT_WFAKE" kernels_unit.hip
cd $GRAFT_REPO_ROOT
python3 tools/power_trace.py --hz 20 --seconds 5 --out gpurun_out/r05_energy_probe.md c5 > /dev/null 2>&1
for v in mfma2 wfake; do
  cp tsdiff_amd/libtsdiff_hip.so /tmp/lib_keep.so; cp tools/bin/lib_$v.so tsdiff_amd/libtsdiff_hip.so
  python3 tools/power_trace.py --hz 20 --seconds 5 --out gpurun_out/r05_energy_probe_$v.md c5 > /dev/null 2>&1
  cp /tmp/lib_keep.so tsdiff_amd/libtsdiff_hip.so
done
for f in gpurun_out/r05_energy_probe.md gpurun_out/r05_energy_probe_mfma2.md gpurun_out/r05_energy_probe_wfake.md; do echo $f; grep "^| c5" $f; done
