#!/bin/bash
# Build a kernel variant of libtsdiff_hip.so for A/B timing:   tools/build_variant.sh NAME "-DFLAG ..." [TU ...]
# Recompiles the listed translation units (default: kernels_combo.hip) with the extra flags, takes the other objects
# from tsdiff_amd/csrc (run `make -C tsdiff_amd/csrc` first) and links tools/bin/lib_NAME.so (git-ignored, ships with gpurun).
set -e
NAME=$1; FLAGS=$2; shift 2 || true
TUS=${@:-kernels_combo.hip}
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/tsdiff_amd/csrc
OUT=$ROOT/tools/bin/obj_$NAME
mkdir -p $OUT
OBJS=""
for f in api kernels_mlp kernels_combo kernels_unit kernels_graph kernels_misc kernels_typed kernels_train train_step; do
  if echo " $TUS " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=on $FLAGS -c $SRC/$f.hip -o $OUT/$f.o &
    OBJS="$OBJS $OUT/$f.o"
  else
    OBJS="$OBJS $SRC/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/bin/lib_$NAME.so
echo built tools/bin/lib_$NAME.so
