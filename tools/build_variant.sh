#!/bin/bash
# Builds a variant of the library for A/B measurements on one box (one source file recompiled with extra flags):
#   bash tools/build_variant.sh <tag> [-DXG_...=...]            ->  xmipp3_amd/libxmipp_hip_<tag>.so   (XMIPP_HIP_LIB selects it)
#   SRC=xh_pm bash tools/build_variant.sh <tag> [-DXH_...=...]   the matcher instead of the reconstruction (SRC=xh_flexalign, ...): ONLY the named
#   file sees the flags -- a variant of another file built without SRC is the product library under another name
set -e
tag=$1; shift
SRC=${SRC:-xh_rf}
cd "$(dirname "$0")/../xmipp3_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build/variants
FPC=""; [ "$SRC" == "xh_rf" ] && FPC="-ffp-contract=off"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result $FPC "$@" -c $SRC.hip -o build/variants/${SRC}_$tag.o
objs=""
for o in xh_ctx xh_pm xh_rf xh_fp xh_fft2d xh_fsc xh_ctfops xh_flexalign xh_estimators; do
  if [ "$o" == "$SRC" ]; then objs="$objs build/variants/${SRC}_$tag.o"; else objs="$objs build/$o.o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libxmipp_hip_$tag.so $objs
echo "built xmipp3_amd/libxmipp_hip_$tag.so"
