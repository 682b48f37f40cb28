#!/bin/bash
# Builds a variant of the library for A/B measurements of the gridding kernel on one box:
#   bash tools/build_variant.sh <tag> [-DXG_...=...]   ->  xmipp3_amd/libxmipp_hip_<tag>.so   (XMIPP_HIP_LIB selects it)
set -e
tag=$1; shift
cd "$(dirname "$0")/../xmipp3_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build/variants
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -ffp-contract=off "$@" -c xh_rf.hip -o build/variants/xh_rf_$tag.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libxmipp_hip_$tag.so build/xh_ctx.o build/xh_pm.o build/xh_fp.o build/xh_fft2d.o build/xh_fsc.o build/xh_ctfops.o build/xh_flexalign.o build/variants/xh_rf_$tag.o
echo "built xmipp3_amd/libxmipp_hip_$tag.so"
