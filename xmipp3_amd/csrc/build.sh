#!/bin/bash
# Builds libxmipp_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result"
# XH_DEBUG_HOOKS=1: the A/B and test knobs of xh_common.h (environment variables, wrong-by-design experiment options) compiled in
[ -n "$XH_DEBUG_HOOKS" ] && COMMON="$COMMON -DXH_DEBUG_HOOKS"
mkdir -p build
pids=()
# geometry of the gridding must round like the reference's scalar code: no FMA contraction
$HIPCC $COMMON -ffp-contract=off -c xh_rf.hip -o build/xh_rf.o & pids+=($!)
$HIPCC $COMMON -c xh_ctx.hip -o build/xh_ctx.o & pids+=($!)
if [ -f xh_pm.hip ]; then $HIPCC $COMMON -c xh_pm.hip -o build/xh_pm.o & pids+=($!); fi
$HIPCC $COMMON -c xh_fp.hip -o build/xh_fp.o & pids+=($!)
$HIPCC $COMMON -c xh_fft2d.hip -o build/xh_fft2d.o & pids+=($!)
$HIPCC $COMMON -c xh_ctfops.hip -o build/xh_ctfops.o & pids+=($!)
$HIPCC $COMMON -c xh_flexalign.hip -o build/xh_flexalign.o & pids+=($!)
$HIPCC $COMMON -c xh_estimators.hip -o build/xh_estimators.o & pids+=($!)
# shell membership of the FSC is decided in double arithmetic that must round like the scalar code
$HIPCC $COMMON -ffp-contract=off -c xh_fsc.hip -o build/xh_fsc.o & pids+=($!)
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
if [ $fail -ne 0 ]; then echo "build.sh: compilation FAILED" >&2; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libxmipp_hip.so build/xh_rf.o build/xh_ctx.o build/xh_pm.o build/xh_fp.o build/xh_fft2d.o build/xh_fsc.o build/xh_ctfops.o build/xh_flexalign.o build/xh_estimators.o
echo "built $(cd .. && pwd)/libxmipp_hip.so"
