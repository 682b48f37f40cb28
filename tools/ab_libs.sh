#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on ONE box: alternates them, three rounds.
#   bash tools/ab_libs.sh tagA tagB ... [-- bench_grid.py args]      ("base" = the product library)
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in 1 2 3; do
  for t in "${tags[@]}"; do
    lib=$PWD/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$PWD/xmipp3_amd/libxmipp_hip.so
    echo -n "$t  "; XMIPP_HIP_LIB=$lib python3 tools/bench_grid.py "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'])"
  done
done
