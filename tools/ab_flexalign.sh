#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on one lane of the FlexAlign bench, alternating, two rounds: movies/s and the stage split
#   bash tools/ab_flexalign.sh tagA tagB ... [-- bench.py args]      ("base" = the product library)
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in 1 2; do
  for t in "${tags[@]}"; do
    lib=$PWD/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$PWD/xmipp3_amd/libxmipp_hip.so
    XMIPP_HIP_LIB=$lib python3 bench.py --mode flexalign --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-8s'%'$t', 'movies/s %.2f'%d['value'], 'ms/movie %.2f'%d['ms_per_step'], d.get('stage_ms_one_lane'), 'parity', d.get('parity_sample',{}).get('max_abs_diff'))"
  done
done
