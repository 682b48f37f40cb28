#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the match-only step of bench.py, alternating, two rounds: the stage split per variant
#   bash tools/ab_match.sh tagA tagB ... [-- bench.py args]      ("base" = the product library)
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in 1 2; do
  for t in "${tags[@]}"; do
    lib=$PWD/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$PWD/xmipp3_amd/libxmipp_hip.so
    XMIPP_HIP_LIB=$lib python3 bench.py --mode match --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms']; n=d['steps']
print('%-8s'%'$t', 'ms/step %.2f'%d['ms_per_step'], ' '.join('%s %.2f'%(k_,v_/n) for k_,v_ in s.items() if v_))"
  done
done
