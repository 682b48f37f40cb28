#!/bin/bash
# A/B of gridding-kernel builds on ONE box (box-to-box spread is +-4 %): alternates the given libraries, three rounds.
#   bash tools/ab_grid.sh xmipp3_amd/libA.so xmipp3_amd/libB.so [bench_grid.py args]
a=$1; b=$2; shift 2
for r in 1 2 3; do
  for l in $a $b; do
    echo -n "$l  "; XMIPP_HIP_LIB=$PWD/$l python3 tools/bench_grid.py "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_launch'])"
  done
done
