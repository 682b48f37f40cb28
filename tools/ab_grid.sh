#!/bin/bash
# A/B of gridding-kernel configurations on ONE box (box-to-box spread is +-4 %): alternates the given option sets, three rounds.
#   bash tools/ab_grid.sh "unit_z=4" "unit_z=8" "unit_z=8,grid_waves=8" [-- bench_grid.py args]
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in 1 2 3; do
  for o in "${sets[@]}"; do
    args=(); IFS=',' read -ra kv <<< "$o"; for x in "${kv[@]}"; do args+=(--opt "$x"); done
    echo -n "$o  "; python3 tools/bench_grid.py "${args[@]}" "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'], d.get('identical_bits',''), d.get('voxel_sets_equal',''))"
  done
done
