#!/bin/bash
# A/B of whole-library variants on ONE box with the default bench: alternates them, N rounds.
#   bash tools/ab_bench.sh <rounds> tagA tagB ... [-- bench args]     ("base" = xmipp3_amd/libxmipp_hip.so, else libxmipp_hip_<tag>.so)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
rounds=$1; shift
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq 1 $rounds); do
  for t in "${tags[@]}"; do
    lib=$root/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$root/xmipp3_amd/libxmipp_hip.so
    XMIPP_HIP_LIB=$lib python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d.get('one_stream_leg',{})
st={k:round(v/d['steps'],2) for k,v in d['stage_ms'].items()}
print('$t', round(d['value']), round(d['ms_per_step'],2), 'steps_only', round(d.get('value_steps_only',0)), 'one_stream', round(o.get('value',0)), 'grid_alone', round(o.get('k_rf_grid_avg_launch_ms',0),2), {k:round(v,2) for k,v in o.get('matcher_stage_ms_per_step',{}).items()}, st, 'parity', d.get('parity_sample_identical'), d.get('parity_volume_rel_err'))
"
  done
done
