#!/bin/bash
# the two-level cut of the contraction against the kind of map: match-only steps of bench.py per K0, for the default gallery and the compact phantom
root=${GRAFT_REPO_ROOT:-.}
for w in phantom compact; do
for k in -1 16 28 48 64 96 128 200 1048576; do
python3 $root/bench.py --mode match --refs $w --k0 $k --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=d['stage_ms']; n=d['steps']
print('$w k0=$k', 'ms/step %.2f'%d['ms_per_step'], 'cut', d['s2_two_level_cut'], 'pruned %.4f'%d['s3_rows_pruned_fraction'], 'rescored %.3f'%d['rescored_fraction'], ' '.join('%s %.2f'%(k_,v_/n) for k_,v_ in s.items() if v_))
"
done; done
