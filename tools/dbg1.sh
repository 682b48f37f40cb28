cd $GRAFT_REPO_ROOT
for r in 1 2; do for w in 12 8; do timeout 300 python3 bench.py --no-cpu-baseline --no-extra-legs --rf-opt grid_waves=$w 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('grid_waves $w', round(d['value']), round(d['ms_per_step'],2), 'grid', round(d['stage_ms']['k_rf_grid']/d['steps'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items() if k in ('prep32','contract','idft_max','rescore_fp64','translate_s6')})"; done; done
