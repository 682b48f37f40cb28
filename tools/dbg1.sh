cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_rf.py tests/test_gpu_pipeline.py tests/test_gpu_pm.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh 2 base base -- --no-extra-legs 2>/dev/null | cut -c1-330
for r in 1 2; do for o in 0 1; do python3 bench.py --no-cpu-baseline --no-extra-legs --rf-opt order_spaces=$o 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('order_spaces $o', round(d['value']), round(d['ms_per_step'],2), 'grid', round(d['stage_ms']['k_rf_grid']/d['steps'],2))"; done; done
