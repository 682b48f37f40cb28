cd $GRAFT_REPO_ROOT
python3 bench.py --mode grid --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('grid mode', round(d['value']), round(d['ms_per_step'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items()})"
bash tools/ab_bench.sh 2 base -- --no-extra-legs 2>/dev/null | cut -c1-120
timeout 600 python3 -m pytest tests/test_gpu_rf.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -2
