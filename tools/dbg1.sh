cd $GRAFT_REPO_ROOT
for r in 1 2; do python3 bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step'],2), 'steps_only', round(d['value_steps_only']), 'finish_s', round(d['finish_and_allreduce_s'],4), d.get('parity_sample_identical'))"; done
python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('driver-style', round(d['value']), round(d['ms_per_step'],2), 'steps_only', round(d['value_steps_only']), 'finish_s', round(d['finish_and_allreduce_s'],4))"
timeout 600 python3 -m pytest tests/test_gpu_rf.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -2
