cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_pm.py -q -x 2>&1 | tail -3
for r in 1 2; do for o in 1 2; do python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 4 --pm-opt fir64_fused=$o 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fir64_fused $o', round(d['value']), round(d['ms_per_step'],2), 'rescore', round(d['stage_ms']['rescore_fp64']/d['steps'],2), d.get('parity_sample_identical'))"; done; done
timeout 600 python3 bench.py --no-cpu-baseline --no-flexalign 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']), d['ms_per_step'], {k:d[k] for k in ('worst_case','noise_gallery','compact_phantom')})"
