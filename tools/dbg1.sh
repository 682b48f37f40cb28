cd $GRAFT_REPO_ROOT
python3 bench.py --no-extra-legs --steps 3 --refs noise 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('noise', round(d['value']), round(d['ms_per_step'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items() if k in ('contract','idft_max','prep32','rescore_fp64')}, d.get('parity_sample_identical'))"
