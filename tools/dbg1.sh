cd $GRAFT_REPO_ROOT
pr() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$1', round(d['value']), round(d['ms_per_step'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items() if k in ('contract','idft_max','prep32','rescore_fp64')}, d.get('parity_sample_identical'), d.get('rescored_fraction'))
"; }
python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 3 --refs noise --pm-opt store_cut=-1 2>/dev/null | pr "noise all stored"
python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 3 --refs noise 2>/dev/null | pr "noise bounds only"
python3 bench.py --no-extra-legs --steps 3 --refs noise 2>/dev/null | pr "noise bounds only, two streams, with parity"
timeout 900 python3 -m pytest tests/test_gpu_pm.py -q -x 2>&1 | tail -3
