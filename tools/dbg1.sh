cd $GRAFT_REPO_ROOT
pr() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$1', round(d['value']), round(d['ms_per_step'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items() if k in ('contract','idft_max','prep32')}, d.get('parity_sample_identical'))
"; }
for r in 1 2; do
for s in 14 141; do
timeout 300 python3 bench.py --no-extra-legs --pipeline 0 --steps 3 --refs noise --cpu-sample 64 --pm-opt contract_shape=$s 2>/dev/null | pr "noise shape $s"
done
for s in 14 141; do
timeout 300 python3 bench.py --no-extra-legs --pipeline 0 --steps 4 --cpu-sample 64 --pm-opt contract_shape=$s 2>/dev/null | pr "default shape $s"
done
done
