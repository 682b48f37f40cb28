cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_pm.py -q -x -k "branch_and_bound or full_size or config2 or any_box or ties or lists" 2>&1 | tail -3
pr() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$1', round(d['value']), round(d['ms_per_step'],2), {k:round(v/d['steps'],2) for k,v in d['stage_ms'].items() if k in ('contract','idft_max','prep32')}, d.get('parity_sample_identical'))
"; }
for r in 1 2; do
XMIPP_HIP_LIB=$PWD/xmipp3_amd/libxmipp_hip_prev.so python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 3 --refs noise 2>/dev/null | pr "noise prev"
python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 3 --refs noise 2>/dev/null | pr "noise new "
XMIPP_HIP_LIB=$PWD/xmipp3_amd/libxmipp_hip_prev.so python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 4 2>/dev/null | pr "default prev"
python3 bench.py --no-cpu-baseline --no-extra-legs --pipeline 0 --steps 4 2>/dev/null | pr "default new "
done
