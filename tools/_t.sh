cd $GRAFT_REPO_ROOT
for e in 2e-5 1e-5 5e-6 2.5e-6; do
python3 bench.py --mode match --pm-opt s6_eps=$e --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms']; n=d['steps']
print('s6_eps $e', 'ms/step %.2f'%d['ms_per_step'], 'translate_s6 %.2f'%(s['translate_s6']/n), 'repeated', d.get('s6_repeated_fraction'))"
done
