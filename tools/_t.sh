cd $GRAFT_REPO_ROOT
for l in 4 5 6 8 4 6; do python3 bench.py --mode flexalign --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs --fa-lanes $l 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lanes $l', 'movies/s %.2f'%d['value'], 'ms/movie %.2f'%d['ms_per_step'])"; done
