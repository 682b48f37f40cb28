cd $GRAFT_REPO_ROOT
for r in 1 2; do timeout 600 python3 bench.py --no-cpu-baseline --no-flexalign 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['warmup'], round(d['value']), round(d['ms_per_step'],2), 'streamed', round(d['value_streamed']), 'one_stream', round(d['one_stream_leg']['value']))"; done
