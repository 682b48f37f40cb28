cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --no-cpu-baseline --no-flexalign 2>gpurun_out/err.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step'],2), 'streamed', round(d['value_streamed']), 'one_stream', round(d['one_stream_leg']['value']), d['config']['timed'], d.get('parity_sample_identical'), {k:round(d[k]['value']) for k in ('worst_case','noise_gallery','compact_phantom')})"
tail -3 gpurun_out/err.txt
timeout 300 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('driver-style', round(d['value']), round(d['ms_per_step'],2), 'tail', d['finish_and_allreduce_s'])"
timeout 300 python3 bench.py --no-cpu-baseline --no-extra-legs --timed streamed 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streamed as value', round(d['value']), round(d['ms_per_step'],2))"
