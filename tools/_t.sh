root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o x -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > /tmp/out_x.json 2>/dev/null
f=$(find /tmp/prof_x -name '*kernel_stats.csv' | head -1)
cp $f $root/gpurun_out/fa_now_kernel_stats.csv
python3 -c "
import json; d=json.loads(open('/tmp/out_x.json').readline()); print(d['value'], d['stage_ms'], d.get('roofline_other_kernels'), d['roofline'])"
