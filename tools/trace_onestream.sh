#!/bin/bash
# kernel-by-kernel listing of one step of the default bench on one stream.  bash tools/trace_onestream.sh <tag> [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/trace1s_$tag
rm -rf $out; mkdir -p $out
cd $root
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --pipeline 0 --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 "$@" > $out/bench.log 2>&1
t=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_step.py "$t" > $root/gpurun_out/${tag}_step_trace.txt
cat $root/gpurun_out/${tag}_step_trace.txt
