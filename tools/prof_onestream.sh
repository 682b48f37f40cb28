#!/bin/bash
# kernel table of the default bench on ONE stream (--pipeline 0): no kernel waits for the other stream's gridding launch, so the
# averages are the kernels' own durations.   bash tools/prof_onestream.sh <tag> [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof1s_$tag
rm -rf $out; mkdir -p $out
cd $root
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --pipeline 0 --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 "$@" > $out/bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats_onestream.csv
grep '^{"metric"' $out/bench.log | tail -1 > $root/gpurun_out/${tag}_bench_onestream.json
rm -rf $out
python3 tools/kstats.py $root/gpurun_out/${tag}_kernel_stats_onestream.csv 60 5
