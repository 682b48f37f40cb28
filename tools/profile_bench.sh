#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command; summary -> gpurun_out/bench_kernel_stats.csv, the idle
# time between kernels (tools/trace_gaps.py) -> gpurun_out/bench_gaps.txt
# Run on the GPU box:  bash tools/profile_bench.sh [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/bench_prof
rm -rf $out; mkdir -p $out
cd $root
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py "$@" > $out/bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/bench_kernel_stats.csv
t=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$t" > $root/gpurun_out/bench_gaps.txt 2>&1
grep '^{"metric"' $out/bench.log | tail -1 > $root/gpurun_out/bench_under_rocprof.json
head -12 $root/gpurun_out/bench_kernel_stats.csv | cut -c1-160
cat $root/gpurun_out/bench_gaps.txt
rm -rf $out/*/*kernel_trace.csv
