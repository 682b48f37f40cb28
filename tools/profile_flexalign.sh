#!/bin/bash
# kernel stats of one K3-sized movie through FlexAlign (tools/bench_flexalign.py) under rocprofv3 -> gpurun_out/<tag>_flexalign_*
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa -o fa -- python3 $root/tools/bench_flexalign.py --reps 1 > $out/${tag}_flexalign_under_rocprof.json 2> $out/${tag}_flexalign_under_rocprof.err
f=$(find /tmp/prof_fa -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_flexalign_kernel_stats.csv
head -25 $out/${tag}_flexalign_kernel_stats.csv | cut -c1-200
