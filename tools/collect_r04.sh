#!/bin/bash
# Round-4 evidence in one call on the GPU box (see tools/collect_r03.sh): the default bench line, kernel stats of the default bench under
# rocprofv3, idle gaps, HBM traffic of the gridding kernel (two PMC passes over the bench command), PMC counters of the gridding kernel,
# and the same for FlexAlign: bench.py --mode flexalign (movies streamed from page-locked memory) + its kernel stats.
#   bash tools/collect_r04.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
bash tools/profile_bench.sh --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_profile.txt 2>&1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/${tag}_kernel_stats_bench_default.csv
cp gpurun_out/bench_gaps.txt gpurun_out/${tag}_idle_between_kernels.txt
cp gpurun_out/bench_under_rocprof.json gpurun_out/${tag}_bench_under_rocprof.json
bash tools/collect_traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
cp gpurun_out/traffic_k_rf_grid.json gpurun_out/${tag}_traffic_k_rf_grid.json
bash tools/pmc_grid.sh ${tag} k_rf_grid > gpurun_out/${tag}_pmc.txt 2>&1
python3 bench.py --mode flexalign --steps 6 --warmup 1 > gpurun_out/${tag}_flexalign_bench.json 2> gpurun_out/${tag}_flexalign_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa -o fa -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_flexalign_under_rocprof.json 2> $root/gpurun_out/${tag}_flexalign_under_rocprof.err
f=$(find /tmp/prof_fa -name '*kernel_stats.csv' | head -1)
cp "$f" $root/gpurun_out/${tag}_flexalign_kernel_stats.csv
cd $root
for m in "--mode match --box 128" "--mode grid" "--refs noise"; do
  n=$(echo $m | tr -d ' -' ); python3 bench.py $m --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_bench_${n}.json 2>/dev/null
done
tail -3 gpurun_out/${tag}_traffic.txt | cut -c1-400
