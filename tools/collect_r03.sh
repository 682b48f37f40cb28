#!/bin/bash
# Round-3 evidence in one call on the GPU box: kernel stats of the default bench under rocprofv3, idle gaps, HBM traffic of the
# gridding kernel (two PMC passes over the bench command), the PMC counters of the gridding kernel (bench_grid.py).
#   bash tools/collect_r03.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r03}
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
tail -3 gpurun_out/${tag}_traffic.txt | cut -c1-400
