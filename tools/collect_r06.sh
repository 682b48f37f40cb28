#!/bin/bash
# Round-6 evidence in one call on the GPU box (every step under `timeout`):
#   the default bench line as the driver runs it (with the FlexAlign and CLI legs), kernel stats of the default bench under rocprofv3
#   (two streams, and one stream: the kernels' own durations), one step kernel by kernel, idle gaps, HBM traffic and PMC counters of the
#   gridding kernel (stamped with the library's source hash), the counters of EVERY main-path kernel, FlexAlign's kernel stats and
#   counters (one lane), the other configurations' bench lines, the two drop-in programs end to end at 256 and 128 px, the host-feed
#   microbenchmark, FlexAlign's precision diagnosis at K3 size, the co-residency experiment.
#   bash tools/collect_r06.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
timeout 600 bash tools/profile_bench.sh --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_profile.txt 2>&1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/${tag}_kernel_stats_bench_default.csv
cp gpurun_out/bench_gaps.txt gpurun_out/${tag}_idle_between_kernels.txt
cp gpurun_out/bench_under_rocprof.json gpurun_out/${tag}_bench_under_rocprof.json
timeout 400 bash tools/prof_onestream.sh ${tag} > /dev/null 2>&1
timeout 400 bash tools/trace_onestream.sh ${tag} > /dev/null 2>&1
timeout 900 bash tools/collect_traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
cp gpurun_out/traffic_k_rf_grid.json gpurun_out/${tag}_traffic_k_rf_grid.json
timeout 900 bash tools/pmc_grid.sh ${tag} k_rf_grid > gpurun_out/${tag}_pmc.txt 2>&1
timeout 1300 bash tools/pmc_all_kernels.sh ${tag}_main bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --pipeline 0 2>&1 | grep -v "at::native" > gpurun_out/${tag}_pmc_all_kernels_main.txt
cp gpurun_out/pmc_all_${tag}_main.json gpurun_out/${tag}_pmc_all_kernels_main.json 2>/dev/null
rm -rf gpurun_out/pmca_${tag}_main
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fa
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa -o fa -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > $root/gpurun_out/${tag}_flexalign_under_rocprof_one_lane.json 2> /dev/null
f=$(find /tmp/prof_fa -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $root/gpurun_out/${tag}_flexalign_kernel_stats_one_lane.csv
cd $root
timeout 600 bash tools/pmc_all_kernels.sh ${tag}_flexalign bench.py --mode flexalign --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 2>&1 | grep -v "at::native" > gpurun_out/${tag}_flexalign_pmc_all_kernels.txt
cp gpurun_out/pmc_all_${tag}_flexalign.json gpurun_out/${tag}_flexalign_pmc_all_kernels.json 2>/dev/null
rm -rf gpurun_out/pmca_${tag}_flexalign
timeout 300 python3 bench.py --mode flexalign --steps 6 --warmup 2 > gpurun_out/${tag}_flexalign_bench.json 2> /dev/null
for m in "--mode match --box 128" "--mode grid" "--refs noise"; do
  n=$(echo $m | tr -d ' -' ); timeout 300 python3 bench.py $m --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_bench_${n}.json 2>/dev/null
done
# the drop-in programs end to end (files under /dev/shm)
timeout 600 python3 tools/bench_cli.py --particles 65536 --unique 16384 > gpurun_out/${tag}_cli_256.json 2> /dev/null
timeout 600 python3 tools/bench_cli.py --box 128 --particles 131072 --unique 32768 > gpurun_out/${tag}_cli_128.json 2> /dev/null
timeout 600 python3 tools/bench_cli.py --particles 32768 --unique 16384 --neighbours 50 > gpurun_out/${tag}_cli_256_50_neighbours.json 2> /dev/null
python3 tools/cli_report.py gpurun_out/${tag}_cli_256.json gpurun_out/${tag}_cli_128.json gpurun_out/${tag}_cli_256_50_neighbours.json > gpurun_out/${tag}_cli_report.txt 2>&1
# what the host can deliver (H2D, page-cache reads, both at once; unbound and bound to either NUMA node)
( hipcc --offload-arch=gfx950 -O2 -pthread tools/ubench_hostfeed.hip -o /tmp/ubench_hostfeed 2>/dev/null && for n in -1 0 1; do echo "== NUMA binding: $n (-1: none)"; timeout 120 /tmp/ubench_hostfeed 2 $n; done ) > gpurun_out/${tag}_ubench_hostfeed.txt 2>&1
timeout 300 python3 tools/diag_fa_precision.py 2>&1 | grep -v "amdgpu.ids\|^built" > gpurun_out/${tag}_flexalign_precision_k3.txt
timeout 300 python3 tools/exp_corun.py --reps 2 --s6-eps 0 2>&1 | grep -v "amdgpu.ids\|^built" > gpurun_out/${tag}_exp_corun.txt
( hipcc --offload-arch=gfx950 -O3 tools/ubench_coresidency.hip -o /tmp/cores 2>/dev/null && timeout 120 /tmp/cores ) > gpurun_out/${tag}_ubench_coresidency.txt 2>&1
tail -3 gpurun_out/${tag}_traffic.txt | cut -c1-400
python3 -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'streamed', d.get('value_streamed'), 'frac', d['roofline']['frac'], 'traffic_stale', d['roofline'].get('traffic_stale'))
print({k:(d[k]['value'] if isinstance(d.get(k),dict) and 'value' in d[k] else None) for k in ('worst_case','noise_gallery','compact_phantom','flexalign')})
"
cat gpurun_out/${tag}_cli_report.txt
