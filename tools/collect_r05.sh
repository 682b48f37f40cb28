#!/bin/bash
# Round-5 evidence in one call on the GPU box (every step under `timeout`): the default bench line (with the FlexAlign leg), kernel stats of
# the default bench under rocprofv3 (two streams, and one stream: the kernels' own durations), one step kernel by kernel, idle gaps, HBM
# traffic and PMC counters of the gridding kernel (stamped with the library's source hash), FlexAlign's kernel stats (ONE lane: rocprofv3
# over the two-lane bench -- two host threads -- hung twice this round), the other configurations' bench lines.
#   bash tools/collect_r05.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
timeout 600 python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
timeout 600 bash tools/profile_bench.sh --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_profile.txt 2>&1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/${tag}_kernel_stats_bench_default.csv
cp gpurun_out/bench_gaps.txt gpurun_out/${tag}_idle_between_kernels.txt
cp gpurun_out/bench_under_rocprof.json gpurun_out/${tag}_bench_under_rocprof.json
timeout 400 bash tools/prof_onestream.sh ${tag} > /dev/null 2>&1
timeout 400 bash tools/trace_onestream.sh ${tag} > /dev/null 2>&1
timeout 900 bash tools/collect_traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
cp gpurun_out/traffic_k_rf_grid.json gpurun_out/${tag}_traffic_k_rf_grid.json
timeout 900 bash tools/pmc_grid.sh ${tag} k_rf_grid > gpurun_out/${tag}_pmc.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fa
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa -o fa -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > $root/gpurun_out/${tag}_flexalign_under_rocprof_one_lane.json 2> /dev/null
f=$(find /tmp/prof_fa -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $root/gpurun_out/${tag}_flexalign_kernel_stats_one_lane.csv
cd $root
# counters of every FlexAlign kernel (one lane, two movies)
timeout 600 bash tools/pmc_all_kernels.sh ${tag}_flexalign bench.py --mode flexalign --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 2>&1 | grep -v "at::native" > gpurun_out/${tag}_flexalign_pmc_all_kernels.txt
cp gpurun_out/pmc_all_${tag}_flexalign.json gpurun_out/${tag}_flexalign_pmc_all_kernels.json 2>/dev/null
rm -rf gpurun_out/pmca_${tag}_flexalign
timeout 300 python3 bench.py --mode flexalign --steps 6 --warmup 2 > gpurun_out/${tag}_flexalign_bench.json 2> /dev/null
for m in "--mode match --box 128" "--mode grid" "--refs noise"; do
  n=$(echo $m | tr -d ' -' ); timeout 300 python3 bench.py $m --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_bench_${n}.json 2>/dev/null
done
tail -3 gpurun_out/${tag}_traffic.txt | cut -c1-400
python3 -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic_stale', d['roofline'].get('traffic_stale'))
print({k:(d[k]['value'] if isinstance(d.get(k),dict) and 'value' in d[k] else None) for k in ('worst_case','noise_gallery','compact_phantom','flexalign')})
"
