#!/bin/bash
# After the last source edit of a round: HBM traffic and PMC counters of the gridding kernel again (stamped with the library's source hash),
# copied to where bench.py reads them, then the default bench line as the driver runs it.   bash tools/restamp.sh <tag>
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
timeout 900 bash tools/collect_traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
cp gpurun_out/traffic_k_rf_grid.json gpurun_out/${tag}_traffic_k_rf_grid.json
timeout 900 bash tools/pmc_grid.sh ${tag} k_rf_grid > gpurun_out/${tag}_pmc.txt 2>&1
cp gpurun_out/pmc_${tag}.json gpurun_out/${tag}_pmc_k_rf_grid.json
cp gpurun_out/${tag}_traffic_k_rf_grid.json profiles/traffic_k_rf_grid.json
cp gpurun_out/${tag}_pmc_k_rf_grid.json profiles/pmc_k_rf_grid.json
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python3 -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().split('\n')[-1])
r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'streamed', d.get('value_streamed'), 'frac', r['frac'], 'traffic', r['traffic'], 'stale', r.get('traffic_stale'), r['second_bound']['stale'], 'valu', r['second_bound']['valu_instructions_per_launch'])
print({k:(d[k]['value'] if isinstance(d.get(k),dict) and 'value' in d[k] else None) for k in ('worst_case','noise_gallery','compact_phantom','flexalign')})
print('cli', {p: d['cli'][p]['particles_per_s_image_loop'] for p in ('xmipp_angular_projection_matching','xmipp_reconstruct_fourier_accel')}, d['flexalign'].get('parity_sample'))
"
