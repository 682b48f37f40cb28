#!/bin/bash
# round-5 run 1: GPU suite, default bench, A/B of the options touched, kernel stats
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
tag=${1:-r05a}
timeout 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/${tag}_tests.txt
python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python3 bench.py --no-cpu-baseline --no-extra-legs --rf-opt records_from_images=1 > gpurun_out/${tag}_bench_rfi1.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-extra-legs --rf-opt ctf_fast=0 > gpurun_out/${tag}_bench_ctfslow.json 2>/dev/null
XH_PREFILTER_FORM=0 python3 bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_bench_fir.json 2>/dev/null
for f in default rfi1 ctfslow fir; do python3 - gpurun_out/${tag}_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], round(d["value"]), d["ms_per_step"], d.get("parity_sample_identical"), d.get("parity_volume_rel_err"), {k:round(v/ d["steps"],2) for k,v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
bash tools/profile_bench.sh --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_profile.txt 2>&1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv

python3 tools/kstats.py gpurun_out/${tag}_kernel_stats.csv 45 9
