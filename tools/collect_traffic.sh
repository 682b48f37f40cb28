#!/bin/bash
# Collects the HBM traffic of the dominant kernel (k_rf_grid) for bench.py's "roofline.traffic":
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the bench
# command itself, per-dispatch means, gfx950 correction (FETCH_SIZE x2, MI355X_MICROARCH.md "HBM").
# Run on the GPU box:  bash tools/collect_traffic.sh   -> gpurun_out/traffic_k_rf_grid.json
set -e
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/traffic
rm -rf $out; mkdir -p $out
cd $root
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $out/$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_rf_grid" in r["Kernel_Name"] and r["Counter_Name"] == c:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = sorted(per.values())
    res[c] = {"dispatches": len(vals), "mean": sum(vals) / max(1, len(vals)), "min": vals[0], "max": vals[-1]}
line = [l for l in open(f"{out}/WRITE_SIZE.log") if l.startswith('{"metric"')][-1]
b = json.loads(line)
B = b["config"]["particles_per_step_per_gpu"]
kb = 1024.0   # rocprofv3 reports both counters in KiB
fetch = 2.0 * res["FETCH_SIZE"]["mean"] * kb      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
write = res["WRITE_SIZE"]["mean"] * kb
doc = {"kernel": "k_rf_grid<4,false>", "command": "rocprofv3 --pmc <C> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs",
       "projections_per_launch": B, "raw_counters_KiB": res,
       "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write, "traffic_MB_per_projection": (fetch + write) / B / 1e6,
       "algorithmic_MB_per_projection": b["roofline"].get("algorithmic_MB_per_projection", 9.65)}
json.dump(doc, open(f"{out}/../traffic_k_rf_grid.json", "w"), indent=1)
print(json.dumps(doc))
PY
python3 $root/tools/libhash.py $root/gpurun_out/traffic_k_rf_grid.json > /dev/null
