#!/bin/bash
# PMC counters of the gridding kernel: separate rocprofv3 --pmc passes (counter collection only, no tracing) over
# tools/bench_grid.py, summed per dispatch and averaged over the dispatches of the named kernel.
# Run on the GPU box:  bash tools/pmc_grid.sh [tag] [kernel substring] [bench_grid.py args...]
#   -> gpurun_out/pmc_<tag>.json   (copy into profiles/ to keep)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-grid}; shift
kern=${1:-k_rf_grid}; shift
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd $root
passes=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_GDS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE"
)
i=0
for p in "${passes[@]}"; do
  timeout 600 rocprofv3 --pmc $p --output-format csv -d $out/p$i -- python3 tools/bench_grid.py --reps 1 "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$out" "$kern" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, kern, tag = sys.argv[1:4]
res = {}
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        v = sorted(d.values())
        res[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "last": d[max(d, key=int)]}
line = None
for l in glob.glob(f"{out}/p*.log"):
    for t in open(l):
        if t.startswith('{"variant"'):
            line = json.loads(t)
doc = {"kernel": kern, "bench_line_under_pmc": line, "counters_per_dispatch": res}
json.dump(doc, open(f"{out}/../pmc_{tag}.json", "w"), indent=1)
for c in sorted(res):
    print(f"{c:34s} {res[c]['last']:.4g}  (n={res[c]['dispatches']})")
PY
python3 $root/tools/libhash.py $root/gpurun_out/pmc_${tag}.json > /dev/null
