#!/bin/bash
# PMC counters of one kernel inside an arbitrary command: separate rocprofv3 --pmc passes (counter collection only).
# Run on the GPU box:  bash tools/pmc_kernel.sh <tag> <kernel substring> <python script> [args...]
#   -> gpurun_out/pmc_<tag>.json (per-dispatch means of the named kernel)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmck_$tag
rm -rf $out; mkdir -p $out
cd $root
passes=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE SQ_WAVES"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"
)
i=0
for p in "${passes[@]}"; do
  timeout 900 rocprofv3 --pmc $p --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
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
        res[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "max": v[-1]}
json.dump({"kernel": kern, "counters_per_dispatch": res}, open(f"{out}/../pmc_{tag}.json", "w"), indent=1)
for c in sorted(res):
    print(f"{c:34s} mean {res[c]['mean']:.4g}  max {res[c]['max']:.4g}  (n={res[c]['dispatches']})")
PY
python3 $root/tools/libhash.py $root/gpurun_out/pmc_${tag}.json > /dev/null
