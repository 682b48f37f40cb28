#!/bin/bash
# PMC counters of EVERY kernel of a command, per kernel name: separate rocprofv3 --pmc passes (counter collection only), per-dispatch means,
# and what they say about the unit a kernel keeps busy.   bash tools/pmc_all_kernels.sh <tag> <python script> [args...]
#   -> gpurun_out/pmc_all_<tag>.json + a table on stdout
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmca_$tag
rm -rf $out; mkdir -p $out
cd $root
passes=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
)
i=0
for p in "${passes[@]}"; do
  timeout 900 rocprofv3 --pmc $p --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections, re
out, tag = sys.argv[1:3]
res = collections.defaultdict(dict)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, cs in per.items():
        for c, d in cs.items():
            v = list(d.values())
            res[k][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(f"{out}/../pmc_all_{tag}.json", "w"), indent=1)
rows = []
for k, c in res.items():
    g = lambda n: c.get(n, {}).get("mean", 0.0)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0                 # summed over the 8 XCDs
    if cyc < 2e4 or k.startswith("at::") or "rocclr" in k or "Cijk" in k: continue
    simd = cyc * 1024 / 4.0                          # SIMD quad-cycles of the launch
    rows.append((cyc, k, c.get("GRBM_GUI_ACTIVE", {}).get("dispatches", 0), g("SQ_ACTIVE_INST_VALU") / simd, g("SQ_VALU_MFMA_BUSY_CYCLES") / (simd * 4), g("SQ_LDS_IDX_ACTIVE") / (cyc * 256),
                 g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")), g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), g("SQ_WAVE_CYCLES") / simd,
                 (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9, g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
print(f"{'kernel':44s} {'n':>3s} {'Mcyc':>7s} {'valu':>5s} {'mfma':>5s} {'lds':>5s} {'conf':>5s} {'wait':>5s} {'wv/S':>5s} {'GB':>7s} {'L2hit':>5s}")
for r in sorted(rows, reverse=True):
    print(f"{r[1][:44]:44s} {r[2]:3d} {r[0] / 1e6:7.3f} {r[3]:5.2f} {r[4]:5.2f} {r[5]:5.2f} {r[6]:5.2f} {r[7]:5.2f} {r[8]:5.2f} {r[9]:7.2f} {r[10]:5.2f}")
PY
