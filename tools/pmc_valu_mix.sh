#!/bin/bash
# Vector-instruction mix of every kernel of a command (add / mul / fma / transcendental / convert / int32 / int64 per wave), two counter
# passes:   bash tools/pmc_valu_mix.sh <tag> <python script> [args...]   -> gpurun_out/<tag>_valu_mix.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmcmix_$tag; rm -rf $out; mkdir -p $out
cd $root
passes=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM")
i=0
for p in "${passes[@]}"; do
  timeout 600 rocprofv3 --pmc $p --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$out" > $root/gpurun_out/${tag}_valu_mix.txt <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, cs in per.items():
        for c, d in cs.items():
            v = list(d.values()); res[k][c] = sum(v) / len(v)
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
        "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"]
print(f"{'kernel (instructions per wave)':44s} " + " ".join(f"{c.replace('SQ_INSTS_', '').replace('VALU_', 'v.'):>9s}" for c in cols))
for k, c in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = c.get("SQ_WAVES", 0)
    if w < 1 or k.startswith("at::") or "rocclr" in k or c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(f"{k[:44]:44s} " + " ".join(f"{c.get(x, 0) / w:9.0f}" for x in cols))
PY
cat $root/gpurun_out/${tag}_valu_mix.txt | head -30
