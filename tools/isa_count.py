#!/usr/bin/env python3
"""Instruction mix of one kernel in a device assembly file (hipcc --cuda-device-only -S):
   python3 tools/isa_count.py file.s <kernel name substring> [label-range-start label-range-end]
Prints counts per class for the whole kernel and per basic block (label), so that loop bodies can be read off."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l) or (l.endswith(":") and kern in l and not l.startswith(".")):
        start = i; break
if start is None:
    sys.exit("kernel not found")
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm") or ".end_amdhsa_kernel" in lines[i])
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("ds_", )): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
tot = collections.Counter(); blocks = []; cur = ["<entry>", collections.Counter()]
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", "//")): continue
    if t.endswith(":") and not t.startswith("."):
        continue
    if re.match(r"^\.LBB\w+:", t):
        blocks.append(cur); cur = [t[:-1], collections.Counter()]; continue
    if t.startswith("."): continue
    op = t.split()[0]
    c = cls(op); tot[c] += 1; cur[1][c] += 1
    if c == "valu" and ("f64" in op): tot["valu_f64"] += 1; cur[1]["valu_f64"] += 1
blocks.append(cur)
print(kern, dict(tot))
for name, c in blocks:
    n = sum(v for k, v in c.items() if k != "valu_f64")
    if n >= 12: print(f"  {name:14s} {n:5d}  {dict(c)}")
