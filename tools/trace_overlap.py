#!/usr/bin/env python3
"""Which kernels ran WHILE a k_rf_grid launch was running?  Reads a rocprofv3 --kernel-trace csv.  Usage: tools/trace_overlap.py <kernel_trace.csv>"""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
grids = [k for k in ker if "k_rf_grid" in k[2]]
print(f"{len(ker)} kernels, {len(grids)} k_rf_grid launches")
for g in grids[-6:]:
    inside = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q in ker:
        if n is g[2] and s == g[0]:
            continue
        ov = min(e, g[1]) - max(s, g[0])
        if ov > 0:
            name = n.split("(")[0].replace("void ", "")[:48]
            inside[(name, q)][0] += 1
            inside[(name, q)][1] += ov / 1e6
    print(f"k_rf_grid {(g[1] - g[0]) / 1e6:7.2f} ms on queue {g[3]}: " + ("nothing ran beside it" if not inside else ""))
    for (name, q), (c, ms) in sorted(inside.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"      {name:48s} queue {q}  x{c:3d}  {ms:7.2f} ms overlapped")
