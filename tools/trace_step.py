#!/usr/bin/env python3
"""The kernels of ONE step of a rocprofv3 kernel trace, in launch order, with start offsets and durations (one-stream runs read best):
    python3 tools/trace_step.py trace.csv [marker kernel substring, default k_rf_grid] [which step, default the last full one]"""
import csv, sys
f = sys.argv[1]
mark = sys.argv[2] if len(sys.argv) > 2 else "k_rf_grid"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if mark in r[2]]
w = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
a, b = idx[w] + 1, idx[w + 1]
t0 = rows[a][0]
busy = 0
for s, e, n in rows[a:b + 1]:
    print("%9.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, n[:110]))
    busy += e - s
print("span %.3f ms, sum of kernels %.3f ms" % ((rows[b][1] - t0) / 1e6, busy / 1e6))
