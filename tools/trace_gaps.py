#!/usr/bin/env python3
"""Idle time of the GPU between the kernels of a rocprofv3 kernel trace (…kernel_trace.csv): how much of the wall time
between the first and last launch of a marker kernel no kernel was running, and the largest gaps with their neighbours.
    python3 tools/trace_gaps.py trace.csv [marker kernel substring, default k_rf_grid]"""
import csv
import sys

f = sys.argv[1]
mark = sys.argv[2] if len(sys.argv) > 2 else "k_rf_grid"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if mark in r[2]]
if len(idx) < 3:
    sys.exit("marker kernel launched fewer than three times")
a, b = idx[1], idx[-1]          # from the second to the last launch of the marker: whole steps
span = rows[b][0] - rows[a][0]
busy_end = rows[a][0]
idle = 0
gaps = []
for i in range(a, b):
    s, e, n = rows[i]
    if s > busy_end:
        idle += s - busy_end
        gaps.append((s - busy_end, rows[i - 1][2][:60], n[:60]))
    busy_end = max(busy_end, e)
nsteps = len(idx) - 2
print("steps %d  span %.2f ms/step  idle %.2f ms/step (%.1f %%)" % (nsteps, span / nsteps / 1e6, idle / nsteps / 1e6, 100.0 * idle / span))
agg = {}
for g, p, n in gaps:
    k = (p, n)
    agg[k] = agg.get(k, 0) + g
for (p, n), g in sorted(agg.items(), key=lambda kv: -kv[1])[:15]:
    print("%8.3f ms/step  after %-60s before %s" % (g / nsteps / 1e6, p, n))
