#!/usr/bin/env python3
"""Prints the top kernels of a rocprofv3 kernel_stats CSV (tools/profile_bench.sh writes gpurun_out/bench_kernel_stats.csv).
    python3 tools/kstats.py [csv] [top-n] [steps]      ms per step when the number of bench steps (warm-up included) is given"""
import csv
import sys

f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_kernel_stats.csv"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f%s" % (tot / 1e6, "  (%.2f per step)" % (tot / 1e6 / steps) if steps else ""))
for r in rows[:top]:
    t = float(r["TotalDurationNs"]) / 1e6
    print("%9.2f ms %s %6d calls %10.1f us  %s" % (t, ("%7.2f/step" % (t / steps)) if steps else "", int(r["Calls"]),
                                                  float(r["AverageNs"]) / 1e3, r["Name"][:96]))
