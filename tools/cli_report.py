#!/usr/bin/env python3
"""One screen of a tools/bench_cli.py JSON: per program the rates and XMIPP_HIP_TIMING's split.  Usage: tools/cli_report.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    c = d["config"]
    print(f"{f}: {c['box']} px, {c['nrefs']} references, {c['rows']} rows over {c['distinct_particles']} particles, {c['neighbour_lists']}")
    for p in ("xmipp_angular_projection_matching", "xmipp_reconstruct_fourier_accel"):
        x = d[p]
        t = x["timing_s"]
        print(f"  {p}: whole process {x['wall_s']:.2f} s = {x['particles_per_s']:.0f} /s; image loop {x['particles_per_s_image_loop']:.0f} /s"
              + (f"; library {x['library_particles_per_s']:.0f} /s -> loop / library {x['vs_library_image_loop']:.2f}" if "library_particles_per_s" in x else ""))
        print("     " + " ".join(f"{k}={t[k]:.3f}" for k in ("total", "setup", "parse", "bank", "loop", "stall", "device", "load", "h2d_wait", "format", "finish", "write") if k in t))
    if "outputs_vs_library" in d:
        print("  outputs vs library:", {k: v for k, v in d["outputs_vs_library"].items() if k != "note"})
