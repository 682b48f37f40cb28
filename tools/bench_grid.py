#!/usr/bin/env python3
"""Times the gridding kernel alone (HIP events around the launch, xh_rf_kernel_ms) on random orientations.

    python3 tools/bench_grid.py [--box 256] [--n 4096] [--reps 3] [--no-ctf] [--opt unit_z=4 --opt grid_waves=12] [--check]

Prints one JSON line: ms per launch, projections/s, fraction of the HBM roofline (B_grid = 4 D^2 + 24 n_vox bytes per
projection).  --check compares the temp spaces with those of the other unit depth (the same taps
summed in the same order: identical bits expected).
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-ctf", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="name=value options passed to xh_rf_set_option")
    args = ap.parse_args()
    import torch
    import xmipp3_amd as xa
    from tests import synth
    ctx = xa.Context(0)
    D, n = args.box, args.n
    g = torch.Generator(device="cuda").manual_seed(3)
    rf = xa.RecFourier(ctx, D)
    f = torch.randn((n, rf.sizeY, rf.sizeX, 2), generator=g, device="cuda")
    c = m = None
    if not args.no_ctf:
        c = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda") + 0.5
        m = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda")
    ang = synth.random_angles(n, np.random.default_rng(4))
    for o in args.opt:
        k, v = o.split("=")
        rf.set_option(k, float(v))

    def run(reps):
        rf.reset()
        rf.insert(f, ang, ctf=c, modulator=m)      # warm-up (allocations, first-touch)
        torch.cuda.synchronize()
        rf.kernel_ms(reset=True)
        rf.reset()
        for _ in range(reps):
            rf.insert(f, ang, ctf=c, modulator=m)
        torch.cuda.synchronize()
        ms, launches = rf.kernel_ms(reset=True)
        return ms / max(1, launches)

    ms = run(args.reps)
    mv = rf.mv
    nvox = math.pi * (mv / 2) ** 2 / 2 * 2 * 1.9
    bgrid = 4 * D * D + 24 * nvox
    out = {"opts": args.opt, "box": D, "n": n, "ctf": not args.no_ctf, "ms_per_launch": round(ms, 3),
           "projections_per_s": round(n / ms * 1e3), "B_grid": round(bgrid),
           "roofline_frac": round(n * bgrid / (ms * 1e-3) / 8e12, 4)}
    if args.check:
        got = rf.temp.clone()
        rf.set_option("unit_z", 4 if any(o.startswith("unit_z=4") for o in args.opt) is False else 8)
        rf.set_option("grid_waves", 0)
        run(1)
        ref = rf.temp
        sc = ref.abs().max().item()
        out["check_vs_other_unit_depth_rel"] = (got / args.reps - ref).abs().max().item() / sc
        d = (got != 0) != (ref != 0)
        out["voxel_sets_equal"] = not bool(d.any().item())
        out["identical_bits"] = bool(args.reps == 1 and (got == ref).all().item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
