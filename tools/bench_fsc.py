"""Times xh_frc_dpr (Fourier shell correlation) on two D^3 fp64 maps resident in HBM and,
with --cpu, the oracle on the host cores beside it. Prints one JSON line. Not part of bench.py: the headline
metric is particles/s of the refine iteration; this is the measurement row of the widened component."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import xmipp3_amd as xa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dpr", action="store_true")
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    D = a.size
    ctx = xa.Context(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    v = torch.randn((D, D, D), generator=g, device="cuda", dtype=torch.float64)
    w = v + torch.randn((D, D, D), generator=g, device="cuda", dtype=torch.float64)
    xa.frc_dpr(ctx, v, w, 1.0, do_dpr=a.dpr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        r = xa.frc_dpr(ctx, v, w, 1.0, do_dpr=a.dpr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    xh = D // 2 + 1
    # compulsory traffic: both maps read once; as implemented: + half spectra written once and read/written by the
    # y and z line passes, then read by the shell pass (16 B per coefficient)
    alg = 2 * 8 * D ** 3
    impl = alg + 2 * 16 * D * D * xh * (1 + 2 + 2 + 1)
    out = {"metric": "fsc_maps_per_s", "size": D, "ms": ms, "do_dpr": a.dpr, "algorithmic_GBps": alg / ms / 1e6,
           "implemented_GBps": impl / ms / 1e6, "frc_shell_10": float(r["frc"][10])}
    if a.cpu:
        from oracle import pyoracle as o
        hv, hw = v.cpu().numpy(), w.cpu().numpy()
        t0 = time.perf_counter()
        e = o.frc_dpr(hv, hw, 1.0, do_dpr=a.dpr)
        out["cpu_ms"] = (time.perf_counter() - t0) * 1e3
        out["max_abs_frc_diff_vs_cpu"] = float(np.nanmax(np.abs(e["frc"] - r["frc"])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
