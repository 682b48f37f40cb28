#!/usr/bin/env python3
"""Where FlexAlign's global alignment loses digits at K3 size: the pair shifts of three 4092 x 5760 frames of the
bench's kind (int8 counts of a smooth field under drift) on the device, in every combination of its transform forms, against the
oracle's double-precision arithmetic.     python tools/diag_fa_precision.py"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def k3_frames(torch, dev, n=3, Y=4092, X=5760, seed=11, counts=True):
    """the first n frames of bench.py's synthetic movie (main_flexalign.make_movie): same field, same drift law, same noise"""
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.randn((Y + 128, X + 128), generator=g, device=dev)
    k = torch.fft.rfft2(base)
    fy = torch.fft.fftfreq(Y + 128, device=dev)[:, None]
    fx = torch.fft.rfftfreq(X + 128, device=dev)[None, :]
    base = torch.fft.irfft2(k * torch.exp(-2 * (math.pi * 4.0) ** 2 * (fx * fx + fy * fy)), s=base.shape) * 30
    del k
    t = np.arange(n, dtype=np.float64)
    sgn = 1.0 if seed % 2 else -1.0
    drift = np.stack([sgn * 14.0 * (1 - np.exp(-t / 9.0)) + 0.04 * t, -9.0 * (1 - np.exp(-t / 14.0)) + 0.06 * t], 1)
    H, W = base.shape
    ys = torch.arange(Y, device=dev, dtype=torch.float32)[:, None]
    xs = torch.arange(X, device=dev, dtype=torch.float32)[None, :]
    u, v = (xs / X - 0.5) * 2, (ys / Y - 0.5) * 2
    mv = torch.empty((n, Y, X), device=dev)
    for i in range(n):
        a = 6.0 * i / 39.0
        sx = xs + 64 + float(drift[i, 0]) + a * (0.7 * u + 0.3 * v)
        sy = ys + 64 + float(drift[i, 1]) + a * (0.8 * v - 0.2 * u)
        grid = torch.stack(((sx + 0.5) * (2.0 / W) - 1, (sy + 0.5) * (2.0 / H) - 1), -1)[None]
        mv[i] = torch.nn.functional.grid_sample(base[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
        mv[i] += 0.5 * torch.randn((Y, X), generator=g, device=dev)
    if counts:
        mv = ((mv - mv.mean()) * (12.0 / mv.std()) + 40.0).round_().clamp_(0, 127)
    return mv.contiguous(), drift


def main():
    import torch
    import __graft_entry__ as ge
    ge.build()
    import xmipp3_amd as xa
    from oracle import pyoracle as o
    dev = torch.device("cuda", 0)
    ctx = xa.Context(0)
    frames, drift = k3_frames(torch, dev)
    Y, X = frames.shape[1:]
    t0 = time.perf_counter()
    og = o.fa_global_alignment(frames.cpu().numpy(), Ts=1.0, max_shift_px=50.0, max_res=30.0)
    print(f"oracle: {time.perf_counter() - t0:.1f} s; pair shifts bX {og['bX']}, bY {og['bY']}")
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 30.0)
    for name, opts in (("product", {}),
                       ("plain pair windows", {"pairwin_form": 0}),
                       ("full inverse transform per pair", {"window": 0}),
                       ("column pass as full lines", {"pruned_columns": 0}),
                       ("column pass on the matrix cores", {"pruned_columns": 2}),
                       ("row pass as three kernels", {"rows_kept": 0}),
                       ("everything as in round 4", {"pruned_columns": 0, "pairwin_form": 0, "rows_kept": 0}),
                       ("round 4 + full inverse", {"pruned_columns": 0, "pairwin_form": 0, "rows_kept": 0, "window": 0})):
        for k_, v_ in (("pruned_columns", 1), ("pairwin_form", 1), ("rows_kept", 1), ("window", 1)):
            fa.set_option(k_, v_)
        for k_, v_ in opts.items():
            fa.set_option(k_, v_)
        dg = fa.global_alignment(frames, 50.0)
        e = max(np.abs(dg["bX"] - og["bX"]).max(), np.abs(dg["bY"] - og["bY"]).max())
        print(f"{name:40s} max |pair shift - oracle| = {e:.2e} px")


if __name__ == "__main__":
    main()
