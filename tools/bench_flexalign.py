"""FlexAlign on one movie of BASELINE config 5's size (40 frames of 4092 x 5760, K3), frames resident on the device:
global alignment, local alignment (program defaults: 500 A patches, 6 x 6 x 5 control points, 3 frames per patch), warp + sum.
python tools/bench_flexalign.py [--frames 40 --y 4092 --x 5760 --reps 3]"""
import argparse
import json
import time

import numpy as np
import torch

import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import xmipp3_amd as xa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--y", type=int, default=4092)
    ap.add_argument("--x", type=int, default=5760)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sampling", type=float, default=1.0)
    ap.add_argument("--max-res", type=float, default=30.0)
    a = ap.parse_args()
    N, Y, X = a.frames, a.y, a.x
    ctx = xa.Context(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn((Y + 128, X + 128), generator=g, device="cuda")
    k = torch.fft.rfft2(base)
    fy = torch.fft.fftfreq(Y + 128, device="cuda")[:, None]
    fx = torch.fft.rfftfreq(X + 128, device="cuda")[None, :]
    base = torch.fft.irfft2(k * torch.exp(-2 * (np.pi * 4.0) ** 2 * (fx * fx + fy * fy)), s=base.shape) * 30
    rng = np.random.default_rng(3)
    drift = np.clip(np.cumsum(rng.integers(-2, 3, (N, 2)), 0), -30, 30)
    drift -= drift[0]
    frames = torch.stack([base[64 + drift[i, 1]:64 + drift[i, 1] + Y, 64 + drift[i, 0]:64 + drift[i, 0] + X] for i in range(N)])
    for i in range(N):
        frames[i] += 0.5 * torch.randn((Y, X), generator=g, device="cuda")
    fa = xa.FlexAlign(ctx, Y, X, a.sampling, a.max_res)
    req = int(500 / a.sampling)
    patches = (int(np.ceil(X / req)), int(np.ceil(Y / req)))
    cp = (6, 6, 5)
    max_shift = 50.0 / a.sampling
    total = torch.zeros((Y, X), device="cuda")
    t = {"global_s": [], "local_s": [], "warp_sum_s": []}
    for rep in range(a.reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gl = fa.global_alignment(frames, max_shift)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loc = fa.local_alignment(frames, gl["shiftX"], gl["shiftY"], gl["ref"], max_shift, patches, (req, req), 3, cp)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        total.zero_()
        for n in range(N):
            fa.apply_bspline(frames[n], loc["coeffsX"], loc["coeffsY"], cp, N, n, total=total)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if rep:
            t["global_s"].append(t1 - t0); t["local_s"].append(t2 - t1); t["warp_sum_s"].append(t3 - t2)
    tt = drift - drift[gl["ref"]]
    out = {k: float(np.median(v)) for k, v in t.items()}
    out["movie_s"] = sum(out.values())
    out.update({"frames": N, "Y": Y, "X": X, "patches": patches, "dims": loc["dims"], "new_dims": fa.new_dims,
                "global_shift_error_px": float(max(np.abs(gl["shiftX"] + tt[:, 0]).max(), np.abs(gl["shiftY"] + tt[:, 1]).max())),
                "movies_per_s": 1.0 / sum(out[k] for k in ("global_s", "local_s", "warp_sum_s"))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
