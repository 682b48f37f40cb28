#!/usr/bin/env python3
"""Error of the fp32 coarse pass of the matcher against its fp64 evaluation, on the workload of bench.py (phantom gallery,
256 px, SNR 0.1): every angle of the correlation rows of 600 particles against their own and a random reference, normalised
by sigma_ref * sigma_img * S (S = sum_r 2 pi r, the largest value a normalised correlation can take). The ambiguity margin
tau_rel of xh_pm_match is chosen from this distribution. Run on the GPU box: python3 tools/measure_tau.py"""
import sys, math, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xmipp3_amd as xa, bench
from tests import synth
ctx = xa.Context(0); dev = torch.device("cuda", 0)
D, nrefs, B = 256, 1000, 600
genr = torch.Generator(device=dev); genr.manual_seed(7)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
dirs = synth.fibonacci_directions(nrefs)
fpj = xa.FourierProjector(ctx, bench.phantom_volume(torch, D, genr, dev), 2.0, 0.5, 3)
refs = fpj.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1)); fpj.close()
refs = ((refs - refs.mean()) / refs.std()).contiguous()
idx = torch.randint(0, nrefs, (B,), generator=gen, device=dev)
th = torch.rand((B,), generator=gen, device=dev) * (2 * math.pi)
rot = torch.zeros((B, 2, 3), device=dev)
rot[:, 0, 0] = torch.cos(th); rot[:, 0, 1] = -torch.sin(th); rot[:, 1, 0] = torch.sin(th); rot[:, 1, 1] = torch.cos(th)
grid = torch.nn.functional.affine_grid(rot, (B, 1, D, D), align_corners=False)
parts = torch.nn.functional.grid_sample(refs[idx][:, None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[:, 0]
parts = (parts + math.sqrt(10.0) * torch.randn((B, D, D), generator=gen, device=dev)).contiguous()
pm = xa.ProjectionMatcher(ctx, refs)
S = 2 * math.pi * sum(range(1, D // 2))
errs = []
rng = np.random.default_rng(0)
for i in range(B):
    for r in (int(idx[i]), int(rng.integers(nrefs))):
        a = pm.debug_corr_rows(parts[i], r, 32); b = pm.debug_corr_rows(parts[i], r, 64)
        errs.append((a - b) / S)          # debug_corr_rows returns rows already divided by sigma_ref sigma_img
e = np.concatenate(errs)
print("samples", e.size, "rms", np.sqrt((e**2).mean()), "max", np.abs(e).max(), "max/rms", np.abs(e).max()/np.sqrt((e**2).mean()))
for q in (0.999, 0.99999, 0.9999999): print(q, np.quantile(np.abs(e), q))
