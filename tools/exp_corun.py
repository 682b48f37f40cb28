#!/usr/bin/env python3
"""Do the matcher's kernels run BESIDE the gridding kernel when it leaves them room?  (round 6 experiment)
k_rf_grid at twelve waves per CU holds 154 of 160 KB of LDS and 3 x 168 registers per SIMD: nothing co-resides.  At eight waves
(xh_rf_set_option grid_waves 8) it leaves 42 KB and 176 registers per lane and SIMD.  The translational alignment (S6: k_pm_tr_build,
k_pm_s6f_*, k_pm_bestshift_coarse -- <= 100 registers, <= 24 KB) fits into that.  Measured here: the gridding of one batch alone, S6 of
one batch alone, and both at once on two streams, for grid_waves 12 and 8.
    python tools/exp_corun.py [--reps 3]"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--nrefs", type=int, default=1000)
    ap.add_argument("--s6-eps", type=float, default=-1, help=">= 0: xh_pm_set_option s6_eps (0: no fp64 repeats, whose kernels do not fit beside the gridding kernel)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    import xmipp3_amd as xa
    import bench
    from tests import synth
    from xmipp3_amd.api import ctf_params
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    D, B, nrefs = args.box, args.batch, args.nrefs
    ctx = xa.Context(0)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        ctx_rf = xa.Context(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    genr = torch.Generator(device=dev); genr.manual_seed(7)
    dirs = synth.fibonacci_directions(nrefs)
    fpj = xa.FourierProjector(ctx, bench.phantom_volume(torch, D, genr, dev), 2.0, 0.5, 3)
    refs = fpj.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1))
    fpj.close()
    refs = ((refs - refs.mean()) / refs.std()).contiguous()
    idx = torch.randint(0, nrefs, (B,), generator=gen, device=dev)
    parts = (refs[idx] + math.sqrt(10.0) * torch.randn((B, D, D), generator=gen, device=dev)).contiguous()
    pm = xa.ProjectionMatcher(ctx, refs)
    refno, psi, flip = pm.match(parts)
    if args.s6_eps >= 0:
        pm.set_option("s6_eps", args.s6_eps)
    rng = np.random.default_rng(1)
    ctf_arr = xa.RecFourier.ctf_param_array([ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d)) for d in rng.uniform(1e4, 3e4, B)])
    ang = torch.cat([torch.from_numpy(np.ascontiguousarray(dirs[:, :2])).to(dev)[refno.long()], (psi.double() * (360.0 / pm.N))[:, None]], 1).contiguous()
    torch.cuda.synchronize()

    def timed(fn_main=None, fn_side=None):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        if fn_side:
            with torch.cuda.stream(side):
                ev[2].record(side); fn_side(); ev[3].record(side)
        if fn_main:
            ev[0].record(); fn_main(); ev[1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        return wall, (ev[0].elapsed_time(ev[1]) if fn_main else 0.0), (ev[2].elapsed_time(ev[3]) if fn_side else 0.0)

    for waves in (12, 8):
        with torch.cuda.stream(side):
            rf = xa.RecFourier(ctx_rf, D, min_ctf=0.01, sampling=1.0)
            rf.set_option("grid_waves", waves)
        grid = lambda: rf.insert_images(parts, ang, ctf_array=ctf_arr)
        s6 = lambda: pm.translate(parts, refno, psi, flip)
        match = lambda: pm.match(parts)
        for f in (grid, s6, match):          # warm-up
            timed(f if f is not grid else None, f if f is grid else None)
        for rep in range(args.reps):
            g = timed(None, grid)
            s = timed(s6, None)
            m = timed(match, None)
            both = timed(s6, grid)
            bothm = timed(match, grid)
            both2 = timed(lambda: (s6(), s6(), s6(), s6()), grid)
            s4 = timed(lambda: (s6(), s6(), s6(), s6()), None)
            print(f"grid_waves {waves:2d}: insert_images alone {g[2]:6.2f} ms | S6 alone {s[1]:5.2f} | match alone {m[1]:5.2f} | together with S6: wall {both[0]:6.2f} (S6 {both[1]:5.2f}, insert {both[2]:6.2f}) | "
                  f"4 x S6 alone {s4[1]:6.2f}, beside insert: wall {both2[0]:6.2f} (S6s {both2[1]:6.2f}, insert {both2[2]:6.2f}) | with match: wall {bothm[0]:6.2f} (match {bothm[1]:6.2f}, insert {bothm[2]:6.2f})", flush=True)
        with torch.cuda.stream(side):
            rf.close() if hasattr(rf, "close") else None
            del rf
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
