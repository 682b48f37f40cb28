#!/usr/bin/env python3
"""Second step of tools/diag_iterative.py: the shift -> rotation half, one round, by hand: where do device and oracle part ways?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import xmipp3_amd as xa
from oracle import pyoracle as o
from test_gpu_estimators import _es_population
for draw in (2, 6):
    D, sh, rot, ref, others = _es_population(o, draw, 100, False)
    others = others[:40]
    n = len(others)
    ctx = xa.Context(0)
    ms = min(20, D // 2 - 1)
    dref, doth = torch.from_numpy(ref).cuda(), torch.from_numpy(others).cuda()
    est = xa.ShiftCorrEstimator(ctx, D, D, ms); est.load_reference(dref)
    s_dev = est.compute_shifts(doth); s_or = o.es_shifts(ref, others, ms)
    print(f"draw {draw} D {D}: shifts equal {np.array_equal(s_dev, s_or)}")
    # pose = [[1,0,sx],[0,1,sy],[0,0,1]]; the transformer applies the inverse
    M = np.tile(np.eye(3, dtype=np.float32), (n, 1, 1)); M[:, 0, 2] = s_or[:, 0]; M[:, 1, 2] = s_or[:, 1]
    inv = np.stack([np.linalg.inv(m.astype(np.float64)).astype(np.float32) for m in M])
    t_dev = xa.apply_geometry2d(ctx, doth, inv).cpu().numpy()
    t_or = np.stack([o.apply_geometry2d(others[i].astype(np.float64), inv[i].astype(np.float64), 1, True, False) for i in range(n)]).astype(np.float32)
    print("   transformed images: max abs diff", float(np.abs(t_dev - t_or).max()), "of a range", float(np.abs(t_or).max()))
    r_dd = xa.rotation_estimate(ctx, dref, torch.from_numpy(t_dev).cuda())
    r_od, c_od = o.es_polar_rotation(ref, t_dev, with_corr=True)
    r_oo, c_oo = o.es_polar_rotation(ref, t_or, with_corr=True)
    print("   rotation: device(dev images) == oracle(dev images):", int((r_dd == r_od.astype(np.float32)).sum()), "of", n,
          "; oracle(dev images) == oracle(oracle images):", int((r_od == r_oo).sum()), "of", n)
    for i in np.nonzero(r_od != r_oo)[0]:
        a, b = c_od[i], c_oo[i]
        ia, ib = int(np.argmax(a)), int(np.argmax(b))
        print(f"      image {i}: arg-max {ia} vs {ib}; oracle-image row: value at {ib} = {b[ib]:.9g}, at {ia} = {b[ia]:.9g} (relative gap {(b[ib]-b[ia])/abs(b[ib]):.2e}); "
              f"device-image row: at {ia} = {a[ia]:.9g}, at {ib} = {a[ib]:.9g}")
