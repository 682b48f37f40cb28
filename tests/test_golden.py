"""Committed regression vectors (tests/golden/, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them; GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_golden_pm(oracle):
    g = np.load(os.path.join(G, "pm_d32.npz"))
    pm = oracle.PM(g["refs"])
    refno, psi, flip, _ = pm.match(g["particles"])
    assert np.array_equal(refno[:, 0], g["refno"]) and np.array_equal(psi[:, 0], g["psi_idx"])
    assert np.array_equal(flip[:, 0], g["flip"])
    sx, sy, cc = pm.translate(g["particles"], g["refno"], g["psi_idx"], g["flip"])
    assert np.allclose(sx, g["shiftX"], atol=1e-9) and np.allclose(cc, g["maxCC"], atol=1e-9)


def test_oracle_reproduces_golden_rf(oracle):
    g = np.load(os.path.join(G, "rf_d32.npz"))
    rf = oracle.RF(32)
    for img, a in zip(g["particles"], g["angles"]):
        rf.insert(rf.prepare_image(img), synth.euler_matrix(*a).T)
    rf.mirror_and_crop()
    assert np.abs(rf.finish() - g["volume"]).max() <= 2e-6 * np.abs(g["volume"]).max()


@pytest.mark.gpu
def test_hip_reproduces_golden():
    import torch
    import xmipp3_amd as xa
    ctx = xa.Context(0)
    g = np.load(os.path.join(G, "pm_d32.npz"))
    pm = xa.ProjectionMatcher(ctx, torch.from_numpy(g["refs"]).cuda())
    p = torch.from_numpy(g["particles"]).cuda()
    refno, psi, flip = pm.match(p)
    assert np.array_equal(refno.cpu().numpy(), g["refno"])
    assert np.array_equal(psi.cpu().numpy(), g["psi_idx"])
    assert np.array_equal(flip.cpu().numpy(), g["flip"])
    sx, sy, cc = pm.translate(p, refno, psi, flip)
    assert np.abs(sx.cpu().numpy() - g["shiftX"]).max() < 1e-3 and np.abs(cc.cpu().numpy() - g["maxCC"]).max() < 1e-5
    h = np.load(os.path.join(G, "rf_d32.npz"))
    rf = xa.RecFourier(ctx, 32)
    rf.insert(rf.prepare_images(torch.from_numpy(h["particles"]).cuda()), h["angles"])
    rf.mirror_and_crop()
    assert np.abs(rf.finish() - h["volume"]).max() <= 1e-4 * np.abs(h["volume"]).max()
