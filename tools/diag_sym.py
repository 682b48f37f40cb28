"""diagnostic: device vs oracle temp spaces with improper symmetry matrices, one at a time"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xmipp3_amd.api as xa
from oracle import pyoracle as oracle
from tests import synth
D = 32
vol = synth.phantom(D, seed=1, nblobs=12)
rng = np.random.default_rng(2)
ang = synth.random_angles(40, rng)
imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
ctx = xa.Context(0)
n_ = np.array([1.0, 2.0, -0.5]) / np.linalg.norm([1.0, 2.0, -0.5])
mats = {"I": np.eye(3), "My": np.diag([1.0, -1.0, 1.0]), "inv": -np.eye(3), "rotrefl": (np.eye(3) - 2 * np.outer(n_, n_)) @ synth.euler_matrix(30, 0, 0),
        "c2": np.diag([-1.0, -1.0, 1.0]), "rot": synth.euler_matrix(30, 40, 50)}
for names in (["I"], ["My"], ["inv"], ["rotrefl"], ["rot"], ["I", "c2"], ["I", "My"], ["I", "inv"], ["I", "My", "inv", "rotrefl"], ["I", "c2", "rot", "My"]):
    sym = np.stack([mats[k] for k in names])
    rf = xa.RecFourier(ctx, D)
    o = oracle.RF(D)
    ffts = np.stack([o.prepare_image(im) for im in imgs])
    for i in range(len(imgs)):
        for R in sym:
            o.insert(ffts[i], synth.euler_matrix(*ang[i]).T, R=R)
    rf.insert(torch.from_numpy(ffts).cuda(), ang, sym=sym)
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
    dw = np.abs(gw - ew)
    k = np.unravel_index(dw.argmax(), dw.shape)
    print(names, "sets equal", bool(((ew != 0) == (gw != 0)).all()), "w err", dw.max() / np.abs(ew).max(), "at", k, gw[k], ew[k], "v err", np.abs(gv - ev).max() / np.abs(ev).max(),
          "n>2e-6", int((dw > 2e-6 * np.abs(ew).max()).sum()))
