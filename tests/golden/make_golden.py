"""Regenerates tests/golden/*.npz from the CPU oracle.

These are REGRESSION vectors of our own oracle (the reference cannot be run here: xmippCore and
FFTW are absent); the reference-derived known answers live in
tests/test_oracle_pins.py.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o  # noqa: E402
from tests import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    D, nrefs, n = 32, 16, 12
    vol = synth.phantom(D, seed=5, nblobs=10)
    refs, dirs = synth.make_refs(vol, nrefs)
    rng = np.random.default_rng(6)
    parts, _ = synth.make_particles(refs, n, rng, snr=0.3, max_shift=2)
    pm = o.PM(refs)
    refno, psi, flip, cc = pm.match(parts, parity=0)
    sx, sy, mcc = pm.translate(parts, refno[:, 0], psi[:, 0], flip[:, 0])
    np.savez_compressed(os.path.join(HERE, "pm_d32.npz"), refs=refs, particles=parts, refno=refno[:, 0],
                        psi_idx=psi[:, 0], flip=flip[:, 0], shiftX=sx, shiftY=sy, maxCC=mcc, N=pm.N)
    ang = synth.random_angles(n, np.random.default_rng(7))
    rf = o.RF(D)
    for i in range(n):
        rf.insert(rf.prepare_image(parts[i]), synth.euler_matrix(*ang[i]).T)
    rf.mirror_and_crop()
    volume = rf.finish()
    np.savez_compressed(os.path.join(HERE, "rf_d32.npz"), particles=parts, angles=ang,
                        volume=volume.astype(np.float32))
    print("wrote pm_d32.npz, rf_d32.npz")


if __name__ == "__main__":
    main()
