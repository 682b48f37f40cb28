"""GPU parity of the FourierProjector (central-slice gallery generation) against the
CPU oracle, through the C ABI. fp64 on the device like the reference; the images leave as float."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import synth  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return xa, xa.Context(0), torch


@pytest.mark.parametrize("D,padding,maxf", [(32, 2.0, 0.5), (32, 1.0, 0.25), (25, 2.0, 0.3)])
def test_coefficients_and_projections_match_the_oracle(gpu, oracle, D, padding, maxf):
    xa, ctx, torch = gpu
    vol = synth.phantom(D, seed=11, nblobs=9).astype(np.float32)
    o = oracle.FP(vol, padding, maxf, 3)
    fp = xa.FourierProjector(ctx, torch.from_numpy(vol).cuda(), padding, maxf, 3)
    assert (fp.P, fp.cdim, fp.cstart) == (o.P, o.cdim, o.cstart)
    gre, gim = fp.coefs()
    ere, eim = o.coefs()
    scale = max(np.abs(ere).max(), np.abs(eim).max())
    # fp64 both sides; the 3-D FFT algorithms differ (radix-2 / Bluestein in LDS vs mixed radix)
    assert np.abs(gre - ere).max() <= 1e-11 * scale and np.abs(gim - eim).max() <= 1e-11 * scale
    ang = np.concatenate([[[0, 0, 0], [90, 90, 0], [10, 170, 33]], synth.random_angles(6, np.random.default_rng(D))])
    got = fp.project(ang).cpu().numpy()
    for a, g in zip(ang, got):
        exp = o.project(*a)
        assert np.abs(g - exp).max() <= 3e-7 * np.abs(exp).max()        # float32 output


def test_ctf_multiplier_and_batching(gpu, oracle):
    xa, ctx, torch = gpu
    D = 32
    vol = synth.phantom(D, seed=4, nblobs=6).astype(np.float32)
    o = oracle.FP(vol, 2.0, 0.5, 3)
    fp = xa.FourierProjector(ctx, torch.from_numpy(vol).cuda(), 2.0, 0.5, 3)
    fy = np.fft.fftfreq(D)[:, None]
    fx = np.fft.rfftfreq(D)[None, :]
    ctf = np.cos(30.0 * (fx * fx + fy * fy)) * np.exp(-4.0 * (fx * fx + fy * fy))
    ang = synth.random_angles(40, np.random.default_rng(1))
    got = fp.project(ang, ctf=torch.from_numpy(ctf).cuda()).cpu().numpy()
    for i in (0, 17, 39):
        exp = o.project(*ang[i], ctf=ctf)
        assert np.abs(got[i] - exp).max() <= 3e-7 * np.abs(exp).max()


def test_gallery_feeds_projection_matching(gpu, oracle):
    """volume -> gallery (device) -> matcher (device): a projection at a gallery direction, rotated in
    plane by a multiple of the angular step, is assigned to that reference with that in-plane angle."""
    xa, ctx, torch = gpu
    D, nrefs = 64, 60
    vol = synth.phantom(D, seed=2, nblobs=14).astype(np.float32)
    dirs = synth.fibonacci_directions(nrefs)
    fp = xa.FourierProjector(ctx, torch.from_numpy(vol).cuda(), 2.0, 0.5, 3)
    gallery = fp.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1))
    pm = xa.ProjectionMatcher(ctx, gallery.contiguous())
    step = 360.0 / pm.N
    picks = [(7, 25), (31, 100), (55, 3)]
    parts = fp.project(np.array([[dirs[r, 0], dirs[r, 1], k * step] for r, k in picks]))
    refno, psi, flip = pm.match(parts.contiguous())
    assert refno.cpu().tolist() == [r for r, _ in picks]
    assert flip.cpu().tolist() == [0, 0, 0]
    got_psi = psi.cpu().numpy()
    for (r, k), g in zip(picks, got_psi):
        assert min((g - k) % pm.N, (k - g) % pm.N) <= 1 or min((g + k) % pm.N, (-g - k) % pm.N) <= 1


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    vol = torch.zeros((16, 16, 16), device="cuda")
    with pytest.raises(xa.XhError):
        xa.FourierProjector(ctx, vol, 2.0, 0.5, 1)       # linear interpolation is not on the device
    with pytest.raises(xa.XhError):
        xa.FourierProjector(ctx, vol, 0.5, 0.5, 3)
