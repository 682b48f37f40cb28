"""GPU parity tests of the CTF pre-steps (xmipp_ctf_phase_flip, xmipp_ctf_correct_wiener2d) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return xa, xa.Context(0), torch


def _ctf(mod, **kw):
    base = dict(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=18000.0, DeltafV=16500.0, azimuthal_angle=35.0)
    base.update(kw)
    return mod(**base)


@pytest.mark.parametrize("shape,kw", [((256, 256), {}), ((192, 320), {}), ((100, 90), {}), ((128, 128), dict(phase_shift=40.0, VPP_radius=0.01)),
                                      ((64, 64), dict(DeltafV=18000.0))])
def test_phase_flip_against_the_oracle(gpu, oracle, shape, kw):
    """actualPhaseFlip (ctf_phase_flip.cpp:88-117): square, non-square, non power-of-two (Bluestein lines), phase plate, round CTF.
    Transforms in fp32 on the device, double in the reference: 1e-5 of the largest value."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    rng = np.random.default_rng(shape[0] + shape[1])
    img = rng.standard_normal(shape).astype(np.float32)
    Tm = 1.4
    exp = oracle.ctf_phase_flip(img, _ctf(oracle.ctf_params, Tm=Tm, **kw))
    op = xa.CtfOps(ctx, *shape)
    got = op.phase_flip(torch.from_numpy(img.copy()).cuda(), _ctf(ctf_params, **kw), Tm).cpu().numpy()
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    assert np.abs(got - img).max() > 0.1            # something was flipped
    if kw.get("DeltafV") == 18000.0:
        # a round CTF has the same sign at (fy, +0.5) and (-fy, +0.5): the flipped half spectrum stays Hermitian on the Nyquist
        # column and flipping twice is the identity (an astigmatic one loses what c2r drops there, in the reference too)
        twice = op.phase_flip(torch.from_numpy(got.copy()).cuda(), _ctf(ctf_params, **kw), Tm).cpu().numpy()
        assert np.abs(twice - img).max() <= 2e-5 * np.abs(img).max()


def test_phase_flip_of_a_delta_is_the_reference_units_pin(gpu):
    """test_ctf_main.cpp:126-149: the phase-flipped delta has standard deviation 0.003906 and maximum 0.017565 (1e-4)."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    d = torch.zeros((256, 256), device="cuda")
    d[128, 128] = 1.0
    op = xa.CtfOps(ctx, 256, 256)
    f = op.phase_flip(d, ctf_params(kV=300.0, DeltafU=20000.0, DeltafV=20000.0, Cs=2.0, Q0=0.1, K=1.0), 1.0).cpu().numpy().astype(np.float64)
    assert abs(f.std() - 0.003906) < 1e-4 and abs(f.max() - 0.017565) < 1e-4


def test_phase_flip_of_a_movie_frame_sized_micrograph(gpu):
    """4096 x 5760 (a K3 frame): lines beyond one LDS transform. Size-independent properties: the power of every Fourier
    coefficient is unchanged (Parseval), flipping twice is the identity."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    g = torch.Generator(device="cuda").manual_seed(2)
    img = torch.randn((4096, 5760), generator=g, device="cuda")
    ref = img.clone()
    op = xa.CtfOps(ctx, 4096, 5760)
    c = _ctf(ctf_params, DeltafV=18000.0)        # round: see test_phase_flip_against_the_oracle
    op.phase_flip(img, c, 0.83)
    assert abs(img.double().pow(2).sum().item() / ref.double().pow(2).sum().item() - 1.0) < 1e-5
    assert (img - ref).abs().max().item() > 0.1
    op.phase_flip(img, c, 0.83)
    assert (img - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()


@pytest.mark.parametrize("kw", [dict(), dict(phase_flipped=True), dict(wiener_constant=0.3), dict(correct_envelope=True, pad=1.5),
                                dict(is_isotropic=True), dict(pad=1.0)])
def test_wiener2d_against_the_oracle(gpu, oracle, kw):
    """Wiener2D::applyWienerFilter (wiener2d.cpp:101-141) on a batch with one CTF per image; every option of the program."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    D, n = 64, 5
    pad = kw.pop("pad", 2.0)
    rng = np.random.default_rng(7)
    imgs = rng.standard_normal((n, D, D)).astype(np.float32)
    env = dict(Ca=2.0, espr=0.6, ispr=0.3, alpha=0.1, DeltaF=3.0, DeltaR=0.5) if kw.get("correct_envelope") else {}
    defoci = rng.uniform(8000.0, 25000.0, n)
    exp = np.stack([oracle.ctf_wiener2d(imgs[i], _ctf(oracle.ctf_params, DeltafU=defoci[i], DeltafV=defoci[i] + 700.0, **env),
                                        sampling_rate=1.3, pad=pad, **kw) for i in range(n)])
    op = xa.CtfOps(ctx, D, D, pad=pad)
    got = op.wiener2d(torch.from_numpy(imgs.copy()).cuda(), [_ctf(ctf_params, DeltafU=defoci[i], DeltafV=defoci[i] + 700.0, **env) for i in range(n)],
                      sampling_rate=1.3, **kw).cpu().numpy()
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    assert np.abs(got - imgs).max() > 0.1


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    with pytest.raises(xa.XhError):
        xa.CtfOps(ctx, 1, 64)
    op = xa.CtfOps(ctx, 32, 32, pad=2.0)
    with pytest.raises(xa.XhError):
        op.phase_flip(torch.zeros((32, 32), device="cuda"), ctf_params(), 1.0)       # a padded handle does not flip
