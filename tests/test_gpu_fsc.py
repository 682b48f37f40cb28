"""GPU parity of the Fourier shell correlation (xh_frc_dpr) against the CPU oracle,
through the C ABI, plus the reference's one known answer and size-independent properties at full size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import synth  # noqa: E402
from tests.test_oracle_pins import _frc_pin_volumes  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return xa, xa.Context(0), torch


def _close(got, exp, rtol, atol=0.0):
    g, e = np.asarray(got), np.asarray(exp)
    assert np.array_equal(np.isnan(g), np.isnan(e))
    m = ~np.isnan(e)
    assert np.all(np.abs(g[m] - e[m]) <= atol + rtol * np.abs(e[m])), (g, e)


def test_reference_rfactor_known_answer(gpu):
    """test_resolution_frc.cpp:120-130 through the device path."""
    xa, ctx, torch = gpu
    v1, v2 = _frc_pin_volumes()
    r = xa.frc_dpr(ctx, torch.from_numpy(v1).cuda(), torch.from_numpy(v2).cuda(), 2.0, do_rfactor=True, min_freq=2.0 / -1.0, max_freq=2.0 / 2.0)
    assert abs(r["rfactor"] - 0.134661) < 0.00001


@pytest.mark.parametrize("shape", [(3, 3, 3), (32, 32, 32), (40, 40, 40), (12, 20, 18), (1, 64, 64), (1, 45, 45), (27, 27, 27)])
def test_matches_oracle(gpu, oracle, shape):
    """fp64 both sides; shell membership is integer-exact, the sums differ by summation order (atomics) and by
    the FFT algorithm (radix-2 / Bluestein lines in LDS vs the oracle's), atan2 by device libm."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(sum(shape))
    a = rng.normal(size=shape) + 3.0
    b = a + 0.7 * rng.normal(size=shape)
    exp = oracle.frc_dpr(a, b, 1.7, do_dpr=True, do_rfactor=True, min_freq=0.05, max_freq=0.4)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    if shape[0] == 1:
        ta, tb = ta[0], tb[0]       # 2-D tensors are images
    got = xa.frc_dpr(ctx, ta, tb, 1.7, do_dpr=True, do_rfactor=True, min_freq=0.05, max_freq=0.4)
    assert np.array_equal(got["freq"], exp["freq"])
    assert np.array_equal(got["frc_noise"], exp["frc_noise"])          # shell counts: exact
    _close(got["frc"], exp["frc"], 1e-11, 1e-13)
    _close(got["error_l2"], exp["error_l2"], 1e-11)
    _close(got["dpr"], exp["dpr"], 1e-9, 1e-9)
    assert abs(got["rfactor"] - exp["rfactor"]) <= 1e-12
    # without the optional outputs nothing else changes
    plain = xa.frc_dpr(ctx, ta, tb, 1.7)
    _close(plain["frc"], got["frc"], 1e-13, 1e-15)
    assert plain["rfactor"] == -1.0 and not plain["dpr"].any()


def test_properties_at_full_size(gpu):
    """256^3 (BASELINE config 3/4 box): FSC(v, v) = 1, FSC(v, -v) = -1, symmetry in its arguments, invariance to a
    common scale, and independent noise stays within the 2/sqrt(n) band the program reports."""
    xa, ctx, torch = gpu
    D = 256
    g = torch.Generator(device="cuda").manual_seed(7)
    v = torch.randn((D, D, D), generator=g, device="cuda", dtype=torch.float64)
    w = torch.randn((D, D, D), generator=g, device="cuda", dtype=torch.float64)
    same = xa.frc_dpr(ctx, v, v, 1.0, do_dpr=True)
    assert np.abs(same["frc"] - 1.0).max() <= 1e-12 and np.abs(same["dpr"]).max() <= 1e-6 and np.abs(same["error_l2"]).max() == 0.0
    neg = xa.frc_dpr(ctx, v, -v, 1.0)
    assert np.abs(neg["frc"][1:] + 1.0).max() <= 1e-12
    ab, ba = xa.frc_dpr(ctx, v, v + w, 1.0), xa.frc_dpr(ctx, v + w, v, 1.0)
    assert np.abs(ab["frc"] - ba["frc"]).max() <= 1e-12
    assert np.abs(ab["frc"][8:] - 1 / np.sqrt(2)).max() < 0.05           # signal and noise of equal power
    sc = xa.frc_dpr(ctx, 3.0 * v, 3.0 * (v + w), 1.0)
    assert np.abs(sc["frc"] - ab["frc"]).max() <= 1e-12
    ind = xa.frc_dpr(ctx, v, w, 1.0)
    # shells with many coefficients: |frc| of independent noise is ~ 1/sqrt(n); the program's band is 2/sqrt(n)
    big = np.arange(len(ind["frc"])) >= 8
    assert np.mean(np.abs(ind["frc"][big]) < ind["frc_noise"][big]) > 0.9
    # shell counts add up to the half-complex coefficients inside the Nyquist sphere
    counts = (2.0 / same["frc_noise"]) ** 2
    k = np.fft.fftfreq(D)
    kx = np.abs(np.fft.fftfreq(D))[:D // 2 + 1]
    kx[-1] = 0.5
    R2 = k[:, None, None] ** 2 + k[None, :, None] ** 2 + kx[None, None, :] ** 2
    assert round(counts.sum()) == int((R2 <= 0.25).sum())


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    z = torch.zeros((4, 4, 1), device="cuda", dtype=torch.float64)
    with pytest.raises(xa.XhError):
        xa.frc_dpr(ctx, z, z, 1.0)                  # X = 1
    z = torch.zeros((2, 2, 1030), device="cuda", dtype=torch.float64)
    with pytest.raises(xa.XhError):
        xa.frc_dpr(ctx, z, z, 1.0)
    z = torch.zeros((8, 8, 8), device="cuda", dtype=torch.float64)
    with pytest.raises(xa.XhError):
        xa.frc_dpr(ctx, z, z, 0.0)
