"""GPU parity tests of FlexAlign's global alignment (xh_fa_*) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return xa, xa.Context(0), torch


def synthetic_movie(N, Y, X, seed, max_step=1.5, noise=0.5, smooth=3.0):
    from tests import synth
    frames, drift, _ = synth.movie(N, Y, X, seed, max_step, noise, smooth)
    return frames, drift


@pytest.mark.parametrize("N,Y,X,ts,res", [(8, 512, 512, 1.0, 8.0), (6, 240, 320, 1.0, 8.0), (5, 384, 300, 1.4, 12.0)])
def test_global_alignment_against_the_oracle(gpu, oracle, N, Y, X, ts, res):
    """Pair shifts and frame shifts of the device against ProgMovieAlignmentCorrelation<double>'s arithmetic (oracle): square
    power-of-two frames, non-square ones whose reduced size is not a power of two (Bluestein lines), another sampling rate.
    fp32 transforms against double: 2e-3 px on every pair, same reference frame."""
    xa, ctx, torch = gpu
    frames, drift = synthetic_movie(N, Y, X, seed=N + Y)
    rng = np.random.default_rng(1)
    dark = (0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    gain = (1.0 + 0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    max_shift = 20.0
    exp = oracle.fa_global_alignment(frames, Ts=ts, max_shift_px=max_shift, max_res=res, dark=dark, igain=gain)
    fa = xa.FlexAlign(ctx, Y, X, ts, res)
    assert fa.new_dims == exp["new_dims"]
    got = fa.global_alignment(torch.from_numpy(frames).cuda(), max_shift, torch.from_numpy(dark).cuda(), torch.from_numpy(gain).cuda())
    assert np.abs(got["bX"] - exp["bX"]).max() <= 2e-3 and np.abs(got["bY"] - exp["bY"]).max() <= 2e-3
    assert got["ref"] == exp["ref"]
    assert np.abs(got["shiftX"] - exp["shiftX"]).max() <= 2e-3 and np.abs(got["shiftY"] - exp["shiftY"]).max() <= 2e-3
    # and the drift that was put in comes out (the stored shift is the negative of the frame's displacement)
    t = drift - drift[exp["ref"]]
    assert np.abs(got["shiftX"] + t[:, 0]).max() < 0.6 and np.abs(got["shiftY"] + t[:, 1]).max() < 0.6


def test_global_alignment_of_k3_sized_frames(gpu):
    """BASELINE config 5 movie (40 frames of 4096 x 5760, the K3 sensor rotated as the FFT test has it; 780 frame pairs): no oracle at this size
    (a double-precision CPU transform of 23.6 Mpixel frames takes minutes); the known drift of the synthetic movie comes out to
    1.5 px -- the frames are correlated at 4.4 px per reduced pixel (30 A at 1 A/px) and bestShift is a centre of mass --."""
    xa, ctx, torch = gpu
    N, Y, X = 40, 4096, 5760
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
    for i in range(N):          # noise frame by frame: a second 3.8 GB tensor is not needed
        frames[i] += 0.5 * torch.randn((Y, X), generator=g, device="cuda")
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 30.0)
    assert fa.new_dims == (int(Y * fa.size_factor), int(X * fa.size_factor))
    got = fa.global_alignment(frames, 40.0)
    t = drift - drift[got["ref"]]
    assert np.abs(got["shiftX"] + t[:, 0]).max() < 1.5 and np.abs(got["shiftY"] + t[:, 1]).max() < 1.5
    assert np.corrcoef(got["shiftX"], -t[:, 0])[0, 1] > 0.95 and np.corrcoef(got["shiftY"], -t[:, 1])[0, 1] > 0.95


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    with pytest.raises(xa.XhError):
        xa.FlexAlign(ctx, 256, 256, 4.0, 8.0)            # scale factor >= 1 (checkSettings)
    fa = xa.FlexAlign(ctx, 256, 256, 1.0, 8.0)
    with pytest.raises(xa.XhError):
        fa.global_alignment(torch.zeros((3, 256, 256), device="cuda"), 500.0)       # --maxShift beyond the reduced frame
