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


@pytest.mark.parametrize("N,Y,X,ts,res", [(8, 512, 512, 1.0, 8.0), (6, 240, 320, 1.0, 8.0), (5, 384, 300, 1.4, 12.0), (5, 241, 321, 1.0, 8.0), (4, 180, 720, 1.0, 8.0)])
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
    # the pair correlations were taken inside the search window (pruned transforms); every pair through the full inverse transform
    # gives the same shifts
    assert fa.last_full_pairs() < len(exp["bX"]) // 4
    fa.set_option("window", 0)
    full = fa.global_alignment(torch.from_numpy(frames).cuda(), max_shift, torch.from_numpy(dark).cuda(), torch.from_numpy(gain).cuda())
    assert fa.last_full_pairs() == len(exp["bX"])
    assert np.abs(full["bX"] - got["bX"]).max() <= 1e-3 and np.abs(full["bY"] - got["bY"]).max() <= 1e-3 and full["ref"] == got["ref"]
    # and the drift that was put in comes out (the stored shift is the negative of the frame's displacement)
    t = drift - drift[exp["ref"]]
    assert np.abs(got["shiftX"] + t[:, 0]).max() < 0.6 and np.abs(got["shiftY"] + t[:, 1]).max() < 0.6


@pytest.mark.parametrize("N,Y,X,ts,res", [(4, 1240, 640, 1.0, 30.0), (4, 992, 700, 1.0, 24.0), (5, 512, 512, 1.0, 8.0)])
def test_pruned_column_pass_and_packed_pair_windows_change_nothing(gpu, oracle, N, Y, X, ts, res):
    """The column pass of the frame transform as two steps that compute the kept rows only (as matrix products: 1240 = 124 x 10 keeps 4 of 10
    second-step frequencies, 992 = 124 x 8 keeps 4 of 8, 512 = 128 x 4 keeps all; as small DFTs: 1240 = 20 x 62 keeps 15 of 62, 992 = 16 x 62
    keeps 18, 512 = 32 x 16 keeps 14) and the pair windows with packed multiply-adds, against
    the full-length line transforms + the plain pair-window kernel (5e-4 px: two fp32 summation orders) and against the oracle."""
    xa, ctx, torch = gpu
    frames, drift = synthetic_movie(N, Y, X, seed=N + Y, smooth=6.0)
    max_shift = 30.0
    d = torch.from_numpy(frames).cuda()
    fa = xa.FlexAlign(ctx, Y, X, ts, res)
    got = fa.global_alignment(d, max_shift)
    fa.set_option("pruned_columns", 0)
    fa.set_option("pairwin_form", 0)
    old = fa.global_alignment(d, max_shift)
    assert np.abs(old["bX"] - got["bX"]).max() <= 5e-4 and np.abs(old["bY"] - got["bY"]).max() <= 5e-4 and old["ref"] == got["ref"]
    for form in (1, 2):          # 1: both steps as small DFTs on the vector ALUs (the default), 2: as products on the matrix cores
        fa.set_option("pruned_columns", form)
        mid = fa.global_alignment(d, max_shift)
        assert np.abs(mid["bX"] - got["bX"]).max() <= 5e-4 and np.abs(mid["bY"] - got["bY"]).max() <= 5e-4
    exp = oracle.fa_global_alignment(frames, Ts=ts, max_shift_px=max_shift, max_res=res)
    assert np.abs(got["bX"] - exp["bX"]).max() <= 2e-3 and np.abs(got["bY"] - exp["bY"]).max() <= 2e-3 and got["ref"] == exp["ref"]


@pytest.mark.parametrize("N,Y,X", [(3, 248, 5760), (3, 249, 5760)])
def test_row_pass_that_writes_the_kept_columns_only(gpu, oracle, N, Y, X):
    """Frames as wide as a K3's (5760 = 45 x 128 points per row) take one kernel for the whole row pass (45-point transforms in registers,
    the 128-point ones for the kept frequencies only, the two rows of a complex row apart): against the three-kernel path it replaces
    (5e-4 px) and against the oracle; an odd number of rows leaves the last complex row half empty."""
    xa, ctx, torch = gpu
    frames, drift = synthetic_movie(N, Y, X, seed=N + Y, smooth=6.0)
    max_shift, res = 30.0, 30.0
    d = torch.from_numpy(frames).cuda()
    fa = xa.FlexAlign(ctx, Y, X, 1.0, res)
    got = fa.global_alignment(d, max_shift)
    fa.set_option("rows_kept", 0)
    old = fa.global_alignment(d, max_shift)
    assert np.abs(old["bX"] - got["bX"]).max() <= 5e-4 and np.abs(old["bY"] - got["bY"]).max() <= 5e-4 and old["ref"] == got["ref"]
    exp = oracle.fa_global_alignment(frames, Ts=1.0, max_shift_px=max_shift, max_res=res)
    assert np.abs(got["bX"] - exp["bX"]).max() <= 2e-3 and np.abs(got["bY"] - exp["bY"]).max() <= 2e-3 and got["ref"] == exp["ref"]


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


def test_global_alignment_of_k3_sized_frames_against_the_oracle(gpu, oracle):
    """The hole round 5 fell through: at K3 size the pruned transforms and the packed pair windows sum 926 and 132 terms per output in
    fp32, and the pair shifts drifted from 1.5e-6 to 7.9e-4 px of the oracle without any test noticing (every oracle test ran on frames
    of at most 1240 rows at 2e-3 px).  What the digits were lost to was not the length of the sums but the map's mean: its (0, 0)
    coefficient, orders of magnitude above all others, was summed into every window element and subtracted again by bestShift.  It now
    stays out (k_fa_pairwin_a / a2, k_fa_pair).  Three 4092 x 5760 frames of the bench's kind (int8 counts of a smooth field under drift
    and a growing dilation; tools/diag_fa_precision.py makes them) against ProgMovieAlignmentCorrelation<double>'s arithmetic: 2e-5 px
    (measured: 2.6e-6 through the windows, 8e-8 with the full inverse transform per pair)."""
    xa, ctx, torch = gpu
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from diag_fa_precision import k3_frames
    frames, drift = k3_frames(torch, torch.device("cuda", 0))
    Y, X = frames.shape[1:]
    exp = oracle.fa_global_alignment(frames.cpu().numpy(), Ts=1.0, max_shift_px=50.0, max_res=30.0)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 30.0)
    assert fa.new_dims == exp["new_dims"]
    got = fa.global_alignment(frames, 50.0)
    assert got["ref"] == exp["ref"]
    assert np.abs(got["bX"] - exp["bX"]).max() <= 2e-5 and np.abs(got["bY"] - exp["bY"]).max() <= 2e-5
    assert np.abs(got["shiftX"] - exp["shiftX"]).max() <= 2e-5 and np.abs(got["shiftY"] - exp["shiftY"]).max() <= 2e-5
    # every pair through the full inverse transform: 2e-6 of the oracle
    fa.set_option("window", 0)
    full = fa.global_alignment(frames, 50.0)
    assert np.abs(full["bX"] - exp["bX"]).max() <= 2e-6 and np.abs(full["bY"] - exp["bY"]).max() <= 2e-6


def test_local_alignment_of_a_k3_movie(gpu, oracle):
    """BASELINE config 5 on the path it names: 40 frames of 4092 x 5760 (K3), the program's defaults -- 12 x 9 patches of 500 px
    (ceil(size / 500), movie_alignment_correlation_base.cpp:218-224), 3 frames per patch, 6 x 6 x 5 control points, 30 A at 1 A/px --
    over a synthetic movie whose drift differs over the field (a dilation plus a shear growing with time, 6 px at the corners of
    the last frame, on top of a global random walk).  computeLocalAlignment (movie_alignment_correlation_gpu.cpp:288-430):
    * the patch layout equals the oracle's for all 108 patches, the patch shifts of a subset (corners, edges, centre: 10 patches x
      40 frames x 780 pairs each through the oracle's double transforms) agree to 1e-2 px;
    * the field comes back as far as the algorithm sees it: the fitted spline at every patch centre and frame against the
      displacement that was put in, relative to the reference frame's.  Measured (and the device equals the oracle to 1e-6 px, so
      this is the algorithm, not the device): rms 0.77 px, 2.7 px at worst, against 1.44 px rms when only the global shifts are
      applied; the local part is seen with a slope below one -- the patches are correlated at 4.4 px per reduced pixel and
      bestShift is a centre of mass over a window of that grid -- (0.64 here: frames are also averaged in threes, which blurs the fast first frames) and correlates with the truth (0.86)."""
    xa, ctx, torch = gpu
    N, Y, X = 40, 4092, 5760
    g = torch.Generator(device="cuda").manual_seed(7)
    base = torch.randn((Y + 128, X + 128), generator=g, device="cuda")
    k = torch.fft.rfft2(base)
    fy = torch.fft.fftfreq(Y + 128, device="cuda")[:, None]
    fx = torch.fft.rfftfreq(X + 128, device="cuda")[None, :]
    base = torch.fft.irfft2(k * torch.exp(-2 * (np.pi * 4.0) ** 2 * (fx * fx + fy * fy)), s=base.shape) * 30
    del k
    # a smooth global drift, as beam-induced motion is (fast at first, then settling): the spline has five control points in time
    # (two intervals over the 40 frames) and cannot follow a frame-to-frame random walk -- nor is it meant to
    t = np.arange(N, dtype=np.float64)
    drift = np.stack([14.0 * (1 - np.exp(-t / 9.0)) + 0.04 * t, -9.0 * (1 - np.exp(-t / 14.0)) + 0.06 * t], 1)
    local = 6.0

    def field(n, x, y):
        u, v = (np.asarray(x, float) / X - 0.5) * 2, (np.asarray(y, float) / Y - 0.5) * 2
        a = local * n / (N - 1)
        return drift[n, 0] + a * (0.7 * u + 0.3 * v), drift[n, 1] + a * (0.8 * v - 0.2 * u)

    H, W = base.shape
    ys = torch.arange(Y, device="cuda", dtype=torch.float32)[:, None]
    xs = torch.arange(X, device="cuda", dtype=torch.float32)[None, :]
    u, v = (xs / X - 0.5) * 2, (ys / Y - 0.5) * 2
    frames = torch.empty((N, Y, X), device="cuda")
    for n in range(N):
        a = local * n / (N - 1)
        sx = xs + 64 + float(drift[n, 0]) + a * (0.7 * u + 0.3 * v)
        sy = ys + 64 + float(drift[n, 1]) + a * (0.8 * v - 0.2 * u)
        grid = torch.stack(((sx + 0.5) * (2.0 / W) - 1, (sy + 0.5) * (2.0 / H) - 1), -1)[None]
        frames[n] = torch.nn.functional.grid_sample(base[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
        frames[n] += 0.5 * torch.randn((Y, X), generator=g, device="cuda")
        del grid, sx, sy
    max_shift, res = 50.0, 30.0
    req = 500
    patches, psize, cp = (int(np.ceil(X / req)), int(np.ceil(Y / req))), (req, req), (6, 6, 5)
    assert patches == (12, 9)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, res)
    gl = fa.global_alignment(frames, max_shift)
    loc = fa.local_alignment(frames, gl["shiftX"], gl["shiftY"], gl["ref"], max_shift, patches, psize, 3, cp)
    # --- a subset of the patches against the oracle (same global shifts, same frames)
    mask = np.zeros((patches[1], patches[0]), np.uint8)
    for (py, px) in ((0, 0), (0, 11), (8, 0), (8, 11), (4, 5), (4, 6), (0, 6), (8, 5), (3, 0), (5, 11)):
        mask[py, px] = 1
    eshifts, ecenters, edims = oracle.fa_local_patch_shifts(frames.cpu().numpy(), gl["shiftX"], gl["shiftY"], gl["ref"], mask, max_shift_px=max_shift, max_res=res,
                                                           patches=patches, patch_size=psize, patches_avg=3)
    assert loc["dims"] == edims and np.array_equal(loc["centers"], ecenters)
    sel = mask.astype(bool)
    d = np.abs(loc["patch_shifts"][sel] - eshifts[sel])
    print("K3 patch shifts against the oracle: max difference", d.max())
    assert np.isfinite(eshifts[sel]).all() and d.max() <= 1e-2
    # --- the field: the spline compensates the displacement of frame n relative to the reference frame's
    r = gl["ref"]
    errs, gerrs, perrs = [], [], []
    for py in range(patches[1]):
        for px in range(patches[0]):
            cx, cy = loc["centers"][py, px]
            for n in range(N):
                bx, by = oracle.fa_bspline_shift(loc["coeffsX"], loc["coeffsY"], cp, X, Y, N, int(cx), int(cy), n)
                fx_, fy_ = field(n, cx, cy)
                rx, ry = field(r, cx, cy)
                errs.append((bx - (fx_ - rx), by - (fy_ - ry)))
                gerrs.append((-gl["shiftX"][n] - (fx_ - rx), -gl["shiftY"][n] - (fy_ - ry)))
                perrs.append((-loc["patch_shifts"][py, px, n, 0] - (fx_ - rx), -loc["patch_shifts"][py, px, n, 1] - (fy_ - ry)))
    errs, gerrs, perrs = np.array(errs), np.array(gerrs), np.array(perrs)
    print("K3 patch shifts against the truth: rms", float(np.sqrt((perrs ** 2).mean())), "worst", float(np.abs(perrs).max()),
          "per frame rms", np.sqrt((perrs.reshape(-1, N, 2) ** 2).mean((0, 2))).round(2).tolist())
    print("K3 spline error per frame rms", np.sqrt((errs.reshape(-1, N, 2) ** 2).mean((0, 2))).round(2).tolist(), "ref", r)
    # computeBSplineCoeffs fits b = -shift (bspline_helper.cpp:79-80): the spline is the displacement itself
    rms, worst = float(np.sqrt((errs ** 2).mean())), float(np.abs(errs).max())
    grms = float(np.sqrt((gerrs ** 2).mean()))
    print("K3 field recovery: rms", rms, "worst", worst, "global only rms", grms)
    # how much of the local part (what the global shift does not explain) the patches see: slope of measured against true
    truth_local = -(gerrs)                                   # displacement minus what the global alignment found
    meas_local = errs - gerrs                                # spline minus what the global alignment found
    slope = float((truth_local * meas_local).sum() / (truth_local ** 2).sum())
    corr = float(np.corrcoef(truth_local.ravel(), meas_local.ravel())[0, 1])
    print("K3 local part: slope of measured against true", slope, "correlation", corr)
    assert rms <= 0.9 and worst <= 3.2 and rms < 0.6 * grms and corr > 0.8 and 0.4 < slope < 1.1


@pytest.mark.parametrize("Y,X,binning", [(300, 420, 1.5), (256, 256, 2.0), (243, 331, 1.3)])
def test_binning_of_a_frame_against_the_oracle(gpu, oracle, Y, X, binning):
    """--bin of the CUDA program (CUDAFlexAlignScale::runScaleIFT + scaleFFT2DKernel): a frame, dark- and gain-corrected, binned by
    cropping its half spectrum; sizes as getMovieSize computes them (float arithmetic, truncated: 243 x 331 at 1.3 gives an odd
    186 x 254 ... whatever comes out, the same on both sides); fp32 transforms against the oracle's double ones: 1e-5 of the range."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import movie_binned_size, movie_bin_frame
    rng = np.random.default_rng(Y)
    fr = rng.standard_normal((Y, X)).astype(np.float32)
    dark = (0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    gain = (1.0 + 0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    Yb, Xb = movie_binned_size(Y, X, binning)
    assert (Yb, Xb) == (int(np.float32(np.float32(np.float32(Y) / np.float32(binning)) / np.float32(2)) * np.float32(2)),
                        int(np.float32(np.float32(np.float32(X) / np.float32(binning)) / np.float32(2)) * np.float32(2)))
    fr_, fb_ = xa.Fft2D(ctx, Y, X), xa.Fft2D(ctx, Yb, Xb)
    got = movie_bin_frame(fr_, fb_, torch.from_numpy(fr).cuda(), torch.from_numpy(dark).cuda(), torch.from_numpy(gain).cuda()).cpu().numpy()
    exp = oracle.fa_bin_frame((fr.astype(np.float64) - dark) * gain, Yb, Xb)
    assert got.shape == exp.shape == (Yb, Xb)
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    # a binned frame keeps the mean of the frame (the zero frequency is copied, times 1 / (X Y), and comes back un-normalised)
    assert abs(got.mean() - ((fr - dark) * gain).mean()) < 1e-4


def test_warp_of_a_whole_movie_in_one_call(gpu):
    """xh_fa_apply_bspline_frames = the per-frame calls: same sums, same aligned frames, bit for bit."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(8)
    N, Y, X, cp = 5, 180, 260, (4, 3, 3)
    fr = torch.from_numpy(rng.standard_normal((N, Y, X)).astype(np.float32)).cuda()
    cx, cy = rng.uniform(-4, 4, 36), rng.uniform(-4, 4, 36)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
    t1, i1 = torch.zeros((Y, X), device="cuda"), torch.zeros((Y, X), device="cuda")
    o1 = torch.empty((3, Y, X), device="cuda")
    for n in range(1, 4):
        fa.apply_bspline(fr[n], cx, cy, cp, N, n, out=o1[n - 1], total=t1, initial=i1)
    t2, i2 = torch.zeros((Y, X), device="cuda"), torch.zeros((Y, X), device="cuda")
    o2 = torch.empty((3, Y, X), device="cuda")
    fa.apply_bspline_frames(fr, cx, cy, cp, 1, 3, out=o2, total=t2, initial=i2)
    assert torch.equal(t1, t2) and torch.equal(i1, i2) and torch.equal(o1, o2) and t1.abs().max().item() > 0


def test_prefilter_ahead_changes_nothing(gpu):
    """xh_fa_set_option "prefilter_ahead": the local alignment leaves the prefiltered frames with the handle (computed while the host
    fits the spline); the warp that follows gives the same bits as the warp that prefilters itself, and a warp with an initial sum,
    other frames or after a new global alignment does not use them."""
    from tests import synth
    xa, ctx, torch = gpu
    N, Y, X = 5, 300, 420
    frames, drift, field = synth.movie(N, Y, X, seed=3, local=3.0)
    d = torch.from_numpy(frames).cuda()
    patches, psize, cp = (4, 4), (100, 90), (3, 3, 3)
    out = {}
    for ahead in (0, 1):
        fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
        fa.set_option("prefilter_ahead", ahead)
        g = fa.global_alignment(d, 25.0)
        loc = fa.local_alignment(d, g["shiftX"], g["shiftY"], g["ref"], 25.0, patches, psize, 2, cp)
        total, aligned = torch.zeros((Y, X), device="cuda"), torch.empty((N, Y, X), device="cuda")
        fa.apply_bspline_frames(d, loc["coeffsX"], loc["coeffsY"], cp, out=aligned, total=total)
        ini = torch.zeros((Y, X), device="cuda")
        t2 = torch.zeros((Y, X), device="cuda")
        fa.apply_bspline_frames(d, loc["coeffsX"], loc["coeffsY"], cp, total=t2, initial=ini)           # with an initial sum: the ordinary path
        other = (d * 2).contiguous()
        t3 = torch.zeros((Y, X), device="cuda")
        fa.apply_bspline_frames(other, loc["coeffsX"], loc["coeffsY"], cp, total=t3)                     # other frames: the ordinary path
        out[ahead] = (total, aligned, t2, ini, t3)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert torch.allclose(out[1][4], 2 * out[1][0], rtol=1e-5, atol=1e-4) and torch.equal(out[1][0], out[1][2])


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    with pytest.raises(xa.XhError):
        xa.FlexAlign(ctx, 256, 256, 4.0, 8.0)            # scale factor >= 1 (checkSettings)
    fa = xa.FlexAlign(ctx, 256, 256, 1.0, 8.0)
    with pytest.raises(xa.XhError):
        fa.global_alignment(torch.zeros((3, 256, 256), device="cuda"), 500.0)       # --maxShift beyond the reduced frame


@pytest.mark.parametrize("N,Y,X,patches,psize,cp,avg", [(8, 384, 384, (5, 5), (128, 128), (3, 3, 3), 3), (6, 300, 420, (4, 5), (100, 90), (3, 4, 3), 1),
                                                        (5, 301, 423, (4, 4), (91, 120), (3, 3, 3), 2),
                                                        (6, 1024, 1100, (4, 4), (400, 380), (3, 3, 3), 3)])
def test_local_alignment_against_the_oracle(gpu, oracle, N, Y, X, patches, psize, cp, avg):
    """computeLocalAlignment of the CUDA program: patch layout, patch shifts (pruned fp32 transforms against the oracle's double
    FFTs: 5e-3 px) and the B-spline fitted to them, evaluated over the field (the coefficients themselves are badly conditioned
    -- end control points barely touch the patch centres -- their spline is not)."""
    from tests import synth
    xa, ctx, torch = gpu
    frames, drift, field = synth.movie(N, Y, X, seed=N + X, local=4.0)
    max_shift, res = 25.0, (8.0 if Y < 1000 else 20.0)            # (the large case at the scale factor of real movies: patches of 400 correlated at 138)
    g = oracle.fa_global_alignment(frames, max_shift_px=max_shift, max_res=res)
    exp = oracle.fa_local_alignment(frames, g["shiftX"], g["shiftY"], g["ref"], max_shift_px=max_shift, max_res=res, patches=patches, patch_size=psize,
                                    patches_avg=avg, control_points=cp)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, res)
    got = fa.local_alignment(torch.from_numpy(frames).cuda(), g["shiftX"], g["shiftY"], g["ref"], max_shift, patches, psize, avg, cp)
    assert got["dims"] == exp["dims"]
    assert np.array_equal(got["centers"], exp["centers"])
    d = np.abs(got["patch_shifts"] - exp["patch_shifts"])
    print("patch shifts: max difference", d.max())
    assert d.max() <= 5e-3
    # windows of at most 32 rows take the kernel with packed multiply-adds (the 1024-pixel case): the plain kernel gives the same shifts
    fa.set_option("pairwin_form", 0)
    old = fa.local_alignment(torch.from_numpy(frames).cuda(), g["shiftX"], g["shiftY"], g["ref"], max_shift, patches, psize, avg, cp)
    assert np.abs(old["patch_shifts"] - got["patch_shifts"]).max() <= 1e-3
    worst = 0.0
    for n in range(N):
        for y in range(0, Y, 37):
            for x in range(0, X, 41):
                a = oracle.fa_bspline_shift(got["coeffsX"], got["coeffsY"], cp, X, Y, N, x, y, n)
                b = oracle.fa_bspline_shift(exp["coeffsX"], exp["coeffsY"], cp, X, Y, N, x, y, n)
                worst = max(worst, abs(a[0] - b[0]), abs(a[1] - b[1]))
    print("spline: max difference over the field", worst)
    assert worst <= 0.05


def test_bspline_warp_against_the_oracle(gpu, oracle):
    """applyBSplineTransform: prefilter (33-tap fp32 convolution against the double recursion), the shift of every pixel from the
    control points with the reference's 1e-4 cut, cubic interpolation with mirrored borders; dark and gain on the way; the two
    sums. Coefficients from a real fit and, second case, random ones of a few pixels (shifts that leave the frame)."""
    from tests import synth
    xa, ctx, torch = gpu
    N, Y, X = 6, 200, 264
    frames, drift, field = synth.movie(N, Y, X, seed=9, local=3.0)
    rng = np.random.default_rng(2)
    dark = (0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    gain = (1.0 + 0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
    g = oracle.fa_global_alignment(frames, max_shift_px=20.0, max_res=8.0)
    cp = (4, 3, 5)
    cx, cy = fa.local_from_global(g["shiftX"], g["shiftY"], patches=(5, 4), patch_size=(64, 64), control_points=cp)
    # localFromGlobal: the spline follows the negative of the frames' global shifts (loosely: five control points in time for six
    # frames, weakly determined end points and the 1e-4 cut of getShift; the fit itself is checked against the oracle above)
    for n in range(N):
        sx, sy = oracle.fa_bspline_shift(cx, cy, cp, X, Y, N, X // 3, Y // 2, n)
        assert abs(sx + g["shiftX"][n]) < 1.0 and abs(sy + g["shiftY"][n]) < 1.0
    cases = [(cx, cy), (rng.uniform(-6, 6, cx.size), rng.uniform(-6, 6, cx.size))]
    for cxx, cyy in cases:
        total = torch.zeros((Y, X), device="cuda")
        initial = torch.zeros((Y, X), device="cuda")
        exp_total = np.zeros((Y, X))
        for n in range(N):
            out = torch.empty((Y, X), device="cuda")
            fa.apply_bspline(torch.from_numpy(frames[n]).cuda(), cxx, cyy, cp, N, n, torch.from_numpy(dark).cuda(), torch.from_numpy(gain).cuda(), out, total, initial)
            corrected = (frames[n].astype(np.float64) - dark) * gain
            exp = oracle.fa_apply_bspline(corrected, cxx.astype(np.float32), cyy.astype(np.float32), cp, N, n)
            exp_total += exp
            err = np.abs(out.cpu().numpy() - exp).max()
            assert err <= 2e-4 * np.abs(exp).max(), (n, err)
        assert np.abs(total.cpu().numpy() - exp_total).max() <= 3e-4 * np.abs(exp_total).max()
        assert np.abs(initial.cpu().numpy() - ((frames.astype(np.float64) - dark) * gain).sum(0)).max() <= 1e-4 * np.abs(exp_total).max()


def test_local_alignment_improves_a_movie_with_a_drift_field(gpu):
    """End to end on the device at a size no oracle run fits in a test (12 frames of 1536 x 1536, 7 x 7 patches of 300 px,
    5 x 5 x 4 control points): global alignment, local alignment, warp; the aligned sum correlates with the clean field
    better than the sum aligned globally only."""
    from scipy import ndimage
    xa, ctx, torch = gpu
    N, Y, X = 12, 1536, 1536
    rng = np.random.default_rng(4)
    base = ndimage.gaussian_filter(rng.standard_normal((Y + 64, X + 64)), 2.0) * 10
    drift = np.cumsum(rng.uniform(-1.5, 1.5, (N, 2)), 0)
    drift -= drift[0]
    yy, xx = np.mgrid[0:Y, 0:X].astype(np.float64)
    u, v = (xx / X - 0.5) * 2, (yy / Y - 0.5) * 2
    frames = np.empty((N, Y, X), np.float32)
    for n in range(N):
        a = 5.0 * n / (N - 1)
        dx, dy = drift[n, 0] + a * (0.7 * u + 0.3 * v), drift[n, 1] + a * (0.8 * v - 0.2 * u)
        frames[n] = ndimage.map_coordinates(base, [yy + 32 + dy, xx + 32 + dx], order=1, mode="wrap") + 0.5 * rng.standard_normal((Y, X))
    d_frames = torch.from_numpy(frames).cuda()
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 10.0)
    g = fa.global_alignment(d_frames, 40.0)
    patches, psize, cp = (7, 7), (300, 300), (5, 5, 4)
    loc = fa.local_alignment(d_frames, g["shiftX"], g["shiftY"], g["ref"], 40.0, patches, psize, 3, cp)
    gcx, gcy = fa.local_from_global(g["shiftX"], g["shiftY"], patches, psize, cp)
    sums = {}
    for name, (cx, cy) in {"global": (gcx, gcy), "local": (loc["coeffsX"], loc["coeffsY"])}.items():
        total = torch.zeros((Y, X), device="cuda")
        for n in range(N):
            fa.apply_bspline(d_frames[n], cx, cy, cp, N, n, total=total)
        sums[name] = total.cpu().numpy()
    # the clean field as the reference frame saw it
    a = 5.0 * g["ref"] / (N - 1)
    dx, dy = drift[g["ref"], 0] + a * (0.7 * u + 0.3 * v), drift[g["ref"], 1] + a * (0.8 * v - 0.2 * u)
    clean = ndimage.map_coordinates(base, [yy + 32 + dy, xx + 32 + dx], order=1, mode="wrap")
    c = {k: np.corrcoef(s[64:-64, 64:-64].ravel(), clean[64:-64, 64:-64].ravel())[0, 1] for k, s in sums.items()}
    print("correlation of the aligned sum with the clean field:", c)
    assert c["local"] > 0.99 and (1 - c["local"]) < 0.5 * (1 - c["global"])


@pytest.mark.parametrize("x,y,n", [(42, 24, 10), (24, 42, 10), (36, 86, 17)])
def test_correlate_known_answers_of_the_reference(gpu, oracle, x, y, n):
    """FlexAlignCorrelateTest of the reference (test_cuda_flexalign_correlate.cpp): frame k = one point at (x/2 + k, y/2 + k); pair
    (i, j) correlates at (i - j, i - j) from the centre to 1e-4 -- the reference's sizes, its maximal distance, its tolerance --
    and the device agrees with the oracle on random frames."""
    xa, ctx, torch = gpu
    fr = np.zeros((n, y, x), np.float32)
    for k in range(n):
        fr[k, k + y // 2, x // 2 + k] = 1
    pos = xa.fa_correlate(ctx, torch.from_numpy(fr).cuda(), np.sqrt(2.0 * n * n))
    idx = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert abs(pos[idx, 0] - x / 2 - (i - j)) <= 1e-4 and abs(pos[idx, 1] - y / 2 - (i - j)) <= 1e-4
            idx += 1
    rng = np.random.default_rng(x)
    from scipy import ndimage
    fr = np.stack([ndimage.shift(ndimage.gaussian_filter(rng.standard_normal((y, x)), 1.5), (0.3 * k, -0.4 * k), mode="wrap") for k in range(6)]).astype(np.float32)
    got = xa.fa_correlate(ctx, torch.from_numpy(fr).cuda(), 8.0)
    exp = oracle.fa_correlate(fr, 8.0)
    assert np.abs(got - exp).max() <= 1e-3


def test_bspline_warp_identities_of_the_reference(gpu):
    """GeoTransformerApplyBSplineTransformTest (test_cuda_geo_transformer_apply_bspline_transform.cpp:107-147): zero coefficients
    leave a random 259 x 311 image unchanged (to the reference's 1e-5 for float), a zero image stays zero under random
    coefficients."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(13)
    img = torch.from_numpy(rng.uniform(-1, 1, (311, 259)).astype(np.float32)).cuda()
    fa = xa.FlexAlign(ctx, 311, 259, 1.0, 8.0)
    out = torch.empty_like(img)
    fa.apply_bspline(img, np.zeros(27), np.zeros(27), (3, 3, 3), 1, 0, out=out)
    assert (out - img).abs().max().item() <= 1e-5
    fa2 = xa.FlexAlign(ctx, 148, 148, 1.0, 8.0)
    z = torch.zeros((148, 148), device="cuda")
    out2 = torch.ones_like(z)
    fa2.apply_bspline(z, rng.uniform(-10, 10, 120), rng.uniform(-10, 10, 120), (6, 5, 4), 4, 0, out=out2)
    assert out2.abs().max().item() == 0


@pytest.mark.parametrize("Y,X", [(128, 160), (90, 150)])
def test_dose_filter_against_the_oracle(gpu, oracle, Y, X):
    """ProgMovieFilterDose on one frame: fp32 transforms against the oracle's double ones, 1e-5 of the frame's range; late frames lose
    their high frequencies altogether (the branch that zeroes a coefficient)."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(Y)
    fr = rng.standard_normal((Y, X)).astype(np.float32)
    fft = xa.Fft2D(ctx, Y, X)
    for n, dose in ((0, 2.0), (7, 2.0), (30, 1.5)):
        got = xa.movie_dose_filter(fft, torch.from_numpy(fr).cuda(), 1.1, 300, n * dose + 0.5, (n + 1) * dose + 0.5).cpu().numpy()
        exp = oracle.dose_filter_frame(fr, 1.1, 300, n * dose + 0.5, (n + 1) * dose + 0.5)
        assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max() + 1e-6
    with pytest.raises(xa.XhError):
        xa.movie_dose_filter(fft, torch.from_numpy(fr).cuda(), 1.1, 120, 0.0, 2.0)


def test_wide_correlation_maxima_take_the_full_transform(gpu, oracle):
    """A very smooth movie: the maximum of a pair correlation is so wide that the square bestShift grows around it leaves the window
    the pruned transforms cover (maxShift + 8); such pairs are flagged on the device and repeated through the full inverse transform,
    and the shifts still equal the oracle's."""
    from tests import synth
    xa, ctx, torch = gpu
    N, Y, X = 5, 256, 320
    frames, drift, _ = synth.movie(N, Y, X, seed=21, smooth=30.0, noise=0.02)
    exp = oracle.fa_global_alignment(frames, max_shift_px=6.0, max_res=8.0)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
    got = fa.global_alignment(torch.from_numpy(frames).cuda(), 6.0)
    print("pairs through the full transform:", fa.last_full_pairs(), "of", len(exp["bX"]))
    assert fa.last_full_pairs() > 0
    assert np.abs(got["bX"] - exp["bX"]).max() <= 5e-3 and np.abs(got["bY"] - exp["bY"]).max() <= 5e-3 and got["ref"] == exp["ref"]


@pytest.mark.parametrize("dtype", ["int8", "int16", "uint16", "uint8", "float32"])
def test_frames_as_counts_become_floats_on_the_device(gpu, dtype):
    """xh_movie_frame_to_float: the cast of Image<float>::read behind the host copy; whole range of every type, sizes that are not
    multiples of the 16 bytes a thread converts, misaligned starts."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(1)
    for n, off in ((4092 * 5760 // 64, 0), (1000003, 0), (777, 3), (5, 1), (0, 0)):
        if dtype == "float32":
            a = rng.standard_normal(n + off).astype(np.float32)
        else:
            info = np.iinfo(dtype)
            a = rng.integers(info.min, info.max + 1, n + off).astype(dtype)
            a[: min(2, n + off)] = (info.min, info.max)[: min(2, n + off)]
        d = torch.from_numpy(a).cuda()[off:]
        if n == 0:
            continue
        out = xa.movie_frames_to_float(ctx, d.contiguous() if off == 0 else d)       # (a slice of a 1-D tensor stays contiguous: misaligned pointer)
        assert out.dtype == torch.float32 and torch.equal(out.cpu(), torch.from_numpy(a[off:].astype(np.float32)))
    o2 = torch.empty(17, device="cuda")
    xa.movie_frames_to_float(ctx, torch.arange(17, dtype=torch.int16, device="cuda"), out=o2)
    assert torch.equal(o2.cpu(), torch.arange(17, dtype=torch.float32))


def test_two_movies_in_flight_on_one_device(gpu):
    """bench.py --mode flexalign keeps two movies in flight per GPU: a host thread per movie, each with its own stream, context and
    FlexAlign handle.  The library keeps no state outside its handles, so the two threads must return what the same calls return
    one after the other: shifts, patch shifts, spline coefficients and the aligned sums, bit for bit."""
    import threading
    xa, ctx, torch = gpu
    N, Y, X = 6, 512, 640
    movies = [synthetic_movie(N, Y, X, seed=31 + i, max_step=2.0)[0] for i in range(2)]
    cp, patches, psize = (4, 4, 3), (3, 3), (256, 256)

    def run(lane_ctx, fa, frames):
        d = torch.from_numpy(frames).cuda()
        g = fa.global_alignment(d, 20.0)
        loc = fa.local_alignment(d, g["shiftX"], g["shiftY"], g["ref"], 20.0, patches, psize, 3, cp)
        total = torch.zeros((Y, X), device="cuda")
        fa.apply_bspline_frames(d, loc["coeffsX"], loc["coeffsY"], cp, total=total)
        lane_ctx.sync()
        return g["shiftX"].copy(), g["shiftY"].copy(), np.asarray(loc["patch_shifts"]).copy(), np.asarray(loc["coeffsX"]).copy(), total.cpu().numpy()

    seq = []
    for m in movies:
        fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
        seq.append(run(ctx, fa, m))
    out, err = [None, None], []

    def lane(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                c = xa.Context(0)
                fa = xa.FlexAlign(c, Y, X, 1.0, 8.0)
                fa.set_option("prefilter_ahead", 1)
                for _ in range(3):              # several rounds: the lanes overlap in every phase
                    out[i] = run(c, fa, movies[i])
        except BaseException as e:
            err.append(e)

    th = [threading.Thread(target=lane, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for i in range(2):
        for a, b in zip(seq[i], out[i]):
            assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("Y,X,cp", [(96, 512, (6, 6, 5)), (130, 448, (5, 4, 4)), (64, 300, (6, 6, 5))])
def test_warp_with_the_usual_control_grid_against_the_oracle(gpu, oracle, Y, X, cp):
    """The kernel a 6 x 6 x 5 control grid takes (k_fa_warp_quads) against the oracle's applyBSplineTransform, every frame of a short
    movie (first and last frames have fewer than four layers).  Rows of 512 and 448 pixels are whole waves, some of which share a
    control cell -- the 64 terms of the shift are then formed once per wave, the 1e-4 cut decided at the wave's two ends and per
    lane only for the terms that change sides in between -- and some of which a control column ends in (every lane its own terms);
    300 pixels leave a partial wave at the end of every row."""
    xa, ctx, torch = gpu
    N = 7
    rng = np.random.default_rng(Y + X)
    nc = cp[0] * cp[1] * cp[2]
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
    for amp in (6.0, 0.05):                 # shifts of pixels, and shifts small enough that many terms sit near the cut
        cx, cy = rng.uniform(-amp, amp, nc), rng.uniform(-amp, amp, nc)
        for n in range(N):
            frame = rng.standard_normal((Y, X)).astype(np.float32)
            out = torch.empty((Y, X), device="cuda")
            total = torch.ones((Y, X), device="cuda")
            fa.apply_bspline(torch.from_numpy(frame).cuda(), cx, cy, cp, N, n, out=out, total=total)
            exp = oracle.fa_apply_bspline(frame.astype(np.float64), cx.astype(np.float32), cy.astype(np.float32), cp, N, n)
            err = np.abs(out.cpu().numpy() - exp).max()
            assert err <= 2e-4 * (frame.max() - frame.min()), (amp, n, err)
            assert np.abs(total.cpu().numpy() - 1.0 - exp).max() <= 2e-4 * (frame.max() - frame.min())


def test_warp_of_a_k3_frame_against_the_oracle(gpu, oracle):
    """BASELINE config 5's own shape: one 4092 x 5760 frame of a 40-frame movie under a 6 x 6 x 5 control grid with shifts of a few
    pixels, against the oracle's applyBSplineTransform (every wave of the kernel shares a control cell here: 5760 / 3 = 1920 = 30 x 64)."""
    xa, ctx, torch = gpu
    Y, X, N, cp = 4092, 5760, 40, (6, 6, 5)
    rng = np.random.default_rng(40)
    nc = cp[0] * cp[1] * cp[2]
    cx, cy = rng.uniform(-5, 5, nc), rng.uniform(-5, 5, nc)
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 30.0)
    frame = rng.standard_normal((Y, X)).astype(np.float32)
    d = torch.from_numpy(frame).cuda()
    for n in (0, 17):
        out = torch.empty((Y, X), device="cuda")
        fa.apply_bspline(d, cx, cy, cp, N, n, out=out)
        exp = oracle.fa_apply_bspline(frame.astype(np.float64), cx.astype(np.float32), cy.astype(np.float32), cp, N, n)
        err = np.abs(out.cpu().numpy() - exp).max()
        print("frame", n, "max difference", err)
        assert err <= 2e-4 * (frame.max() - frame.min()), (n, err)


def test_warp_with_a_large_control_grid_takes_the_plain_kernel(gpu, oracle):
    """the quad form of the warp kernel stages 32 (lX - 3) lY bytes per layer pair in LDS; a control grid whose quads
    exceed 64 KB (here 30 x 30 x 5, as `--controlPoints 30 30 5` would ask for) must fall back to the plain kernel, not fail the
    launch.  Checked against the oracle's applyBSplineTransform."""
    xa, ctx, torch = gpu
    Y, X, N, cp = 192, 224, 6, (30, 30, 5)
    rng = np.random.default_rng(3)
    frame = rng.standard_normal((Y, X)).astype(np.float32)
    cx, cy = rng.uniform(-3, 3, cp[0] * cp[1] * cp[2]), rng.uniform(-3, 3, cp[0] * cp[1] * cp[2])
    fa = xa.FlexAlign(ctx, Y, X, 1.0, 8.0)
    out = torch.empty((Y, X), device="cuda")
    fa.apply_bspline(torch.from_numpy(frame).cuda(), cx, cy, cp, N, 2, out=out)
    exp = oracle.fa_apply_bspline(frame.astype(np.float64), cx.astype(np.float32), cy.astype(np.float32), cp, N, 2)
    assert np.abs(out.cpu().numpy() - exp).max() <= 2e-4 * (frame.max() - frame.min())
