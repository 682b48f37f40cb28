"""Pin the CPU oracle against the reference's own known-answer unit tests.

Every expected value below is re-typed from the reference's gtest sources
(/root/reference/src/xmipp/applications/tests/function_tests/...), cited per test.
"""
import numpy as np
import pytest

from tests import synth

M3 = np.array([[1., 2., 3.], [3., 2., 1.], [4., 4., 5.]])  # fixture used by all four suites


def test_fft_forward_values_and_normalisation(oracle):
    # test_fftw_main.cpp:35-51  (FourierTransformer: forward divided by N)
    F = oracle.fft2d_r2c(M3)
    exp = np.array([[2.77778 + 0j, -0.0555556 + 0.096225j],
                    [-0.388889 + 0.673575j, -0.388889 - 0.096225j],
                    [-0.388889 - 0.673575j, -0.0555556 + 0.288675j]])
    assert np.allclose(F, exp, atol=1e-5)
    assert np.allclose(F, np.fft.rfft2(M3) / 9, atol=1e-14)
    # inverse is un-normalised: round trip restores the input
    assert np.allclose(oracle.fft2d_c2r(F, 3), M3, atol=1e-13)


def test_fft_idx2digfreq(oracle):
    # test_fftw_main.cpp:80-109
    f = oracle.lib().xo_fft_idx2digfreq
    assert f(0, 128) == 0 and f(1, 128) == 1 / 128 and f(64, 128) == 0.5
    assert f(65, 128) == -63 / 128 and f(127, 128) == -1 / 128
    assert f(64, 129) == 64 / 129 and f(65, 129) == -64 / 129 and f(128, 129) == -1 / 129
    assert f(255, 256) == -1 / 256


@pytest.mark.parametrize("n", [1, 2, 3, 6, 12, 50, 97, 194, 197, 394, 398, 796, 1024, 1000])
def test_fft_any_length_matches_numpy(oracle, n):
    rng = np.random.default_rng(n)
    v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.allclose(oracle.fft1d_c2c(v, -1), np.fft.fft(v), atol=1e-10)
    assert np.allclose(oracle.fft1d_c2c(v, +1), np.fft.ifft(v) * n, atol=1e-10)
    r = rng.standard_normal(n)
    assert np.allclose(oracle.fft1d_r2c(r), np.fft.rfft(r) / n, atol=1e-12)
    if n > 1:
        assert np.allclose(oracle.fft1d_c2r(np.fft.rfft(r), n), r * n, atol=1e-9)


def test_polar_average_and_stddev_pin(oracle):
    # test_polar_main.cpp:32-40: raw 3x3 (STARTING = 0), rings 0..1, tolerance 1e-6
    rings = oracle.polar_from_cartesian(M3, 0, 1, 0, 0)
    mean, std = oracle.polar_avg_std(rings, 0, 1)
    assert abs(mean - 1.886528450043468) < 1e-6
    assert abs(std - 0.49643800057938808) < 1e-6


def test_rotate_bspline3_pin(oracle):
    # test_transformation_main.cpp:76-95 (MultidimArray == compares within 1e-6)
    out = oracle.rotate2d(M3, 10.0, degree=3, wrap=False)
    exp = np.array([[0, 2.1950049, 0], [2.6541736, 2, 1.3803737], [0, 3.9039731, 0]])
    assert np.allclose(out, exp, atol=1e-6)


def test_translate_wrap_is_row_roll(oracle):
    # test_transformation_main.cpp:97-113: translate by (0,1) with wrap moves row i to i+1
    out = oracle.translate2d(M3, 0.0, 1.0, degree=3, wrap=True)
    assert np.allclose(out, np.roll(M3, 1, axis=0), atol=1e-6)


def test_correlation_matrix_pin(oracle):
    # test_filters_main.cpp:71-92
    R = oracle.correlation_matrix(M3, M3)
    exp = np.array([[64., 62, 66], [78, 85, 78], [66, 62, 64]])
    assert np.allclose(R, exp, atol=1e-6)


def test_best_shift_self_is_zero(oracle):
    # test_filters_main.cpp:59-69
    sx, sy, _ = oracle.best_shift(M3, M3)
    assert sx == 0.0 and sy == 0.0


def test_correlation_index_self_is_one(oracle):
    # test_filters_main.cpp:94-103 (EXPECT_DOUBLE_EQ => 4 ulp)
    assert abs(oracle.correlation_index(M3, M3) - 1.0) < 1e-15


def test_euler_matrix_closed_form(oracle):
    # test_geometry_main.cpp:26-67
    for rot in (0., 30., 150., 270.):
        for tilt in (0., 60., 120.):
            for psi in (0., 90., 210.):
                A = oracle.euler_matrix(rot, tilt, psi)
                a, b, g = np.radians([rot, tilt, psi])
                assert abs(A[0, 0] - (np.cos(g) * np.cos(b) * np.cos(a) - np.sin(g) * np.sin(a))) < 1e-12
                assert abs(A[0, 1] - (np.cos(g) * np.cos(b) * np.sin(a) + np.sin(g) * np.cos(a))) < 1e-12
                assert abs(A[0, 2] - (-np.cos(g) * np.sin(b))) < 1e-12
                assert abs(A[1, 1] - (-np.sin(g) * np.cos(b) * np.sin(a) + np.cos(g) * np.cos(a))) < 1e-12
                assert abs(A[1, 2] - (np.sin(g) * np.sin(b))) < 1e-12
                assert abs(A[2, 2] - np.cos(b)) < 1e-12
                assert np.allclose(A @ A.T, np.eye(3), atol=1e-12)


def test_ctf_wavelength_and_first_zero_pin(oracle):
    # test_ctf_main.cpp:77-96: errorMaxFreqCTFs = 7.6852355 at 300 kV, defoci 6000/7500;
    # ctf.cpp:187-211: 1/sqrt((pi/2)/(K1*|dU-dV|)), K1 = pi*lambda
    p = oracle.ctf_params(kV=300.0)
    lam = oracle.lib().xo_ctf_lambda(p)
    assert abs(lam - 0.0196876) < 1e-6
    val = 1.0 / np.sqrt((np.pi / 2) / (np.pi * lam * abs(6000.0 - 7500.0)))
    assert abs(val - 7.6852355) < 2e-6


def test_ctf_phase_flip_pin(oracle):
    """test_ctf_main.cpp:126-149: a centred delta in a 256 x 256 image, phase flipped (CTFDescription::correctPhase, the
    variant with the damping envelope; without envelope parameters it flips the same coefficients as actualPhaseFlip, the
    restatement the programs use): standard deviation 0.003906, maximum 0.017565, EXPECT_NEAR 1e-4."""
    c = oracle.ctf_params(Tm=1.0, kV=300.0, DeltafU=20000.0, DeltafV=20000.0, Cs=2.0, Q0=0.1, K=1.0)
    delta = np.zeros((256, 256))
    delta[128, 128] = 1.0
    for damping in (True, False):
        f = oracle.ctf_phase_flip(delta, c, with_damping=damping)
        assert abs(f.std() - 0.003906) < 1e-4
        assert abs(f.max() - 0.017565) < 1e-4
    # and it is a sign pattern: every coefficient keeps its modulus
    assert np.allclose(np.abs(np.fft.rfft2(f)), np.abs(np.fft.rfft2(delta)), atol=1e-12)


def test_flexalign_global_alignment_recovers_a_known_drift(oracle):
    """The reference holds no known answer for the movie alignment that runs without CUDA (PARITY UNPINNED, oracle/xo_flexalign.cpp);
    the restatement is checked on physics: a synthetic movie with a known drift. The shifts it reports are the negatives of
    the frames' displacements from the reference frame (storeGlobalShifts negates them once more into "the shift to apply"),
    pair shifts are consistent, and the solver alone reproduces exact shifts from exact pair shifts, outlier included."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    Y, X, N = 240, 320, 6
    base = ndimage.gaussian_filter(rng.standard_normal((Y + 64, X + 64)), 3.0) * 10
    drift = np.cumsum(rng.uniform(-1.5, 1.5, (N, 2)), 0)
    drift -= drift[0]
    frames = np.stack([ndimage.shift(base, (-drift[i, 1], -drift[i, 0]), order=3, mode="wrap")[32:32 + Y, 32:32 + X]
                       + 0.5 * rng.standard_normal((Y, X)) for i in range(N)])
    r = oracle.fa_global_alignment(frames, Ts=1.0, max_shift_px=20.0, max_res=8.0)
    assert r["new_dims"] == (int(Y * 0.8493218 * 8 / 8.0), int(X * 0.8493218 * 8 / 8.0))
    t = drift - drift[r["ref"]]
    assert np.abs(r["shiftX"] + t[:, 0]).max() < 0.6 and np.abs(r["shiftY"] + t[:, 1]).max() < 0.6
    assert np.corrcoef(r["shiftX"], -t[:, 0])[0, 1] > 0.95
    # the solver: exact pair shifts (one of them corrupted) give back the exact frame-to-frame shifts
    s = rng.uniform(-2, 2, (7, 2))                         # shifts between successive frames of an 8-frame movie
    bx, by = [], []
    for i in range(7):
        for j in range(i + 1, 8):
            bx.append(s[i:j, 0].sum()); by.append(s[i:j, 1].sum())
    bx, by = np.array(bx), np.array(by)
    bx[5] += 40.0                                          # an outlier: rejected by the 3-sigma round
    sx, sy, ref = oracle.fa_solve(bx, by, 8)
    for j in range(8):
        tx = -s[ref:j, 0].sum() if ref < j else s[j:ref, 0].sum()
        assert abs(sx[j] - tx) < 1e-9
    assert ref == int(np.argmin([max(abs(-s[i:j, 0].sum() if i < j else s[j:i, 0].sum()) for j in range(8)) for i in range(8)]))


def test_prefilter_inverts_bspline_sampling(oracle):
    # coefficients c reproduce the samples: s[k] = (c[k-1] + 4 c[k] + c[k+1]) / 6 with
    # half-sample mirror (c[-1] = c[0], c[n] = c[n-1])  
    rng = np.random.default_rng(0)
    img = rng.standard_normal((17, 23))
    c = oracle.prefilter2d(img)
    cp = np.pad(c, 1, mode="symmetric")
    k = np.array([1, 4, 1]) / 6.0
    rows = cp[:, :-2] * k[0] + cp[:, 1:-1] * k[1] + cp[:, 2:] * k[2]
    rec = rows[:-2] * k[0] + rows[1:-1] * k[1] + rows[2:] * k[2]
    assert np.allclose(rec, img, atol=1e-12)


def test_fourier_projector_against_analytic_gaussians(oracle):
    """FourierProjector (data/fourier_projection.cpp) has no unit test in the reference; the restatement is
    pinned on physics instead: the projection of a sum of isotropic Gaussians is the sum of 2-D Gaussians of
    weight sigma*sqrt(2 pi) centred at the first two rows of Euler(rot,tilt,psi) applied to the 3-D centres.
    That fixes the axis order, the centring (Xmipp origin), the Euler convention and the normalisation."""
    D = 32
    z, y, x = np.mgrid[-(D // 2):D - D // 2, -(D // 2):D - D // 2, -(D // 2):D - D // 2].astype(float)
    cen = [(3.0, -2.0, 4.0), (-5.0, 1.0, -2.0)]
    sig = [2.0, 2.5]
    amp = [1.0, 0.7]
    vol = sum(a * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * s * s)) for a, c, s in zip(amp, cen, sig))
    fp = oracle.FP(vol, 2.0, 0.5, 3)
    assert (fp.P, fp.cdim, fp.cstart) == (64, 63, -31)
    yy, xx = np.mgrid[-(D // 2):D - D // 2, -(D // 2):D - D // 2].astype(float)
    for ang, tol in (((0, 0, 0), 1e-6), ((30, 40, 50), 5e-4), ((90, 90, 0), 1e-4), ((123, 67, -20), 5e-4)):
        E = synth.euler_matrix(*ang)
        ana = np.zeros((D, D))
        for a, c, s in zip(amp, cen, sig):
            pc = E @ np.array(c)
            ana += a * s * np.sqrt(2 * np.pi) * np.exp(-((xx - pc[0]) ** 2 + (yy - pc[1]) ** 2) / (2 * s * s))
        got = fp.project(*ang)
        assert np.abs(got - ana).max() <= tol * ana.max(), ang
    # a CTF image multiplies the slice: a constant 0.5 halves the projection; the crop follows max_freq
    half = fp.project(30, 40, 50, ctf=np.full((D, D // 2 + 1), 0.5))
    assert np.allclose(half, 0.5 * fp.project(30, 40, 50), rtol=0, atol=1e-12)
    assert oracle.FP(vol, 2.0, 0.25, 3).cdim == 2 * (int(0.25 * 64 + 10)) + 1


def _frc_pin_volumes():
    # test_resolution_frc.cpp:20-88
    v1 = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26.4, 27.5, 28.5],
                  float).reshape(3, 3, 3)
    v2 = np.zeros((3, 3, 3))
    v2[2] = np.array([1.5, 2.4, 3.3, 4.6, 5.7, 6.4, 7.3, 8.2, 9.5]).reshape(3, 3)
    v2[1] = np.array([10.2, 11.4, 12.5, 13.6, 14.5, 15.7, 17.3, 18.2, 19.4]).reshape(3, 3)
    v2[0] = np.array([20.3, 21.4, 22.5, 23.4, 24.5, 25.6, 26.7, 24, 23]).reshape(3, 3)
    return v1, v2


def test_frc_rfactor_pin(oracle):
    """test_resolution_frc.cpp:120-130: sam=2, min_sam=-1, max_sam=2 -> minFreq=-2, maxFreq=1, Rfactor 0.134661 +- 1e-5."""
    v1, v2 = _frc_pin_volumes()
    sam, min_sam, max_sam = 2.0, -1.0, 2.0
    r = oracle.frc_dpr(v1, v2, sam, do_dpr=False, do_rfactor=True, min_freq=sam / min_sam, max_freq=sam / max_sam)
    assert abs(r["rfactor"] - 0.134661) < 0.00001
    assert r["frc"][0] == pytest.approx(1.0, abs=1e-14)


@pytest.mark.parametrize("shape", [(3, 3, 3), (8, 8, 8), (6, 10, 9), (1, 12, 12)])
def test_frc_shells_against_numpy(oracle, shape):
    """Independent numpy statement of the same definition (rfftn / size, shells of round(R * xsize))."""
    rng = np.random.default_rng(sum(shape))
    a = rng.normal(size=shape)
    b = a + 0.5 * rng.normal(size=shape)
    Z, Y, X = shape
    F1, F2 = np.fft.rfftn(a) / a.size, np.fft.rfftn(b) / b.size
    fz, fy, fx = np.fft.fftfreq(Z)[:, None, None], np.fft.fftfreq(Y)[None, :, None], np.abs(np.fft.fftfreq(X))[None, None, :X // 2 + 1]
    R = np.sqrt(fz ** 2 + fy ** 2 + fx ** 2)
    idx = np.floor(R * X + 0.5).astype(int)      # C round(): halves away from zero
    L = X // 2 + 1
    keep = (R ** 2 <= 0.25) & (idx < L)
    def shell(v):
        return np.bincount(idx[keep], weights=v[keep], minlength=L)
    frc = shell((np.conj(F1) * F2).real) / np.sqrt(shell(np.abs(F1) ** 2) * shell(np.abs(F2) ** 2))
    cnt = shell(np.ones(R.shape))
    dphi = np.degrees(np.angle(F1) - np.angle(F2))
    dphi = (dphi + 180.0) % 360.0 - 180.0
    w = np.abs(F1) + np.abs(F2)
    r = oracle.frc_dpr(a.reshape(shape), b.reshape(shape), 1.5, do_dpr=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.allclose(r["frc"], frc, atol=1e-12, equal_nan=True)
        assert np.allclose(r["frc_noise"], 2 / np.sqrt(cnt), equal_nan=True)
        assert np.allclose(r["error_l2"], shell(np.abs(F1 - F2)) / cnt, atol=1e-14, equal_nan=True)
        assert np.allclose(r["dpr"], np.sqrt(shell(w * dphi ** 2) / shell(w)), atol=1e-7, equal_nan=True)
    assert np.allclose(r["freq"], np.arange(L) / (X * 1.5))


def test_the_two_reference_reconstruction_programs_agree_to_float_level():
    """xmipp_reconstruct_fourier (ProgRecFourier, RF: double accumulators, image-driven scatter into the FFTW layout,
    wrap + conjugate, correctWeight) and xmipp_reconstruct_fourier_accel (RFA: float, voxel-driven traversal,
    mirrorAndCrop) are two arithmetics for the same pairs and weights. The device implements RFA under both names;
    this pins, on the two CPU restatements, what that choice costs: 1e-3 of the peak at the edges, correlation
    0.99999+. And --iter (NiterWeight) changes nothing wherever the weight exceeds 1e-3 (RF:1056-1101: the
    re-processing pass multiplies the 1/w estimate by w)."""
    from oracle import pyoracle as o
    from tests import synth
    D = 32
    vol = synth.phantom(D, seed=1, nblobs=12)
    rng = np.random.default_rng(2)
    ang = synth.random_angles(200, rng)
    imgs = np.stack([synth.project(vol, *a) for a in ang])
    a, b, b3 = o.RF(D), o.RF2(D), o.RF2(D, niter_weight=3)
    for i in range(len(ang)):
        A = synth.euler_matrix(*ang[i]).T
        a.insert(a.prepare_image(imgs[i]), A)
        b.insert(imgs[i], A)
        b3.insert(imgs[i], A)
    a.mirror_and_crop()
    va, vb, vb3 = a.finish(), b.finish(), b3.finish()
    peak = np.abs(vb).max()
    assert np.abs(va - vb).max() <= 2e-3 * peak
    assert np.corrcoef(va.ravel(), vb.ravel())[0, 1] > 0.99999
    assert np.abs(vb3 - vb).max() <= 1e-12 * peak
    assert np.corrcoef(vb.ravel(), vol.ravel())[0, 1] > 0.999


def test_flexalign_local_alignment_recovers_a_known_field(oracle):
    """The patch alignment of the CUDA program (computeLocalAlignment) has no known answer in the reference either (its tests need
    a CUDA device): physics again. A movie whose drift differs over the field: the patch shifts follow the local drift at the
    patch centres, the B-spline of BSplineHelper gives -(patch shift) back at the centres (the rows of the fit), its value
    anywhere in the field is the local displacement from the reference frame, and warping a frame by it (applyShiftTransform's
    arithmetic) brings it onto the reference frame."""
    from tests import synth
    N, Y, X = 8, 384, 384
    frames, drift, field = synth.movie(N, Y, X, seed=3, local=4.0)
    g = oracle.fa_global_alignment(frames, max_shift_px=30.0, max_res=8.0)
    ref, cp, pt = g["ref"], (3, 3, 3), (5, 5)
    loc = oracle.fa_local_alignment(frames, g["shiftX"], g["shiftY"], ref, max_shift_px=30.0, max_res=8.0, patches=pt, patch_size=(128, 128),
                                    control_points=cp)
    assert loc["dims"] == (128, 128, 110, 110)        # getCorrelationHint: the smallest even size with size / 128 >= 0.8493
    ps, cen = loc["patch_shifts"], loc["centers"]
    e_patch, e_fit = [], []
    for j in range(pt[1]):
        for i in range(pt[0]):
            for n in range(N):
                dx, dy = field(n, cen[j, i, 0], cen[j, i, 1])
                d0x, d0y = field(ref, cen[j, i, 0], cen[j, i, 1])
                e_patch.append((ps[j, i, n, 0] + dx - d0x, ps[j, i, n, 1] + dy - d0y))
                sx, sy = oracle.fa_bspline_shift(loc["coeffsX"], loc["coeffsY"], cp, X, Y, N, int(cen[j, i, 0]), int(cen[j, i, 1]), n)
                e_fit.append((sx + ps[j, i, n, 0], sy + ps[j, i, n, 1]))
    e_patch, e_fit = np.array(e_patch), np.array(e_fit)
    assert np.sqrt((e_patch ** 2).mean()) < 0.6 and np.abs(e_patch).max() < 1.5
    assert np.sqrt((e_fit ** 2).mean()) < 0.8          # 27 coefficients for 200 rows, and getShift drops terms below 1e-4
    e_field, raw = [], []
    for n in range(N):
        for y in range(64, Y - 63, 64):
            for x in range(64, X - 63, 64):
                sx, sy = oracle.fa_bspline_shift(loc["coeffsX"], loc["coeffsY"], cp, X, Y, N, x, y, n)
                dx, dy = field(n, x, y)
                d0x, d0y = field(ref, x, y)
                e_field.append((sx - (dx - d0x), sy - (dy - d0y)))
                raw.append((dx - d0x, dy - d0y))
    e_field, raw = np.array(e_field), np.array(raw)
    assert np.sqrt((e_field ** 2).mean()) < 0.7 < 1.2 < np.sqrt((raw ** 2).mean())
    n = N - 1 if ref < N // 2 else 0
    w = oracle.fa_apply_bspline(frames[n], loc["coeffsX"], loc["coeffsY"], cp, N, n)
    w0 = oracle.fa_apply_bspline(frames[ref], loc["coeffsX"], loc["coeffsY"], cp, N, ref)
    c = lambda a, b: np.corrcoef(a[40:-40, 40:-40].ravel(), b[40:-40, 40:-40].ravel())[0, 1]
    assert c(frames[n], frames[ref]) < 0.75 < 0.8 < c(w, w0)


@pytest.mark.parametrize("x,y,n", [(42, 24, 10), (24, 42, 10), (36, 86, 17)])
def test_flexalign_correlate_known_answers(oracle, x, y, n):
    """PINNED: FlexAlignCorrelateTest (applications/tests/function_tests/test_cuda_flexalign_correlate.cpp:17-70,118-140): frame k
    holds one point at (x/2 + k, y/2 + k); the correlation maximum of the pair (i, j) lies at (i - j, i - j) from the centre,
    to 1e-4, for the reference's three sizes (Dimensions(42, 24, 1, 10), (24, 42, 1, 10), (36, 86, 1, 17)) and its maximal
    distance sqrt(2 n^2). This fixes the sign of the pair shifts, the centring by (-1)^(x+y) and the 3 x 3 refinement of the
    stage the local alignment is built on."""
    fr = np.zeros((n, y, x))
    for k in range(n):
        fr[k, k + y // 2, x // 2 + k] = 1
    pos = oracle.fa_correlate(fr, np.sqrt(2.0 * n * n))
    idx = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert abs(pos[idx, 0] - x / 2 - (i - j)) <= 1e-4 and abs(pos[idx, 1] - y / 2 - (i - j)) <= 1e-4
            idx += 1


def test_bspline_warp_identities(oracle):
    """PINNED on the properties GeoTransformerApplyBSplineTransformTest asserts (test_cuda_geo_transformer_apply_bspline_transform.cpp:
    107-147): zero coefficients leave a random image of 259 x 311 unchanged (prefilter, then interpolation at the pixel itself),
    and a zero image stays zero under random coefficients of (6, 5, 4) control points."""
    rng = np.random.default_rng(13)
    img = rng.uniform(-1, 1, (311, 259))
    out = oracle.fa_apply_bspline(img, np.zeros(27), np.zeros(27), (3, 3, 3), 1, 0)
    assert np.abs(out - img).max() <= 1e-12
    z = oracle.fa_apply_bspline(np.zeros((147, 147)), rng.uniform(-10, 10, 120), rng.uniform(-10, 10, 120), (6, 5, 4), 4, 0)
    assert np.abs(z).max() == 0


def test_movie_filter_dose_known_answers(oracle):
    """PINNED: MovieFilterDoseTest (applications/tests/function_tests/test_movie_filter_dose.cpp:15-90), every value of it:
    doseFilter, the voltage scaling factors, criticalDose, optimalDoseGivenCriticalDose (EXPECT_FLOAT_EQ = 4 float ulps)."""
    vs, dose_filter, critical, optimal = oracle.dose_scalars()
    feq = lambda a, b: abs(np.float32(a) - np.float32(b)) <= 4 * np.spacing(np.float32(b))
    assert feq(dose_filter(4.0, 412084.3), 0.9999952) and feq(dose_filter(4.0, 12.82717), 0.8556285)
    assert vs(300) == 1.0 and vs(200) == 0.8 and vs(250) < 0
    assert int(critical(1.8219448E-04, 1.0)) == int(412084.3) and feq(critical(0.3587903, 1.0), 4.163977)
    assert feq(optimal(38.49693), 96.73663)
    # and the frame filter: linear, removes the mean (the "infinite" critical dose at the origin makes both distances to the optimal
    # dose equal, so the origin is zeroed: "It forces the origin to 0", movie_filter_dose.cpp:152-156) and damps the rest
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((40, 56)), rng.standard_normal((40, 56))
    f = lambda v: oracle.dose_filter_frame(v, 1.0, 300, 2.0, 4.0)
    assert np.abs(f(a + 2 * b) - f(a) - 2 * f(b)).max() < 1e-12
    assert abs(f(a).mean()) < 1e-12 and f(a).std() < a.std()
