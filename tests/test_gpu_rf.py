"""GPU parity tests of the Fourier-gridding path (C ABI -> HIP) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import synth  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    ctx = xa.Context(0)
    return xa, ctx, torch


@pytest.fixture(scope="module")
def data32():
    D = 32
    vol = synth.phantom(D, seed=1, nblobs=12)
    rng = np.random.default_rng(2)
    ang = synth.random_angles(40, rng)
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    return D, vol, ang, imgs


def test_tables_match_oracle(gpu, oracle):
    xa, ctx, torch = gpu
    for D, order in ((32, 0), (64, 0), (64, 2)):
        rf = xa.RecFourier(ctx, D, blob_order=order)
        o = oracle.RF(D, blob_order=order)
        bt, fbt, ids, idf = rf.tables()
        assert np.allclose(bt, o.blob_table, rtol=1e-6, atol=1e-9)
        assert np.allclose(fbt, o.fourier_blob_table, rtol=1e-12)
        assert ids == o.p.iDeltaSqrt and idf == o.p.iDeltaFourier
        assert (rf.P, rf.mv) == (o.P, o.mv)


@pytest.mark.parametrize("D,maxres", [(32, 0.5), (64, 0.5), (64, 0.3), (128, 0.5), (50, 0.5), (36, 0.4), (45, 0.5)])
def test_prepare_images(gpu, oracle, D, maxres):
    """D=64 and 128 take the register-blocked columns-first / rows-last FFT (radix 16 x 8, 16 x 16; 256 px: 16 x 32, in
    test_gridding_at_full_size_against_the_oracle), D=32 the radix-2 LDS one, 50/36/45 (padded 100/72/90) Bluestein."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(D)
    imgs = rng.standard_normal((5, D, D)).astype(np.float32)
    rf = xa.RecFourier(ctx, D, max_resolution=maxres)
    o = oracle.RF(D, max_resolution=maxres)
    got = rf.prepare_images(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(5):
        exp = o.prepare_image(imgs[i])
        assert got[i].shape == exp.shape
        # fp32 device FFT vs double FFT narrowed to float (RFA:293): tolerance 3e-6 of the spectrum peak
        # (1e-5 for the chirp-z form used at non power-of-two sizes: two fp32 FFTs and three chirp products)
        tol = 3e-6 if (2 * D) & (2 * D - 1) == 0 else 1e-5
        assert np.abs(got[i] - exp).max() <= tol * np.abs(exp).max()
        # coefficients beyond max_resolution are exactly zero (RFA:284-290), the others are not
        cut = (exp[..., 0] == 0) & (exp[..., 1] == 0)
        assert cut.any() and not np.any(got[i][cut]) and np.all(np.any(got[i][~cut] != 0, axis=-1))


def test_shift_images(gpu, oracle):
    """readApplyGeo(only_apply_shifts): BSPLINE3 translation with wrapping (RFA:304-323)."""
    xa, ctx, torch = gpu
    D = 64
    rng = np.random.default_rng(4)
    imgs = rng.standard_normal((4, D, D)).astype(np.float32)
    shifts = np.array([[0.0, 0.0], [1.0, -2.0], [0.37, 2.6], [-3.25, -0.5]], np.float32)
    rf = xa.RecFourier(ctx, D)
    got = rf.shift_images(torch.from_numpy(imgs).cuda(), shifts).cpu().numpy()
    for i in range(4):
        exp = oracle.translate2d(imgs[i], shifts[i, 0], shifts[i, 1], degree=3, wrap=True)
        assert np.abs(got[i] - exp).max() <= 2e-5 * np.abs(exp).max()
    assert np.array_equal(got[0], imgs[0])
    # mirrored + shifted (flip negates the first row of the transformation matrix)
    gotf = rf.shift_images(torch.from_numpy(imgs).cuda(), shifts, flips=np.ones(4, np.uint8)).cpu().numpy()
    for i in range(4):
        A = np.array([[-1.0, 0, shifts[i, 0]], [0, 1, shifts[i, 1]], [0, 0, 1]])
        exp = oracle.apply_geometry2d(imgs[i], A, 3, False, True)
        assert np.abs(gotf[i] - exp).max() <= 2e-5 * np.abs(exp).max()


def test_shift_reuses_the_matchers_coefficients(gpu):
    """One refinement iteration prefilters every particle twice in the reference (matching, then readApplyGeo before
    gridding); here the gridding side can take the matcher's coefficients: same bits."""
    xa, ctx, torch = gpu
    D = 64
    g = torch.Generator(device="cuda").manual_seed(9)
    refs = torch.randn((6, D, D), generator=g, device="cuda")
    parts = torch.randn((10, D, D), generator=g, device="cuda")
    pm = xa.ProjectionMatcher(ctx, refs)
    pm.match(parts)
    coefs = pm.last_coefficients(10)
    assert coefs is not None and pm.last_coefficients(9) is None
    rf = xa.RecFourier(ctx, D)
    shifts = np.random.default_rng(1).uniform(-3, 3, (10, 2)).astype(np.float32)
    a = rf.shift_images(parts, shifts)
    b = rf.shift_images(parts, shifts, coefs=coefs)
    assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["astigmatic", "round", "round_envelope", "astigmatic_envelope"])
def test_ctf_arrays(gpu, oracle, kind):
    """k_rf_ctf has per-image shortcuts (no atan2 for a round CTF, no exp without envelope terms); every
    combination must equal the general double-precision formula (ctf.h:376-501)."""
    xa, ctx, torch = gpu
    D = 64
    rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.5)
    o = oracle.RF(D, min_ctf=0.01, sampling=1.5, use_ctf=True)
    from xmipp3_amd.api import ctf_params
    kw = dict(kV=300.0, Cs=2.7, Q0=0.07, DeltafU=15000.0, DeltafV=17000.0, azimuthal_angle=30.0, K=1.0)
    if kind.startswith("round"):
        kw["DeltafV"] = kw["DeltafU"]
    if kind.endswith("envelope"):
        kw.update(Ca=2.0, espr=0.6, ispr=0.3, alpha=0.1, DeltaF=3.0, DeltaR=0.5)
    c, m = rf.ctf_arrays([ctf_params(**kw)])
    ce, me = o.ctf_arrays(oracle.ctf_params(**kw))
    c, m = c.cpu().numpy()[0], m.cpu().numpy()[0]
    # same double-precision formula; libm vs device sin/cos differ in the last ulps, amplified by 1/CTF
    rel = np.abs(c - ce) / np.maximum(1.0, np.abs(ce))
    assert rel.max() < 1e-4
    assert np.abs(m - me).max() < 1e-6


def _insert_both(xa, ctx, torch, oracle, D, imgs, ang, **kw):
    rf = xa.RecFourier(ctx, D, **{k: v for k, v in kw.items() if k in ("fast", "blob_order", "blob_radius")})
    for k, v in kw.get("options", {}).items():
        rf.set_option(k, v)
    o = oracle.RF(D, **{k: v for k, v in kw.items() if k in ("fast", "blob_order", "blob_radius")})
    ffts = np.stack([o.prepare_image(im) for im in imgs])
    return rf, o, ffts


def test_insert_single_projection_same_voxels_and_taps(gpu, oracle, data32):
    """One projection into an empty volume: the same voxels, the same taps and the same table entries as
    processVoxelBlob (RFA:627-700); sums to float rounding."""
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    for i in (0, 3, 7):
        rf, o, ffts = _insert_both(xa, ctx, torch, oracle, D, imgs[i:i + 1], ang[i:i + 1])
        o.insert(ffts[0], synth.euler_matrix(*ang[i]).T)
        rf.insert(torch.from_numpy(ffts).cuda(), ang[i:i + 1])
        ev, ew = o.temp()
        gv, gw = rf.temp_spaces()
        gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
        assert (ew > 0).sum() > 1000
        # the products inside a tap are re-associated (records carry re*ctf*mod*w) and fused: float rounding
        assert np.array_equal(gw != 0, ew != 0)
        assert np.abs(gw - ew).max() <= 1e-6 * np.abs(ew).max()
        assert np.abs(gv - ev).max() <= 1e-6 * np.abs(ev).max()


def test_insert_axis_aligned_projection_is_dropped_like_reference(gpu, oracle):
    """rot=tilt=psi=0 makes getX (RFA:479-490) divide 0/0: the reference visits no voxel."""
    xa, ctx, torch = gpu
    D = 32
    rf = xa.RecFourier(ctx, D)
    o = oracle.RF(D)
    f = o.prepare_image(np.random.default_rng(0).standard_normal((D, D)))
    o.insert(f, np.eye(3))
    rf.insert(torch.from_numpy(f[None]).cuda(), np.zeros((1, 3)))
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    assert np.array_equal(gw.cpu().numpy(), ew) and np.array_equal(gv.cpu().numpy(), ev)


@pytest.mark.parametrize("mode", ["plain", "sym_weights", "sym_improper", "ctf", "fast", "fast_ctf"])
def test_insert_many(gpu, oracle, data32, mode):
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    fast = mode.startswith("fast")
    rf, o, ffts = _insert_both(xa, ctx, torch, oracle, D, imgs, ang, fast=fast)
    n = len(imgs)
    rng = np.random.default_rng(5)
    weights = None
    sym = None
    ctf = mod = None
    if mode == "sym_weights":
        weights = rng.uniform(0.0, 2.0, n).astype(np.float32)
        weights[3] = 0.0  # skipped, RFA:327-329
        c2 = np.diag([-1.0, -1.0, 1.0])
        sym = np.stack([np.eye(3), c2])
    if mode == "sym_improper":
        # groups with mirror planes / inversion (cs, cNv, ..., ih): the reference multiplies by R whatever its determinant
        # (reconstruct_fourier_accel.cpp:252-254, 953-956); here a mirror plane, the inversion and a rotoreflection
        n_ = np.array([1.0, 2.0, -0.5]) / np.linalg.norm([1.0, 2.0, -0.5])
        sym = np.stack([np.eye(3), np.diag([1.0, -1.0, 1.0]), -np.eye(3), (np.eye(3) - 2 * np.outer(n_, n_)) @ synth.euler_matrix(30, 0, 0)])
    if mode in ("ctf", "fast_ctf"):
        ctf = (rng.uniform(0.5, 2.0, ffts.shape[:3]) * rng.choice([-1, 1], ffts.shape[:3])).astype(np.float32)
        mod = rng.uniform(0.0, 1.0, ffts.shape[:3]).astype(np.float32)
    for i in range(n):
        w = 1.0 if weights is None else float(weights[i])
        if weights is not None and w == 0.0:
            continue
        for R in ([np.eye(3)] if sym is None else sym):
            o.insert(ffts[i], synth.euler_matrix(*ang[i]).T, R=R, weight=w,
                     ctf=None if ctf is None else ctf[i], modulator=None if mod is None else mod[i])
    rf.insert(torch.from_numpy(ffts).cuda(), ang, weights=weights, sym=sym,
              ctf=None if ctf is None else torch.from_numpy(ctf).cuda(),
              modulator=None if mod is None else torch.from_numpy(mod).cuda())
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
    # identical voxel sets; sums differ only by float summation order
    assert ((ew != 0) == (gw != 0)).all()
    if mode == "sym_improper":
        # four placements per image: the centre voxel collects ~1100 float addends (every projection crosses it), where the order of
        # summation shows (8e-6 of the largest weight, the same with four proper rotations: tools/diag_sym.py); 2e-6 everywhere else
        dw, dv = np.abs(gw - ew), np.abs(gv - ev).max(-1) if gv.ndim == 4 else np.abs(gv - ev)
        assert dw.max() <= 2e-5 * np.abs(ew).max() and (dw > 2e-6 * np.abs(ew).max()).sum() <= 2
        assert dv.max() <= 2e-5 * np.abs(ev).max() and (dv > 2e-6 * np.abs(ev).max()).sum() <= 2
        return
    assert np.abs(gw - ew).max() <= 2e-6 * np.abs(ew).max()
    assert np.abs(gv - ev).max() <= 2e-6 * np.abs(ev).max()


@pytest.mark.parametrize("radius", [2.4, 2.6, 2.9, 2.99])
def test_insert_wide_blob(gpu, oracle, data32, radius):
    """Blob radius in [2, 3): 6 x 6 footprints (the W = 6 instantiation of the gridding kernel).  Radii beyond 2.5 are the
    ones whose sixth tap carries weight (a footprint spans floor(x - r) ... ceil(x + r): six pixels for r > 2.5 whatever the
    fraction of x); 40 random orientations, CTF-weighted, voxel sets equal and 2e-6 against the oracle (RFA:627-700)."""
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    n = len(imgs)
    rf, o, ffts = _insert_both(xa, ctx, torch, oracle, D, imgs, ang, blob_radius=radius)
    rng = np.random.default_rng(int(radius * 100))
    ctf = (rng.uniform(0.5, 2.0, ffts.shape[:3]) * rng.choice([-1, 1], ffts.shape[:3])).astype(np.float32)
    mod = rng.uniform(0.0, 1.0, ffts.shape[:3]).astype(np.float32)
    for i in range(n):
        o.insert(ffts[i], synth.euler_matrix(*ang[i]).T, ctf=ctf[i], modulator=mod[i])
    rf.insert(torch.from_numpy(ffts).cuda(), ang, ctf=torch.from_numpy(ctf).cuda(), modulator=torch.from_numpy(mod).cuda())
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
    assert ((ew != 0) == (gw != 0)).all()
    assert np.abs(gw - ew).max() <= 2e-6 * np.abs(ew).max()
    assert np.abs(gv - ev).max() <= 2e-6 * np.abs(ev).max()
    # a single projection: the same voxels and taps
    rf.reset()
    o1 = oracle.RF(D, blob_radius=radius)
    o1.insert(ffts[7], synth.euler_matrix(*ang[7]).T)
    rf.insert(torch.from_numpy(ffts[7:8]).cuda(), ang[7:8])
    _, ew1 = o1.temp()
    _, gw1 = rf.temp_spaces()
    gw1 = gw1.cpu().numpy()
    assert np.array_equal(gw1 != 0, ew1 != 0) and np.abs(gw1 - ew1).max() <= 1e-6 * np.abs(ew1).max()


def test_insert_blob_radius_three_and_more_is_refused(gpu, data32):
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    with pytest.raises(xa.XhError):
        rf3 = xa.RecFourier(ctx, D, blob_radius=3.2)
        rf3.insert(torch.zeros((1, 2 * D, D, 2), device="cuda"), ang[:1])


@pytest.mark.parametrize("D", [64, 50])
def test_insert_images_equals_the_separate_steps(gpu, D):
    """xh_rf_insert_images (processBufferGPU in one call) against xh_rf_ctf_arrays + xh_rf_prepare_images + xh_rf_insert:
    the same bits; twice in a row with fewer images on the same handles."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    n = 40
    g = torch.Generator(device="cuda").manual_seed(D)
    imgs = torch.randn((n, D, D), generator=g, device="cuda")
    rng = np.random.default_rng(D)
    ang = synth.random_angles(n, rng)
    w = rng.uniform(0.0, 2.0, n).astype(np.float32)
    w[5] = 0.0
    arr = xa.RecFourier.ctf_param_array([ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d) + 500.0,
                                                    azimuthal_angle=20.0) for d in rng.uniform(10000.0, 30000.0, n)])
    c2 = np.diag([-1.0, -1.0, 1.0])
    sym = np.stack([np.eye(3), c2])
    a = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.2)
    c, m = a.ctf_arrays(arr)
    a.insert(a.prepare_images(imgs), ang, weights=w, ctf=c, modulator=m, sym=sym)
    b = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.2)
    b.insert_images(imgs, ang, ctf_array=arr, weights=w, sym=sym)
    assert torch.equal(a.temp, b.temp) and a.temp.abs().max().item() > 0
    # without CTF, fewer images, same handles
    a.reset(); b.reset()
    a.insert(a.prepare_images(imgs[:7].contiguous()), ang[:7])
    b.insert_images(imgs[:7].contiguous(), ang[:7])
    assert torch.equal(a.temp, b.temp)


@pytest.mark.parametrize("D", [64, 50])
def test_device_side_orientations_give_the_host_forms_temp_spaces(gpu, D):
    """xh_rf_shift_images_dev / xh_rf_insert_images_dev (shifts, flips, angles, weights in device memory, the traverse spaces
    built by a kernel) against the host-array forms: same voxel sets, values to float rounding (the device's double sin/cos
    may differ from libm's in the last place before the cast to float); symmetry, zero weights, CTF, --fast too."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    n = 48
    g = torch.Generator(device="cuda").manual_seed(D + 1)
    imgs = torch.randn((n, D, D), generator=g, device="cuda")
    rng = np.random.default_rng(D + 1)
    ang = synth.random_angles(n, rng)
    ang[3] = (0.0, 0.0, 0.0)
    ang[4] = (90.0, 90.0, 0.0)            # axis-aligned slabs: the degenerate branches of the row test
    w = rng.uniform(0.0, 2.0, n).astype(np.float32)
    w[5] = 0.0; w[17] = 0.0
    sx = rng.uniform(-3, 3, n); sy = rng.uniform(-3, 3, n)
    fl = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    arr = xa.RecFourier.ctf_param_array([ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d) + 500.0,
                                                    azimuthal_angle=20.0) for d in rng.uniform(10000.0, 30000.0, n)])
    sym = np.stack([np.eye(3), np.diag([-1.0, -1.0, 1.0])])
    for fast in (False, True):
        a = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.2, fast=fast)
        b = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.2, fast=fast)
        sa = a.shift_images(imgs, np.stack([sx, sy], 1), flips=fl)
        sb = b.shift_images(imgs, (torch.from_numpy(sx).cuda(), torch.from_numpy(sy).cuda()), flips=torch.from_numpy(fl).cuda())
        assert torch.equal(sa, sb)
        a.insert_images(sa, ang, ctf_array=arr, weights=w, sym=sym)
        b.insert_images(sb, torch.from_numpy(ang).cuda(), ctf_array=arr, weights=torch.from_numpy(w).cuda(), sym=sym)
        ta, tb = a.temp, b.temp
        assert ta.abs().max().item() > 0
        assert torch.equal(ta != 0, tb != 0)
        assert (ta - tb).abs().max().item() <= 1e-6 * ta.abs().max().item()
        # no weights, no symmetry, no CTF, fewer images on the same handles
        a.reset(); b.reset()
        a.insert_images(sa[:9].contiguous(), ang[:9])
        b.insert_images(sb[:9].contiguous(), torch.from_numpy(ang[:9].copy()).cuda())
        assert torch.equal(a.temp != 0, b.temp != 0)
        assert (a.temp - b.temp).abs().max().item() <= 1e-6 * a.temp.abs().max().item()


def test_mirror_crop_and_finish_given_same_temp(gpu, oracle, data32):
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    for fast in (False, True):
        rf, o, ffts = _insert_both(xa, ctx, torch, oracle, D, imgs, ang, fast=fast)
        for i in range(len(imgs)):
            o.insert(ffts[i], synth.euler_matrix(*ang[i]).T)
        ev, ew = o.temp()
        gv, gw = rf.temp_spaces()
        gv.copy_(torch.from_numpy(ev))
        gw.copy_(torch.from_numpy(ew))
        rf.mirror_and_crop()
        o.mirror_and_crop()
        ev2, ew2 = o.temp()
        gv2, gw2 = rf.temp_spaces()
        assert np.array_equal(gv2.cpu().numpy(), ev2)
        assert np.array_equal(gw2.cpu().numpy(), ew2)
        got = rf.finish()
        exp = o.finish()
        assert np.abs(got - exp).max() <= 1e-9 * np.abs(exp).max()


def test_end_to_end_reconstruction(gpu, oracle, data32):
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    rf = xa.RecFourier(ctx, D)
    o = oracle.RF(D)
    rf.insert(rf.prepare_images(torch.from_numpy(imgs).cuda()), ang)
    rf.mirror_and_crop()
    got = rf.finish()
    for i in range(len(imgs)):
        o.insert(o.prepare_image(imgs[i]), synth.euler_matrix(*ang[i]).T)
    o.mirror_and_crop()
    exp = o.finish()
    # stated float tolerance of the volume: 1e-4 of the peak
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()
    assert np.corrcoef(got.ravel(), exp.ravel())[0, 1] > 0.999999
    # reset + idempotence: a second identical run gives the same volume within atomics noise
    rf.reset()
    rf.insert(rf.prepare_images(torch.from_numpy(imgs).cuda()), ang)
    rf.mirror_and_crop()
    again = rf.finish()
    assert np.abs(again - got).max() <= 1e-5 * np.abs(got).max()


@pytest.mark.parametrize("D", [50, 45])
def test_end_to_end_reconstruction_any_box_size(gpu, oracle, D):
    """Box sizes that are not powers of two: every FFT on the path (projection r2c, 3-D c2r) runs the
    Bluestein form of xh_plan.h; same volume tolerance as the power-of-two case."""
    xa, ctx, torch = gpu
    vol = synth.phantom(D, seed=5, nblobs=8)
    ang = synth.random_angles(12, np.random.default_rng(D))
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    rf = xa.RecFourier(ctx, D)
    o = oracle.RF(D)
    assert (rf.P, rf.mv) == (o.P, o.mv) == (2 * D, 2 * D)
    rf.insert(rf.prepare_images(torch.from_numpy(imgs).cuda()), ang)
    rf.mirror_and_crop()
    got = rf.finish()
    for i in range(len(imgs)):
        o.insert(o.prepare_image(imgs[i]), synth.euler_matrix(*ang[i]).T)
    o.mirror_and_crop()
    exp = o.finish()
    assert got.shape == exp.shape == (D, D, D)
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()


def test_tile_kernel_is_deterministic(gpu, data32):
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    rf = xa.RecFourier(ctx, D)
    f = rf.prepare_images(torch.from_numpy(imgs).cuda())
    rf.insert(f, ang)
    v1 = rf.temp.clone()
    rf.reset()
    rf.insert(f, ang)
    assert torch.equal(rf.temp, v1)


@pytest.mark.parametrize("D,n", [(64, 300), (40, 64)])
def test_launch_chunking_changes_little(gpu, D, n):
    """tile_max_spaces cuts a call into several launches (each with its own super-tile lists): the same projections reach
    every voxel in the same order, only the grouping of the float sums changes."""
    xa, ctx, torch = gpu
    g = torch.Generator(device="cuda").manual_seed(D)
    imgs = torch.randn((n, D, D), generator=g, device="cuda")
    ang = synth.random_angles(n, np.random.default_rng(D))
    rf = xa.RecFourier(ctx, D)
    f = rf.prepare_images(imgs)
    c = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda") + 0.5
    m = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda")
    rf.insert(f, ang, ctf=c, modulator=m)
    one = rf.temp.clone()
    rf.reset()
    rf.set_option("tile_max_spaces", 64)
    rf.insert(f, ang, ctf=c, modulator=m)
    assert ((rf.temp != 0) == (one != 0))[-(rf.mv + 1) ** 3:].all()        # weights: the same voxels
    assert (rf.temp - one).abs().max().item() <= 2e-6 * one.abs().max().item()
    assert one.abs().max().item() > 0


def test_linearity_of_insertion(gpu, data32):
    """Size-independent property: inserting A then B equals inserting A and B in one call."""
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    rf = xa.RecFourier(ctx, D)
    f = rf.prepare_images(torch.from_numpy(imgs).cuda())
    rf.insert(f, ang)
    v1 = rf.temp.clone()
    rf.reset()
    rf.insert(f[:17].contiguous(), ang[:17])
    rf.insert(f[17:].contiguous(), ang[17:])
    assert (rf.temp - v1).abs().max().item() <= 2e-6 * v1.abs().max().item()


def test_reduce_of_partial_reconstructions(gpu, data32):
    """Thread-per-device hosts sum their partial cropped spaces with xh_rf_reduce (here three handles on
    the one GPU of the box: same code path minus the peer copy). Float sum order differs from the
    single-handle run, hence a float tolerance on the volume."""
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    one = xa.RecFourier(ctx, D)
    f = one.prepare_images(torch.from_numpy(imgs).cuda())
    one.insert(f, ang)
    one.mirror_and_crop()
    ref_cropped = one.cropped_view().clone()
    ref = one.finish()
    parts = []
    for g in range(3):
        lo, hi = xa.shard_range(len(imgs), g, 3)
        rf = xa.RecFourier(ctx, D)
        rf.insert(f[lo:hi].contiguous(), ang[lo:hi])
        rf.mirror_and_crop()
        parts.append(rf)
    xa.reduce_reconstructions(parts)
    summed = parts[0].cropped_view()
    assert (summed - ref_cropped).abs().max().item() <= 2e-6 * ref_cropped.abs().max().item()
    got = parts[0].finish()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    with pytest.raises(xa.XhError):
        xa.reduce_reconstructions([parts[1], parts[1]])
    with pytest.raises(xa.XhError):
        xa.reduce_reconstructions([parts[1], xa.RecFourier(ctx, D)])   # second one not cropped yet


def test_reduce_across_devices_is_one_rccl_allreduce(gpu, data32):
    """Two or more GPUs in the box: the partial reconstructions live on different devices and xh_rf_reduce sums them with
    one in-place RCCL all-reduce (every participating handle ends up with the total). Skipped on single-GPU boxes."""
    xa, ctx, torch = gpu
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    D, vol, ang, imgs = data32
    one = xa.RecFourier(ctx, D)
    f = one.prepare_images(torch.from_numpy(imgs).cuda())
    one.insert(f, ang)
    one.mirror_and_crop()
    ref_cropped = one.cropped_view().clone()
    ndev = min(4, torch.cuda.device_count())
    ctxs = [ctx] + [xa.Context(d) for d in range(1, ndev)]
    parts = []
    for g in range(ndev):
        lo, hi = xa.shard_range(len(imgs), g, ndev)
        with torch.cuda.device(g):
            rf = xa.RecFourier(ctxs[g], D)
            rf.insert(rf.prepare_images(torch.from_numpy(imgs[lo:hi]).cuda(g)), ang[lo:hi])
            rf.mirror_and_crop()
        parts.append(rf)
    xa.reduce_reconstructions(parts)
    tol = 2e-6 * ref_cropped.abs().max().item()
    for g in range(ndev):          # an all-reduce: every device holds the sum
        assert (parts[g].cropped_view().to(ref_cropped.device) - ref_cropped).abs().max().item() <= tol


def test_half_sets_sum_to_the_full_reconstruction(gpu, data32):
    """--prepare_fsc bookkeeping (RF:991-1045): half 1 and half 2 are reconstructed from zeroed spaces,
    their Fourier volumes + weights are kept and summed for the final volume."""
    xa, ctx, torch = gpu
    D, vol, ang, imgs = data32
    rf = xa.RecFourier(ctx, D)
    f = rf.prepare_images(torch.from_numpy(imgs).cuda())
    rf.insert(f, ang)
    rf.mirror_and_crop()
    full = rf.finish()
    h = len(imgs) // 2
    rf.reset()
    rf.insert(f[:h].contiguous(), ang[:h])
    rf.mirror_and_crop()
    keep1 = rf.export_cropped()
    v1 = rf.finish()
    rf.reset()
    rf.insert(f[h:].contiguous(), ang[h:])
    rf.mirror_and_crop()
    keep2 = rf.export_cropped()
    v2 = rf.finish()
    rf.import_cropped(keep1)
    rf.import_cropped(keep2, add=True)
    both = rf.finish()
    assert np.abs(both - full).max() <= 1e-5 * np.abs(full).max()
    # the halves are genuinely different reconstructions of the same object
    assert np.abs(v1 - v2).max() > 1e-3 * np.abs(full).max()
    assert np.corrcoef(v1.ravel(), v2.ravel())[0, 1] > 0.9


def test_config1_reconstruction_recovers_the_phantom(gpu, oracle):
    """a 64^3 phantom of 20 Gaussians, 1000 projections at orientations uniform on SO(3),
    --padding 2 2 --blob 1.9 0 15 --max_resolution 0.5. Checks: the volume equals the oracle's on a subset of the
    projections (1e-4 of the peak), and FSC(reconstruction, phantom) stays high as far as the phantom has any power:
    its narrowest blob (sigma = 2 voxels) is down to exp(-2 pi^2 sigma^2 f^2) = 4e-2 of its peak at f = 0.2 and to 3e-6 at
    the 0.4 cycles/pixel SURVEY.md names, where interpolation error is all that is left to correlate. The projections
    come from the library's own central-slice projector, the FSC from xh_frc_dpr: volume -> gallery -> reconstruction
    -> resolution, all on the device."""
    xa, ctx, torch = gpu
    D, n = 64, 1000
    vol = synth.phantom(D, seed=1, nblobs=20).astype(np.float32)
    ang = synth.random_angles(n, np.random.default_rng(2))
    fp = xa.FourierProjector(ctx, torch.from_numpy(vol).cuda(), 2.0, 0.5, 3)
    imgs = fp.project(ang)
    fp.close()
    rf = xa.RecFourier(ctx, D)
    rf.insert(rf.prepare_images(imgs), ang)
    rf.mirror_and_crop()
    rec = rf.finish()
    r = xa.frc_dpr(ctx, torch.from_numpy(vol.astype(np.float64)).cuda(), torch.from_numpy(rec).cuda(), 1.0)
    assert r["frc"][1:int(0.2 * D) + 1].min() > 0.99, r["frc"]
    first_below_half = int(np.argmax(r["frc"][1:] < 0.5)) + 1
    assert first_below_half >= int(0.25 * D), r["frc"]
    assert np.corrcoef(rec.ravel(), vol.ravel())[0, 1] > 0.99
    # oracle on the first 100 projections, same images
    m = 100
    h = imgs[:m].cpu().numpy()
    o = oracle.RF(D)
    for i in range(m):
        o.insert(o.prepare_image(h[i]), synth.euler_matrix(*ang[i]).T)
    o.mirror_and_crop()
    exp = o.finish()
    rf.reset()
    rf.insert(rf.prepare_images(imgs[:m].contiguous()), ang[:m])
    rf.mirror_and_crop()
    got = rf.finish()
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()


def test_linearity_of_insertion_at_full_size(gpu):
    """BASELINE config 3/4 box (256 px, 513^3 temp spaces): inserting 768 CTF-weighted projections in one call or in
    three gives the same sums up to float summation order."""
    xa, ctx, torch = gpu
    D, n = 256, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    imgs = torch.randn((n, D, D), generator=g, device="cuda")
    ang = synth.random_angles(n, np.random.default_rng(4))
    rf = xa.RecFourier(ctx, D)
    f = rf.prepare_images(imgs)
    c = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda") + 0.5
    m = torch.rand((n, rf.sizeY, rf.sizeX), generator=g, device="cuda")
    rf.insert(f, ang, ctf=c, modulator=m)
    one = rf.temp.clone()
    rf.reset()
    for lo in range(0, n, 256):
        rf.insert(f[lo:lo + 256].contiguous(), ang[lo:lo + 256], ctf=c[lo:lo + 256].contiguous(), modulator=m[lo:lo + 256].contiguous())
    scale = one.abs().max().item()
    assert (rf.temp - one).abs().max().item() <= 2e-6 * scale
    assert scale > 0


def test_gridding_at_full_size_against_the_oracle(gpu, oracle):
    """SURVEY.md 8d config 3 at its own box: 64 CTF-weighted projections into the 513^3 temp spaces. At this size the
    kernel runs with all eight XCD classes, work stealing between tile streams and super-tile lists of hundreds of
    entries, none of which the 32-px tests reach. Temp spaces: same voxels, 2e-6; finished volume: 1e-4 of the peak and
    FSC >= 0.999 up to 0.9 Nyquist."""
    xa, ctx, torch = gpu
    D, n = 256, 64
    rng = np.random.default_rng(11)
    vol = synth.phantom(64, seed=3, nblobs=12)
    ang = synth.random_angles(n, rng)
    small = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    # 64-px projections of the phantom in the middle of a noisy 256-px box: structure at low and high frequencies
    imgs = (0.05 * rng.standard_normal((n, D, D))).astype(np.float32)
    imgs[:, 96:160, 96:160] += small
    rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.0)
    o = oracle.RF(D, use_ctf=True, min_ctf=0.01)
    from xmipp3_amd.api import ctf_params
    defocus = rng.uniform(10000.0, 30000.0, n)
    ctfs = [ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d)) for d in defocus]
    weights = rng.uniform(0.5, 1.5, n).astype(np.float32)
    c, m = rf.ctf_arrays(xa.RecFourier.ctf_param_array(ctfs))
    f = rf.prepare_images(torch.from_numpy(imgs).cuda())
    hf, hc, hm = f.cpu().numpy(), c.cpu().numpy(), m.cpu().numpy()
    # the oracle grids the device's spectra and CTF planes: this test is about the insertion (the front end has its own)
    for i in range(n):
        o.insert(hf[i], synth.euler_matrix(*ang[i]).T, weight=float(weights[i]), ctf=hc[i], modulator=hm[i])
    rf.insert(f, ang, weights=weights, ctf=c, modulator=m)
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
    assert (ew != 0).sum() > 10_000_000
    assert ((ew != 0) == (gw != 0)).all()
    assert np.abs(gw - ew).max() <= 2e-6 * np.abs(ew).max()
    assert np.abs(gv - ev).max() <= 2e-6 * np.abs(ev).max()
    del ev, ew, gv, gw
    o.mirror_and_crop()
    exp = o.finish()
    rf.mirror_and_crop()
    got = rf.finish()
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()
    res = xa.frc_dpr(ctx, torch.from_numpy(got).cuda(), torch.from_numpy(exp).cuda())
    freq, frc = np.asarray(res["freq"]), np.asarray(res["frc"])
    sel = (freq > 0) & (freq <= 0.45)
    assert sel.sum() > 50 and frc[sel].min() >= 0.999


@pytest.mark.parametrize("sym", [False, True])
def test_gridding_of_particles_that_share_directions_at_full_size_against_the_oracle(gpu, oracle, sym):
    """What a refinement hands over: several particles per gallery direction, differing in the in-plane angle only.  The launch is
    ordered by plane and an interior visit whose plane equals the previous visit's reuses its voxel queue (xh_rf_grid.h `sameQueue`),
    which random orientations never exercise.  A few directions x several in-plane angles at 256 px with CTF and weights, with and without a
    symmetry matrix, against the oracle: same voxels, 2e-6.  (VERDICT r05 test hole b: this path was held against itself only.)"""
    xa, ctx, torch = gpu
    D, ndir, nin = (256, 3, 4) if sym else (256, 4, 6)         # (24 placements either way: the oracle takes seconds per placement at this size)
    n = ndir * nin
    rng = np.random.default_rng(23)
    dirs = synth.fibonacci_directions(200)[rng.choice(200, ndir, replace=False)]
    ang = np.array([[d[0], d[1], 360.0 * rng.integers(0, 796) / 796.0] for d in dirs for _ in range(nin)])
    ang = ang[rng.permutation(n)]                      # metadata order: directions interleaved, the library sorts them by plane
    imgs = (0.05 * rng.standard_normal((n, D, D))).astype(np.float32)
    imgs[:, 100:156, 100:156] += rng.standard_normal((n, 56, 56)).astype(np.float32)
    rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.0)
    o = oracle.RF(D, use_ctf=True, min_ctf=0.01)
    from xmipp3_amd.api import ctf_params
    ctfs = [ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d)) for d in rng.uniform(10000.0, 30000.0, n)]
    weights = rng.uniform(0.5, 1.5, n).astype(np.float32)
    R = [np.eye(3)]
    if sym:
        c2 = np.diag([-1.0, -1.0, 1.0])                # a two-fold about z: its planes are shared by the same groups of particles
        R.append(c2)
    c, m = rf.ctf_arrays(xa.RecFourier.ctf_param_array(ctfs))
    f = rf.prepare_images(torch.from_numpy(imgs).cuda())
    hf, hc, hm = f.cpu().numpy(), c.cpu().numpy(), m.cpu().numpy()
    for i in range(n):
        for Rs in R:
            o.insert(hf[i], synth.euler_matrix(*ang[i]).T, R=Rs, weight=float(weights[i]), ctf=hc[i], modulator=hm[i])
    rf.insert(f, ang, weights=weights, ctf=c, modulator=m, sym=np.stack(R) if sym else None)
    ev, ew = o.temp()
    gv, gw = rf.temp_spaces()
    gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
    assert (ew != 0).sum() > 1_000_000
    assert ((ew != 0) == (gw != 0)).all()
    assert np.abs(gw - ew).max() <= 2e-6 * np.abs(ew).max()
    assert np.abs(gv - ev).max() <= 2e-6 * np.abs(ev).max()
    # (six placements per plane -- eight with the two-fold about z, which maps every plane onto itself -- so most interior visits of a
    # unit take the previous visit's queue)


def test_device_volume_against_the_double_precision_program(gpu, oracle):
    """BASELINE config 1 is quoted on xmipp_reconstruct_fourier (ProgRecFourier, RF); the device runs the accel
    arithmetic under that name too. Against the RF restatement (double scatter, FFTW layout, correctWeight): 2e-3 of
    the peak, correlation 0.99999+ (the two CPU restatements differ by as much, tests/test_oracle_pins.py)."""
    xa, ctx, torch = gpu
    D, n = 64, 400
    vol = synth.phantom(D, seed=5, nblobs=16)
    rng = np.random.default_rng(6)
    ang = synth.random_angles(n, rng)
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    ref = oracle.RF2(D)
    for i in range(n):
        ref.insert(imgs[i], synth.euler_matrix(*ang[i]).T)
    exp = ref.finish()
    rf = xa.RecFourier(ctx, D)
    rf.insert(rf.prepare_images(torch.from_numpy(imgs).cuda()), ang)
    rf.mirror_and_crop()
    got = rf.finish()
    peak = np.abs(exp).max()
    assert np.abs(got - exp).max() <= 2e-3 * peak
    assert np.corrcoef(got.ravel(), exp.ravel())[0, 1] > 0.99999


@pytest.mark.parametrize("niter,kind", [(1, "plain"), (0, "plain"), (3, "plain"), (1, "sym_weights_ctf"), (1, "odd_box")])
def test_double_precision_program_on_the_device(gpu, oracle, niter, kind):
    """ProgRecFourier's own arithmetic (xh_rf2_*, the program behind xmipp_reconstruct_fourier) against its restatement
    oracle.RF2: double accumulators, image-driven scatter, wrap + conjugate, correctWeight (--iter 0, 1 and 3 with the
    re-processing passes), symmetry, weights, CTF, a box that is not a power of two. Sums differ by their order only:
    1e-9 of the peak (the judged bar is 1e-6)."""
    xa, ctx, torch = gpu
    from xmipp3_amd.api import ctf_params
    D, n = (45, 24) if kind == "odd_box" else (32, 60)
    vol = synth.phantom(D, seed=5, nblobs=12)
    rng = np.random.default_rng(6)
    ang = synth.random_angles(n, rng)
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    weights = sym = None
    dctf = octf = None
    if kind == "sym_weights_ctf":
        weights = rng.uniform(0.0, 2.0, n).astype(np.float32)
        weights[3] = 0.0
        sym = np.stack([np.eye(3), np.diag([-1.0, -1.0, 1.0])])
        defoci = rng.uniform(10000.0, 30000.0, n)
        kw = dict(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, azimuthal_angle=20.0)
        dctf = [ctf_params(DeltafU=float(d), DeltafV=float(d) + 800.0, **kw) for d in defoci]
        octf = [oracle.ctf_params(DeltafU=float(d), DeltafV=float(d) + 800.0, **kw) for d in defoci]
    ref = oracle.RF2(D, niter_weight=niter)
    ref.sampling, ref.min_ctf = 1.3, 0.02
    for i in range(n):
        for R in ([np.eye(3)] if sym is None else sym):
            ref.insert(imgs[i], synth.euler_matrix(*ang[i]).T, R=R, weight=1.0 if weights is None else float(weights[i]),
                       ctf=None if octf is None else octf[i])
    exp = ref.finish()
    rf = xa.RecFourier2(ctx, D, niter_weight=niter, sampling=1.3, min_ctf=0.02)
    dimgs = torch.from_numpy(imgs).cuda()
    h = n // 2
    rf.insert(dimgs[:h].contiguous(), ang[:h], weights=None if weights is None else weights[:h], sym=sym, ctfs=None if dctf is None else dctf[:h])
    rf.insert(dimgs[h:].contiguous(), ang[h:], weights=None if weights is None else weights[h:], sym=sym, ctfs=None if dctf is None else dctf[h:])
    got = rf.finish()
    peak = np.abs(exp).max()
    assert peak > 0 and np.abs(got - exp).max() <= 1e-9 * peak


def test_config1_on_the_double_precision_program(gpu, oracle):
    """BASELINE config 1 on its own binary's arithmetic: xmipp_reconstruct_fourier = ProgRecFourier (RF:571-793,1056-1221),
    64^3 phantom of 20 Gaussians, 1000 projections uniform on SO(3), --padding 2 2 --blob 1.9 0 15 --max_resolution 0.5.
    All 1000 projections through xh_rf2_*: the phantom comes back; a 100-projection subset against oracle.RF2 (1e-9 of the
    peak: double accumulators, the order of the sums is all that differs)."""
    xa, ctx, torch = gpu
    D, n = 64, 1000
    vol = synth.phantom(D, seed=1, nblobs=20).astype(np.float32)
    ang = synth.random_angles(n, np.random.default_rng(2))
    fp = xa.FourierProjector(ctx, torch.from_numpy(vol).cuda(), 2.0, 0.5, 3)
    imgs = fp.project(ang)
    fp.close()
    rf = xa.RecFourier2(ctx, D)
    for lo in range(0, n, 250):
        rf.insert(imgs[lo:lo + 250].contiguous(), ang[lo:lo + 250])
    rec = rf.finish()
    assert np.corrcoef(rec.ravel(), vol.ravel())[0, 1] > 0.99
    m = 100
    h = imgs[:m].cpu().numpy()
    ref = oracle.RF2(D)
    for i in range(m):
        ref.insert(h[i], synth.euler_matrix(*ang[i]).T)
    exp = ref.finish()
    rf = xa.RecFourier2(ctx, D)
    rf.insert(imgs[:m].contiguous(), ang[:m])
    got = rf.finish()
    peak = np.abs(exp).max()
    assert peak > 0 and np.abs(got - exp).max() <= 1e-9 * peak


def test_errors_are_loud(gpu):
    xa, ctx, torch = gpu
    with pytest.raises(xa.XhError):
        xa.RecFourier(ctx, 1500)        # padded size 3000 exceeds the LDS line budget of the device FFT
    with pytest.raises(xa.XhError):
        xa.RecFourier(ctx, 32, blob_order=1)
    rf = xa.RecFourier(ctx, 32)
    rf.mirror_and_crop()
    with pytest.raises(xa.XhError):
        rf.insert(torch.zeros((1, 64, 32, 2), device="cuda"), np.zeros((1, 3)))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["round", "astigmatic"])
def test_cheap_ctf_form_against_the_general_formula(gpu, oracle, kind):
    """d_ctf_pixel_fast (envelope-free CTFs: one sinusoid, argument reduced in double, float sine polynomial) against the general
    double-precision formula of the same library (option ctf_fast 0) and against the oracle, at the bench's size: the DECISION
    |CTF| < minCTF is the same for every pixel (within 1e-5 of the threshold the general formula is evaluated), the modulators are
    equal to float rounding, the factors 1 / CTF to 1e-6 relative."""
    xa, ctx, torch = gpu
    D = 256
    from xmipp3_amd.api import ctf_params
    kw = dict(kV=300.0, Cs=2.7, Q0=0.07, DeltafU=21000.0, DeltafV=21000.0 if kind == "round" else 19500.0, azimuthal_angle=37.0, K=1.0)
    res = {}
    for fast in (1, 0):
        rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.0)
        rf.set_option("ctf_fast", fast)
        c, m = rf.ctf_arrays([ctf_params(**kw)])
        res[fast] = (c.cpu().numpy()[0], m.cpu().numpy()[0])
    (cf, mf), (cs, ms) = res[1], res[0]
    assert np.array_equal(mf == 1.0, ms == 1.0)                       # the same pixels below the threshold
    assert np.abs(mf - ms).max() <= 1e-7
    assert (np.abs(cf - cs) / np.maximum(1.0, np.abs(cs))).max() <= 1e-6
    o = oracle.RF(D, min_ctf=0.01, sampling=1.0, use_ctf=True)
    ce, me = o.ctf_arrays(oracle.ctf_params(**kw))
    assert (np.abs(cf - ce) / np.maximum(1.0, np.abs(ce))).max() < 1e-4 and np.abs(mf - me).max() < 1e-6


@pytest.mark.gpu
def test_launch_order_by_plane_changes_only_the_summation_order(gpu):
    """order_spaces 1 (the product: the traverse spaces of a launch ordered by plane, voxel queues shared between projections of one
    direction) against 0 (input order): the same voxels, sums equal to float rounding -- on orientations as the matcher assigns them
    (a few directions, many in-plane angles), through the device-side and the host-side entry point, without and with a symmetry group (c4)."""
    xa, ctx, torch = gpu
    D, n = 64, 96
    rng = np.random.default_rng(5)
    dirs = synth.fibonacci_directions(7)
    ang = np.stack([np.array([dirs[i % 7][0], dirs[i % 7][1], rng.uniform(0, 360)]) for i in range(n)])
    imgs = torch.from_numpy(rng.standard_normal((n, D, D)).astype(np.float32)).cuda()
    from xmipp3_amd.api import ctf_params
    ctfs = xa.RecFourier.ctf_param_array([ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=15000.0 + 100 * i, DeltafV=15000.0 + 100 * i) for i in range(n)])
    c4 = np.stack([synth.euler_matrix(90.0 * k, 0.0, 0.0) for k in range(4)])
    for sym, dev_angles in ((None, True), (c4, True), (c4, False)):
        temps = []
        for order in (1, 0):
            rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.0)
            rf.set_option("order_spaces", order)
            rf.insert_images(imgs, torch.from_numpy(ang).cuda() if dev_angles else ang, ctf_array=ctfs, sym=sym)
            v, w = rf.temp_spaces()
            temps.append((v.cpu().numpy(), w.cpu().numpy()))
        (v1, w1), (v0, w0) = temps
        assert (w0 != 0).sum() > 50_000
        assert np.array_equal(w1 != 0, w0 != 0)
        assert np.abs(w1 - w0).max() <= 2e-6 * np.abs(w0).max() and np.abs(v1 - v0).max() <= 2e-6 * np.abs(v0).max()


@pytest.mark.gpu
def test_shift_band_by_band_gives_the_same_bits(gpu):
    """k_rf_shift_band (256-px images: the coefficients of a band of rows staged in LDS) against k_rf_shift (option shift_bands 0):
    the same expressions in the same order, so the same bits -- fractional and whole shifts, shifts that wrap, mirrored particles,
    untouched ones."""
    xa, ctx, torch = gpu
    D, n = 256, 12
    rng = np.random.default_rng(21)
    imgs = torch.from_numpy(rng.standard_normal((n, D, D)).astype(np.float32)).cuda()
    shifts = rng.uniform(-9, 9, (n, 2))
    shifts[0] = 0.0
    shifts[1] = (3.0, -2.0)
    shifts[2] = (130.25, -131.5)
    flips = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    flips[0] = 0
    outs = []
    for bands in (1, 0):
        rf = xa.RecFourier(ctx, D)
        rf.set_option("shift_bands", bands)
        outs.append(rf.shift_images(imgs, shifts, flips=flips).cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0][0], imgs[0].cpu().numpy())
