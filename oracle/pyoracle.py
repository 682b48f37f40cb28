"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (xmipp3_amd) never imports it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=sys.stderr)
    return so


class CtfParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "Tm", "kV", "DeltafU", "DeltafV", "azimuthal_angle", "Cs", "Ca", "espr", "ispr", "alpha",
        "DeltaF", "DeltaR", "Q0", "K", "envR0", "envR1", "envR2", "phase_shift", "VPP_radius")]


class RfParams(C.Structure):
    _fields_ = [("imgSize", C.c_int), ("padding_proj", C.c_double), ("padding_vol", C.c_double),
                ("maxResolution", C.c_double), ("blob_radius", C.c_double), ("blob_order", C.c_int),
                ("blob_alpha", C.c_double), ("useFast", C.c_int), ("useCTF", C.c_int),
                ("isPhaseFlipped", C.c_int), ("minCTF", C.c_double), ("iTs", C.c_double),
                ("paddedImgSize", C.c_int), ("maxVolumeIndexX", C.c_int),
                ("maxVolumeIndexYZ", C.c_int), ("iDeltaSqrt", C.c_float),
                ("iDeltaFourier", C.c_float)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(os.environ.get("XO_ORACLE_LIB") or build())          # XO_ORACLE_LIB: another build of the same sources (sanitizer runs)
    vp = C.c_void_p
    i, d = C.c_int, C.c_double
    sig = {
        "xo_fft1d_r2c": (None, [c_double_p, i, c_double_p]),
        "xo_fft1d_c2r": (None, [c_double_p, i, c_double_p]),
        "xo_fft2d_r2c": (None, [c_double_p, i, i, c_double_p]),
        "xo_fft2d_c2r": (None, [c_double_p, i, i, c_double_p]),
        "xo_fft3d_c2r": (None, [c_double_p, i, i, i, c_double_p]),
        "xo_fft1d_c2c": (None, [c_double_p, i, i, c_double_p]),
        "xo_fft_idx2digfreq": (d, [i, i]),
        "xo_bspline3_prefilter2d": (None, [c_double_p, i, i, c_double_p]),
        "xo_bspline3_interp2d": (d, [c_double_p, i, i, i, i, d, d]),
        "xo_polar_nsam": (i, [i]),
        "xo_polar_layout": (None, [i, i, c_int32_p, c_int32_p, c_int32_p]),
        "xo_polar_from_cartesian_bspline": (None, [c_double_p, i, i, i, i, i, i, d, d, c_double_p]),
        "xo_polar_avg_std": (None, [c_double_p, i, i, c_double_p, c_double_p]),
        "xo_polar_fft_rings": (None, [c_double_p, i, i, i, c_double_p]),
        "xo_rotational_correlation": (None, [c_double_p, c_double_p, i, i, c_double_p]),
        "xo_euler_matrix": (None, [d, d, d, c_double_p]),
        "xo_apply_geometry2d": (None, [i, c_double_p, i, i, c_double_p, i, i, c_double_p]),
        "xo_rotate2d": (None, [i, c_double_p, i, i, d, i, c_double_p]),
        "xo_translate2d": (None, [i, c_double_p, i, i, d, d, i, c_double_p]),
        "xo_correlation_matrix": (None, [c_double_p, c_double_p, i, i, c_double_p]),
        "xo_best_shift_mcorr": (d, [c_double_p, i, i, i, c_double_p, c_double_p]),
        "xo_best_shift": (d, [c_double_p, c_double_p, i, i, i, c_double_p, c_double_p]),
        "xo_correlation_index": (d, [c_double_p, c_double_p, C.c_size_t]),
        "xo_pm_create": (vp, [i, i, i, i, c_double_p, c_double_p, i]),
        "xo_pm_destroy": (None, [vp]),
        "xo_pm_nsam_outer": (i, [vp]),
        "xo_pm_ncoef": (i, [vp]),
        "xo_pm_ref_coefs": (c_double_p, [vp, i]),
        "xo_pm_ref_sigma": (d, [vp, i]),
        "xo_pm_prepare_particle": (None, [vp, c_double_p, d, d, c_double_p, c_double_p, c_double_p]),
        "xo_pm_match": (None, [vp, c_double_p, i, c_int32_p, c_int32_p, i, i, c_int32_p, c_int32_p,
                               i, i, c_int32_p, c_int32_p, c_uint8_p, c_double_p]),
        "xo_pm_match_thr": (None, [vp, c_double_p, i, c_int32_p, c_int32_p, i, i, c_int32_p, c_int32_p,
                                   i, i, i, c_int32_p, c_int32_p, c_uint8_p, c_double_p]),
        "xo_pm_corr_rows": (None, [vp, c_double_p, i, c_double_p]),
        "xo_pm_translate": (None, [vp, c_double_p, i, c_int32_p, c_int32_p, c_uint8_p, d, i,
                                   c_double_p, c_double_p, c_double_p]),
        "xo_kaiser_value": (d, [d, d, d, i]),
        "xo_kaiser_fourier_value": (d, [d, d, d, i]),
        "xo_bessi0": (d, [d]),
        "xo_bessi1": (d, [d]),
        "xo_ctf_defaults": (None, [C.POINTER(CtfParams)]),
        "xo_ctf_value_pure_nok": (d, [C.POINTER(CtfParams), d, d]),
        "xo_ctf_lambda": (d, [C.POINTER(CtfParams)]),
        "xo_fa_global_alignment": (C.c_int, [c_double_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, C.c_float, C.c_float, C.c_float,
                                             c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "xo_fa_local_alignment": (C.c_int, [c_double_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                            c_double_p, C.POINTER(C.c_int)]),
        "xo_fa_bspline_shift": (None, [c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
        "xo_fa_apply_bspline": (None, [c_double_p, C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p]),
        "xo_fa_solve": (None, [c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int)]),
        "xo_ctf_phase_flip": (None, [c_double_p, C.c_int, C.c_int, C.POINTER(CtfParams), C.c_int]),
        "xo_ctf_wiener2d": (None, [c_double_p, C.c_int, C.c_int, C.POINTER(CtfParams), d, d, C.c_int, C.c_int, d, C.c_int]),
        "xo_rf_create": (vp, [C.POINTER(RfParams)]),
        "xo_rf_destroy": (None, [vp]),
        "xo_rf_blob_table_sqrt": (c_float_p, [vp]),
        "xo_rf_fourier_blob_table": (c_double_p, [vp]),
        "xo_rf_prepare_image": (None, [vp, c_double_p, c_float_p]),
        "xo_rf_ctf_arrays": (None, [vp, C.POINTER(CtfParams), c_float_p, c_float_p]),
        "xo_rf_insert": (None, [vp, c_float_p, c_float_p, c_float_p, c_double_p, c_double_p, C.c_float]),
        "xo_rf_temp_volume": (c_float_p, [vp]),
        "xo_rf_temp_weights": (c_float_p, [vp]),
        "xo_rf_reset": (None, [vp]),
        "xo_rf_mirror_and_crop": (None, [vp]),
        "xo_rf_finish": (None, [vp, c_double_p]),
        "xo_rf_hermitian_and_weights": (None, [vp]),
        "xo_num_threads": (i, []),
        "xo_rf2_create": (vp, [i, d, d, d, d, i, d, i]),
        "xo_rf2_destroy": (None, [vp]),
        "xo_rf2_weights": (c_double_p, [vp]),
        "xo_rf2_fourier": (c_double_p, [vp]),
        "xo_rf2_vol_pad": (i, [vp]),
        "xo_rf2_insert": (None, [vp, c_double_p, c_double_p, c_double_p, d, C.POINTER(CtfParams), d, d, i, i]),
        "xo_rf2_weights_begin": (None, [vp]),
        "xo_rf2_weights_iter_begin": (None, [vp]),
        "xo_rf2_weights_iter_end": (None, [vp]),
        "xo_rf2_weights_end": (None, [vp]),
        "xo_rf2_finish": (None, [vp, c_double_p]),
        "xo_fp_create": (vp, [c_double_p, i, d, d, i]),
        "xo_fp_destroy": (None, [vp]),
        "xo_fp_padded_size": (i, [vp]),
        "xo_fp_coef_dim": (i, [vp]),
        "xo_fp_coef_start": (i, [vp]),
        "xo_fp_coefs": (c_double_p, [vp, i]),
        "xo_fp_project": (None, [vp, d, d, d, c_double_p, c_double_p]),
        "xo_frc_dpr": (i, [c_double_p, c_double_p, i, i, i, d, i, i, d, d, c_double_p, c_double_p, c_double_p,
                           c_double_p, c_double_p, c_double_p]),
        "xo_fa_local_patch_shifts_f32": (i, [c_float_p, i, i, i, c_double_p, c_double_p, i, C.c_float, C.c_float, C.c_float, i, i, i, i, i, c_uint8_p,
                                             c_double_p, c_double_p, c_int32_p]),
        "xo_fa_bin_frame": (None, [c_double_p, i, i, i, i, c_double_p]),
        "xo_es_rotation_corr_len": (i, [i, i]),
        "xo_es_polar_rotation": (None, [c_double_p, c_double_p, i, i, i, i, c_double_p, c_double_p]),
        "xo_es_shifts": (None, [c_float_p, c_float_p, i, i, i, i, c_float_p]),
        "xo_es_rotations": (None, [c_float_p, c_float_p, i, i, i, i, c_float_p]),
        "xo_es_iterative_pass": (None, [c_float_p, c_float_p, i, i, i, i, i, i, i, c_float_p, c_float_p]),
        "xo_es_iterative_alignment": (None, [c_float_p, c_float_p, i, i, i, i, i, i, c_float_p, c_float_p]),
        "xo_es_iterative_pass_ties": (None, [c_float_p, c_float_p, i, i, i, i, i, i, c_int32_p, c_int32_p, c_float_p, c_float_p]),
        "xo_es_test_population": (i, [i, i, c_float_p, c_float_p]),
        "xo_es_test_add_noise": (None, [c_float_p, C.c_size_t]),
        "xo_es_test_make_others": (None, [c_float_p, i, i, c_float_p, c_float_p, c_float_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _fp(a):
    return None if a is None else a.ctypes.data_as(c_float_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------- FFT
def fft2d_r2c(img):
    img = f64(img)
    y, x = img.shape
    out = np.empty((y, x // 2 + 1, 2))
    lib().xo_fft2d_r2c(_dp(img), y, x, _dp(out))
    return out[..., 0] + 1j * out[..., 1]


def fft2d_c2r(F, xdim):
    y = F.shape[0]
    buf = np.ascontiguousarray(np.stack([F.real, F.imag], -1), dtype=np.float64)
    out = np.empty((y, xdim))
    lib().xo_fft2d_c2r(_dp(buf), y, xdim, _dp(out))
    return out


def fft1d_r2c(v):
    v = f64(v)
    out = np.empty((len(v) // 2 + 1, 2))
    lib().xo_fft1d_r2c(_dp(v), len(v), _dp(out))
    return out[:, 0] + 1j * out[:, 1]


def fft1d_c2r(F, n):
    buf = np.ascontiguousarray(np.stack([F.real, F.imag], -1), dtype=np.float64)
    out = np.empty(n)
    lib().xo_fft1d_c2r(_dp(buf), n, _dp(out))
    return out


def fft1d_c2c(v, sign):
    buf = np.ascontiguousarray(np.stack([np.real(v), np.imag(v)], -1), dtype=np.float64)
    out = np.empty_like(buf)
    lib().xo_fft1d_c2c(_dp(buf), len(v), sign, _dp(out))
    return out[:, 0] + 1j * out[:, 1]


# ---------------------------------------------------------------- B-spline / geometry
def prefilter2d(img):
    img = f64(img)
    out = np.empty_like(img)
    lib().xo_bspline3_prefilter2d(_dp(img), img.shape[0], img.shape[1], _dp(out))
    return out


def rotate2d(img, ang, degree=3, wrap=False):
    img = f64(img)
    out = np.empty_like(img)
    lib().xo_rotate2d(degree, _dp(img), img.shape[0], img.shape[1], float(ang), int(wrap), _dp(out))
    return out


def translate2d(img, sx, sy, degree=1, wrap=True):
    img = f64(img)
    out = np.empty_like(img)
    lib().xo_translate2d(degree, _dp(img), img.shape[0], img.shape[1], float(sx), float(sy),
                         int(wrap), _dp(out))
    return out


def apply_geometry2d(img, A, degree, is_inv, wrap):
    img = f64(img)
    A = f64(A)
    out = np.empty_like(img)
    lib().xo_apply_geometry2d(degree, _dp(img), img.shape[0], img.shape[1], _dp(A), int(is_inv),
                              int(wrap), _dp(out))
    return out


def correlation_matrix(a, b):
    a, b = f64(a), f64(b)
    out = np.empty_like(a)
    lib().xo_correlation_matrix(_dp(a), _dp(b), a.shape[0], a.shape[1], _dp(out))
    return out


def best_shift(a, b, max_shift=-1):
    a, b = f64(a), f64(b)
    sx, sy = C.c_double(0), C.c_double(0)
    m = lib().xo_best_shift(_dp(a), _dp(b), a.shape[0], a.shape[1], max_shift, C.byref(sx), C.byref(sy))
    return sx.value, sy.value, m


def correlation_index(a, b):
    a, b = f64(a), f64(b)
    return lib().xo_correlation_index(_dp(a), _dp(b), a.size)


def euler_matrix(rot, tilt, psi):
    A = np.empty((3, 3))
    lib().xo_euler_matrix(rot, tilt, psi, _dp(A))
    return A


# ---------------------------------------------------------------- polar
def polar_layout(Ri, Ro):
    n = Ro - Ri + 1
    nsam = np.zeros(n, np.int32)
    ts, tc = C.c_int32(0), C.c_int32(0)
    lib().xo_polar_layout(Ri, Ro, _ip(nsam), C.byref(ts), C.byref(tc))
    return nsam, ts.value, tc.value


def polar_from_cartesian(coef, Ri, Ro, starty, startx, xoff=0.0, yoff=0.0):
    coef = f64(coef)
    _, ts, _ = polar_layout(Ri, Ro)
    out = np.empty(ts)
    lib().xo_polar_from_cartesian_bspline(_dp(coef), coef.shape[0], coef.shape[1], starty, startx,
                                          Ri, Ro, xoff, yoff, _dp(out))
    return out


def polar_avg_std(rings, Ri, Ro):
    rings = f64(rings)
    a, s = C.c_double(0), C.c_double(0)
    lib().xo_polar_avg_std(_dp(rings), Ri, Ro, C.byref(a), C.byref(s))
    return a.value, s.value


def polar_fft_rings(rings, Ri, Ro, conj):
    rings = f64(rings)
    _, _, tc = polar_layout(Ri, Ro)
    out = np.empty((tc, 2))
    lib().xo_polar_fft_rings(_dp(rings), Ri, Ro, int(conj), _dp(out))
    return out[:, 0] + 1j * out[:, 1]


def rotational_correlation(F1, F2, Ri, Ro):
    b1 = np.ascontiguousarray(np.stack([F1.real, F1.imag], -1), dtype=np.float64)
    b2 = np.ascontiguousarray(np.stack([F2.real, F2.imag], -1), dtype=np.float64)
    N = lib().xo_polar_nsam(Ro)
    out = np.empty(N)
    lib().xo_rotational_correlation(_dp(b1), _dp(b2), Ri, Ro, _dp(out))
    return out


# ---------------------------------------------------------------- projection matching
class PM:
    def __init__(self, refs, Ri=1, Ro=-1, Mctf=None, paddim=0):
        refs = f64(refs)
        self.nrefs, self.D, _ = refs.shape
        self.Ri = max(1, Ri)
        self.Ro = Ro if Ro >= 0 else self.D // 2 - 1
        m = None if Mctf is None else _dp(f64(Mctf))
        self._keep = (refs, Mctf)
        self.h = lib().xo_pm_create(self.D, Ri, Ro, self.nrefs, _dp(refs), m, paddim)
        self.N = lib().xo_pm_nsam_outer(self.h)
        self.ncoef = lib().xo_pm_ncoef(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xo_pm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def ref_coefs(self, r):
        p = lib().xo_pm_ref_coefs(self.h, r)
        a = np.ctypeslib.as_array(p, shape=(self.ncoef, 2)).copy()
        return a[:, 0] + 1j * a[:, 1]

    def ref_sigma(self, r):
        return lib().xo_pm_ref_sigma(self.h, r)

    def prepare_particle(self, img, xoff=0.0, yoff=0.0):
        img = f64(img)
        fP = np.empty((self.ncoef, 2))
        fPm = np.empty((self.ncoef, 2))
        s = C.c_double(0)
        lib().xo_pm_prepare_particle(self.h, _dp(img), xoff, yoff, _dp(fP), _dp(fPm), C.byref(s))
        return fP[:, 0] + 1j * fP[:, 1], fPm[:, 0] + 1j * fPm[:, 1], s.value

    def corr_rows(self, img, ref):
        img = f64(img)
        out = np.empty(2 * self.N)
        lib().xo_pm_corr_rows(self.h, _dp(img), ref, _dp(out))
        return out

    def match(self, particles, nbr_off=None, nbr_ids=None, parity=0, n_orient=1, xoff5d=None,
              yoff5d=None, nthreads=0, ref_threads=1):
        """ref_threads: the program's --thr (worker c takes the list positions i % ref_threads == c, the per-worker lists are merged,
        APM:631,1063-1108); it only matters where two correlation values are exactly equal"""
        particles = f64(particles)
        n = particles.shape[0]
        refno = np.empty((n, n_orient), np.int32)
        psi = np.empty((n, n_orient), np.int32)
        flip = np.empty((n, n_orient), np.uint8)
        cc = np.empty((n, n_orient))
        if nbr_off is not None:
            nbr_off = np.ascontiguousarray(nbr_off, np.int32)
            nbr_ids = np.ascontiguousarray(nbr_ids, np.int32)
        nt = 0
        if xoff5d is not None:
            xoff5d = np.ascontiguousarray(xoff5d, np.int32)
            yoff5d = np.ascontiguousarray(yoff5d, np.int32)
            nt = len(xoff5d)
        lib().xo_pm_match_thr(self.h, _dp(particles), n, _ip(nbr_off), _ip(nbr_ids), parity, n_orient,
                              _ip(xoff5d), _ip(yoff5d), nt, nthreads, int(ref_threads), _ip(refno), _ip(psi),
                              flip.ctypes.data_as(c_uint8_p), _dp(cc))
        return refno, psi, flip, cc

    def translate(self, particles, refno, psi_idx, flip, max_shift=-1.0, nthreads=0):
        particles = f64(particles)
        n = particles.shape[0]
        refno = np.ascontiguousarray(refno, np.int32).reshape(-1)
        psi_idx = np.ascontiguousarray(psi_idx, np.int32).reshape(-1)
        flip = np.ascontiguousarray(flip, np.uint8).reshape(-1)
        sx, sy, cc = np.empty(n), np.empty(n), np.empty(n)
        lib().xo_pm_translate(self.h, _dp(particles), n, _ip(refno), _ip(psi_idx),
                              flip.ctypes.data_as(c_uint8_p), float(max_shift), nthreads, _dp(sx),
                              _dp(sy), _dp(cc))
        return sx, sy, cc


# ---------------------------------------------------------------- reconstruction
def fa_global_alignment(frames, Ts=1.0, max_shift_px=50.0, max_res=30.0, dark=None, igain=None):
    """ProgMovieAlignmentCorrelation<double>::computeGlobalAlignment on frames [N, Y, X]: dict with the pair shifts, the
    frame shifts from the reference frame (movie pixels), the reference frame and the size of the reduced frames."""
    fr = f64(frames)
    N, Y, X = fr.shape
    rows = N * (N - 1) // 2
    bx, by, sx, sy = np.empty(rows), np.empty(rows), np.empty(N), np.empty(N)
    ref, nd = C.c_int(0), (C.c_int * 2)()
    dk = None if dark is None else f64(dark)
    ig = None if igain is None else f64(igain)
    rc = lib().xo_fa_global_alignment(_dp(fr), N, Y, X, None if dk is None else _dp(dk), None if ig is None else _dp(ig), Ts, max_shift_px,
                                      max_res, _dp(bx), _dp(by), _dp(sx), _dp(sy), C.byref(ref), nd)
    if rc != 0:
        raise ValueError("the correlation scale factor is >= 1 (checkSettings)")
    return {"bX": bx, "bY": by, "shiftX": sx, "shiftY": sy, "ref": ref.value, "new_dims": (nd[0], nd[1])}


def fa_local_alignment(frames, g_shift_x, g_shift_y, ref, Ts=1.0, max_shift_px=50.0, max_res=30.0, patches=(7, 7), patch_size=(500, 500),
                       patches_avg=3, control_points=(6, 6, 5)):
    """computeLocalAlignment of the CUDA program (movie_alignment_correlation_gpu.cpp:288-430) on corrected frames [N, Y, X]."""
    fr = f64(frames)
    N, Y, X = fr.shape
    px, py = patches
    lX, lY, lT = control_points
    shifts, centers = np.empty((py, px, N, 2)), np.empty((py, px, 2))
    cx, cy = np.empty(lX * lY * lT), np.empty(lX * lY * lT)
    dims = (C.c_int * 4)()
    gx, gy = f64(g_shift_x), f64(g_shift_y)
    rc = lib().xo_fa_local_alignment(_dp(fr), N, Y, X, _dp(gx), _dp(gy), int(ref), Ts, max_shift_px, max_res, px, py, patch_size[0], patch_size[1],
                                     patches_avg, lX, lY, lT, _dp(shifts), _dp(centers), _dp(cx), _dp(cy), dims)
    if rc != 0:
        raise ValueError("Movie is too small for local alignment.")
    return {"patch_shifts": shifts, "centers": centers, "coeffsX": cx, "coeffsY": cy, "dims": tuple(dims)}


def fa_local_patch_shifts(frames, g_shift_x, g_shift_y, ref, mask, Ts=1.0, max_shift_px=50.0, max_res=30.0, patches=(7, 7), patch_size=(500, 500), patches_avg=3):
    """xo_fa_local_alignment on float32 frames for the patches mask [py, px] marks: (patch_shifts [py, px, N, 2] -- NaN where
    not computed --, centers [py, px, 2], dims)"""
    fr = np.ascontiguousarray(frames, dtype=np.float32)
    N, Y, X = fr.shape
    px, py = patches
    gx, gy = f64(g_shift_x), f64(g_shift_y)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    assert m.shape == (py, px)
    shifts, centers = np.full((py, px, N, 2), np.nan), np.empty((py, px, 2))
    dims = np.zeros(4, np.int32)
    rc = lib().xo_fa_local_patch_shifts_f32(_fp(fr), N, Y, X, _dp(gx), _dp(gy), int(ref), Ts, max_shift_px, max_res, px, py, int(patch_size[0]), int(patch_size[1]),
                                            int(patches_avg), m.ctypes.data_as(c_uint8_p), _dp(shifts), _dp(centers), _ip(dims))
    if rc != 0:
        raise ValueError("movie too small for the patches")
    return shifts, centers, tuple(int(v) for v in dims)


def fa_bin_frame(frame, Yb, Xb):
    """--bin of the CUDA FlexAlign program (xo_fa_bin_frame)"""
    fr = f64(frame)
    out = np.empty((Yb, Xb))
    lib().xo_fa_bin_frame(_dp(fr), fr.shape[0], fr.shape[1], int(Yb), int(Xb), _dp(out))
    return out


def fa_bspline_shift(coeffsX, coeffsY, control_points, X, Y, N, x, y, n):
    sx, sy = C.c_double(), C.c_double()
    cx, cy = f64(coeffsX), f64(coeffsY)
    lib().xo_fa_bspline_shift(_dp(cx), _dp(cy), control_points[0], control_points[1], control_points[2], X, Y, N, int(x), int(y), int(n), C.byref(sx), C.byref(sy))
    return sx.value, sy.value


def fa_apply_bspline(frame, coeffsX, coeffsY, control_points, N, n):
    fr = f64(frame)
    out = np.empty_like(fr)
    cx, cy = f64(coeffsX), f64(coeffsY)
    lib().xo_fa_apply_bspline(_dp(fr), fr.shape[0], fr.shape[1], _dp(cx), _dp(cy), control_points[0], control_points[1], control_points[2], N, int(n), _dp(out))
    return out


def dose_filter_frame(frame, pixel_size, acc_voltage, dose_start, dose_finish):
    """One frame through ProgMovieFilterDose's filter."""
    L = lib()
    L.xo_dose_voltage_scaling.restype = C.c_double
    vs = L.xo_dose_voltage_scaling(C.c_double(acc_voltage))
    if vs < 0:
        raise ValueError("Bad acceleration voltage (must be 200 or 300 kV")
    out = f64(frame).copy()
    L.xo_dose_filter_frame(_dp(out), out.shape[0], out.shape[1], C.c_double(pixel_size), C.c_double(vs), C.c_double(dose_start), C.c_double(dose_finish))
    return out


def dose_scalars():
    L = lib()
    for n in ("xo_dose_voltage_scaling", "xo_dose_filter", "xo_dose_critical", "xo_dose_optimal"):
        getattr(L, n).restype = C.c_double
    return (lambda v: L.xo_dose_voltage_scaling(C.c_double(v)), lambda d, c: L.xo_dose_filter(C.c_double(d), C.c_double(c)),
            lambda f, s: L.xo_dose_critical(C.c_double(f), C.c_double(s)), lambda c: L.xo_dose_optimal(C.c_double(c)))


def fa_correlate(frames, max_dist):
    """CUDAFlexAlignCorrelate::run on real frames [N, Y, X]: positions (x, y) of the correlation maxima of all pairs i < j."""
    fr = f64(frames)
    N, Y, X = fr.shape
    pos = np.empty((N * (N - 1) // 2, 2))
    lib().xo_fa_correlate(_dp(fr), N, Y, X, C.c_double(max_dist), _dp(pos))
    return pos


def fa_solve(bX, bY, N, iterations=2):
    bx, by = f64(bX), f64(bY)
    sx, sy = np.empty(N), np.empty(N)
    ref = C.c_int(0)
    lib().xo_fa_solve(_dp(bx), _dp(by), N, iterations, _dp(sx), _dp(sy), C.byref(ref))
    return sx, sy, ref.value


def ctf_phase_flip(img, ctf, with_damping=False):
    """actualPhaseFlip (ctf_phase_flip.cpp:88-117); ctf.Tm = sampling rate of the image. with_damping: CTFDescription::correctPhase."""
    out = f64(img).copy()
    lib().xo_ctf_phase_flip(_dp(out), out.shape[0], out.shape[1], C.byref(ctf), int(with_damping))
    return out


def ctf_wiener2d(img, ctf, sampling_rate=1.0, pad=2.0, phase_flipped=False, is_isotropic=False, wiener_constant=-1.0, correct_envelope=False):
    """Wiener2D::applyWienerFilter (data/wiener2d.cpp:101-141) on one image"""
    out = f64(img).copy()
    lib().xo_ctf_wiener2d(_dp(out), out.shape[0], out.shape[1], C.byref(ctf), sampling_rate, pad, int(phase_flipped), int(is_isotropic),
                          wiener_constant, int(correct_envelope))
    return out


def ctf_params(**kw):
    p = CtfParams()
    lib().xo_ctf_defaults(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class RF:
    def __init__(self, D, padding_proj=2.0, padding_vol=2.0, max_resolution=0.5, blob_radius=1.9,
                 blob_order=0, blob_alpha=15.0, fast=False, use_ctf=False, phase_flipped=False,
                 min_ctf=0.01, sampling=1.0):
        p = RfParams()
        p.imgSize = D
        p.padding_proj, p.padding_vol = padding_proj, padding_vol
        p.maxResolution = max_resolution
        p.blob_radius, p.blob_order, p.blob_alpha = blob_radius, blob_order, blob_alpha
        p.useFast, p.useCTF, p.isPhaseFlipped = int(fast), int(use_ctf), int(phase_flipped)
        p.minCTF, p.iTs = min_ctf, 1.0 / sampling
        self.p = p
        self.h = lib().xo_rf_create(C.byref(p))
        self.D = D
        self.P = p.paddedImgSize
        self.mv = p.maxVolumeIndexYZ
        self.cropped = False

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xo_rf_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def blob_table(self):
        return np.ctypeslib.as_array(lib().xo_rf_blob_table_sqrt(self.h), shape=(10000,)).copy()

    @property
    def fourier_blob_table(self):
        return np.ctypeslib.as_array(lib().xo_rf_fourier_blob_table(self.h), shape=(10000,)).copy()

    def prepare_image(self, img):
        img = f64(img)
        out = np.empty((self.mv, self.mv // 2, 2), np.float32)
        lib().xo_rf_prepare_image(self.h, _dp(img), _fp(out))
        return out

    def ctf_arrays(self, ctf):
        c = np.empty((self.mv, self.mv // 2), np.float32)
        m = np.empty((self.mv, self.mv // 2), np.float32)
        lib().xo_rf_ctf_arrays(self.h, C.byref(ctf), _fp(c), _fp(m))
        return c, m

    def insert(self, fft, euler_T, R=None, weight=1.0, ctf=None, modulator=None):
        fft = np.ascontiguousarray(fft, np.float32)
        A = f64(euler_T)
        R = f64(np.eye(3) if R is None else R)
        if ctf is not None:
            ctf = np.ascontiguousarray(ctf, np.float32)
            modulator = np.ascontiguousarray(modulator, np.float32)
        lib().xo_rf_insert(self.h, _fp(fft), _fp(ctf), _fp(modulator), _dp(A), _dp(R), weight)

    def temp(self):
        mv = self.mv
        nx = (mv // 2 + 1) if self.cropped else (mv + 1)
        v = np.ctypeslib.as_array(lib().xo_rf_temp_volume(self.h), shape=(mv + 1, mv + 1, nx, 2)).copy()
        w = np.ctypeslib.as_array(lib().xo_rf_temp_weights(self.h), shape=(mv + 1, mv + 1, nx)).copy()
        return v, w

    def set_temp(self, v, w):
        mv = self.mv
        nx = (mv // 2 + 1) if self.cropped else (mv + 1)
        dv = np.ctypeslib.as_array(lib().xo_rf_temp_volume(self.h), shape=(mv + 1, mv + 1, nx, 2))
        dw = np.ctypeslib.as_array(lib().xo_rf_temp_weights(self.h), shape=(mv + 1, mv + 1, nx))
        dv[...] = v
        dw[...] = w

    def reset(self):
        lib().xo_rf_reset(self.h)
        self.cropped = False

    def mirror_and_crop(self):
        lib().xo_rf_mirror_and_crop(self.h)
        self.cropped = True

    def finish(self):
        out = np.empty((self.D, self.D, self.D))
        lib().xo_rf_finish(self.h, _dp(out))
        return out


def frc_dpr(m1, m2, sampling_rate=1.0, do_dpr=False, do_rfactor=False, min_freq=0.0, max_freq=0.5):
    """frc_dpr of xmippCore as called by resolution_fsc.cpp:179-203 -> dict of shell arrays (+ rfactor)."""
    m1, m2 = f64(m1), f64(m2)
    assert m1.shape == m2.shape
    shp = (1,) * (3 - m1.ndim) + m1.shape
    L = shp[2] // 2 + 1
    out = {k: np.zeros(L) for k in ("freq", "frc", "frc_noise", "dpr", "error_l2")}
    rf = np.full(1, -1.0)
    n = lib().xo_frc_dpr(_dp(m1), _dp(m2), shp[0], shp[1], shp[2], sampling_rate, int(do_dpr), int(do_rfactor),
                         min_freq, max_freq, _dp(out["freq"]), _dp(out["frc"]), _dp(out["frc_noise"]),
                         _dp(out["dpr"]), _dp(out["error_l2"]), _dp(rf))
    assert n == L
    out["rfactor"] = float(rf[0])
    return out


class FP:
    """FourierProjector (data/fourier_projection.cpp): vol is [z][y][x] with the Xmipp origin at D//2."""

    def __init__(self, vol, padding=2.0, max_freq=0.5, degree=3):
        vol = f64(vol)
        self.D = vol.shape[0]
        assert vol.shape == (self.D,) * 3
        self.h = lib().xo_fp_create(_dp(vol), self.D, float(padding), float(max_freq), int(degree))
        self.P = lib().xo_fp_padded_size(self.h)
        self.cdim = lib().xo_fp_coef_dim(self.h)
        self.cstart = lib().xo_fp_coef_start(self.h)

    def coefs(self):
        n = self.cdim ** 3
        re = np.ctypeslib.as_array(lib().xo_fp_coefs(self.h, 0), shape=(n,)).reshape((self.cdim,) * 3).copy()
        im = np.ctypeslib.as_array(lib().xo_fp_coefs(self.h, 1), shape=(n,)).reshape((self.cdim,) * 3).copy()
        return re, im

    def project(self, rot, tilt, psi, ctf=None):
        out = np.empty((self.D, self.D))
        c = None if ctf is None else f64(ctf)
        lib().xo_fp_project(self.h, float(rot), float(tilt), float(psi), _dp(c), _dp(out))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xo_fp_destroy(self.h)
                self.h = None
        except Exception:
            pass


class RF2:
    """ProgRecFourier (reconstruction/reconstruct_fourier.cpp), the double-precision scatter variant: images are kept so
    that the weight correction (--iter, NiterWeight) can replay them."""

    def __init__(self, D, padding_proj=2.0, padding_vol=2.0, max_resolution=0.5, blob_radius=1.9, blob_order=0,
                 blob_alpha=15.0, niter_weight=1):
        self.D, self.niter = D, niter_weight
        self.h = lib().xo_rf2_create(D, padding_proj, padding_vol, max_resolution, blob_radius, blob_order, blob_alpha,
                                     niter_weight)
        self.images = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xo_rf2_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _insert(self, img, A, R, weight, reprocess, ctf=None):
        lib().xo_rf2_insert(self.h, _dp(img), _dp(A), _dp(R), float(weight), None if ctf is None else C.byref(ctf), 1.0 / self.sampling,
                            self.min_ctf, int(self.phase_flipped), int(reprocess))

    sampling, min_ctf, phase_flipped = 1.0, 0.01, False      # --sampling, --minCTF, --phaseFlipped (RF:48-56)

    def insert(self, img, euler_T, R=None, weight=1.0, ctf=None):
        img = f64(img)
        A = f64(euler_T)
        R = f64(np.eye(3) if R is None else R)
        self.images.append((img, A, R, weight))
        self._insert(img, A, R, weight, 0, ctf)

    def finish(self, correct_weight=True):
        """correct_weight=False: finishComputations as --prepare_fsc calls it for the two halves (RF:991-1045), before correctWeight"""
        L = lib()
        if correct_weight:
            L.xo_rf2_weights_begin(self.h)                      # correctWeight, RF:1056-1101
            for _ in range(1, self.niter):
                L.xo_rf2_weights_iter_begin(self.h)
                for img, A, R, w in self.images:
                    self._insert(img, A, R, w, 1)
                L.xo_rf2_weights_iter_end(self.h)
            L.xo_rf2_weights_end(self.h)
        out = np.empty((self.D,) * 3, np.float64)
        L.xo_rf2_finish(self.h, _dp(out))
        return out


# ---------------------------------------------------------------- estimator chain (xo_estimators.cpp, xo_polar.cpp)
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def es_default_rings(D):
    """RotationEstimationSetting::getDefaultFirstRing / getDefaultLastRing (arotation_estimator.h:58-64)"""
    return max(2, D // 20), (D - 3) // 2


def es_polar_rotation(ref, others, first_ring=None, last_ring=None, with_corr=False):
    """PolarRotationEstimator OneToN (polar_rotation_estimator.cpp:49-99): degrees per image; optionally the 2N - 1 correlation rows"""
    ref, others = f64(ref), f64(others)
    n, D = others.shape[0], ref.shape[0]
    a, b = es_default_rings(D)
    first_ring, last_ring = first_ring or a, last_ring or b
    rot = np.empty(n)
    corr = np.empty((n, lib().xo_es_rotation_corr_len(first_ring, last_ring))) if with_corr else None
    lib().xo_es_polar_rotation(_dp(ref), _dp(others), n, D, first_ring, last_ring, _dp(rot), _dp(corr))
    return (rot, corr) if with_corr else rot


def es_shifts(ref, others, max_shift):
    """ShiftCorrEstimator::computeShift2DOneToN (shift_corr_estimator.cpp:201-283): [n][2] = (x, y) of the correlation maximum"""
    ref, others = _f32(ref), _f32(others)
    n, Y, X = others.shape
    out = np.empty((n, 2), np.float32)
    lib().xo_es_shifts(_fp(ref), _fp(others), n, Y, X, int(max_shift), _fp(out))
    return out


def es_iterative_alignment(ref, others, max_shift, iters=3, first_ring=None, last_ring=None, order=None):
    """IterativeAlignmentEstimator::compute(others, iters) (iterative_alignment_estimator.cpp:96-169); order = "RS" / "SR" runs one
    half only.  Returns poses [n][3][3] (float) and merits [n]."""
    ref, others = _f32(ref), _f32(others)
    n, D = others.shape[0], ref.shape[0]
    a, b = es_default_rings(D)
    first_ring, last_ring = first_ring or a, last_ring or b
    poses, merit = np.empty((n, 3, 3), np.float32), np.empty(n, np.float32)
    if order is None:
        lib().xo_es_iterative_alignment(_fp(ref), _fp(others), n, D, int(max_shift), first_ring, last_ring, int(iters), _fp(poses), _fp(merit))
    else:
        lib().xo_es_iterative_pass(_fp(ref), _fp(others), n, D, int(max_shift), first_ring, last_ring, int(iters), int(order == "RS"), _fp(poses), _fp(merit))
    return poses, merit


def es_iterative_reachable(ref, other, max_shift, iters=3, first_ring=None, last_ring=None, limit=64):
    """Every (pose, merit) IterativeAlignmentEstimator::compute can end at for ONE image when the arg-max ties of its shift steps --
    positions whose correlation value, as the float the reference compares, lies within two ulps of the maximum -- are resolved either
    way: the reference's own choice among them is made by the rounding of its float FFT.  Returns the list of (pose [3][3], merit, ties
    met) over both orders' choice trees combined the way compute() combines them (the better merit of RS / SR)."""
    ref, other = _f32(ref), _f32(other)
    D = ref.shape[0]
    a, b = es_default_rings(D)
    first_ring, last_ring = first_ring or a, last_ring or b

    def half(rs):
        out, todo = [], [()]
        while todo and len(out) < limit:
            pre = todo.pop()
            picks = np.zeros(iters, np.int32); picks[:len(pre)] = pre
            counts = np.zeros(iters, np.int32)
            pose, merit = np.empty((3, 3), np.float32), np.empty(1, np.float32)
            lib().xo_es_iterative_pass_ties(_fp(ref), _fp(other), D, int(max_shift), first_ring, last_ring, int(iters), int(rs),
                                            picks.ctypes.data_as(c_int32_p), counts.ctypes.data_as(c_int32_p), _fp(pose), _fp(merit))
            out.append((pose.copy(), float(merit[0]), int((counts > 1).sum())))
            # branch at the first step beyond the prefix that had a tie
            for k in range(len(pre), iters):
                if counts[k] > 1:
                    for c in range(1, int(counts[k])):
                        todo.append(tuple(picks[:k]) + (c,))
                    # (deeper steps of the c = 0 branch are explored through their own ties below)
            # ties after the first tied step on the all-zero continuation
            first = next((k for k in range(len(pre), iters) if counts[k] > 1), None)
            if first is not None:
                for k in range(first + 1, iters):
                    if counts[k] > 1:
                        for c in range(1, int(counts[k])):
                            todo.append(tuple(picks[:k]) + (c,))
        return out
    rs, sr = half(True), half(False)
    res = []
    for pr, mr, tr in rs:
        for ps, ms_, ts in sr:
            res.append((ps, ms_, tr + ts) if mr < ms_ else (pr, mr, tr + ts))
    return res


def es_test_population(draw, n):
    """Size, shifts and rotations of draw `draw` of IterativeAlignmentEstimator_Test's engine (aiterative_alignment_tests.h:107-121)"""
    sh, rot = np.empty((n, 2), np.float32), np.empty(n, np.float32)
    size = lib().xo_es_test_population(int(draw), int(n), _fp(sh), _fp(rot))
    return size, sh, rot


def es_test_make_others(ref, shifts, rotations):
    ref, shifts, rotations = _f32(ref), _f32(shifts), _f32(rotations)
    n, D = rotations.shape[0], ref.shape[0]
    out = np.empty((n, D, D), np.float32)
    lib().xo_es_test_make_others(_fp(ref), D, n, _fp(shifts), _fp(rotations), _fp(out))
    return out


def es_test_add_noise(data):
    data = _f32(data).copy()
    lib().xo_es_test_add_noise(_fp(data), data.size)
    return data
