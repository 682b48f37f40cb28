"""Host-side Python mirror of the two Xmipp programs' inner interfaces.

Names follow the reference (ProgRecFourierAccel / ProgAngularProjectionMatching members);
every method is a thin call into libxmipp_hip.so.  torch provides device memory (tensors),
the stream and torch.distributed -- plumbing only.
"""
import ctypes as C
import weakref

import numpy as np

from ._lib import CtfParams, RfParams, XhError, check, lib


def _torch():
    import torch
    return torch


def search5d_offsets(search5d_shift, search5d_step=2):
    """Translations of the 5-D search in the reference's order (APM:321-348)."""
    if search5d_step == 0:
        search5d_step = 1
    fin = search5d_shift + search5d_shift % search5d_step
    xs, ys = [], []
    for x in range(-fin, fin + 1, search5d_step):
        for y in range(-fin, fin + 1, search5d_step):
            if x * x + y * y <= search5d_shift * search5d_shift:
                xs.append(x)
                ys.append(y)
    return np.asarray(xs, np.int32), np.asarray(ys, np.int32)


def shard_range(n, rank, world):
    """Contiguous particle range of `rank` ([g*N/G, (g+1)*N/G))."""
    return (rank * n) // world, ((rank + 1) * n) // world


class Context:
    """xh_ctx bound to torch's current stream on `device`."""

    def __init__(self, device=0):
        torch = _torch()
        L = lib()
        if not torch.cuda.is_available():
            raise XhError("no HIP device visible to torch; xmipp3_amd has no CPU fallback")
        self.device = int(device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        check(L.xh_ctx_create(self.device, C.c_void_p(stream), C.byref(h)))
        self.h = h
        self.torch_device = torch.device("cuda", self.device)
        self._children = weakref.WeakSet()   # handles must be destroyed before their context

    def sync(self):
        check(lib().xh_ctx_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            for c in list(self._children):
                c.close()
            lib().xh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # HIP-event timer on the library stream (bench.py roofline)
    def timer(self):
        return _Timer(self)


class _Timer:
    def __init__(self, ctx):
        self.ctx = ctx
        self.t = C.c_void_p()
        check(lib().xh_timer_create(ctx.h, C.byref(self.t)))

    def start(self):
        check(lib().xh_timer_start(self.ctx.h, self.t))

    def stop(self):
        check(lib().xh_timer_stop(self.ctx.h, self.t))

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(lib().xh_timer_elapsed_ms(self.ctx.h, self.t, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self.ctx, "h", None):
                lib().xh_timer_destroy(self.ctx.h, self.t)
        except Exception:
            pass


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise XhError("device arguments must be contiguous cuda tensors")
    if dtype is not None and t.dtype != dtype:
        raise XhError(f"expected a {dtype} tensor, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ctf_params(**kw):
    p = CtfParams()
    lib().xh_ctf_defaults(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, float(v))
    return p


class RecFourier:
    """Device side of ProgRecFourierAccel (reconstruction/reconstruct_fourier_accel.cpp)."""

    def __init__(self, ctx, imgSize, padding_proj=2.0, padding_vol=2.0, max_resolution=0.5,
                 blob_radius=1.9, blob_order=0, blob_alpha=15.0, fast=False, phase_flipped=False,
                 min_ctf=0.01, sampling=1.0):
        torch = _torch()
        self.ctx = ctx
        p = RfParams(int(imgSize), padding_proj, padding_vol, max_resolution, blob_radius,
                     int(blob_order), blob_alpha, int(fast), int(phase_flipped), min_ctf, sampling)
        h = C.c_void_p()
        check(lib().xh_rf_create(ctx.h, C.byref(p), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        P, mv, sx, sy = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().xh_rf_sizes(h, C.byref(P), C.byref(mv), C.byref(sx), C.byref(sy)))
        self.D, self.P, self.mv, self.sizeX, self.sizeY = int(imgSize), P.value, mv.value, sx.value, sy.value
        # temp spaces are a torch tensor so that torch.distributed can all-reduce them in place
        self.temp = torch.zeros(lib().xh_rf_temp_floats(h), dtype=torch.float32, device=ctx.torch_device)
        check(lib().xh_rf_attach_temp(h, _ptr(self.temp)))
        self.cropped = False

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_rf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        check(lib().xh_rf_set_option(self.h, name.encode(), float(value)))

    def tables(self):
        bt = np.empty(10000, np.float32)
        fbt = np.empty(10000, np.float64)
        a, b = C.c_float(), C.c_float()
        check(lib().xh_rf_tables(self.h, _np_ptr(bt), _np_ptr(fbt), C.byref(a), C.byref(b)))
        return bt, fbt, a.value, b.value

    def reset(self):
        check(lib().xh_rf_reset(self.h))
        self.cropped = False

    def shift_images(self, imgs, shifts, flips=None, coefs=None):
        """Image::readApplyGeo(only_apply_shifts): shifts [n,2] = (shiftX, shiftY) on the host. coefs: device pointer
        (int) to the fp32 B-spline coefficients of imgs when the caller has them (ProjectionMatcher.last_coefficients)."""
        torch = _torch()
        n = imgs.shape[0]
        if isinstance(shifts, tuple) and torch.is_tensor(shifts[0]) and shifts[0].is_cuda:
            # (shiftX, shiftY) float64 tensors as ProjectionMatcher.translate returns them; flips: a uint8 tensor
            out = torch.empty_like(imgs)
            check(lib().xh_rf_shift_images_dev(self.h, _ptr(imgs, torch.float32), C.c_void_p(coefs) if coefs else None,
                                               _ptr(shifts[0], torch.float64), _ptr(shifts[1], torch.float64),
                                               _ptr(flips, torch.uint8), n, _ptr(out)))
            return out
        sh = np.ascontiguousarray(shifts, np.float32).reshape(n, 2)
        fl = None if flips is None else np.ascontiguousarray(flips, np.uint8).reshape(n)
        out = torch.empty_like(imgs)
        check(lib().xh_rf_shift_images_coefs(self.h, _ptr(imgs, torch.float32), C.c_void_p(coefs) if coefs else None,
                                             _np_ptr(sh), _np_ptr(fl), n, _ptr(out)))
        return out

    def prepare_images(self, imgs, out=None):
        """imgs: cuda float32 [n,D,D] -> [n, mv, mv/2, 2] float32 (centred half spectra)."""
        torch = _torch()
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        n = imgs.shape[0]
        if out is None:
            out = torch.empty((n, self.sizeY, self.sizeX, 2), dtype=torch.float32, device=imgs.device)
        check(lib().xh_rf_prepare_images(self.h, _ptr(imgs), n, _ptr(out)))
        return out

    @staticmethod
    def ctf_param_array(ctfs):
        """Pack a list of CtfParams once (reusable across calls)."""
        return (CtfParams * len(ctfs))(*ctfs)

    def ctf_arrays(self, ctfs):
        torch = _torch()
        n = len(ctfs)
        arr = ctfs if isinstance(ctfs, C.Array) else (CtfParams * n)(*ctfs)
        c = torch.empty((n, self.sizeY, self.sizeX), dtype=torch.float32, device=self.ctx.torch_device)
        m = torch.empty_like(c)
        check(lib().xh_rf_ctf_arrays(self.h, arr, n, _ptr(c), _ptr(m)))
        return c, m

    def insert(self, fft, angles, weights=None, ctf=None, modulator=None, sym=None):
        """angles: [n,3] (rot,tilt,psi) degrees on the host."""
        n = fft.shape[0]
        ang = np.ascontiguousarray(angles, np.float64).reshape(n, 3)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        s = None if sym is None else np.ascontiguousarray(sym, np.float64).reshape(-1, 9)
        f32 = _torch().float32
        check(lib().xh_rf_insert(self.h, _ptr(fft, f32), _ptr(ctf, f32), _ptr(modulator, f32), _np_ptr(ang), _np_ptr(w), n,
                                 _np_ptr(s), 0 if s is None else s.shape[0]))

    def insert_images(self, imgs, angles, ctf_array=None, weights=None, sym=None):
        """Shifted images [n,D,D] + CTF parameters (ctf_param_array) + orientations -> temp spaces in one call
        (xh_rf_insert_images: CTF planes, FFT and insertion on scratch owned by the handle)."""
        n = imgs.shape[0]
        torch = _torch()
        s = None if sym is None else np.ascontiguousarray(sym, np.float64).reshape(-1, 9)
        if torch.is_tensor(angles) and angles.is_cuda:
            # orientations (float64 [n,3]) and weights (float32 [n]) that never left the device
            assert angles.shape == (n, 3) and angles.is_contiguous()
            check(lib().xh_rf_insert_images_dev(self.h, _ptr(imgs, torch.float32), ctf_array, _ptr(angles, torch.float64),
                                                _ptr(weights, torch.float32), n, _np_ptr(s), 0 if s is None else s.shape[0]))
            return
        ang = np.ascontiguousarray(angles, np.float64).reshape(n, 3)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        check(lib().xh_rf_insert_images(self.h, _ptr(imgs, _torch().float32), ctf_array, _np_ptr(ang), _np_ptr(w), n,
                                        _np_ptr(s), 0 if s is None else s.shape[0]))

    def insert_matrices(self, fft, ainv, weights=None, ctf=None, modulator=None, sym=None):
        n = fft.shape[0]
        a = np.ascontiguousarray(ainv, np.float64).reshape(n, 9)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        s = None if sym is None else np.ascontiguousarray(sym, np.float64).reshape(-1, 9)
        f32 = _torch().float32
        check(lib().xh_rf_insert_matrices(self.h, _ptr(fft, f32), _ptr(ctf, f32), _ptr(modulator, f32), _np_ptr(a), _np_ptr(w),
                                          n, _np_ptr(s), 0 if s is None else s.shape[0]))

    def kernel_ms(self, reset=True):
        """(total ms, launches) of the gridding kernel since the last reset (HIP events on the stream)."""
        ms, n = C.c_double(), C.c_int64()
        check(lib().xh_rf_kernel_ms(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def temp_spaces(self):
        """(volume [mv+1,mv+1,nx,2], weights [mv+1,mv+1,nx]) views of the temp tensor."""
        d = self.mv + 1
        nx = (self.mv // 2 + 1) if self.cropped else d
        tot = d * d * nx
        return self.temp[:2 * tot].view(d, d, nx, 2), self.temp[2 * tot:3 * tot].view(d, d, nx)

    def mirror_and_crop(self):
        check(lib().xh_rf_mirror_and_crop(self.h))
        self.cropped = True

    def cropped_view(self):
        """Flat view [volume | weights] of the cropped spaces: the all-reduce payload."""
        assert self.cropped
        return self.temp[:lib().xh_rf_cropped_floats(self.h)]

    def export_cropped(self):
        """Copy of the cropped spaces (a torch tensor on the handle's device); finish() consumes the
        originals, so this is how a half-set is kept for the later sum (RF:991-1045)."""
        torch = _torch()
        out = torch.empty(lib().xh_rf_cropped_floats(self.h), dtype=torch.float32, device=self.ctx.torch_device)
        check(lib().xh_rf_cropped_export(self.h, _ptr(out)))
        return out

    def import_cropped(self, buf, add=False):
        assert buf.dtype == _torch().float32 and buf.numel() == lib().xh_rf_cropped_floats(self.h)
        check(lib().xh_rf_cropped_import(self.h, _ptr(buf), 1 if add else 0))
        self.cropped = True

    def finish(self, reuse=False):
        """The finaliser; returns the D^3 volume as a numpy float64 array in page-locked memory (134 MB at D=256; from pageable
        memory the copy is the longest part of the finaliser).  Page-locking takes 25-30 ms, three times the finaliser's kernels:
        a caller that is done with the previous result before it asks for the next one says reuse=True and gets the SAME buffer
        again (its previous contents are overwritten); by default every call returns a fresh array."""
        torch = _torch()
        out = getattr(self, "_fin_out", None) if reuse else None
        if out is None:
            try:
                pin = torch.empty((self.D, self.D, self.D), dtype=torch.float64, pin_memory=True)
                out = pin.numpy()
            except RuntimeError:
                out = np.empty((self.D, self.D, self.D), np.float64)
            if reuse:
                self._fin_out = out
        check(lib().xh_rf_finish(self.h, _np_ptr(out)))
        return out


class RecFourier2:
    """Device side of ProgRecFourier (reconstruction/reconstruct_fourier.cpp), the double-precision program behind the name
    xmipp_reconstruct_fourier. Keeps what it inserted so that finish() can replay it for --iter > 1 (correctWeight)."""

    def __init__(self, ctx, imgSize, padding_proj=2.0, padding_vol=2.0, max_resolution=0.5, blob_radius=1.9, blob_order=0, blob_alpha=15.0,
                 niter_weight=1, phase_flipped=False, min_ctf=0.01, sampling=1.0):
        self.ctx, self.D, self.niter = ctx, int(imgSize), int(niter_weight)
        p = RfParams(self.D, padding_proj, padding_vol, max_resolution, blob_radius, int(blob_order), blob_alpha, 0, int(phase_flipped), min_ctf, sampling)
        h = C.c_void_p()
        check(lib().xh_rf2_create(ctx.h, C.byref(p), self.niter, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        self._calls = []

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_rf2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _insert(self, imgs, ctfs, ang, w, s, reprocess):
        n = ang.shape[0]
        arr = None if ctfs is None else (ctfs if isinstance(ctfs, C.Array) else (CtfParams * n)(*ctfs))
        check(lib().xh_rf2_insert(self.h, _ptr(imgs), arr, _np_ptr(ang), _np_ptr(w), n, _np_ptr(s), 0 if s is None else s.shape[0], int(reprocess)))

    def insert(self, imgs, angles, weights=None, sym=None, ctfs=None):
        torch = _torch()
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        ang = np.ascontiguousarray(angles, np.float64).reshape(-1, 3)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        s = None if sym is None else np.ascontiguousarray(sym, np.float64).reshape(-1, 9)
        self._insert(imgs, ctfs, ang, w, s, 0)
        self._calls.append((ang, w, s))

    def finish(self):
        L = lib()
        check(L.xh_rf2_weights_step(self.h, 0))
        for _ in range(1, self.niter):
            check(L.xh_rf2_weights_step(self.h, 1))
            for ang, w, s in self._calls:
                self._insert(None, None, ang, w, s, 1)
            check(L.xh_rf2_weights_step(self.h, 2))
        check(L.xh_rf2_weights_step(self.h, 3))
        out = np.empty((self.D,) * 3, np.float64)
        check(L.xh_rf2_finish(self.h, _np_ptr(out)))
        return out


def reduce_reconstructions(rfs):
    """Sum the cropped spaces of several handles of this process (one per device) into rfs[0]:
    the thread-per-device counterpart of allreduce_reconstruction (xh_rf_reduce)."""
    arr = (C.c_void_p * len(rfs))(*[r.h for r in rfs])
    check(lib().xh_rf_reduce(arr, len(rfs)))


def frc_dpr(ctx, m1, m2, sampling_rate=1.0, do_dpr=False, do_rfactor=False, min_freq=0.0, max_freq=0.5):
    """Fourier ring/shell correlation of two float64 device tensors of equal shape (2-D or 3-D): xh_frc_dpr,
    the core of xmipp_resolution_fsc (resolution_fsc.cpp:179-203). m1 is the reference map."""
    torch = _torch()
    assert m1.shape == m2.shape and m1.dtype == m2.dtype == torch.float64 and m1.is_cuda and m2.is_cuda
    m1, m2 = m1.contiguous(), m2.contiguous()
    shp = (1,) * (3 - m1.dim()) + tuple(m1.shape)
    L = shp[2] // 2 + 1
    out = {k: np.zeros(L) for k in ("freq", "frc", "frc_noise", "dpr", "error_l2")}
    rfac = np.full(1, -1.0)
    check(lib().xh_frc_dpr(ctx.h, _ptr(m1), _ptr(m2), shp[0], shp[1], shp[2], sampling_rate, int(do_dpr), int(do_rfactor),
                           min_freq, max_freq, _np_ptr(out["freq"]), _np_ptr(out["frc"]), _np_ptr(out["frc_noise"]),
                           _np_ptr(out["dpr"]), _np_ptr(out["error_l2"]), _np_ptr(rfac)))
    out["rfactor"] = float(rfac[0])
    return out


def allreduce_reconstruction(rf):
    """The one exchange step of the sharded path: SUM of [volume | weights] over ranks
    (replaces the per-row MPI_Reduce of parallel/mpi_reconstruct_fourier_accel.cpp:249-267)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(rf.cropped_view(), op=dist.ReduceOp.SUM)


class Fft2D:
    """In-place complex 2-D FFT of an [ny, nx] complex64 cuda tensor, lines of any factorisable length (xh_fft2d_*: the
    four-step transform FlexAlign's movie frames need)."""

    def __init__(self, ctx, ny, nx):
        self.ctx, self.ny, self.nx = ctx, int(ny), int(nx)
        h = C.c_void_p()
        check(lib().xh_fft2d_create(ctx.h, self.ny, self.nx, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        f = np.zeros(4, np.int32)
        check(lib().xh_fft2d_factors(h, _np_ptr(f)))
        self.factors = tuple(int(v) for v in f)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_fft2d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, data, inverse=False):
        torch = _torch()
        assert data.is_cuda and data.dtype == torch.complex64 and data.is_contiguous() and tuple(data.shape) == (self.ny, self.nx)
        check(lib().xh_fft2d_exec(self.h, C.c_void_p(data.data_ptr()), int(bool(inverse))))
        return data


class FlexAlign:
    """Global alignment of a movie (ProgMovieAlignmentCorrelation*::computeGlobalAlignment) for frames of one size."""

    def __init__(self, ctx, Y, X, sampling_rate=1.0, max_res=30.0):
        self.ctx, self.Y, self.X = ctx, int(Y), int(X)
        h = C.c_void_p()
        check(lib().xh_fa_create(ctx.h, self.Y, self.X, float(sampling_rate), float(max_res), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        a, b, f = C.c_int32(), C.c_int32(), C.c_double()
        check(lib().xh_fa_info(h, C.byref(a), C.byref(b), C.byref(f)))
        self.new_dims, self.size_factor = (a.value, b.value), f.value

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_fa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        check(lib().xh_fa_set_option(self.h, name.encode(), float(value)))

    def last_full_pairs(self):
        return lib().xh_fa_last_full_pairs(self.h)

    def global_alignment(self, frames, max_shift_px, dark=None, gain=None):
        """frames [N, Y, X] float32 on the device -> dict(bX, bY, shiftX, shiftY, ref)"""
        torch = _torch()
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and tuple(frames.shape[1:]) == (self.Y, self.X)
        N = frames.shape[0]
        rows = N * (N - 1) // 2
        bx, by, sx, sy = np.empty(rows), np.empty(rows), np.empty(N), np.empty(N)
        ref = C.c_int32(0)
        check(lib().xh_fa_global_alignment(self.h, _ptr(frames), N, _ptr(dark, torch.float32), _ptr(gain, torch.float32), float(max_shift_px),
                                           _np_ptr(bx), _np_ptr(by), _np_ptr(sx), _np_ptr(sy), C.byref(ref)))
        return {"bX": bx, "bY": by, "shiftX": sx, "shiftY": sy, "ref": ref.value}

    def local_alignment(self, frames, g_shift_x, g_shift_y, ref, max_shift_px, patches=(7, 7), patch_size=(500, 500), patches_avg=3,
                        control_points=(6, 6, 5), dark=None, gain=None):
        """computeLocalAlignment of the CUDA program: dict(patch_shifts [py, px, N, 2], centers [py, px, 2], coeffsX, coeffsY, dims)"""
        torch = _torch()
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and tuple(frames.shape[1:]) == (self.Y, self.X)
        N = frames.shape[0]
        px, py = patches
        lX, lY, lT = control_points
        gx, gy = np.ascontiguousarray(g_shift_x, np.float64), np.ascontiguousarray(g_shift_y, np.float64)
        assert gx.shape == (N,) and gy.shape == (N,)
        shifts, centers = np.empty((py, px, N, 2)), np.empty((py, px, 2))
        cx, cy = np.empty(lX * lY * lT), np.empty(lX * lY * lT)
        dims = np.zeros(4, np.int32)
        check(lib().xh_fa_local_alignment(self.h, _ptr(frames), N, _ptr(dark, torch.float32), _ptr(gain, torch.float32), _np_ptr(gx), _np_ptr(gy), int(ref),
                                          float(max_shift_px), px, py, int(patch_size[0]), int(patch_size[1]), int(patches_avg), lX, lY, lT,
                                          _np_ptr(shifts), _np_ptr(centers), _np_ptr(cx), _np_ptr(cy), _np_ptr(dims)))
        return {"patch_shifts": shifts, "centers": centers, "coeffsX": cx, "coeffsY": cy, "dims": tuple(int(v) for v in dims)}

    def local_from_global(self, g_shift_x, g_shift_y, patches=(7, 7), patch_size=(500, 500), control_points=(6, 6, 5)):
        """localFromGlobal: the B-spline of a movie aligned globally only -> (coeffsX, coeffsY)"""
        gx, gy = np.ascontiguousarray(g_shift_x, np.float64), np.ascontiguousarray(g_shift_y, np.float64)
        N = gx.shape[0]
        px, py = patches
        lX, lY, lT = control_points
        centers = np.empty((py, px, 2))
        cx, cy = np.empty(lX * lY * lT), np.empty(lX * lY * lT)
        check(lib().xh_fa_local_from_global(self.h, N, _np_ptr(gx), _np_ptr(gy), px, py, int(patch_size[0]), int(patch_size[1]), lX, lY, lT, _np_ptr(centers),
                                            _np_ptr(cx), _np_ptr(cy)))
        return cx, cy

    def apply_bspline_frames(self, frames, coeffsX, coeffsY, control_points, n0=0, n1=None, dark=None, gain=None, out=None, total=None, initial=None):
        """Frames n0 .. n1 of frames [N, Y, X] warped by the B-spline in one call (the loop of applyShiftsComputeAverage): total += the
        aligned frames, initial += the corrected unaligned ones, out [n1 - n0 + 1, Y, X] = the aligned frames (each optional)."""
        torch = _torch()
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and tuple(frames.shape[1:]) == (self.Y, self.X)
        N = frames.shape[0]
        n1 = N - 1 if n1 is None else n1
        for t in (total, initial):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (self.Y, self.X))
        assert out is None or (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n1 - n0 + 1, self.Y, self.X))
        lX, lY, lT = control_points
        cx, cy = np.ascontiguousarray(coeffsX, np.float64), np.ascontiguousarray(coeffsY, np.float64)
        assert cx.size == lX * lY * lT and cy.size == lX * lY * lT
        check(lib().xh_fa_apply_bspline_frames(self.h, _ptr(frames), N, int(n0), int(n1), _ptr(dark, torch.float32), _ptr(gain, torch.float32), _np_ptr(cx), _np_ptr(cy),
                                               lX, lY, lT, _ptr(out, torch.float32), _ptr(total, torch.float32), _ptr(initial, torch.float32)))

    def apply_bspline(self, frame, coeffsX, coeffsY, control_points, N, n, dark=None, gain=None, out=None, total=None, initial=None):
        """Frame n of N ([Y, X] float32 on the device) warped by the B-spline (applyBSplineTransform): out = aligned frame,
        total += it, initial += the corrected unaligned frame (each optional)."""
        torch = _torch()
        assert frame.is_cuda and frame.dtype == torch.float32 and frame.is_contiguous() and tuple(frame.shape) == (self.Y, self.X)
        for t in (out, total, initial):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (self.Y, self.X))
        lX, lY, lT = control_points
        cx, cy = np.ascontiguousarray(coeffsX, np.float64), np.ascontiguousarray(coeffsY, np.float64)
        assert cx.size == lX * lY * lT and cy.size == lX * lY * lT
        check(lib().xh_fa_apply_bspline(self.h, _ptr(frame), _ptr(dark, torch.float32), _ptr(gain, torch.float32), _np_ptr(cx), _np_ptr(cy), lX, lY, lT, int(N), int(n),
                                        _ptr(out), _ptr(total), _ptr(initial)))
        return out


EXTREMA_MAX, EXTREMA_LOWEST, EXTREMA_MAX_AROUND_CENTER, EXTREMA_LOWEST_AROUND_CENTER = 0, 1, 2, 3


def extrema_find(ctx, data, search_type, max_dist=0.0):
    """SingleExtremaFinder on signals [n, (z,) (y,) x] float32 on the device -> (positions float32 [n], values float32 [n])"""
    torch = _torch()
    assert data.is_cuda and data.dtype == torch.float32 and data.is_contiguous() and 2 <= data.dim() <= 4
    shp = (data.shape[0],) + (1,) * (4 - data.dim()) + tuple(data.shape[1:])
    pos, val = np.empty(shp[0], np.float32), np.empty(shp[0], np.float32)
    check(lib().xh_extrema_find(ctx.h, _ptr(data), shp[0], shp[1], shp[2], shp[3], int(search_type), float(max_dist), _np_ptr(pos), _np_ptr(val)))
    return pos, val


def rotation_estimate(ctx, ref, others, first_ring=None, last_ring=None):
    """PolarRotationEstimator (OneToN): ref [D, D], others [n, D, D] float32 on the device -> rotations [n] in degrees as the
    reference returns them. Default rings: RotationEstimationSetting::getDefaultFirstRing / getDefaultLastRing."""
    torch = _torch()
    D = ref.shape[-1]
    assert ref.is_cuda and ref.dtype == torch.float32 and ref.is_contiguous() and tuple(ref.shape) == (D, D)
    assert others.is_cuda and others.dtype == torch.float32 and others.is_contiguous() and tuple(others.shape[1:]) == (D, D)
    if first_ring is None:
        first_ring = max(2, D // 20)
    if last_ring is None:
        last_ring = (D - 3) // 2
    out = np.empty(others.shape[0], np.float32)
    check(lib().xh_rotation_estimate(ctx.h, _ptr(ref), _ptr(others), others.shape[0], D, int(first_ring), int(last_ring), _np_ptr(out)))
    return out


def apply_geometry2d(ctx, src, matrices):
    """BSplineGeoTransformer::interpolate: src [n, y, x] float32 on the device, matrices [n, 3, 3] (applyGeometry LINEAR, IS_INV, DONT_WRAP)"""
    torch = _torch()
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 3
    m = np.ascontiguousarray(matrices, np.float32).reshape(src.shape[0], 9)
    out = torch.empty_like(src)
    check(lib().xh_apply_geometry2d(ctx.h, _ptr(src), src.shape[0], src.shape[1], src.shape[2], _np_ptr(m), _ptr(out)))
    return out


def correlation_merit(ctx, ref, others):
    """CorrelationComputer (OneToN, normalised): correlationIndex(ref, others[i]) -> float32 [n]"""
    torch = _torch()
    assert ref.is_cuda and others.is_cuda and ref.dtype == others.dtype == torch.float32 and ref.is_contiguous() and others.is_contiguous()
    out = np.empty(others.shape[0], np.float32)
    check(lib().xh_correlation_merit(ctx.h, _ptr(ref), _ptr(others), others.shape[0], others.shape[1], others.shape[2], _np_ptr(out)))
    return out


def iterative_alignment(ctx, ref, others, max_shift, iters=3, first_ring=None, last_ring=None):
    """IterativeAlignmentEstimator::compute: ref [D, D], others [n, D, D] -> (poses [n, 3, 3] float32, merit [n])"""
    torch = _torch()
    D = ref.shape[-1]
    assert ref.is_cuda and ref.dtype == torch.float32 and ref.is_contiguous() and tuple(ref.shape) == (D, D)
    assert others.is_cuda and others.dtype == torch.float32 and others.is_contiguous() and tuple(others.shape[1:]) == (D, D)
    if first_ring is None:
        first_ring = max(2, D // 20)
    if last_ring is None:
        last_ring = (D - 3) // 2
    n = others.shape[0]
    poses, merit = np.empty((n, 9), np.float32), np.empty(n, np.float32)
    check(lib().xh_iterative_alignment(ctx.h, _ptr(ref), _ptr(others), n, D, int(max_shift), int(first_ring), int(last_ring), int(iters), _np_ptr(poses), _np_ptr(merit)))
    return poses.reshape(n, 3, 3), merit


class ShiftCorrEstimator:
    """Alignment::ShiftCorrEstimator<float>, AlignType::OneToN, for images of x by y pixels (even)."""

    def __init__(self, ctx, x, y, max_shift):
        self.ctx, self.x, self.y = ctx, int(x), int(y)
        h = C.c_void_p()
        check(lib().xh_shiftcorr_create(ctx.h, self.x, self.y, int(max_shift), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_shiftcorr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_reference(self, ref):
        torch = _torch()
        assert ref.is_cuda and ref.dtype == torch.float32 and ref.is_contiguous() and tuple(ref.shape) == (self.y, self.x)
        check(lib().xh_shiftcorr_load_reference(self.h, _ptr(ref)))

    def compute_shifts(self, others):
        """others [n, y, x] -> shifts [n, 2] (x, y) as getShifts2D returns them (the image's shift is the negative)"""
        torch = _torch()
        assert others.is_cuda and others.dtype == torch.float32 and others.is_contiguous() and tuple(others.shape[1:]) == (self.y, self.x)
        out = np.empty((others.shape[0], 2), np.float32)
        check(lib().xh_shiftcorr_compute_shifts(self.h, _ptr(others), others.shape[0], _np_ptr(out)))
        return out

    @staticmethod
    def correlate(ctx, inout, ref, center):
        """computeCorrelations2DOneToN: inout [n, fy, fx] complex64 <- ref conj(inout) (times (-1)^(x+y) when center), in place"""
        torch = _torch()
        assert inout.is_cuda and inout.dtype == torch.complex64 and inout.is_contiguous() and ref.dtype == torch.complex64 and ref.is_contiguous()
        assert tuple(ref.shape) == tuple(inout.shape[1:])
        check(lib().xh_shiftcorr_correlate(ctx.h, C.c_void_p(inout.data_ptr()), C.c_void_p(ref.data_ptr()), inout.shape[0], inout.shape[1], inout.shape[2], int(bool(center))))
        return inout


def movie_binned_size(Y, X, binning):
    """AProgMovieAlignmentCorrelation::getMovieSize (movie_alignment_correlation_base.cpp:356-370): float arithmetic, truncated"""
    f = np.float32
    return int(f(f(f(Y) / f(binning)) / f(2)) * f(2)), int(f(f(f(X) / f(binning)) / f(2)) * f(2))


def movie_frames_to_float(ctx, raw, out=None):
    """Frames as the detector stores them (int8, int16, uint16, uint8 or float32 tensor on the device, any shape) -> float32, the
    cast of Image<float>::read (xh_movie_frame_to_float): copy the counts to the device, convert there."""
    torch = _torch()
    modes = {torch.int8: 0, torch.int16: 1, torch.float32: 2, torch.uint16: 6, torch.uint8: 100}
    assert raw.is_cuda and raw.is_contiguous() and raw.dtype in modes
    if out is None:
        out = torch.empty(raw.shape, dtype=torch.float32, device=raw.device)
    assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == raw.numel()
    check(lib().xh_movie_frame_to_float(ctx.h, C.c_void_p(raw.data_ptr()), modes[raw.dtype], raw.numel(), _ptr(out)))
    return out


def movie_bin_frame(fft_raw, fft_binned, frame, dark=None, gain=None):
    """--bin of the CUDA FlexAlign program: frame [Y, X] -> [Yb, Xb] by cropping its half spectrum (xh_movie_bin_frame)"""
    torch = _torch()
    assert frame.is_cuda and frame.dtype == torch.float32 and frame.is_contiguous() and tuple(frame.shape) == (fft_raw.ny, fft_raw.nx)
    out = torch.empty((fft_binned.ny, fft_binned.nx), device=frame.device)
    check(lib().xh_movie_bin_frame(fft_raw.ctx.h, fft_raw.h, fft_binned.h, _ptr(frame), _ptr(dark, torch.float32), _ptr(gain, torch.float32), fft_raw.ny, fft_raw.nx,
                                   _ptr(out), fft_binned.ny, fft_binned.nx))
    return out


def movie_dose_filter(fft, frame, pixel_size, acc_voltage, dose_start, dose_finish):
    """ProgMovieFilterDose on one frame ([Y, X] float32 on the device, in place); fft = Fft2D(ctx, Y, X)."""
    torch = _torch()
    assert frame.is_cuda and frame.dtype == torch.float32 and frame.is_contiguous() and tuple(frame.shape) == (fft.ny, fft.nx)
    check(lib().xh_movie_dose_filter(fft.ctx.h, fft.h, _ptr(frame), fft.ny, fft.nx, float(pixel_size), float(acc_voltage), float(dose_start), float(dose_finish)))
    return frame


def fa_correlate(ctx, frames, max_dist):
    """CUDAFlexAlignCorrelate::run: frames [N, Y, X] float32 on the device (even sizes) -> positions [N (N-1)/2, 2] (x, y) of the
    correlation maxima of all pairs i < j."""
    torch = _torch()
    assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and frames.dim() == 3
    N, Y, X = frames.shape
    pos = np.empty((N * (N - 1) // 2, 2))
    check(lib().xh_fa_correlate(ctx.h, _ptr(frames), N, Y, X, float(max_dist), _np_ptr(pos)))
    return pos


class CtfOps:
    """CTF pre-steps on the device: actualPhaseFlip (reconstruction/ctf_phase_flip.cpp:88-117) and
    Wiener2D::applyWienerFilter (data/wiener2d.cpp:101-141) for images of one size."""

    def __init__(self, ctx, ydim, xdim, pad=1.0):
        self.ctx, self.ydim, self.xdim = ctx, int(ydim), int(xdim)
        h = C.c_void_p()
        check(lib().xh_ctfop_create(ctx.h, self.ydim, self.xdim, float(pad), C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_ctfop_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def phase_flip(self, img, ctf, sampling_rate):
        """img: [ydim, xdim] float32 on the device, flipped in place"""
        torch = _torch()
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and tuple(img.shape) == (self.ydim, self.xdim)
        check(lib().xh_ctfop_phase_flip(self.h, _ptr(img), C.byref(ctf), float(sampling_rate)))
        return img

    def wiener2d(self, imgs, ctfs, sampling_rate=1.0, phase_flipped=False, is_isotropic=False, wiener_constant=-1.0, correct_envelope=False):
        """imgs: [n, ydim, xdim] float32 on the device, corrected in place; ctfs: list of CtfParams or ctf_param_array"""
        torch = _torch()
        n = imgs.shape[0]
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous() and tuple(imgs.shape[1:]) == (self.ydim, self.xdim)
        arr = ctfs if isinstance(ctfs, C.Array) else (CtfParams * n)(*ctfs)
        check(lib().xh_ctfop_wiener2d(self.h, _ptr(imgs), n, arr, float(sampling_rate), int(phase_flipped), int(is_isotropic),
                                      float(wiener_constant), int(correct_envelope)))
        return imgs


class FourierProjector:
    """Device side of FourierProjector (data/fourier_projection.cpp): central-slice projections of a
    volume `[z][y][x]` (float32, cuda) with cubic B-spline interpolation in Fourier space."""

    def __init__(self, ctx, vol, padding=2.0, max_freq=0.5, degree=3):
        torch = _torch()
        assert vol.is_cuda and vol.dtype == torch.float32 and vol.is_contiguous() and vol.dim() == 3
        self.ctx = ctx
        self.D = vol.shape[0]
        h = C.c_void_p()
        check(lib().xh_fp_create(ctx.h, _ptr(vol), self.D, float(padding), float(max_freq), int(degree), C.byref(h)))
        self.h = h
        ctx._children.add(self)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().xh_fp_info(h, C.byref(a), C.byref(b), C.byref(c)))
        self.P, self.cdim, self.cstart = a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_fp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def coefs(self):
        re = np.empty((self.cdim,) * 3, np.float64)
        im = np.empty((self.cdim,) * 3, np.float64)
        check(lib().xh_fp_coefs(self.h, _np_ptr(re), _np_ptr(im)))
        return re, im

    def project(self, angles, ctf=None):
        """angles: [n, 3] (rot, tilt, psi) in degrees; ctf: optional float64 cuda tensor [D, D//2+1]."""
        torch = _torch()
        ang = np.ascontiguousarray(angles, np.float64).reshape(-1, 3)
        n = ang.shape[0]
        out = torch.empty((n, self.D, self.D), dtype=torch.float32, device=self.ctx.torch_device)
        check(lib().xh_fp_project(self.h, _np_ptr(ang), n, None if ctf is None else _ptr(ctf, torch.float64), _ptr(out)))
        return out


class ProjectionMatcher:
    """Device side of ProgAngularProjectionMatching
    (reconstruction/angular_projection_matching.cpp)."""

    def __init__(self, ctx, refs, Ri=1, Ro=-1, Mctf=None, paddim=0):
        torch = _torch()
        assert refs.is_cuda and refs.dtype == torch.float32 and refs.is_contiguous()
        self.ctx = ctx
        self.nrefs, self.D, _ = refs.shape
        h = C.c_void_p()
        m = None if Mctf is None else np.ascontiguousarray(Mctf, np.float64)
        check(lib().xh_pm_create(ctx.h, self.D, Ri, Ro, self.nrefs, _ptr(refs), _np_ptr(m), paddim, C.byref(h)))
        self.h = h
        ctx._children.add(self)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().xh_pm_info(h, C.byref(a), C.byref(b), C.byref(c)))
        self.N, self.ncoef, self.nsamples = a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                lib().xh_pm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        check(lib().xh_pm_set_option(self.h, name.encode(), float(value)))

    def match(self, particles, nbr_off=None, nbr_ids=None, parity=0, n_orient=1, shifts5d=None):
        """Rotational search (APM:530-760). shifts5d: (xoff[], yoff[]) integer 5-D search translations
        (`search5d_offsets`); n_orient > 1 returns [n, n_orient] arrays (refno -1 = rank not filled)."""
        torch = _torch()
        assert particles.is_cuda and particles.dtype == torch.float32 and particles.is_contiguous()
        n = particles.shape[0]
        dev = particles.device
        shape = (n,) if n_orient == 1 else (n, n_orient)
        refno = torch.empty(shape, dtype=torch.int32, device=dev)
        psi = torch.empty(shape, dtype=torch.int32, device=dev)
        flip = torch.empty(shape, dtype=torch.uint8, device=dev)
        off = ids = xo = yo = None
        nt = 0
        if nbr_off is not None:
            off = np.ascontiguousarray(nbr_off, np.int32)
            ids = np.ascontiguousarray(nbr_ids, np.int32)
        if shifts5d is not None:
            xo = np.ascontiguousarray(shifts5d[0], np.int32)
            yo = np.ascontiguousarray(shifts5d[1], np.int32)
            nt = len(xo)
            assert len(yo) == nt and nt > 0
        check(lib().xh_pm_match_ex(self.h, _ptr(particles), n, _np_ptr(off), _np_ptr(ids), int(parity), int(n_orient), nt,
                                   _np_ptr(xo), _np_ptr(yo), _ptr(refno), _ptr(psi), _ptr(flip)))
        return refno, psi, flip

    def translate(self, particles, refno, psi, flip, max_shift=-1.0):
        torch = _torch()
        n = particles.shape[0]
        dev = particles.device
        sx = torch.empty(n, dtype=torch.float64, device=dev)
        sy = torch.empty_like(sx)
        cc = torch.empty_like(sx)
        check(lib().xh_pm_translate(self.h, _ptr(particles, torch.float32), n, _ptr(refno, torch.int32), _ptr(psi, torch.int32),
                                    _ptr(flip, torch.uint8), float(max_shift),
                                    _ptr(sx), _ptr(sy), _ptr(cc)))
        return sx, sy, cc

    def translate_repeated(self):
        """Particles the last translate() repeated in double precision (its first pass is fp32)."""
        r = C.c_int64()
        check(lib().xh_pm_translate_stats(self.h, C.byref(r)))
        return r.value

    def stage_ms(self, reset=True):
        ms = np.zeros(8, np.float64)
        check(lib().xh_pm_stage_ms(self.h, _np_ptr(ms), int(reset)))
        return dict(zip(("prep32", "contract", "idft_max", "select", "rescore_fp64"), ms[:5].tolist()))

    def two_level_cut(self):
        """(K0, nk): the contraction of the dense search stops at angular frequency K0 (xh_pm_two_level_cut)."""
        a, b = C.c_int32(), C.c_int32()
        check(lib().xh_pm_two_level_cut(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_coefficients(self, n):
        """Device pointer to the fp32 B-spline coefficients of the n particles of the last match() call, or None when
        that call ran in several chunks (or with the recursive prefilter): valid until the next call on this matcher."""
        p, a, b = C.c_void_p(), C.c_int32(), C.c_int32()
        check(lib().xh_pm_last_coefficients(self.h, C.byref(p), C.byref(a), C.byref(b)))
        return p.value if (p.value and a.value == 0 and b.value == n) else None

    def last_stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().xh_pm_last_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        d = C.c_int64()
        check(lib().xh_pm_rows_pruned(self.h, C.byref(d)))
        return {"rows": a.value, "rescored_particles": b.value, "rescored_rows": c.value, "pruned_rows": d.value}

    # ---- test hooks
    def debug_prepare(self, particles, precision=32):
        n = particles.shape[0]
        coefs = np.empty((n, self.ncoef, 2), np.float64)
        sigma = np.empty(n, np.float64)
        check(lib().xh_pm_debug_prepare(self.h, _ptr(particles), n, precision, _np_ptr(coefs), _np_ptr(sigma)))
        return coefs[..., 0] + 1j * coefs[..., 1], sigma

    def debug_ref(self, r):
        coefs = np.empty((self.ncoef, 2), np.float64)
        s = C.c_double()
        check(lib().xh_pm_debug_ref(self.h, r, _np_ptr(coefs), C.byref(s)))
        return coefs[:, 0] + 1j * coefs[:, 1], s.value

    def debug_corr_rows(self, particle, ref, precision=32):
        out = np.empty(2 * self.N, np.float64)
        check(lib().xh_pm_debug_corr_rows(self.h, _ptr(particle), ref, precision, _np_ptr(out)))
        return out

    def debug_s6_maps(self, n):
        """correlation maps left by the last translate() under set_option("s6_capture", 32 | 64): [n, D, D] float64"""
        out = np.empty((n, self.D, self.D), np.float64)
        check(lib().xh_pm_debug_s6_maps(self.h, n, _np_ptr(out)))
        return out

    def get_option(self, name):
        v = C.c_double()
        check(lib().xh_pm_get_option(self.h, name.encode(), C.byref(v)))
        return v.value
