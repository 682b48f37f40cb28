"""The two-stream choreography of bench.py (reconstruction half of batch k on a second library context / stream beside the matching
of batch k+1) against the same steps on one stream: same orientations, same shifts, same temp spaces -- a race between the streams
(the matcher's coefficient buffer, the batch buffers, the orientation arrays) would show here."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _run(xa, torch, refs, batches, dirs, ctfs, pipelined):
    D = refs.shape[-1]
    dev = torch.device("cuda", 0)
    ctx = xa.Context(0)
    side = torch.cuda.Stream(device=dev) if pipelined else None
    ctx_rf = ctx
    if pipelined:
        with torch.cuda.stream(side):
            ctx_rf = xa.Context(0)
    pm = xa.ProjectionMatcher(ctx, refs)
    rf = xa.RecFourier(ctx_rf, D, min_ctf=0.01, sampling=1.0)
    dirs_dev = torch.from_numpy(np.ascontiguousarray(dirs[:, :2], np.float64)).to(dev)
    ctf_arr = xa.RecFourier.ctf_param_array(ctfs)
    outs = []
    dbuf = [torch.empty_like(batches[0]), torch.empty_like(batches[0])]
    for k, b in enumerate(batches):
        parts = dbuf[k & 1]
        parts.copy_(b)                                                   # the buffer is reused two steps later, like the bench's
        refno, psi, flip = pm.match(parts)
        sx, sy, cc = pm.translate(parts, refno, psi, flip)
        ang = torch.cat([dirs_dev[refno.long()], (psi.double() * (360.0 / pm.N))[:, None]], 1).contiguous()
        n = parts.shape[0]
        if pipelined:
            matched = torch.cuda.Event()
            matched.record()
            shifted = torch.cuda.Event()
            with torch.cuda.stream(side):
                side.wait_event(matched)
                for t_ in (parts, sx, sy, flip, ang):
                    t_.record_stream(side)
                imgs = rf.shift_images(parts, (sx, sy), flips=flip, coefs=pm.last_coefficients(n))
                shifted.record(side)
                rf.insert_images(imgs, ang, ctf_array=ctf_arr)
            torch.cuda.current_stream().wait_event(shifted)
        else:
            imgs = rf.shift_images(parts, (sx, sy), flips=flip, coefs=pm.last_coefficients(n))
            rf.insert_images(imgs, ang, ctf_array=ctf_arr)
        outs.append((refno.cpu().numpy(), psi.cpu().numpy(), flip.cpu().numpy(), sx.cpu().numpy(), sy.cpu().numpy()))
    torch.cuda.synchronize()
    if pipelined:
        with torch.cuda.stream(side):
            v, w = rf.temp_spaces()
            v, w = v.cpu().numpy(), w.cpu().numpy()
    else:
        v, w = rf.temp_spaces()
        v, w = v.cpu().numpy(), w.cpu().numpy()
    rf.close(); pm.close()
    return outs, v, w


def test_two_streams_give_the_one_stream_results():
    import torch
    import xmipp3_amd as xa
    from xmipp3_amd.api import ctf_params
    assert torch.cuda.is_available()
    D, nrefs, B, nb = 64, 60, 512, 5
    vol = synth.phantom(D, seed=3, nblobs=12)
    refs_np, dirs = synth.make_refs(vol, nrefs)
    refs = torch.from_numpy(refs_np.astype(np.float32)).cuda()
    refs = ((refs - refs.mean()) / refs.std()).contiguous()
    rng = np.random.default_rng(8)
    batches = []
    for _ in range(nb):
        parts, _t = synth.make_particles(refs_np, B, rng, snr=0.2, max_shift=2)
        batches.append(torch.from_numpy(parts.astype(np.float32)).cuda().contiguous())
    ctfs = [ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d)) for d in rng.uniform(10000.0, 30000.0, B)]
    o1, v1, w1 = _run(xa, torch, refs, batches, dirs, ctfs, False)
    for rep in range(3):                                                # a race need not show the first time
        o2, v2, w2 = _run(xa, torch, refs, batches, dirs, ctfs, True)
        for a, b in zip(o1, o2):
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))                   # reference, psi, flip
            # (the shifts of two one-stream runs already differ in the last bit: the double-precision repeats of S6 pair the
            # flagged particles in the order an atomic counter hands them out)
            assert all(np.abs(x - y).max() <= 1e-12 for x, y in zip(a[3:], b[3:]))
        assert np.array_equal(v1, v2) and np.array_equal(w1, w2)                             # temp spaces: the same bits


def test_host_feed_entry_points():
    """The ABI's pieces for a host that feeds the device while it computes (xmipp3_amd/host/fastio.h uses them from C++): page-locked
    memory, copies that only enqueue, one context's stream waiting for another's on the device, the device's NUMA node."""
    import ctypes as C
    import torch
    from xmipp3_amd import _lib
    L = _lib.lib()
    a, b = C.c_void_p(), C.c_void_p()
    assert L.xh_ctx_create_private(0, C.byref(a)) == 0 and L.xh_ctx_create_private(0, C.byref(b)) == 0
    n = 1 << 22
    hp, hq, d1, d2 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.xh_host_alloc(a, n * 4, C.byref(hp)) == 0 and L.xh_host_alloc(a, n * 4, C.byref(hq)) == 0
    assert L.xh_malloc(a, n * 4, C.byref(d1)) == 0 and L.xh_malloc(a, n * 4, C.byref(d2)) == 0
    src = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), (n,))
    dst = np.ctypeslib.as_array(C.cast(hq, C.POINTER(C.c_float)), (n,))
    src[:] = np.arange(n, dtype=np.float32)
    dst[:] = -1
    # a: host -> d1 (enqueued only); b waits for a on the device, copies d1 -> host through its own stream
    assert L.xh_memcpy_h2d_async(a, d1, hp, n * 4) == 0
    assert L.xh_ctx_wait_for(b, a) == 0
    assert L.xh_memcpy_d2h_async(b, hq, d1, n * 4) == 0
    assert L.xh_ctx_sync(b) == 0
    assert np.array_equal(dst, src)
    node = C.c_int(-5)
    assert L.xh_device_numa_node(0, C.byref(node)) == 0 and node.value >= -1
    # contexts on different devices cannot wait for each other; null arguments are refused
    assert L.xh_ctx_wait_for(b, None) != 0 and L.xh_host_alloc(a, 16, None) != 0
    assert L.xh_host_free(a, hp) == 0 and L.xh_host_free(a, hq) == 0 and L.xh_host_free(a, None) == 0
    assert L.xh_free(a, d1) == 0 and L.xh_free(a, d2) == 0
    # the crop used by the size-search stand-in of the movie program
    fr = torch.arange(2 * 6 * 8, dtype=torch.float32, device="cuda").reshape(2, 6, 8)
    out = torch.empty((2, 4, 5), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    assert L.xh_movie_crop_frames(a, C.c_void_p(fr.data_ptr()), 2, 6, 8, 4, 5, C.c_void_p(out.data_ptr())) == 0
    assert L.xh_ctx_sync(a) == 0
    assert torch.equal(out, fr[:, :4, :5])
    assert L.xh_movie_crop_frames(a, C.c_void_p(fr.data_ptr()), 2, 6, 8, 7, 5, C.c_void_p(out.data_ptr())) != 0
    assert L.xh_ctx_destroy(a) == 0 and L.xh_ctx_destroy(b) == 0
