#!/usr/bin/env python3
"""bench.py -- one refine iteration of the Xmipp hot path on MI355X.

Workload (BASELINE.json metric "particles/s projection-matched+reconstructed, 256x256 box",
config "Full refine iteration (match + CTF + reconstruct)"): each rank owns a contiguous
shard of synthetic 256x256 particles resident in HBM and, per step, pushes one batch through

    projection matching vs the 1000-reference gallery (rotational search + mirror, exact
    arg-max, translational alignment)  ->  apply the found shifts  ->  CTF planes  ->
    2-D FFT + crop  ->  Kaiser-Bessel gridding into the rank's Fourier volume

and, after the K steps, mirror+crop, ONE all-reduce (RCCL) of [volume | weights] and the
finaliser (3-D inverse FFT + blob correction) on every rank.  All of it is inside the timed
region.  Per the task statement `value` is measured with the inputs resident in HBM when the clock starts
(four distinct batches, 16 384 particles, cycled; every step's results -- reference, in-plane angle,
mirror, shifts, maxCC -- still go back to page-locked host memory inside the clock); the PCIe-inclusive
rate of SURVEY.md 8d -- every batch copied from page-locked host memory inside the clock, two device
buffers, the copy of batch k+1 on its own stream under step k -- is printed beside it as `value_streamed`
(`--timed streamed` makes it the timed region, as it was in rounds 3-4).  Weak scaling: per-GPU work is
fixed as N grows.

One JSON line on rank 0 (see the driver contract in the task statement), with
  roofline      -- the kernel that dominates the timed region, algorithmic work / HIP-event time
  worst_case    -- the same step with the data-dependent shortcuts of the matcher switched off
  value_streamed -- the same steps + finish with every batch H2D inside the clock
  match_only_leg -- the matching half alone (xh_pm_match + xh_pm_translate), the rate north_star's 2 M / 8 GPUs is stated for
  noise_gallery / compact_phantom / flexalign -- the data dependence of the headline, and BASELINE config 5
  cpu_baseline  -- the CPU oracle ("port" of the reference algorithm; Xmipp itself cannot be
                   built here: xmippCore/FFTW absent) timed on a bounded sample on rank 0.

`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh child processes, one per
GPU, rendezvous on 127.0.0.1) and relays rank 0's line; under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)         # = the number of distinct batches: every batch has been a step once before the clock starts
    ap.add_argument("--box", type=int, default=256, help="particle box D")
    ap.add_argument("--nrefs", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=4096, help="particles per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="particles in the CPU sample (0 = auto)")
    ap.add_argument("--mode", default="full", choices=["full", "match", "grid", "flexalign"],
                    help="full / match / grid: the refine iteration or one half of it; flexalign: BASELINE config 5, one K3 movie per step "
                         "(global + local alignment, warp + sum), movies streamed from page-locked host memory")
    ap.add_argument("--movie", default="40x4092x5760", help="flexalign mode: frames x rows x columns of a movie")
    ap.add_argument("--movie-mode", type=int, default=0, choices=[0, 1, 2, 6], help="flexalign mode: how the frames lie in host memory, as an MRC data mode: 0 int8 counts "
                    "(default: what a K3 writes, 0.94 GB per movie cross the link), 1 int16 / 6 uint16 counts -- all cast to float on the device "
                    "(xh_movie_frame_to_float) --, 2 float32 (3.77 GB per movie: the link then bounds the rate)")
    ap.add_argument("--fa-opt", action="append", default=[], metavar="NAME=VALUE", help="flexalign mode: xh_fa_set_option on every lane (A/B runs), "
                    "e.g. --fa-opt pruned_columns=0 --fa-opt pairwin_form=0")
    ap.add_argument("--fa-shared-copy", type=int, default=1, help="flexalign mode: 1 one copy stream for all lanes, 0 one per lane")
    ap.add_argument("--fa-lanes", type=int, default=4, help="flexalign mode: movies in flight per GPU (each lane: a host thread with its own stream, library "
                    "handle and pair of device buffers; the kernels of one lane fill the device while another lane's host solves its shifts / fits its spline)")
    ap.add_argument("--refs", default="phantom", choices=["phantom", "noise", "compact"],
                    help="reference gallery: projections of a Gaussian-blob phantom (BASELINE config 2/4) or unrelated band-limited noise images")
    ap.add_argument("--tr-chunk-mb", type=int, default=0, help="S6 scratch per pass in MB (0: library default)")
    ap.add_argument("--chunk-rows", type=float, default=0, help="correlation rows per chunk of the matcher (0: library default)")
    ap.add_argument("--k0", type=int, default=-1, help="two-level contraction cut (0 auto, >= nk off); default: automatic")
    ap.add_argument("--neighbours", type=int, default=0,
                    help="local search (APM:615-631): every particle is matched against the K references nearest to its true "
                         "direction only (ascending lists, as a sampling file holds them); 0: the whole bank")
    ap.add_argument("--pm-opt", action="append", default=[], help="name=value passed to xh_pm_set_option (A/B runs)")
    ap.add_argument("--main-priority", type=int, default=0, help="experiment: priority of the matcher's stream (-1 high, 0 default)")
    ap.add_argument("--side-priority", type=int, default=0, help="experiment: priority of the reconstruction stream (-1 high, 0 default)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment: q of every 4 CU groups to the matcher's stream, the others to the "
                    "reconstruction stream (hipExtStreamCreateWithCUMask); 0: no masks")
    ap.add_argument("--rf-opt", action="append", default=[], help="name=value passed to xh_rf_set_option (A/B runs)")
    ap.add_argument("--no-prune", action="store_true", help="transform every correlation row (S3 branch and bound off)")
    ap.add_argument("--tau-rel", type=float, default=0, help="ambiguity margin of the coarse pass relative to S (0: library default)")
    ap.add_argument("--timed", default="resident", choices=["resident", "streamed"],
                    help="what the timed region of `value` feeds on: resident = the batches lie in HBM when the clock starts (the task statement's contract: "
                         "the PCIe-inclusive rate is never `value`); streamed = every batch H2D from page-locked memory inside the timed region (SURVEY.md 8d's "
                         "metric, `value` of rounds 3-4). The other one is printed beside it (`value_streamed` / `value_resident`)")
    ap.add_argument("--no-flexalign", action="store_true", help="skip the FlexAlign leg (config 5, a child process) of the default line")
    ap.add_argument("--no-cli", action="store_true", help="skip the program-level leg (tools/bench_cli.py, a child process) of the default line")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the worst-case and host-streaming legs after the timed region")
    ap.add_argument("--pipeline", type=int, default=1, help="1: the reconstruction half (shift, CTF, FFT, gridding) of batch k runs on a second "
                    "stream beside the matching of batch k+1")
    ap.add_argument("--stream-test", type=int, default=0, help="diagnosis of the host-traffic cost: 1 no result copies, 2 batches copied device to device, 3 host copies in 16 chunks")
    ap.add_argument("--sdma", type=int, default=-1, help="1 / 0: HSA_ENABLE_SDMA for this process (copy engines or shader copies for the host traffic); -1: leave the environment alone")
    ap.add_argument("--unique-batches", type=int, default=0, help="distinct particle batches cycled from host memory (0: 4 = 16384 particles at --batch 4096)")
    return ap.parse_args(argv)


# bounds the FlexAlign leg asserts: device vs oracle pair shifts of three K3 frames (measured 2.6e-6 px in round 6, once the mean of the
# correlation map stays out of the fp32 sums; 7.9e-4 in round 5; tests/test_gpu_flexalign.py holds the same figure), and the error of the
# recovered global drift of the synthetic movie (measured 2.1 px)
FA_PARITY_BOUND_PX = 2e-5
FA_DRIFT_BOUND_PX = 2.5


def spawn_ranks(script, argv, n, python=sys.executable, extra_env=None, timeout=None):
    """Start n fresh processes of `script argv` as ranks 0..n-1 of one node (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR =
    127.0.0.1, MASTER_PORT = a free port). Rank 0 inherits stdout, the others write theirs to stderr. Returns 0 when
    every rank exits 0, else the first non-zero exit code (the surviving ranks are terminated). The caller must not have
    touched the GPU: the children are new processes, nothing is exec'ed over an initialised runtime."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([python, script] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    t_end = None if timeout is None else time.time() + timeout
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:
                    other.terminate()
        if t_end is not None and time.time() > t_end:
            for other in live:
                other.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


def smooth_noise(torch, n, D, gen, device, sigma_px=3.0):
    """Band-limited random images (stand-in for projections); data generation only."""
    x = torch.randn((n, D, D), generator=gen, device=device)
    f = torch.fft.rfft2(x)
    ky = torch.fft.fftfreq(D, device=device)[:, None]
    kx = torch.fft.rfftfreq(D, device=device)[None, :]
    f = f * torch.exp(-2 * (math.pi * sigma_px) ** 2 * (kx * kx + ky * ky))
    y = torch.fft.irfft2(f, s=(D, D))
    r2 = (torch.arange(D, device=device) - D // 2) ** 2
    mask = ((r2[:, None] + r2[None, :]) < (0.42 * D) ** 2).float()
    y = y * mask
    return (y / y.std()).contiguous()


def phantom_volume(torch, D, gen, device, nblobs=20, rmax=20.0):
    """BASELINE config 1/3 phantom: 20 3-D Gaussians, sigma in [2,5] voxels and centres within radius rmax = 20 of a 64^3
    box, scaled to D; amplitudes U[0.5,1]. Data generation only.  (rmax 9.6 = 0.3 of the box radius: the compact phantom of the tests.)"""
    ax = torch.arange(D, device=device, dtype=torch.float32) - D // 2
    sc = D / 64.0
    vol = torch.zeros((D, D, D), device=device)
    u = torch.rand((nblobs, 6), generator=gen, device=device).cpu().numpy()
    for b in range(nblobs):
        r = rmax * sc * u[b, 0] ** (1.0 / 3.0)
        ct, ph = 2 * u[b, 1] - 1, 2 * math.pi * u[b, 2]
        st = math.sqrt(max(0.0, 1 - ct * ct))
        c = (r * st * math.cos(ph), r * st * math.sin(ph), r * ct)
        sg = (2.0 + 3.0 * u[b, 3]) * sc
        amp = 0.5 + 0.5 * u[b, 4]
        g = [torch.exp(-(ax - ci) ** 2 / (2 * sg * sg)) for ci in c]
        vol += amp * g[2][:, None, None] * g[1][None, :, None] * g[0][None, None, :]
    return vol.contiguous()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"



def main_flexalign(args):
    """BASELINE config 5: FlexAlign on K3 movies (40 frames of 4092 x 5760), one movie per step and rank.  Inside the timed region, per movie: the
    movie's 3.8 GB come from page-locked host memory (two device buffers: the copy of movie k+1 flies on its own stream under the
    alignment of movie k; only the first copy precedes the clock), global alignment (movie_alignment_correlation_gpu.cpp:633-725), local
    alignment with the program's defaults (12 x 9 patches of 500 px, three frames per patch, 6 x 6 x 5 control points; :288-430), B-spline
    warp of every frame into the aligned sum (:460-560), and the sum's copy back to page-locked host memory.  N ranks = N movies at a
    time, no exchange ("replicas only")."""
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        if local == 0:
            ge.build()
        dist.barrier()
    else:
        ge.build()
    import xmipp3_amd as xa
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ctx = xa.Context(local)
    N, Y, X = (int(v) for v in args.movie.lower().split("x"))
    Ts, max_res, max_shift = 1.0, 30.0, 50.0
    req = int(500 / Ts)
    patches, psize, cp = (int(math.ceil(X / req)), int(math.ceil(Y / req))), (req, req), (6, 6, 5)

    # ---- two synthetic movies: a smooth random field under a global drift (fast at first, settling) plus a dilation / shear that grows
    # with time, white noise per frame; generated on the device, kept in page-locked host memory
    raw_dtype = {0: torch.int8, 1: torch.int16, 2: torch.float32, 6: torch.uint16}[args.movie_mode]

    def make_movie(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        base = torch.randn((Y + 128, X + 128), generator=g, device=dev)
        k = torch.fft.rfft2(base)
        fy = torch.fft.fftfreq(Y + 128, device=dev)[:, None]
        fx = torch.fft.rfftfreq(X + 128, device=dev)[None, :]
        base = torch.fft.irfft2(k * torch.exp(-2 * (math.pi * 4.0) ** 2 * (fx * fx + fy * fy)), s=base.shape) * 30
        del k
        t = np.arange(N, dtype=np.float64)
        sgn = 1.0 if seed % 2 else -1.0
        drift = np.stack([sgn * 14.0 * (1 - np.exp(-t / 9.0)) + 0.04 * t, -9.0 * (1 - np.exp(-t / 14.0)) + 0.06 * t], 1)
        H, W = base.shape
        ys = torch.arange(Y, device=dev, dtype=torch.float32)[:, None]
        xs = torch.arange(X, device=dev, dtype=torch.float32)[None, :]
        u, v = (xs / X - 0.5) * 2, (ys / Y - 0.5) * 2
        mv = torch.empty((N, Y, X), device=dev)
        for n in range(N):
            a = 6.0 * n / max(1, N - 1)
            sx = xs + 64 + float(drift[n, 0]) + a * (0.7 * u + 0.3 * v)
            sy = ys + 64 + float(drift[n, 1]) + a * (0.8 * v - 0.2 * u)
            grid = torch.stack(((sx + 0.5) * (2.0 / W) - 1, (sy + 0.5) * (2.0 / H) - 1), -1)[None]
            mv[n] = torch.nn.functional.grid_sample(base[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
            mv[n] += 0.5 * torch.randn((Y, X), generator=g, device=dev)
            del grid, sx, sy
        if args.movie_mode != 2:
            # detector counts: the same frames scaled to mean 40, sigma 12 and rounded (they fit every integer mode)
            mv = ((mv - mv.mean()) * (12.0 / mv.std()) + 40.0).round_().clamp_(0, 127).to(raw_dtype)
        hb = torch.empty(mv.shape, dtype=mv.dtype, pin_memory=True)
        hb.copy_(mv)
        del mv, base
        return hb, drift
    nuniq = 2
    host, drifts = zip(*[make_movie(11 + 2 * rank + u) for u in range(nuniq)])
    torch.cuda.empty_cache()
    import threading
    nlanes = max(1, args.fa_lanes)

    class Lane:
        """one movie in flight: a stream, a library context bound to it, a FlexAlign handle, two device buffers (the copy of this lane's
        next movie flies under the alignment of its current one) and the aligned sum"""
        def __init__(self, idx):
            self.idx = idx
            self.stream = torch.cuda.current_stream(dev) if idx == 0 else torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self.stream):
                self.ctx = ctx if idx == 0 else xa.Context(local)
                self.fa = xa.FlexAlign(self.ctx, Y, X, Ts, max_res)
                self.fa.set_option("prefilter_ahead", 1)        # the warp's prefilter of the frames runs while the host fits the spline
                for kv in args.fa_opt:
                    self.fa.set_option(kv.split("=")[0], float(kv.split("=")[1]))
                self.dbuf = [torch.empty((N, Y, X), device=dev), torch.empty((N, Y, X), device=dev)]
                # frames that arrive as counts land here and are cast into dbuf by the lane's first kernel
                self.rbuf = [torch.empty((N, Y, X), device=dev, dtype=raw_dtype) for _ in range(2)] if args.movie_mode != 2 else self.dbuf
                self.src = host
                self.total = torch.zeros((Y, X), device=dev)
            self.h_avg = [torch.empty((Y, X), dtype=torch.float32, pin_memory=True) for _ in range(2)]
            # one copy stream for all lanes: the movies cross the link one after the other anyway, and two host copies in flight on
            # two streams make the runtime move one of them with a shader kernel
            self.copy_stream = Lane.shared_copy_stream if args.fa_shared_copy else torch.cuda.Stream(device=dev)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.timers = {"global_alignment": [], "local_alignment": [], "warp_and_sum": []}
            self.results = []
            self.error = None

        def timed(self, name, record, fn):
            if not record:
                return fn()
            t = self.ctx.timer()
            t.start()
            r = fn()
            t.stop()
            self.timers[name].append(t)
            return r

        def fetch(self, j, movie):
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.done[j & 1])
                self.rbuf[j & 1].copy_(self.src[movie % nuniq], non_blocking=True)
                self.ready[j & 1].record(self.copy_stream)

        def align(self, frames, record, slot):
            fa_ = self.fa
            gl = self.timed("global_alignment", record, lambda: fa_.global_alignment(frames, max_shift))
            loc = self.timed("local_alignment", record, lambda: fa_.local_alignment(frames, gl["shiftX"], gl["shiftY"], gl["ref"], max_shift, patches, psize, 3, cp))

            def warp():
                self.total.zero_()
                fa_.apply_bspline_frames(frames, loc["coeffsX"], loc["coeffsY"], cp, total=self.total)
            self.timed("warp_and_sum", record, warp)
            self.h_avg[slot].copy_(self.total, non_blocking=True)              # the aligned micrograph goes back to the host
            return gl, loc

        def prime(self, movies):
            for e in self.done:
                e.record(self.stream)
            if movies:
                self.fetch(0, movies[0])

        def run(self, movies, record):
            # movies: the indices this lane aligns, in order
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(self.stream):
                    for j, k in enumerate(movies):
                        if j + 1 < len(movies):
                            self.fetch(j + 1, movies[j + 1])
                        self.stream.wait_event(self.ready[j & 1])
                        if self.rbuf is not self.dbuf:
                            xa.movie_frames_to_float(self.ctx, self.rbuf[j & 1], out=self.dbuf[j & 1])
                        r = self.align(self.dbuf[j & 1], record, j & 1)
                        self.done[j & 1].record(self.stream)
                        if record:
                            self.results.append((k % nuniq, r[0]))
            except BaseException as e:          # re-raised on the main thread
                self.error = e

    Lane.shared_copy_stream = torch.cuda.Stream(device=dev)
    lanes = [Lane(i) for i in range(nlanes)]
    fa = lanes[0].fa

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def streamed(nsteps, record, use=None):
        use = lanes if use is None else use
        work = [list(range(i, nsteps, len(use))) for i in range(len(use))]
        for ln, w in zip(use, work):
            ln.prime(w)
        barrier()
        ta = time.perf_counter()
        if len(use) == 1:
            use[0].run(work[0], record)
        else:
            th = [threading.Thread(target=ln.run, args=(w, record)) for ln, w in zip(use, work)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
        for ln in use:
            if ln.error is not None:
                raise ln.error
        barrier()
        return time.perf_counter() - ta

    if args.warmup:
        streamed(max(args.warmup, nlanes), False)
    elapsed = streamed(args.steps, True)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    results = [r for ln in lanes for r in ln.results]
    counts_leg = None
    if args.movie_mode == 2 and not args.no_extra_legs:
        # the same movies as 8-bit counts (what a K3 writes; MRC mode 0): a quarter of the bytes cross the link, the cast runs on the device
        host8 = []
        for hb in host:
            d = hb.to(dev)
            d = ((d - d.mean()) * (12.0 / d.std()) + 40.0).round_().clamp_(0, 127).to(torch.int8)
            h8 = torch.empty(d.shape, dtype=torch.int8, pin_memory=True)
            h8.copy_(d)
            host8.append(h8)
            del d
        for ln in lanes:
            with torch.cuda.stream(ln.stream):
                ln.rbuf = [torch.empty((N, Y, X), device=dev, dtype=torch.int8) for _ in range(2)]
            ln.src = host8
            ln.results = []
        streamed(nlanes, False)
        e8 = streamed(args.steps, False)
        if world > 1:
            t = torch.tensor([e8], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e8 = t.item()
        counts_leg = {"value": args.steps * world / e8, "unit": "movies/s", "ms_per_movie": 1e3 * e8 / args.steps, "steps": args.steps,
                      "what": f"the same streamed steps with the frames in host memory as int8 counts (MRC mode 0, {N * Y * X / 1e9:.2f} GB per movie), "
                              "cast to float on the device by xh_movie_frame_to_float; `--movie-mode 0` makes this the timed region"}
        for ln in lanes:
            ln.rbuf = ln.dbuf
            ln.src = host
        del host8
    stage_overlapped = {k_: float(sum(t.elapsed_ms() for ln in lanes for t in ln.timers[k_])) for k_ in lanes[0].timers}
    # one lane alone, movies resident in HBM: what the host traffic and the second lane are worth, and the stages' own durations (the
    # HIP-event times of the timed region include waiting for the other lane's kernels)
    for ln in lanes:
        ln.timers = {k_: [] for k_ in ln.timers}
    nres = max(2, min(args.steps, 3))
    barrier()
    tr0 = time.perf_counter()
    with torch.cuda.stream(lanes[0].stream):
        for k in range(nres):
            lanes[0].align(lanes[0].dbuf[k & 1], True, k & 1)
    barrier()
    resident = nres * world / (time.perf_counter() - tr0)
    timers = lanes[0].timers
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    stage = {k_: float(sum(t.elapsed_ms() for t in ts)) for k_, ts in timers.items()}      # of the one-lane resident leg: nres movies
    # algorithmic bytes per movie of each stage: the global alignment reads every frame once and keeps the low-frequency columns of its
    # transform (complex, nY x (nX / 2 + 1)); the local alignment reads every patch of every frame and of the two frames averaged with it;
    # the warp reads a frame and writes (accumulates) a frame per frame
    nY, nX = fa.new_dims
    by = {"global_alignment": N * (Y * X * 4 + nY * (nX // 2 + 1) * 8),
          "local_alignment": patches[0] * patches[1] * psize[0] * psize[1] * N * 3 * 4,
          "warp_and_sum": N * Y * X * 8}
    kern = {"global_alignment": "k_fft2d rows/columns of the frame transforms + pair correlations (xh_fa_global_alignment)",
            "local_alignment": "k_fa_gather + k_fa_gemm_mfma + pair windows (xh_fa_local_alignment)",
            "warp_and_sum": "k_fa_prefilter + k_fa_warp (xh_fa_apply_bspline, 40 frames)"}
    dom = max(stage, key=lambda k_: stage[k_])
    mk = lambda k_: {"kernel": kern[k_], "bound": "hbm", "achieved": nres * by[k_] / (stage[k_] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": nres * by[k_] / (stage[k_] * 1e-3) / 1e9 / 8000.0, "traffic": None, "ms_one_lane_leg": stage[k_], "movies_one_lane_leg": nres,
                     "avg_ms_per_movie": stage[k_] / nres, "algorithmic_bytes_per_movie": by[k_]}
    # physics check of the timed movies: the drift that was put in comes out of the global alignment
    err = 0.0
    for u, gl in results:
        tt = drifts[u] - drifts[u][gl["ref"]]
        err = max(err, float(np.abs(gl["shiftX"] + tt[:, 0]).max()), float(np.abs(gl["shiftY"] + tt[:, 1]).max()))
    out = {"metric": f"movies/s FlexAlign (global + local alignment, B-spline warp + sum), {X}x{Y}x{N} frames", "value": args.steps * world / elapsed,
           "unit": "movies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (transforms, correlations, warp) + f64 (shift solves, spline fit on the host)",
           "data": "synthetic",
           "config": {"workload": f"FlexAlign movie alignment, {N} frames of {Y}x{X} (BASELINE config 5), patch cross-correlation path: {patches[0]}x{patches[1]} patches of "
                                  f"{req} px, control points {cp}, {max_res} A at {Ts} A/px", "mode": "flexalign", "movies_total": args.steps * world,
                      "unique_movies_per_gpu": nuniq,
                      "host_traffic": f"every movie ({N * Y * X * host[0].element_size() / 1e9:.2f} GB of {str(raw_dtype).replace('torch.', '')} frames) H2D from page-locked memory inside the timed region (two device buffers, "
                                      "copy stream per lane), the aligned sum D2H", "lanes_per_gpu": nlanes, "HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA", "unset"),
                      "parallelism": f"movie replicas x{world}, no exchange; {nlanes} movies in flight per GPU (host threads, one stream and library handle each)"},
           "roofline": mk(dom), "roofline_other_kernels": {k_: mk(k_) for k_ in stage if k_ != dom},
           "stage_ms": {k_: v / nres for k_, v in stage.items()},
           "stage_ms_note": "per movie, HIP events of one lane alone on movies resident in HBM (the one-lane leg after the timed region); "
                            "stage_ms_timed_region are the same events inside the timed region, per movie, where they include waiting for the other lanes' kernels",
           "stage_ms_timed_region": {k_: v / args.steps for k_, v in stage_overlapped.items()},
           "value_resident_one_lane": resident, "global_shift_error_px": err}
    if counts_leg:
        out["int8_frames_leg"] = counts_leg
    if not args.no_cpu_baseline and world == 1:
        # the oracle's global alignment (ProgMovieAlignmentCorrelation<double>'s arithmetic, one thread) on the first 2 and the first 3 frames
        # of movie 0 at full size: T(n) = n t_frame + n (n - 1) / 2 t_pair, extrapolated to the movie's frames.  The local alignment and
        # the warp are NOT in it (the reference has no CPU form of the patch path): a lower bound of the CPU cost of a movie.
        from oracle import pyoracle as o
        f3 = host[0][:3].to(torch.float32).numpy()
        tb = []
        for n in (2, 3):
            t0 = time.perf_counter()
            og = o.fa_global_alignment(f3[:n], Ts=Ts, max_shift_px=max_shift, max_res=max_res)
            tb.append(time.perf_counter() - t0)
        # T(2) = 2 f + p, T(3) = 3 f + 3 p  =>  p = (2 T(3) - 3 T(2)) / 3, f = (T(2) - p) / 2
        pp = max(0.0, (2 * tb[1] - 3 * tb[0]) / 3.0)
        ff = (tb[0] - pp) / 2.0
        movie_s = N * ff + N * (N - 1) / 2 * pp
        dg = fa.global_alignment(torch.from_numpy(f3).to(dev), max_shift)
        out["cpu_baseline"] = {"value": 1.0 / movie_s, "unit": "movies/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                               "frame_s": ff, "pair_s": pp,
                               "sample": f"oracle global alignment of the first 2 ({tb[0]:.1f} s) and 3 ({tb[1]:.1f} s) frames of movie 0 at full size, one thread, "
                                         f"extrapolated to {N} frames and {N * (N - 1) // 2} pairs; local alignment and warp not included"}
        pdiff = float(max(np.abs(dg["bX"] - og["bX"]).max(), np.abs(dg["bY"] - og["bY"]).max()))
        out["parity_sample"] = {"what": "pair shifts of the device against the oracle on those 3 frames (px)", "max_abs_diff": pdiff,
                                "bound_px": FA_PARITY_BOUND_PX, "ok": pdiff <= FA_PARITY_BOUND_PX}
    # the drift that was put in comes out: the frames are correlated at 4.4 px per reduced pixel and bestShift is a centre of mass
    out["global_shift_error_bound_px"] = FA_DRIFT_BOUND_PX
    out["global_shift_error_ok"] = err <= FA_DRIFT_BOUND_PX
    print(json.dumps(out), flush=True)
    if not out["global_shift_error_ok"] or not out.get("parity_sample", {"ok": True})["ok"]:
        # a parity figure that drifts is a failure of this leg, not a footnote (round 5: 1.5e-6 -> 7.9e-4 px went by unasserted)
        raise SystemExit(f"bench.py --mode flexalign: parity sample {out.get('parity_sample')} / recovered drift error {err:.2f} px exceed their bounds")
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: build once (a child process, nothing here touches the GPU), then start the ranks
        rc = subprocess.call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, stdout=sys.stderr)
        if rc != 0:
            sys.exit(rc)
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.sdma >= 0:
        os.environ["HSA_ENABLE_SDMA"] = str(args.sdma)          # read when the runtime starts: before torch is imported
    elif args.mode == "flexalign" and "HSA_ENABLE_SDMA" not in os.environ:
        # the 3.77 GB of a movie go to the device while another movie is being aligned: with the variable unset the runtime moves them
        # with a shader kernel (__amd_rocclr_copyBuffer in the kernel trace) that takes CUs from the alignment -- 12.2 movies/s; on the
        # copy engines 14.1 (and 8.9 with the engines switched off).  The main path's copies (1 GB per 43 ms step) do not care.
        os.environ["HSA_ENABLE_SDMA"] = "1"
    if args.mode == "flexalign":
        return main_flexalign(args)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself; under torchrun pass the same N)")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus
        # one rank per node builds (csrc/build.sh and host/build.sh write into shared directories), the others wait
        if local == 0:
            ge.build()
        dist.barrier()
    else:
        ge.build()
    import xmipp3_amd as xa
    from tests import synth

    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    masked_side = None
    if args.cu_split > 0:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")

        def masked_stream(pred):
            mask = (ctypes.c_uint32 * 8)()
            for i in range(256):
                if pred(i):
                    mask[i // 32] |= 1 << (i % 32)
            sp = ctypes.c_void_p()
            rc_ = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, mask)
            assert rc_ == 0, rc_
            return torch.cuda.ExternalStream(sp.value, device=dev)
        q_ = args.cu_split
        torch.cuda.set_stream(masked_stream(lambda i: (i // 8) % 4 < q_))
        masked_side = masked_stream(lambda i: (i // 8) % 4 >= q_)
    if args.main_priority != 0 and args.cu_split == 0:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=args.main_priority))
    ctx = xa.Context(local)
    D, nrefs, B = args.box, args.nrefs, args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    genr = torch.Generator(device=dev)
    genr.manual_seed(7)

    # ---- synthetic inputs, resident in HBM before the timed region
    dirs = synth.fibonacci_directions(nrefs)
    dirs_dev = torch.from_numpy(np.ascontiguousarray(dirs[:, :2], np.float64)).to(dev)
    if args.refs in ("phantom", "compact"):
        # BASELINE config 2/4: the gallery = projections of the phantom at Fibonacci-sphere directions, made with the
        # library's own central-slice projector (xh_fp_*, the xmipp_angular_project_library path)
        # (--refs compact: every blob within 0.3 of the box radius, the nearly rotation-invariant map of the compact_phantom leg)
        fpj = xa.FourierProjector(ctx, phantom_volume(torch, D, genr, dev, **({"rmax": 9.6} if args.refs == "compact" else {})), 2.0, 0.5, 3)
        refs = fpj.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1))
        fpj.close()
        refs = ((refs - refs.mean()) / refs.std()).contiguous()
    else:
        refs = smooth_noise(torch, nrefs, D, genr, dev)
    def make_batch(refs=refs):
        """particle = reference, random in-plane rotation, mirror with p = 0.5, shift U{-3..3}^2, white noise at SNR 0.1"""
        idx = torch.randint(0, nrefs, (B,), generator=gen, device=dev)
        th = torch.rand((B,), generator=gen, device=dev) * (2 * math.pi)
        mir = (torch.rand((B,), generator=gen, device=dev) < 0.5).float() * 2 - 1          # -1: mirrored in x
        shf = torch.randint(-3, 4, (B, 2), generator=gen, device=dev).float()
        rot = torch.zeros((B, 2, 3), device=dev)
        rot[:, 0, 0] = torch.cos(th) * mir; rot[:, 0, 1] = -torch.sin(th); rot[:, 1, 0] = torch.sin(th) * mir; rot[:, 1, 1] = torch.cos(th)
        rot[:, :, 2] = shf * (2.0 / D)
        out = torch.empty((B, D, D), device=dev)
        for b0 in range(0, B, 512):
            sl = slice(b0, min(B, b0 + 512))
            grid = torch.nn.functional.affine_grid(rot[sl], (rot[sl].shape[0], 1, D, D), align_corners=False)
            out[sl] = torch.nn.functional.grid_sample(refs[idx[sl]][:, None], grid, mode="bilinear", padding_mode="zeros",
                                                      align_corners=False)[:, 0]
        return (out + math.sqrt(10.0) * torch.randn((B, D, D), generator=gen, device=dev)).contiguous(), idx

    # >= 16384 distinct particles cycled (BASELINE config 3: fresh orientations per use); they live in page-locked HOST memory and
    # are streamed through two device buffers inside the timed region; batch 0 also stays on the device for the resident leg,
    # the CPU sample and the parity checks
    nuniq = args.unique_batches or 4
    host, h_truth = [], []
    particles = None
    for u in range(nuniq):
        b_, i_ = make_batch()
        hb = torch.empty(b_.shape, dtype=b_.dtype, pin_memory=True)
        hb.copy_(b_)
        host.append(hb)
        h_truth.append(i_.cpu().numpy())
        if u == 0:
            particles = b_
        del b_
    # --neighbours K: the K references nearest (angular distance of the projection directions) to the particle's own
    nbr = None
    if args.neighbours > 0 and args.mode != "grid":
        K = min(args.neighbours, nrefs)
        rt = np.radians(dirs[:, :2])
        v = np.stack([np.sin(rt[:, 1]) * np.cos(rt[:, 0]), np.sin(rt[:, 1]) * np.sin(rt[:, 0]), np.cos(rt[:, 1])], 1)
        table = np.sort(np.argsort(-(v @ v.T), axis=1, kind="stable")[:, :K], axis=1).astype(np.int32)
        nbrs = [((np.arange(B + 1) * K).astype(np.int32), np.ascontiguousarray(table[h_truth[u]].ravel())) for u in range(nuniq)]
        nbr = nbrs[0]
    rng = np.random.default_rng(100 + rank)
    from xmipp3_amd.api import ctf_params
    ctfs = [ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(d), DeltafV=float(d))
            for d in rng.uniform(10000.0, 30000.0, B)]
    ctf_arr = xa.RecFourier.ctf_param_array(ctfs)

    pm = xa.ProjectionMatcher(ctx, refs) if args.mode != "grid" else None
    for o_ in (args.pm_opt if pm is not None else []):
        k_, v_ = o_.split("=")
        pm.set_option(k_, float(v_))
    if pm is not None and args.no_prune:
        pm.set_option("prune", 0)
    if pm is not None and args.tr_chunk_mb > 0:
        pm.set_option("tr_chunk_mb", args.tr_chunk_mb)
    if pm is not None and args.chunk_rows > 0:
        pm.set_option("chunk_rows", args.chunk_rows)
    if pm is not None and args.k0 >= 0:
        pm.set_option("k0", args.k0)
    if pm is not None and args.tau_rel > 0:
        pm.set_option("tau_rel", args.tau_rel)
    rows_seen = [0, 0]      # correlation rows searched / skipped by the S3 branch and bound, timed steps only
    s6_rep = [0]            # particles whose translational alignment was repeated in double precision, timed steps only
    # --pipeline: the reconstructor lives on a second stream (its own library context), so that the gridding of batch k
    # (LDS / VALU bound) runs beside the matching of batch k+1 (mostly HBM bound)
    import contextlib
    pipelined = bool(args.pipeline) and args.mode == "full"
    side = (masked_side or torch.cuda.Stream(device=dev, priority=args.side_priority)) if pipelined else None
    ctx_rf = ctx
    if pipelined:
        with torch.cuda.stream(side):
            ctx_rf = xa.Context(local)

    def on_rf_stream():
        return torch.cuda.stream(side) if pipelined else contextlib.nullcontext()
    rf = xa.RecFourier(ctx_rf, D, min_ctf=0.01, sampling=1.0) if args.mode != "match" else None
    for o_ in (args.rf_opt if rf is not None else []):
        k_, v_ = o_.split("=")
        rf.set_option(k_, float(v_))
    # HIP-event pairs on the library's stream around the stages xh_pm_stage_ms does not cover; read after the timed region
    timers = {"translate_s6": [], "shift_images": [], "gridding_insert_images": []}

    def timed(name, record, fn, c=None):
        if not record:
            return fn()
        t = (c or ctx).timer()
        t.start()
        r = fn()
        t.stop()
        timers[name].append(t)
        return r

    # results of a step as a host would receive them: page-locked, copied behind the step's kernels without a sync
    h_out = None
    if pm is not None:
        h_out = [{"refno": torch.empty(B, dtype=torch.int32, pin_memory=True), "psi": torch.empty(B, dtype=torch.int32, pin_memory=True),
                  "flip": torch.empty(B, dtype=torch.uint8, pin_memory=True), "sx": torch.empty(B, dtype=torch.float64, pin_memory=True),
                  "sy": torch.empty(B, dtype=torch.float64, pin_memory=True), "cc": torch.empty(B, dtype=torch.float64, pin_memory=True)}
                 for _ in range(2)]

    def step(record, parts=None, u=0, slot=None, pipe=True, pm=pm):
        ang = flips = None
        if parts is None:
            parts = particles
        imgs = parts
        if pm is not None:
            nb = nbrs[u] if nbr is not None else None
            refno, psi, flip = pm.match(parts, *nb) if nb is not None else pm.match(parts)
            if record:
                st = pm.last_stats()
                rows_seen[0] += st["rows"]; rows_seen[1] += st["pruned_rows"]
            sx, sy, cc = timed("translate_s6", record, lambda: pm.translate(parts, refno, psi, flip))
            if record:
                s6_rep[0] += pm.translate_repeated()
            if slot is not None and args.stream_test != 1:
                ho = h_out[slot]
                for k_, t_ in (("refno", refno), ("psi", psi), ("flip", flip), ("sx", sx), ("sy", sy), ("cc", cc)):
                    ho[k_].copy_(t_, non_blocking=True)
            # the orientations stay on the device: (rot, tilt) of the matched reference, psi from the sample index
            ang = torch.cat([dirs_dev[refno.long()], (psi.double() * (360.0 / pm.N))[:, None]], 1).contiguous()
            shifts = (sx, sy)
            flips = flip
        else:
            ang = synth.random_angles(B, rng)
            shifts = rng.uniform(-3, 3, (B, 2))
        if rf is not None and pipelined:
            matched = torch.cuda.Event()
            matched.record()
            shifted = torch.cuda.Event()
            with torch.cuda.stream(side):
                side.wait_event(matched)
                for t_ in (parts, sx, sy, flip, ang):
                    t_.record_stream(side)
                imgs = timed("shift_images", record, lambda: rf.shift_images(parts, shifts, flips=flips, coefs=pm.last_coefficients(B)), ctx_rf)
                shifted.record(side)
                timed("gridding_insert_images", record, lambda: rf.insert_images(imgs, ang, ctf_array=ctf_arr), ctx_rf)
            # the next match overwrites the coefficients (and the copy stream the batch) the shift has just read
            torch.cuda.current_stream().wait_event(shifted)
            if not pipe:                                    # one-stream leg: the matching waits for the gridding too
                gridded = torch.cuda.Event()
                gridded.record(side)
                torch.cuda.current_stream().wait_event(gridded)
        elif rf is not None:
            # the matcher has just computed the particles' B-spline coefficients: the shift reuses them
            imgs = timed("shift_images", record, lambda: rf.shift_images(parts, shifts, flips=flips, coefs=pm.last_coefficients(B) if pm is not None else None))
            timed("gridding_insert_images", record, lambda: rf.insert_images(imgs, ang, ctf_array=ctf_arr))      # CTF planes + FFT + records + gridding

    fin_ev = []          # (start, end) event pairs of the tail on the reconstruction stream: its duration on the DEVICE

    def finish():
        if rf is None:
            return
        with on_rf_stream():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rf.mirror_and_crop()
            xa.allreduce_reconstruction(rf)
            rf.finish(reuse=True)          # (the volume is not looked at here: one page-locked result buffer for every call)
            rf.reset()
            e1.record()
            fin_ev.append((e0, e1))

    # the batches of the resident form: all of them in HBM before the clock starts (4 GB at the defaults), cycled like the host ones;
    # the warm-up steps cycle through them too (a batch the device has never read costs its first reader page-table walks)
    dres = [particles] + [h.to(dev) for h in host[1:]] if args.timed == "resident" or not args.no_extra_legs else [particles]
    for w_ in range(args.warmup):
        step(False, dres[w_ % len(dres)], w_ % len(dres))
    for b_ in dres[min(len(dres), max(1, args.warmup)):]:
        b_.sum()                                          # (fewer warm-up steps than batches: one read of the rest, outside the clock)
    if args.warmup:
        finish()
    if pm is not None:
        pm.stage_ms(reset=True)
    if rf is not None:
        rf.kernel_ms(reset=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: batches streamed from page-locked host memory through two device buffers
    dbuf = [torch.empty_like(particles), torch.empty_like(particles)]
    copy_stream = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]

    def fetch(k):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[k & 1])          # the step that last used this buffer has finished
            if args.stream_test == 2:
                dbuf[k & 1].copy_(particles, non_blocking=True)
            elif args.stream_test == 3:
                for c0 in range(0, B, B // 16):
                    dbuf[k & 1][c0:c0 + B // 16].copy_(host[k % nuniq][c0:c0 + B // 16], non_blocking=True)
            else:
                dbuf[k & 1].copy_(host[k % nuniq], non_blocking=True)
            ready[k & 1].record(copy_stream)

    def streamed_steps(nsteps, record):
        for e in done:
            e.record()
        fetch(0)                                          # the prologue copy: before the clock starts
        barrier()
        ta = time.perf_counter()
        for k in range(nsteps):
            if k + 1 < nsteps:
                fetch(k + 1)                              # flies under step k
            torch.cuda.current_stream().wait_event(ready[k & 1])
            step(record, dbuf[k & 1], k % nuniq, k & 1)
            done[k & 1].record()
        tb = time.perf_counter()
        finish()
        barrier()
        return ta, tb, time.perf_counter()

    def resident_steps(nsteps, record):
        barrier()
        ta = time.perf_counter()
        for k in range(nsteps):
            step(record, dres[k % len(dres)], k % len(dres), k & 1)
        tb = time.perf_counter()
        finish()
        barrier()
        return ta, tb, time.perf_counter()

    t0, t_fin0, t1 = (resident_steps if args.timed == "resident" else streamed_steps)(args.steps, True)
    elapsed = t1 - t0
    # the once-per-run tail (mirror/crop + all-reduce + finaliser + reset) as the device saw it: events on the reconstruction stream
    # around it.  (The host clock from "all steps queued" to the end -- what rounds 3-4 printed here -- also holds the steps the
    # device had not finished yet: 37 ms where the tail itself is 8.)
    finish_s = fin_ev[-1][0].elapsed_time(fin_ev[-1][1]) * 1e-3 if fin_ev else 0.0
    finish_host_s = t1 - t_fin0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    stage = pm.stage_ms(reset=False) if pm is not None else {}
    stage = dict(stage)
    k_ms, k_launches = rf.kernel_ms(reset=False) if rf is not None else (0.0, 0)
    stats_timed = pm.last_stats() if pm is not None else None

    # ---- extra legs, outside the timed region (every rank runs them: same collectives, same barriers)
    extra = {}
    if not args.no_extra_legs:
        # (1) the same steps + finish fed the other way: what the host traffic costs is value_resident - value_streamed
        nres = max(2, min(args.steps, 4))
        nleg = max(2, args.steps)          # the same number of steps as `value`: both forms amortise the tail alike
        ta_, _, tb_ = (streamed_steps if args.timed == "resident" else resident_steps)(nleg, False)
        el = tb_ - ta_
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        if args.timed == "resident":
            extra["value_streamed"] = nleg * B * world / el
            extra["ms_per_step_streamed"] = 1e3 * el / nleg
            extra["streamed_leg"] = {"steps": nleg, "what": "SURVEY.md 8d's PCIe-inclusive rate (`value` of rounds 3-4): the same steps + finish with every batch copied "
                                                             "from page-locked host memory inside the clock (two device buffers, the copy of batch k + 1 under step k), results "
                                                             "copied back; on a box whose host link is shared with other jobs it falls below `value` (seen: 85 k against 102 k)"}
        else:
            extra["value_resident"] = nleg * B * world / el
            extra["ms_per_step_resident"] = 1e3 * el / nleg
            extra["resident_leg"] = {"steps": nleg, "what": "same steps + finish on batches resident in HBM, no batch copies"}
        if pipelined:
            # (1b) the same without the second stream: every stage of a step behind the one before it
            if rf is not None:
                rf.kernel_ms(reset=True)
            stage_timed = pm.stage_ms(reset=True) if pm is not None else {}       # (the timed region's, kept; the timers restart)
            barrier()
            ts0 = time.perf_counter()
            for _ in range(nres):
                step(False, pipe=False)
            finish()
            barrier()
            el = time.perf_counter() - ts0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            k1, n1 = rf.kernel_ms(reset=False)
            st1 = pm.stage_ms(reset=True) if pm is not None else {}
            extra["one_stream_leg"] = {"value": nres * B * world / el, "steps": nres, "k_rf_grid_avg_launch_ms": k1 / max(1, n1),
                                       "matcher_stage_ms_per_step": {k_: v_ / nres for k_, v_ in st1.items()},
                                       "what": "resident batch, reconstruction half behind the matching on the same timeline: what the second stream buys, "
                                               "and the gridding kernel's duration without the other stream's kernels beside it"}
        # (1c) the matching half alone (rotational search + translational alignment, results copied back): north_star's target is stated
        # for it -- >= 2 M particles/s projection-matched at 8 GPUs = 250 k per GPU -- while `value` is the whole refine iteration
        if pm is not None and args.mode == "full":
            barrier()
            pm.stage_ms(reset=True)
            nm = 6
            outs = None
            tm0 = time.perf_counter()
            for _ in range(nm):
                refno_, psi_, flip_ = pm.match(particles)
                outs = (refno_, psi_, flip_) + tuple(pm.translate(particles, refno_, psi_, flip_))
                host_ = [t_.cpu() for t_ in outs]            # every step's results back on the host
            torch.cuda.synchronize()
            barrier()
            el = time.perf_counter() - tm0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            extra["match_only_leg"] = {"value": nm * B * world / el, "unit": "particles/s projection-matched", "steps": nm, "ms_per_step": 1e3 * el / nm,
                                       "matcher_stage_ms_per_step": {k_: v_ / nm for k_, v_ in pm.stage_ms(reset=True).items()},
                                       "north_star_target_per_gpu": 250000.0,
                                       "what": "the same resident batch through xh_pm_match + xh_pm_translate only (no shift, CTF, FFT, gridding): the rate the "
                                               "north_star's 2 M particles/s at 8 GPUs is stated for; `--mode match` times the same thing as the headline"}
            del host_, outs
        # (2) the matcher without its data-dependent shortcuts: every correlation row contracted over all frequencies and
        # transformed (S3 branch and bound off, two-level cut off)
        if pm is not None and not args.no_prune:
            pm.set_option("prune", 0)
            pm.set_option("k0", 1 << 20)
            step(False)            # warm-up of the other code path
            finish()
            barrier()
            pm.stage_ms(reset=True)
            tw0 = time.perf_counter()
            nw = 2
            for _ in range(nw):
                step(False)
            finish()
            barrier()
            el = time.perf_counter() - tw0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            extra["worst_case"] = {"value": nw * B * world / el, "unit": "particles/s", "steps": nw,
                                   "matcher_stage_ms_per_step": {k_: v_ / nw for k_, v_ in pm.stage_ms(reset=True).items()},
                                   "what": "branch and bound of the row transforms off, contraction over all frequencies: "
                                           "what a gallery and particles with flat correlation peaks would cost"}
            pm.set_option("prune", 1)
            pm.set_option("k0", 0 if args.k0 < 0 else args.k0)
        # (3) the other end of the data dependence: a gallery of unrelated band-limited noise images (sharp correlation peaks, no band
        # limit for the two-level cut to use) -- the same step on a resident batch made from that gallery
        if pm is not None and args.refs == "phantom" and args.neighbours == 0:
            refs_n = smooth_noise(torch, nrefs, D, genr, dev)
            pm_n = xa.ProjectionMatcher(ctx, refs_n)
            parts_n, _ = make_batch(refs_n)
            step(False, parts_n, pm=pm_n)
            finish()
            barrier()
            tn0 = time.perf_counter()
            nn = 2
            for _ in range(nn):
                step(False, parts_n, pm=pm_n)
            finish()
            barrier()
            el = time.perf_counter() - tn0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            extra["noise_gallery"] = {"value": nn * B * world / el, "unit": "particles/s", "steps": nn,
                                      "what": "the same step (resident batch) with 1000 unrelated band-limited noise images as references and "
                                              "particles made from them: the value of `--refs noise`; with worst_case the honest range of the headline"}
            pm_n.close()
            del pm_n, refs_n, parts_n
            # (4) a map whose correlation peaks are FLAT in the in-plane angle: the compact phantom of the tests (every blob within 0.3 of
            # the box radius of the centre) is nearly rotation invariant, neighbouring in-plane angles differ by less than the fp32 margin
            # and most particles take the fp64 re-score (the default gallery: 13 %).  The headline assumes a friendlier map; this is
            # what it costs when the map is not.
            fpc = xa.FourierProjector(ctx, phantom_volume(torch, D, genr, dev, rmax=9.6), 2.0, 0.5, 3)
            refs_c = fpc.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1))
            fpc.close()
            refs_c = ((refs_c - refs_c.mean()) / refs_c.std()).contiguous()
            pm_c = xa.ProjectionMatcher(ctx, refs_c)
            parts_c, _ = make_batch(refs_c)
            step(False, parts_c, pm=pm_c)
            finish()
            barrier()
            pm_c.stage_ms(reset=True)
            tc0 = time.perf_counter()
            for _ in range(2):
                step(False, parts_c, pm=pm_c)
            finish()
            barrier()
            el = time.perf_counter() - tc0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            extra["compact_phantom"] = {"value": 2 * B * world / el, "unit": "particles/s", "steps": 2,
                                        "rescored_fraction_compact_phantom": pm_c.last_stats()["rescored_particles"] / float(B),
                                        "matcher_stage_ms_per_step": {k_: v_ / 2 for k_, v_ in pm_c.stage_ms(reset=True).items()},
                                        "matcher_stats": {k_: v_ for k_, v_ in pm_c.last_stats().items() if not hasattr(v_, "__len__")},
                                        "what": "the same step (resident batch) with the gallery of a compact, nearly rotation-invariant phantom "
                                                "(blobs within 0.3 of the box radius): flat correlation peaks in the in-plane angle, most particles re-scored in fp64"}
            pm_c.close()
            del pm_c, refs_c, parts_c

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_particles = args.steps * B * world
    value = total_particles / elapsed
    for name, ts in timers.items():                         # translate (S6), shift, insert_images = CTF + FFT + records + spaces + kernel
        if ts:
            stage[name] = float(sum(t.elapsed_ms() for t in ts))
    stage["k_rf_grid"] = k_ms                                # HIP events around the kernel launches only
    # ---- roofline of the dominant kernel (per launch = per chunk; reported per particle-second)
    N = pm.N if pm is not None else 2 * int(math.pi * (D // 2 - 1))
    ncoef = pm.ncoef if pm is not None else 0
    rows = args.steps * B * nrefs
    mv = rf.mv if rf is not None else 2 * D
    nvox = math.pi * (mv / 2) ** 2 / 2 * 2 * 1.9          # slab voxels per projection
    bytes_grid = 4 * D * D + nvox * 24                     # image read + 12 B read + 12 B write per voxel
    cand = {}
    if pm is not None and stage.get("idft_max", 0) > 0:
        # EXECUTED work only. Rows transformed: 5 N log2 N flops of a packed complex inverse DFT each; contraction:
        # 4 real FMAs per (row, ring coefficient below the two-level cut K0) -- the rows that survive the branch and
        # bound get their remaining frequencies inside k_pm_idft_max, counted there
        K0, nk = pm.two_level_cut()
        surv = rows - rows_seen[1]
        frac_low = min(1.0, K0 / float(nk))
        fl = surv * (5.0 * N * math.log2(N) + 8.0 * ncoef * (1.0 - frac_low))
        cand["k_pm_idft_max"] = ("valu", fl / (stage["idft_max"] * 1e-3) / 1e12, 157.3, "TFLOP/s", stage["idft_max"],
                                 f"{surv} of {rows} rows transformed (the rest pruned by the branch and bound)")
        fl2 = rows * 8.0 * ncoef * frac_low
        cand["k_pm_contract"] = ("mfma", fl2 / (stage["contract"] * 1e-3) / 1e12, 157.3, "TFLOP/s", stage["contract"],
                                 f"frequencies below the two-level cut K0 = {K0} of {nk}")
    if rf is not None and k_ms > 0:
        by = args.steps * B * bytes_grid
        cand["k_rf_grid"] = ("hbm", by / (k_ms * 1e-3) / 1e9, 8000.0, "GB/s", k_ms, "algorithmic bytes of SURVEY.md 8d")
    dom = max(cand, key=lambda k: cand[k][4]) if cand else None
    roofline = None
    if dom:
        b, ach, peak, unit, ms, note = cand[dom]
        roofline = {"kernel": dom, "bound": b, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                    "traffic": None, "ms_in_timed_region": ms, "counted": note}
        if dom == "k_rf_grid":
            roofline["launches"] = k_launches
            roofline["avg_launch_ms"] = k_ms / max(1, k_launches)
            roofline["algorithmic_bytes_per_launch"] = B * bytes_grid
            roofline["algorithmic_MB_per_projection"] = bytes_grid / 1e6
            # HBM bytes per launch from the PMC passes of tools/collect_traffic.sh over this same command
            # (FETCH_SIZE x2 on gfx950 + WRITE_SIZE); PMC cannot be read from inside the process
            # Both side files carry the hash of the library sources they were collected with (tools/libhash.py); collected with other
            # sources than the ones this run was built from they are printed with "stale": true
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            from libhash import source_hash
            lib_sha = source_hash()
            roofline["library_source_sha16"] = lib_sha
            tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_k_rf_grid.json")
            if os.path.exists(tf):
                tj = json.load(open(tf))
                if tj.get("projections_per_launch") == B:
                    roofline["traffic"] = tj["traffic_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/traffic_k_rf_grid.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
                    roofline["traffic_stale"] = tj.get("library_source_sha16") != lib_sha
            # What the formulation itself can reach (the kernel is closed, its ceiling printed beside the HBM fraction):
            # the USEFUL arithmetic of a launch -- a (voxel, pixel) pair inside the blob is three multiply-adds (re, im, weight) -- against
            # the fp32 vector peak.  A slab voxel has pi rho^2 pixels within reach, rho^2 = r^2 - z^2, z uniform in [-r, r]: 2 pi r^2 / 3.
            r_blob = 1.9
            pairs = B * nvox * (2.0 * math.pi * r_blob * r_blob / 3.0)
            roofline["arithmetic"] = {
                "useful_pairs_per_launch": pairs, "useful_flops_per_launch": 6.0 * pairs,
                "achieved_TFLOPs": 6.0 * pairs / (k_ms / max(1, k_launches) * 1e-3) / 1e12, "fp32_vector_peak_TFLOPs": 157.3,
                "frac": 6.0 * pairs / (k_ms / max(1, k_launches) * 1e-3) / 1e12 / 157.3,
                "what": "useful tap multiply-adds of a launch / fp32 vector peak.  The kernel issues ~43 lane-instructions per useful (voxel, pixel) pair "
                        "(profiles/pmc_k_rf_grid.json: SQ_INSTS_VALU x 64 / pairs) where the three multiply-adds are 2 (one packed): every tap slot of the "
                        "4 x 4 footprint pays distance, table index, compare + select, convert, shift and the table read whether the pixel lies inside the "
                        "blob or not (47 % do), batches are 67 % full, and the reference's float expressions are kept operation for operation so that the "
                        "voxel sets and table entries stay the reference's -- that, not HBM, is the ceiling of this formulation (DESIGN.md 5)"}
            # the kernel's own bound is not HBM: vector issue and the LDS pipeline, from the committed PMC passes
            # of tools/pmc_grid.sh over the same 4096-projection launch (counters cannot be read from inside the process)
            pf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_k_rf_grid.json")
            if os.path.exists(pf) and B == 4096 and D == 256:
                pj = json.load(open(pf))
                c = {k_: v_["mean"] for k_, v_ in pj["counters_per_dispatch"].items()}
                if all(k_ in c for k_ in ("SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")):
                    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the eight XCDs
                    roofline["arithmetic"]["lane_instructions_per_useful_pair"] = 64.0 * c["SQ_INSTS_VALU"] / pairs
                    roofline["second_bound"] = {
                        "bound": "vector issue + LDS pipeline", "kernel_cycles": cyc,
                        "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
                        "valu_issue_frac_at_4_cycles_per_instruction": 4.0 * c["SQ_INSTS_VALU"] / (1024.0 * cyc),
                        "lds_pipeline_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc),
                        "waves_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                        "stale": pj.get("library_source_sha16") != lib_sha,
                        "source": "profiles/pmc_k_rf_grid.json (rocprofv3 --pmc, separate passes, tools/pmc_grid.sh); a packed fp32 "
                                  "instruction costs 5.2 and a conversion 4+ cycles (profiles/experiments/r03_ubench_valu_lds.txt), so the issue "
                                  "fraction at the real mix is ~1.15 x the 4-cycle figure"}
    others = {k: {"bound": v[0], "achieved": v[1], "peak": v[2], "unit": v[3], "frac": v[1] / v[2], "ms": v[4], "counted": v[5]}
              for k, v in cand.items() if k != dom}

    out = {
        "metric": "particles/s projection-matched+reconstructed, 256x256 box",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "value_definition": args.timed, "value_definition_changed_in": "r05 (rounds 1-4: streamed; both forms are in every line since r05, over equal step counts since r06)",
        "vs_baseline": None, "dtype": "f32 (coarse search, gridding) + f64 (exact re-score, shifts, finaliser)",
        "data": "synthetic",
        "config": {"workload": {"full": f"full refine iteration (match + CTF + reconstruct), {D}x{D} particles vs {nrefs} references (BASELINE config 4 per GPU)",
                                "match": f"projection matching only (rotational + translational), {D}x{D} particles vs {nrefs} references (BASELINE config 2 shape)",
                                "grid": f"Fourier gridding only (shift + CTF + FFT + insertion + finish), {D}x{D} projections into a {D}^3 volume (BASELINE config 3 shape)"}[args.mode],
                   "mode": args.mode, "box": D, "nrefs": nrefs, "references": args.refs, "neighbours": args.neighbours or None,
                   "particles_per_step_per_gpu": B,
                   "particles_total": total_particles, "unique_particles_per_gpu": nuniq * B,
                   "timed": args.timed,
                   "host_traffic": ("batches resident in HBM when the clock starts (the task statement's `value`), results D2H inside the timed region; the rate with "
                                    "every batch H2D inside the clock is `value_streamed`" if args.timed == "resident" else
                                    "every batch H2D from page-locked memory (double-buffered, second stream), results D2H, inside the timed region"),
                   "pipeline": ("reconstruction half of batch k on a second stream beside the matching of batch k+1" if pipelined else "one stream"),
                   "parallelism": f"particle shards x{world}, one all-reduce"},
        "roofline": roofline, "roofline_other_kernels": others,
        "stage_ms": stage, "finish_and_allreduce_s": finish_s, "host_clock_from_last_step_queued_to_end_s": finish_host_s,
        # the step rate without the once-per-run tail (mirror/crop + all-reduce + finaliser): config 4 is 1 M particles = 30.5 steps of
        # 4096 per GPU on 8 GPUs, so a run of fewer steps over-weights the tail in `value` -- run --steps 31 for that comparison
        "value_steps_only": total_particles / max(1e-9, elapsed - finish_s),
        "steps_for_config4_per_gpu": int(math.ceil(1e6 / 8 / B)),
    }
    if pipelined:
        out["stage_ms_note"] = ("HIP-event times of the matcher's stages (prep32 ... translate_s6) and of gridding_insert_images are taken on two streams that "
                                "share the device: they include WAITING for the other stream's kernels (k_rf_grid holds every CU), not work alone; "
                                "one_stream_leg.matcher_stage_ms_per_step and one_stream_leg.k_rf_grid_avg_launch_ms are the same stages without a neighbour")
    out.update(extra)
    if pm is not None:
        out["rescored_fraction"] = stats_timed["rescored_particles"] / float(B)
        # exact branch and bound of the row transforms: rows whose coefficient moduli cannot reach the
        # particle's best value are not transformed; the fraction depends on the data
        out["s3_rows_pruned_fraction"] = rows_seen[1] / float(max(1, rows_seen[0]))
        out["s2_two_level_cut"] = dict(zip(("K0", "nk"), pm.two_level_cut()))
        # S6 runs in fp32 first; particles whose arg-max / window decision is within 2e-5 |max| of flipping are repeated in fp64
        out["s6_repeated_fraction"] = s6_rep[0] / float(max(1, args.steps * B))

    # ---- CPU baseline: the oracle on a bounded sample of the same workload, host cores of rank 0
    if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N=1 only
        from oracle import pyoracle as o
        ns = min(B, args.cpu_sample or (256 if D >= 256 else 512))
        ncores = o.lib().xo_num_threads()
        h_refs = refs.cpu().numpy()
        h_parts = particles[:ns].cpu().numpy()
        tb0 = time.perf_counter()
        opm = o.PM(h_refs) if args.mode != "grid" else None
        tb_setup = time.perf_counter() - tb0
        # gridding: one private volume per thread, like RF's worker threads (1.6 GB each at 256 px): at most 8
        nthr = max(1, min(8, ncores, ns)) if args.mode != "match" else 0
        orfs = [o.RF(D, use_ctf=True, min_ctf=0.01) for _ in range(nthr)]
        tb1 = time.perf_counter()
        if opm is not None:
            er, ep, ef, _ = opm.match(h_parts, nbr[0][:ns + 1], nbr[1][:nbr[0][ns]]) if nbr is not None else opm.match(h_parts)
            ex, ey, ec = opm.translate(h_parts, er[:, 0], ep[:, 0], ef[:, 0])
            oang = np.stack([dirs[er[:, 0], 0], dirs[er[:, 0], 1], ep[:, 0] * (360.0 / opm.N)], 1)
        else:
            oang = synth.random_angles(ns, rng)
            ex = ey = np.zeros(ns)
        tb_match = time.perf_counter()

        def grid_worker(t):
            orf = orfs[t]
            for i in range(t, ns, nthr):            # the ctypes calls release the GIL
                img = o.translate2d(h_parts[i], ex[i], ey[i], degree=3, wrap=True)
                cpar = o.ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=ctfs[i].DeltafU, DeltafV=ctfs[i].DeltafV)
                c, m = orf.ctf_arrays(cpar)
                orf.insert(orf.prepare_image(img), synth.euler_matrix(*oang[i]).T, ctf=c, modulator=m)
        thr = [threading.Thread(target=grid_worker, args=(t,)) for t in range(nthr)]
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        tb2 = time.perf_counter()
        out["cpu_baseline"] = {
            "value": ns / (tb2 - tb1), "unit": "particles/s", "cores": max(min(ncores, ns), nthr), "kind": "port",
            "cpu": cpu_model(), "match_s": tb_match - tb1, "gridding_s": tb2 - tb_match, "gridding_threads": nthr,
            "sample": f"{ns} of the same {D}x{D} particles vs the same {nrefs} references: oracle match+translate "
                      f"(OpenMP over particles, {ncores} threads) then shift+CTF+FFT+gridding on {nthr} threads with a private "
                      f"volume each; library setup {tb_setup:.1f}s excluded on both sides; finaliser excluded"}
        if opm is not None and pm is not None:
            # the sample doubles as a parity spot check at full size: orientations ...
            g_ref, g_psi, g_flip = (pm.match(particles[:ns].contiguous(), nbr[0][:ns + 1], nbr[1][:nbr[0][ns]]) if nbr is not None
                                    else pm.match(particles[:ns].contiguous()))
            out["parity_sample_identical"] = bool(np.array_equal(g_ref.cpu().numpy(), er[:, 0]) and
                                                  np.array_equal(g_psi.cpu().numpy(), ep[:, 0]) and
                                                  np.array_equal(g_flip.cpu().numpy(), ef[:, 0]))
            out["parity_sample_particles"] = ns
        if rf is not None and nthr > 0:
            # ... and the gridded Fourier volume: the device grids the first particles of the sample with the oracle's
            # orientations and shifts (shift, CTF planes, FFT, insertion) and the temp spaces are compared
            nv = min(ns, 32)
            chk = o.RF(D, use_ctf=True, min_ctf=0.01)
            for i in range(nv):
                img = o.translate2d(h_parts[i], ex[i], ey[i], degree=3, wrap=True)
                cpar = o.ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=ctfs[i].DeltafU, DeltafV=ctfs[i].DeltafV)
                c, m = chk.ctf_arrays(cpar)
                chk.insert(chk.prepare_image(img), synth.euler_matrix(*oang[i]).T, ctf=c, modulator=m)
            ev, ew = chk.temp()
            torch.cuda.synchronize()
            with on_rf_stream():
                rf.reset()
                # through the calls the timed region makes: shifts and orientations as device arrays, CTF + FFT + records + gridding in one call
                sx_t = torch.from_numpy(np.ascontiguousarray(ex[:nv], np.float64)).to(dev)
                sy_t = torch.from_numpy(np.ascontiguousarray(ey[:nv], np.float64)).to(dev)
                simg = rf.shift_images(particles[:nv].contiguous(), (sx_t, sy_t))
                ang_t = torch.from_numpy(np.ascontiguousarray(oang[:nv], np.float64)).to(dev)
                rf.insert_images(simg, ang_t, ctf_array=xa.RecFourier.ctf_param_array(ctfs[:nv]))
                gv, gw = rf.temp_spaces()
                gv, gw = gv.cpu().numpy(), gw.cpu().numpy()
            out["parity_volume_rel_err"] = float(np.abs(gv - ev).max() / np.abs(ev).max())
            out["parity_weights_rel_err"] = float(np.abs(gw - ew).max() / np.abs(ew).max())
            out["parity_volume_voxel_sets_equal"] = bool(((gw != 0) == (ew != 0)).all())
            out["parity_volume_particles"] = nv
            with on_rf_stream():
                rf.reset()
    if args.mode == "full" and world == 1 and not args.no_extra_legs and not args.no_flexalign:
        out["flexalign"] = flexalign_leg()
    if args.mode == "full" and world == 1 and not args.no_extra_legs and not args.no_cli:
        out["cli"] = cli_leg(D, nrefs, B)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cli_leg(D, nrefs, B):
    """The deliverable north_star names -- the two drop-in programs -- end to end in the default line: tools/bench_cli.py as a CHILD
    process (files under /dev/shm, xmipp_angular_projection_matching then xmipp_reconstruct_fourier_accel --useCTF, the library on the
    same data beside them), reduced to the rates and the host-side split a reader needs."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_cli.py"), "--box", str(D), "--nrefs", str(nrefs), "--batch", str(B),
           "--particles", str(16 * B), "--unique", str(4 * B)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"what"')]
        if r.returncode != 0 or not line:
            return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-400:]}
        d = json.loads(line[-1])
        out = {"what": d["what"], "config": d["config"], "outputs_vs_library": d.get("outputs_vs_library")}
        for p in ("xmipp_angular_projection_matching", "xmipp_reconstruct_fourier_accel"):
            x = d[p]
            out[p] = {k: x.get(k) for k in ("particles_per_s", "particles_per_s_image_loop", "library_particles_per_s", "vs_library_image_loop", "wall_s")}
            out[p]["timing_s"] = {k: v for k, v in x["timing_s"].items() if k in ("total", "setup", "parse", "bank", "loop", "stall", "device", "load", "finish", "write")}
        out["note"] = ("particles_per_s: rows / wall clock of the whole process (start-up, side info, image loop, output); particles_per_s_image_loop: "
                       "the loop alone; library_particles_per_s: the same library calls from Python on batches resident in HBM. At 256 px the matcher's "
                       "loop is bound by the host link: 1.07 GB per 4096 particles at the 56 GB/s this link gives = 214 k particles/s")
        return out
    except Exception as e:      # the headline must not depend on this leg
        return {"error": repr(e)}


def flexalign_leg():
    """BASELINE config 5 in the default line: `bench.py --mode flexalign` (two K3 movies of int8 counts streamed from page-locked memory, four
    in flight) as a CHILD process -- its own runtime, its own contexts; nothing is exec'ed over this process's GPU state -- reduced to the
    figures a reader needs: movies/s and the three stages' times and roofline fractions."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "flexalign", "--steps", "12", "--warmup", "4", "--no-extra-legs"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
        if r.returncode != 0 or not line:
            return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-400:]}
        d = json.loads(line[-1])
        fr = {k_: v_["frac"] for k_, v_ in d.get("roofline_other_kernels", {}).items()}
        fr.update({k_: d["roofline"]["frac"] for k_ in d["stage_ms"] if k_ not in fr})      # the dominant stage is the one left over
        return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_movie": d["ms_per_step"],
                "frames": d["config"]["host_traffic"].split(" H2D")[0], "lanes_per_gpu": d["config"]["lanes_per_gpu"],
                "stage_ms_one_lane": d["stage_ms"],
                "stage_roofline_frac": fr,
                "global_shift_error_px": d.get("global_shift_error_px"), "global_shift_error_bound_px": d.get("global_shift_error_bound_px"),
                "parity_sample": d.get("parity_sample"), "cpu_baseline": d.get("cpu_baseline"),
                "what": "`python bench.py --mode flexalign --steps 12 --warmup 4` (four movies in flight) run as a child process after the timed region of the refine iteration"}
    except Exception as e:      # the headline must not depend on this leg
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
