#!/usr/bin/env python3
"""The drop-in binaries end to end: xmipp_angular_projection_matching, then
xmipp_reconstruct_fourier_accel --useCTF on its output, on files as a Scipion protocol would hand them over --
a Spider stack of particles, a gallery stack with its real `_sampling.xmd` (every particle's neighbour list = the
whole gallery: 5 KB of text per particle at 1000 references), `.xmd` tables -- all under /dev/shm, so that what is timed is the
programs' host side (read / parse / H2D / write) on top of the kernels and not a disk.

Per program: wall-clock particles/s of the whole process (runtime start-up, side info, image loop, output) and of the image loop
alone, next to XMIPP_HIP_TIMING's split, and next to the LIBRARY driven from Python on the same data (batches resident in HBM,
the same calls the programs make): `vs_library` = image-loop rate / library rate is what the host side costs.
Outputs are compared with the library path's: orientation indices identical, shifts to the 1e-6 the table prints, volume 1e-6.

    python tools/bench_cli.py                       # 256 px, 1000 references, 32768 rows over 16384 distinct particles
    python tools/bench_cli.py --box 128 --particles 65536
One JSON object on stdout (bench.py's default line embeds it as "cli")."""
import argparse
import json
import math
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "xmipp3_amd", "bin")


def spider_stack_memmap(path, n, D):
    """An empty Spider stack of n D x D float images as a float32 memmap [n, hw + D*D] (+ the file header), see tests/xmipp_io.py"""
    from tests import xmipp_io
    lenbyt = D * 4
    labrec = (1024 + lenbyt - 1) // lenbyt
    hw = labrec * lenbyt // 4
    per = hw + D * D
    a = np.memmap(path, np.float32, "w+", shape=(hw + n * per,))
    a[:hw] = xmipp_io._spider_header(D, D, 1, 1, 2, n, 0)
    body = a[hw:].reshape(n, per)
    h = xmipp_io._spider_header(D, D, 1, 1, 0, 0, 1)
    body[:, :hw] = h[None, :]
    body[:, 26] = np.arange(1, n + 1, dtype=np.float32)          # IMGNUM
    return a, body[:, hw:].reshape(n, D, D)


def parse_timing(stderr):
    m = [l for l in stderr.splitlines() if l.startswith("timing ")]
    if not m:
        return {}
    return {k: float(v) for k, v in re.findall(r"(\w+)=([-+0-9.eE]+)", m[-1])}


def run_program(cmd, env):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise SystemExit(f"{cmd[0]} exited with {r.returncode}:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    return dt, parse_timing(r.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--nrefs", type=int, default=1000)
    ap.add_argument("--particles", type=int, default=32768, help="rows of the input table")
    ap.add_argument("--unique", type=int, default=16384, help="distinct particle images in the stack (rows cycle through them)")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--dir", default="/dev/shm/xmipp3_amd_cli")
    ap.add_argument("--readers", type=int, default=0)
    ap.add_argument("--keep", action="store_true", help="leave the files in --dir")
    ap.add_argument("--no-library", action="store_true", help="skip the library legs (rates and output comparison)")
    ap.add_argument("--env", action="append", default=[], metavar="NAME=VALUE", help="environment of the two programs (A/B runs, e.g. HSA_ENABLE_SDMA=0)")
    ap.add_argument("--repeat", type=int, default=1, help="run the pair of programs this many times on the same files; the last run is reported")
    ap.add_argument("--neighbours", type=int, default=0, help="K > 0: every particle lists its K nearest references instead of the whole gallery")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    ge.build()
    import xmipp3_amd as xa
    from xmipp3_amd.api import ctf_params
    from tests import synth, xmipp_io
    import bench

    D, nrefs, n, nu, B = args.box, args.nrefs, args.particles, min(args.unique, args.particles), args.batch
    tmp = args.dir
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    free = shutil.disk_usage(tmp).free
    need = (nu + nrefs) * (D * D * 4 + 1024) + n * (nrefs * 5 + 400) + 3 * D ** 3 * 4
    if need * 1.2 > free:
        raise SystemExit(f"bench_cli: {tmp} has {free / 1e9:.1f} GB free, the files need {need / 1e9:.1f} GB")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = xa.Context(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    genr = torch.Generator(device=dev); genr.manual_seed(7)

    # ---- the files (not timed): gallery = central-slice projections of the bench phantom, particles as bench.py makes them
    t_gen0 = time.perf_counter()
    dirs = synth.fibonacci_directions(nrefs)
    fpj = xa.FourierProjector(ctx, bench.phantom_volume(torch, D, genr, dev), 2.0, 0.5, 3)
    refs = fpj.project(np.concatenate([dirs, np.zeros((nrefs, 1))], 1))
    fpj.close()
    refs = ((refs - refs.mean()) / refs.std()).contiguous()
    xmipp_io.write_stack(f"{tmp}/ref.stk", refs.cpu().numpy())
    mm, parts_view = spider_stack_memmap(f"{tmp}/parts.stk", nu, D)
    for b0 in range(0, nu, 512):
        m = min(512, nu - b0)
        idx = torch.randint(0, nrefs, (m,), generator=gen, device=dev)
        th = torch.rand((m,), generator=gen, device=dev) * (2 * math.pi)
        mir = (torch.rand((m,), generator=gen, device=dev) < 0.5).float() * 2 - 1
        shf = torch.randint(-3, 4, (m, 2), generator=gen, device=dev).float()
        rot = torch.zeros((m, 2, 3), device=dev)
        rot[:, 0, 0] = torch.cos(th) * mir; rot[:, 0, 1] = -torch.sin(th); rot[:, 1, 0] = torch.sin(th) * mir; rot[:, 1, 1] = torch.cos(th)
        rot[:, :, 2] = shf * (2.0 / D)
        grid = torch.nn.functional.affine_grid(rot, (m, 1, D, D), align_corners=False)
        p = torch.nn.functional.grid_sample(refs[idx][:, None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[:, 0]
        p = p + math.sqrt(10.0) * torch.randn((m, D, D), generator=gen, device=dev)
        parts_view[b0:b0 + m] = p.cpu().numpy()
    mm.flush()
    del mm, parts_view
    ids = list(range(nrefs))
    if args.neighbours > 0:
        K = min(args.neighbours, nrefs)
        rt = np.radians(dirs[:, :2])
        v = np.stack([np.sin(rt[:, 1]) * np.cos(rt[:, 0]), np.sin(rt[:, 1]) * np.sin(rt[:, 0]), np.cos(rt[:, 1])], 1)
        table = np.sort(np.argsort(-(v @ v.T), axis=1, kind="stable")[:, :K], axis=1)
        rngl = np.random.default_rng(5)
        centre = rngl.integers(0, nrefs, n)
        lists = [" " + " ".join(map(str, table[c])) + " " for c in centre]
    else:
        whole = " " + " ".join(str(i) for i in ids) + " "
        lists = None
    with open(f"{tmp}/ref_sampling.xmd", "w") as f:
        f.write("# XMIPP_STAR_1 * \n# \ndata_extra\n _sampling_rate 0.05\n _neighborhoodRadius -1.01\n _pointsAsymmetricUnit %d\n" % nrefs)
        f.write("data_neighbors\nloop_\n _neighbor\n _neighbors\n")
        for i in range(n):
            f.write(f"{i + 1:10d} '{whole if lists is None else lists[i]}' \n")
        f.write("data_projectionDirections\nloop_\n _neighbor\n _angleRot\n _angleTilt\n _anglePsi\n _X\n _Y\n _Z\n")
        for i in range(nrefs):
            rt_, tt_ = math.radians(dirs[i][0]), math.radians(dirs[i][1])
            f.write(f"{ids[i]:10d} {dirs[i][0]:12.6f} {dirs[i][1]:12.6f} {0.0:12.6f} {math.sin(tt_) * math.cos(rt_):12.6f} {math.sin(tt_) * math.sin(rt_):12.6f} {math.cos(tt_):12.6f} \n")
    rng = np.random.default_rng(100)
    defocus = rng.uniform(10000.0, 30000.0, n)
    with open(f"{tmp}/exp.xmd", "w") as f:
        f.write("# XMIPP_STAR_1 * \n# \ndata_noname\nloop_\n _itemId\n _image\n")
        for i in range(n):
            f.write(f"{i + 1:10d} {(i % nu) + 1:06d}@{tmp}/parts.stk \n")
    sizes = {k: os.path.getsize(f"{tmp}/{k}") for k in ("parts.stk", "ref.stk", "ref_sampling.xmd", "exp.xmd")}
    t_gen = time.perf_counter() - t_gen0

    # ---- the two programs, as a pipeline would start them
    env = dict(os.environ, XMIPP_HIP_TIMING="1")
    for kv in args.env:
        k_, v_ = kv.split("=", 1)
        env[k_] = v_
    rd = ["--readers", str(args.readers)] if args.readers else []
    t_apm, tm_apm = run_program([f"{BIN}/xmipp_angular_projection_matching", "-i", f"{tmp}/exp.xmd", "-o", f"{tmp}/out.xmd", "--ref", f"{tmp}/ref.stk",
                                 "--batch", str(B)] + rd, env)
    # the join a protocol does between the two steps (not timed): the CTF columns of the micrograph next to the assigned angles
    labels, rows = xmipp_io.read_xmd(f"{tmp}/out.xmd")
    assert len(rows) == n, (len(rows), n)
    with open(f"{tmp}/rec_in.xmd", "w") as f:
        f.write("# XMIPP_STAR_1 * \n# \ndata_noname\nloop_\n" + "".join(f" _{l}\n" for l in labels) +
                " _ctfVoltage\n _ctfSphericalAberration\n _ctfQ0\n _ctfK\n _ctfDefocusU\n _ctfDefocusV\n _ctfSamplingRate\n")
        for i, r in enumerate(rows):
            f.write(" " + " ".join(r) + f" 300.000000 2.700000 0.070000 1.000000 {defocus[i]:.6f} {defocus[i]:.6f} 1.000000 \n")
    t_rfa, tm_rfa = run_program([f"{BIN}/xmipp_reconstruct_fourier_accel", "-i", f"{tmp}/rec_in.xmd", "-o", f"{tmp}/rec.vol", "--useCTF", "--sampling", "1",
                                 "--batch", str(B)] + rd, env)
    out = {
        "what": "the two drop-in programs end to end on files under /dev/shm (tools/bench_cli.py): whole-process wall clock and the image loop alone",
        "config": {"box": D, "nrefs": nrefs, "rows": n, "distinct_particles": nu, "batch": B,
                   "neighbour_lists": "whole gallery per particle" if lists is None else f"{args.neighbours} nearest references per particle",
                   "files_GB": {k: round(v / 1e9, 3) for k, v in sizes.items()}, "files_written_in_s": round(t_gen, 1)},
        "xmipp_angular_projection_matching": {"wall_s": t_apm, "particles_per_s": n / t_apm, "particles_per_s_image_loop": tm_apm.get("images_per_s_loop"),
                                              "timing_s": tm_apm},
        "xmipp_reconstruct_fourier_accel": {"wall_s": t_rfa, "particles_per_s": n / t_rfa, "particles_per_s_image_loop": tm_rfa.get("images_per_s_loop"),
                                            "timing_s": tm_rfa},
        "timing_note": "timing_s = XMIPP_HIP_TIMING: setup (contexts, page-locked buffers), parse (tables, neighbour lists), bank (gallery read + xh_pm_create), "
                       "loop (image loop wall clock) = stall (main thread waiting for the loader) + device (main thread inside library calls); load / h2d_wait / "
                       "format run on worker threads UNDER the loop; finish (mirror/crop + finaliser + volume D2H); write (output files)",
    }

    if not args.no_library:
        # ---- the library on the same data, resident in HBM: its rate for each half, and the outputs the programs must reproduce
        stack = xmipp_io.read_stack(f"{tmp}/parts.stk") if nu * D * D * 4 < 24e9 else None
        col = {l: i for i, l in enumerate(labels)}
        c_ref = np.array([int(r[col["ref"]]) for r in rows]); c_flip = np.array([int(r[col["flip"]]) for r in rows])
        c_psi = np.array([float(r[col["anglePsi"]]) for r in rows])
        c_sx = np.array([float(r[col["shiftX"]]) for r in rows]); c_sy = np.array([float(r[col["shiftY"]]) for r in rows])
        c_rot = np.array([float(r[col["angleRot"]]) for r in rows]); c_tilt = np.array([float(r[col["angleTilt"]]) for r in rows])
        pm = xa.ProjectionMatcher(ctx, refs)
        rf = xa.RecFourier(ctx, D, min_ctf=0.01, sampling=1.0)
        d_batches = [torch.from_numpy(stack[b0:b0 + B]).to(dev) for b0 in range(0, nu, B)]
        nbt = len(d_batches)
        per_batch = [d.shape[0] for d in d_batches]
        l_ref, l_psi, l_flip, l_sx, l_sy = [], [], [], [], []

        def rows_of(k):           # table rows of device batch k of the program (rows b0 .. b0 + B - 1 cycle through the stack)
            b0 = k * B
            return np.arange(b0, min(n, b0 + B))
        nbatch = (n + B - 1) // B
        # warm-up + results: the program's batches are consecutive table rows; with nu a multiple of B they are whole stack batches
        aligned = nu % B == 0
        torch.cuda.synchronize()
        parity = 0
        tm0 = None
        for rep in range(2):
            parity = 0
            if rep == 1:
                torch.cuda.synchronize(); tm0 = time.perf_counter()
            for k in range(nbatch):
                r_ = rows_of(k)
                if aligned:
                    d = d_batches[(k * B % nu) // B][:len(r_)]
                else:
                    d = torch.from_numpy(stack[r_ % nu]).to(dev)
                nb_ = None
                if lists is not None:
                    off = (np.arange(len(r_) + 1) * K).astype(np.int32)
                    nb_ = (off, np.ascontiguousarray(table[centre[r_]].ravel().astype(np.int32)))
                refno, psi, flip = pm.match(d, *nb_, parity=parity) if nb_ is not None else pm.match(d, parity=parity)
                sx, sy, cc = pm.translate(d, refno, psi, flip)
                parity ^= len(r_) & 1
                if rep == 0:
                    l_ref.append(refno.cpu().numpy()); l_psi.append(psi.cpu().numpy()); l_flip.append(flip.cpu().numpy())
                    l_sx.append(sx.cpu().numpy()); l_sy.append(sy.cpu().numpy())
        torch.cuda.synchronize()
        lib_match = n / (time.perf_counter() - tm0)
        l_ref, l_psi, l_flip, l_sx, l_sy = (np.concatenate(x) for x in (l_ref, l_psi, l_flip, l_sx, l_sy))
        same_ref = bool(np.array_equal(l_ref, c_ref))          # (gallery numbers == stack positions in these files)
        same_psi = bool(np.abs(l_psi * (360.0 / pm.N) - c_psi).max() < 1e-6)
        same_flip = bool(np.array_equal(l_flip, c_flip))
        d_shift = float(max(np.abs(l_sx - c_sx).max(), np.abs(l_sy - c_sy).max()))
        # reconstruction from the table the program read (its printed values), twice: first for the volume, then timed
        ang = np.stack([c_rot, c_tilt, c_psi], 1)
        ctfs_all = [ctf_params(kV=300.0, Cs=2.7, Q0=0.07, K=1.0, DeltafU=float(f"{d_:.6f}"), DeltafV=float(f"{d_:.6f}"), Tm=1.0) for d_ in defocus]
        vol = None
        for rep in range(2):
            rf.reset()
            torch.cuda.synchronize(); tr0 = time.perf_counter()
            for k in range(nbatch):
                r_ = rows_of(k)
                d = d_batches[(k * B % nu) // B][:len(r_)] if aligned else torch.from_numpy(stack[r_ % nu]).to(dev)
                sh = np.stack([c_sx[r_], c_sy[r_]], 1)
                imgs = rf.shift_images(d, sh, flips=c_flip[r_].astype(np.uint8))
                rf.insert_images(imgs, ang[r_], ctf_array=xa.RecFourier.ctf_param_array([ctfs_all[i] for i in r_]))
            torch.cuda.synchronize()
            lib_rec = n / (time.perf_counter() - tr0)
            if rep == 0:
                rf.mirror_and_crop()
                vol = rf.finish().copy()
        cvol = xmipp_io.read_volume(f"{tmp}/rec.vol")
        vol_err = float(np.abs(cvol - vol.astype(np.float32)).max() / np.abs(vol).max())
        a, r2 = out["xmipp_angular_projection_matching"], out["xmipp_reconstruct_fourier_accel"]
        a["library_particles_per_s"] = lib_match
        a["vs_library_image_loop"] = (a["particles_per_s_image_loop"] or 0) / lib_match
        a["vs_library_whole_process"] = a["particles_per_s"] / lib_match
        r2["library_particles_per_s"] = lib_rec
        r2["vs_library_image_loop"] = (r2["particles_per_s_image_loop"] or 0) / lib_rec
        r2["vs_library_whole_process"] = r2["particles_per_s"] / lib_rec
        out["library_note"] = ("library_particles_per_s: the same calls from Python on batches resident in HBM (match + translate with result copies; "
                               "shift + CTF + FFT + gridding), host arrays prepared inside the clock as the programs do")
        out["outputs_vs_library"] = {"ref_identical": same_ref, "psi_identical": same_psi, "flip_identical": same_flip, "max_shift_difference_px": d_shift,
                                     "volume_rel_err": vol_err,
                                     "note": "the table prints shifts and angles with six decimals; the library reconstruction is fed the printed values"}
        pm.close()
    if not args.keep:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
