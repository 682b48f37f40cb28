"""The drop-in process-level contract: same binary names, flags, error behaviour and output
files as xmipp_angular_projection_matching / xmipp_reconstruct_fourier_accel."""
import os
import subprocess

import numpy as np
import pytest

from tests import synth, xmipp_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "xmipp3_amd", "bin")


@pytest.fixture(scope="module")
def bins():
    import __graft_entry__ as g
    g.build()
    return BIN


def _run(args, timeout=600, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, **kw)


def test_help_and_argument_errors(bins):
    for prog in ("xmipp_angular_projection_matching", "xmipp_reconstruct_fourier_accel", "xmipp_reconstruct_fourier",
                 "xmipp_angular_project_library", "xmipp_resolution_fsc", "xmipp_ctf_phase_flip", "xmipp_ctf_correct_wiener2d", "xmipp_movie_alignment_correlation", "xmipp_movie_filter_dose"):
        r = _run([os.path.join(bins, prog), "--help"])
        assert r.returncode == 0 and "USAGE" in r.stderr
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-o", "x.xmd"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "-i is mandatory" in r.stderr
    r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", "nonexistent.xmd", "--bogus"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr


def test_reference_flag_spellings_are_accepted(bins, tmp_path):
    # the argument strings of test_programs.py:150-159,881-889 parse; failure must come from the
    # missing input file (or, without a GPU, from the device), never from the parser
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", "input/aFewProjections.sel", "-o", str(tmp_path / "o.xmd"),
              "--ref", "ref.stk", "--thr", "3", "--search5d_shift", "0", "--search5d_step", "2", "--Ri", "1", "--Ro", "20",
              "--max_shift", "10", "--mem", "1", "--append"])
    assert r.returncode != 0 and "cannot open" in r.stderr
    r = _run([os.path.join(bins, "xmipp_reconstruct_fourier"), "-i", "input/aFewProjections.sel", "-o", str(tmp_path / "rec.vol"),
              "--sym", "c1", "--padding", "2", "2", "--max_resolution", "0.5", "--blob", "1.9", "0", "15", "--thr", "1", "--weight"])
    assert r.returncode != 0 and "cannot open" in r.stderr


def test_resolution_fsc_argument_rules(bins):
    """resolution_fsc.cpp:34-118: -i needs --ref, --set_of_images excludes both ("or" alternative of the parameter DSL)."""
    fsc = os.path.join(bins, "xmipp_resolution_fsc")
    r = _run([fsc, "-s", "2"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "-i is mandatory" in r.stderr
    r = _run([fsc, "-i", "a.vol"])
    assert r.returncode != 0 and "requires --ref" in r.stderr
    r = _run([fsc, "--set_of_images", "s.sel", "-i", "x", "--ref", "y"])
    assert r.returncode != 0 and "should not be provided with -i or --ref" in r.stderr
    r = _run([fsc, "--set_of_images", "s.sel"])
    assert r.returncode != 0 and "not available" in r.stderr
    r = _run([fsc, "--ref", "missing_a.vol", "-i", "missing_b.vol", "-s", "5.6", "--do_dpr", "--oroot", "x"])
    assert r.returncode != 0 and "cannot open" in r.stderr          # the flags of test_programs.py:898-900 parse
    r = _run([fsc, "--help"])
    assert r.returncode == 0 and "or --set_of_images" in r.stderr


def _write_dataset(tmp, D=32, nrefs=12, n=9, seed=4):
    vol = synth.phantom(D, seed=seed, nblobs=10)
    refs, dirs = synth.make_refs(vol, nrefs)
    rng = np.random.default_rng(seed)
    parts, _ = synth.make_particles(refs, n, rng, snr=0.5, max_shift=1)
    xmipp_io.write_stack(str(tmp / "ref.stk"), refs)
    xmipp_io.write_stack(str(tmp / "parts.stk"), parts)
    # reference ids are 1-based and deliberately not in stack order
    ids = [3 * i + 1 for i in range(nrefs)]
    nbrs = [" ".join(str(ids[j]) for j in rng.permutation(nrefs)[:rng.integers(4, nrefs + 1)]) for _ in range(n)]
    xmipp_io.write_xmd(str(tmp / "ref_sampling.xmd"), [
        ("extra", ["sampling_rate", "neighborhoodRadius", "pointsAsymmetricUnit"], [[0.05, -1.01, max(ids) + 1]]),
        ("neighbors", ["neighbor", "neighbors"], [[i + 1, " " + nbrs[i] + " "] for i in range(n)]),
        ("projectionDirections", ["neighbor", "angleRot", "angleTilt", "anglePsi", "X", "Y", "Z"],
         [[ids[i], f"{dirs[i][0]:.6f}", f"{dirs[i][1]:.6f}", "0.000000", 0, 0, 1] for i in range(nrefs)])])
    xmipp_io.write_xmd(str(tmp / "exp.xmd"), [("noname", ["itemId", "image"], [[100 + i, f"{i + 1}@{tmp}/parts.stk"] for i in range(n)])])
    return refs, dirs, parts, ids, nbrs


def test_no_gpu_means_xmipp_error_not_fallback(bins, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _write_dataset(tmp_path)
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp.xmd"), "-o", str(tmp_path / "out.xmd"),
              "--ref", str(tmp_path / "ref.stk")])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "no CPU fallback" in r.stderr
    assert not (tmp_path / "out.xmd").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("box", [32, 40])
def test_cli_pipeline_matches_oracle(bins, tmp_path, oracle, box):
    """project-match with the CLI, reconstruct from its output with the CLI; compare both with
    the oracle run on the same files' contents. box=40: no power of two anywhere on the path."""
    refs, dirs, parts, ids, nbrs = _write_dataset(tmp_path, D=box)
    n, nrefs, D = len(parts), len(refs), refs.shape[1]
    # (box 40 with --thr 3, as the reference's own test of the program runs it, test_programs.py:150-159)
    thr = 3 if box == 40 else 1
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp.xmd"), "-o", str(tmp_path / "out.xmd"),
              "--ref", str(tmp_path / "ref.stk"), "--max_shift", "6", "--batch", "4", "--thr", str(thr)])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"))
    assert labels == ["itemId", "image", "angleRot", "angleTilt", "anglePsi", "shiftX", "shiftY", "ref", "flip", "scale", "maxCC"]
    assert len(rows) == n
    # oracle on the same neighbour lists (stack positions), same visiting-order parity
    pos = {ids[i]: i for i in range(nrefs)}
    lists = [[pos[int(v)] for v in s.split()] for s in nbrs]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    pm = oracle.PM(refs)
    er, ep, ef, _ = pm.match(parts, off, np.concatenate(lists).astype(np.int32), ref_threads=thr)
    ex, ey, ec = pm.translate(parts, er[:, 0], ep[:, 0], ef[:, 0], 6.0)
    c = {l: i for i, l in enumerate(labels)}
    for i, row in enumerate(rows):
        assert int(row[c["itemId"]]) == 100 + i
        assert int(row[c["ref"]]) == ids[er[i, 0]]
        assert int(row[c["flip"]]) == ef[i, 0]
        assert abs(float(row[c["anglePsi"]]) - ep[i, 0] * 360.0 / pm.N) < 1e-5
        assert abs(float(row[c["angleRot"]]) - dirs[er[i, 0]][0]) < 1e-5 and abs(float(row[c["angleTilt"]]) - dirs[er[i, 0]][1]) < 1e-5
        assert abs(float(row[c["shiftX"]]) - ex[i]) < 1e-3 and abs(float(row[c["shiftY"]]) - ey[i]) < 1e-3
        assert abs(float(row[c["maxCC"]]) - ec[i]) < 1e-5
    # reconstruction from the assigned angles (shifts and flips applied by the program)
    r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "out.xmd"), "-o", str(tmp_path / "rec.vol"),
              "--batch", "5", "--sym", "c2"])
    assert r.returncode == 0, r.stderr
    got = xmipp_io.read_volume(str(tmp_path / "rec.vol"))
    assert got.shape == (D, D, D)
    rf = oracle.RF(D)
    c2 = np.diag([-1.0, -1.0, 1.0])
    for i, row in enumerate(rows):
        sx, sy, fl = float(row[c["shiftX"]]), float(row[c["shiftY"]]), int(row[c["flip"]])
        A = np.array([[-1.0 if fl else 1.0, 0, sx], [0, 1, sy], [0, 0, 1]])
        img = oracle.apply_geometry2d(parts[i], A, 3, False, True) if (fl or sx or sy) else parts[i]
        ang = [float(row[c["angleRot"]]), float(row[c["angleTilt"]]), float(row[c["anglePsi"]])]
        f = rf.prepare_image(img)
        for R in (np.eye(3), c2):
            rf.insert(f, synth.euler_matrix(*ang).T, R=R)
    rf.mirror_and_crop()
    exp = rf.finish()
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()


@pytest.mark.gpu
def test_cli_loader_failures_end_the_programs_cleanly(bins, tmp_path):
    """The image loops are pipelines (reader threads, copier threads, workers that assemble and format batches: host/fastio.h): an
    image that cannot be read -- a missing stack, a stack that ends early, an image of another size -- somewhere in the MIDDLE of a run
    has to surface as the reference's XMIPP_ERROR exit, promptly, with no output file and no thread left waiting."""
    refs, dirs, parts, ids, nbrs = _write_dataset(tmp_path, D=32, n=9)
    n = len(parts)
    apm = os.path.join(bins, "xmipp_angular_projection_matching")
    rfa = os.path.join(bins, "xmipp_reconstruct_fourier_accel")
    ok = _run([apm, "-i", str(tmp_path / "exp.xmd"), "-o", str(tmp_path / "good.xmd"), "--ref", str(tmp_path / "ref.stk"), "--batch", "2"], timeout=120)
    assert ok.returncode == 0, ok.stderr
    # a table whose seventh row names a stack that does not exist, whose eighth an image beyond the end, whose fifth an image of another size
    xmipp_io.write_stack(str(tmp_path / "other.stk"), np.zeros((2, 16, 16), np.float32))
    cases = {"missing": (6, f"1@{tmp_path}/nowhere.stk", "cannot open"), "beyond": (7, f"{n + 5}@{tmp_path}/parts.stk", "beyond the end"),
             "size": (4, f"1@{tmp_path}/other.stk", "different size")}
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "good.xmd"))
    ci = labels.index("image")
    for name, (row, image, msg) in cases.items():
        xmipp_io.write_xmd(str(tmp_path / f"exp_{name}.xmd"), [("noname", ["itemId", "image"],
                           [[100 + i, image if i == row else f"{i + 1}@{tmp_path}/parts.stk"] for i in range(n)])])
        r = _run([apm, "-i", str(tmp_path / f"exp_{name}.xmd"), "-o", str(tmp_path / f"out_{name}.xmd"), "--ref", str(tmp_path / "ref.stk"), "--batch", "2"], timeout=120)
        assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and msg in r.stderr, (name, r.stderr[-400:])
        assert not (tmp_path / f"out_{name}.xmd").exists()
        bad = [list(x) for x in rows]
        bad[row][ci] = image
        xmipp_io.write_xmd(str(tmp_path / f"rec_{name}.xmd"), [("noname", labels, bad)])
        r = _run([rfa, "-i", str(tmp_path / f"rec_{name}.xmd"), "-o", str(tmp_path / f"rec_{name}.vol"), "--batch", "2"], timeout=120)
        assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and msg in r.stderr, (name, r.stderr[-400:])
        assert not (tmp_path / f"rec_{name}.vol").exists()


@pytest.mark.gpu
def test_cli_second_iteration_applies_the_previous_shifts(bins, tmp_path, oracle):
    """A second refinement iteration: the input metadata carries shiftX / shiftY from the first. getCurrentImage
    (APM:1194-1236) builds the transformation from them, inverts it and applies it with BSPLINE3 + WRAP and IS_INV before
    anything else; the shifts written out are the newly found ones plus the previous ones (APM:1142-1144). Oracle: the same
    geometry applied with the oracle's applyGeometry, then its match and translate on the moved images."""
    refs, dirs, parts, ids, nbrs = _write_dataset(tmp_path, D=32, n=9, seed=6)
    n, nrefs = len(parts), len(refs)
    rng = np.random.default_rng(12)
    prev = np.round(rng.uniform(-2.5, 2.5, (n, 2)), 3)
    prev[0] = (0.0, 0.0)                       # an identity transformation is skipped (APM:1231)
    prev[1] = (1.0, -2.0)                      # whole pixels
    xmipp_io.write_xmd(str(tmp_path / "exp2.xmd"), [("noname", ["itemId", "image", "shiftX", "shiftY", "scale"],
                       [[100 + i, f"{i + 1}@{tmp_path}/parts.stk", f"{prev[i, 0]:.3f}", f"{prev[i, 1]:.3f}", "1.000000"] for i in range(n)])])
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp2.xmd"), "-o", str(tmp_path / "out2.xmd"),
              "--ref", str(tmp_path / "ref.stk"), "--max_shift", "6", "--batch", "4"])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out2.xmd"))
    c = {l: i for i, l in enumerate(labels)}
    moved = np.empty_like(parts)
    for i in range(n):
        T = np.array([[1.0, 0, prev[i, 0]], [0, 1, prev[i, 1]], [0, 0, 1]])
        moved[i] = parts[i] if not prev[i].any() else oracle.apply_geometry2d(parts[i], np.linalg.inv(T), 3, True, True)
    assert np.abs(moved[1] - np.roll(parts[1], (-2, 1), (0, 1))).max() < 1e-9 * np.abs(parts[1]).max() + 1e-6     # content moves by +shift
    pos = {ids[i]: i for i in range(nrefs)}
    lists = [[pos[int(v)] for v in s.split()] for s in nbrs]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    pm = oracle.PM(refs)
    er, ep, ef, _ = pm.match(moved, off, np.concatenate(lists).astype(np.int32))
    ex, ey, ec = pm.translate(moved, er[:, 0], ep[:, 0], ef[:, 0], 6.0)
    assert len(rows) == n
    for i, row in enumerate(rows):
        assert int(row[c["ref"]]) == ids[er[i, 0]] and int(row[c["flip"]]) == ef[i, 0]
        assert abs(float(row[c["anglePsi"]]) - ep[i, 0] * 360.0 / pm.N) < 1e-5
        assert abs(float(row[c["shiftX"]]) - (ex[i] + prev[i, 0])) < 1e-3 and abs(float(row[c["shiftY"]]) - (ey[i] + prev[i, 1])) < 1e-3
        assert abs(float(row[c["maxCC"]]) - ec[i]) < 1e-5
    # a scaled input row is refused loudly (the reference would resample it, APM:1222-1233)
    xmipp_io.write_xmd(str(tmp_path / "exp3.xmd"), [("noname", ["itemId", "image", "scale"], [[1, f"1@{tmp_path}/parts.stk", "1.050000"]])])
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp3.xmd"), "-o", str(tmp_path / "out3.xmd"),
              "--ref", str(tmp_path / "ref.stk")])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "scale" in r.stderr


@pytest.mark.gpu
def test_cli_several_devices_give_the_single_device_answer(bins, tmp_path):
    """--gpus / --devices: one host thread per device slot, contiguous particle ranges, the visiting-order
    parity carried across the ranges, one tree reduction of the volumes. The box has one GPU, so the slots
    all sit on device 0 (--devices 0,0,0): same threads, contexts, streams and reduction as on a node."""
    _write_dataset(tmp_path, D=32, n=11)
    apm = os.path.join(bins, "xmipp_angular_projection_matching")
    base = [apm, "-i", str(tmp_path / "exp.xmd"), "--ref", str(tmp_path / "ref.stk"), "--max_shift", "6", "--batch", "2"]
    r1 = _run(base + ["-o", str(tmp_path / "one.xmd")])
    r3 = _run(base + ["-o", str(tmp_path / "three.xmd"), "--devices", "0,0,0"])
    assert r1.returncode == 0 and r3.returncode == 0, r1.stderr + r3.stderr
    assert "Devices                 : 0,0,0" in r3.stdout
    assert open(tmp_path / "one.xmd").read() == open(tmp_path / "three.xmd").read()
    rfa = os.path.join(bins, "xmipp_reconstruct_fourier_accel")
    a = _run([rfa, "-i", str(tmp_path / "one.xmd"), "-o", str(tmp_path / "one.vol"), "--batch", "3"])
    b = _run([rfa, "-i", str(tmp_path / "one.xmd"), "-o", str(tmp_path / "three.vol"), "--batch", "3", "--devices", "0,0,0"])
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    v1, v3 = xmipp_io.read_volume(str(tmp_path / "one.vol")), xmipp_io.read_volume(str(tmp_path / "three.vol"))
    assert np.abs(v1 - v3).max() <= 1e-5 * np.abs(v1).max()
    # more device slots than images: the empty ranges contribute nothing
    c = _run([rfa, "-i", str(tmp_path / "one.xmd"), "-o", str(tmp_path / "many.vol"), "--devices", ",".join(["0"] * 13)])
    assert c.returncode == 0, c.stderr
    assert np.abs(xmipp_io.read_volume(str(tmp_path / "many.vol")) - v1).max() <= 1e-5 * np.abs(v1).max()
    # a device this node does not have is an argument error, not a silent clamp
    d = _run(base + ["-o", str(tmp_path / "x.xmd"), "--gpus", "64"])
    assert d.returncode != 0 and "XMIPP_ERROR" in d.stderr and "requested but this node has" in d.stderr
    e = _run(base + ["-o", str(tmp_path / "x.xmd"), "--devices", "0,a"])
    assert e.returncode != 0 and "XMIPP_ERROR" in e.stderr


@pytest.mark.gpu
def test_cli_prepare_fsc_writes_the_two_half_set_volumes(bins, tmp_path):
    """xmipp_reconstruct_fourier_accel --prepare_fsc <root> (the bookkeeping of RF:846,991-1045 on the accel arithmetic): images 0..(n-1)/2 -> <root>_1_recons.vol,
    the rest -> <root>_2_recons.vol, output volume = reconstruction from the summed halves."""
    D, n = 32, 9
    vol = synth.phantom(D, seed=8, nblobs=10)
    ang = synth.random_angles(n, np.random.default_rng(5))
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "p.stk"), imgs)

    def write(name, idx):
        xmipp_io.write_xmd(str(tmp_path / name), [("noname", ["image", "angleRot", "angleTilt", "anglePsi"],
                           [[f"{i + 1}@{tmp_path}/p.stk"] + [f"{v:.6f}" for v in ang[i]] for i in idx])])
    write("all.xmd", range(n))
    split = (n - 1) // 2          # FSCIndex, included in half 1
    write("h1.xmd", range(0, split + 1))
    write("h2.xmd", range(split + 1, n))
    rf = os.path.join(bins, "xmipp_reconstruct_fourier_accel")
    r = _run([rf, "-i", str(tmp_path / "all.xmd"), "-o", str(tmp_path / "all.vol"), "--prepare_fsc", str(tmp_path / "fsc"), "--devices", "0,0"])
    assert r.returncode == 0, r.stderr
    for name in ("plain", "h1", "h2"):
        q = _run([rf, "-i", str(tmp_path / ("all.xmd" if name == "plain" else name + ".xmd")), "-o", str(tmp_path / (name + ".vol"))])
        assert q.returncode == 0, q.stderr
    rd = lambda f: xmipp_io.read_volume(str(tmp_path / f))
    full = rd("plain.vol")
    tol = 1e-5 * np.abs(full).max()
    assert np.abs(rd("all.vol") - full).max() <= tol
    assert np.abs(rd("fsc_1_recons.vol") - rd("h1.vol")).max() <= tol
    assert np.abs(rd("fsc_2_recons.vol") - rd("h2.vol")).max() <= tol
    # the reference deletes its intermediate <root>_{1,2}_{Fourier,Weights}.vol; nothing of the kind is left here
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("fsc")) == ["fsc_1_recons.vol", "fsc_2_recons.vol"]


@pytest.mark.gpu
def test_cli_half_sets_then_resolution_fsc(bins, tmp_path, oracle):
    """reconstruct_fourier --prepare_fsc -> resolution_fsc on the two half maps (the use SURVEY.md 8f rank 2 names):
    the .frc file carries the oracle's curve for the same two volumes; flags of test_programs.py:898-900."""
    D, n = 32, 40
    vol = synth.phantom(D, seed=3, nblobs=10)
    ang = synth.random_angles(n, np.random.default_rng(9))
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "p.stk"), imgs)
    xmipp_io.write_xmd(str(tmp_path / "all.xmd"), [("noname", ["image", "angleRot", "angleTilt", "anglePsi"],
                       [[f"{i + 1}@{tmp_path}/p.stk"] + [f"{v:.6f}" for v in ang[i]] for i in range(n)])])
    r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "all.xmd"), "-o", str(tmp_path / "all.vol"),
              "--prepare_fsc", str(tmp_path / "fsc")])
    assert r.returncode == 0, r.stderr
    fsc = os.path.join(bins, "xmipp_resolution_fsc")
    r = _run([fsc, "--ref", str(tmp_path / "fsc_1_recons.vol"), "-i", str(tmp_path / "fsc_2_recons.vol"), "-s", "5.6", "--do_dpr",
              "--oroot", str(tmp_path / "halves")])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "halves.frc"))
    assert labels == ["resolutionFreqFourier", "resolutionFRC", "resolutionDPR", "resolutionErrorL2", "resolutionFRCRandomNoise",
                      "resolutionFreqReal"]
    assert len(rows) == D // 2                      # shell 0 is not written (resolution_fsc.cpp:136)
    got = np.array(rows, float)
    v1, v2 = xmipp_io.read_volume(str(tmp_path / "fsc_1_recons.vol")), xmipp_io.read_volume(str(tmp_path / "fsc_2_recons.vol"))
    exp = oracle.frc_dpr(v1, v2, 5.6, do_dpr=True, max_freq=0.5)
    for col, key, tol in ((0, "freq", 1e-6), (1, "frc", 2e-6), (2, "dpr", 1e-4), (3, "error_l2", 1e-6), (4, "frc_noise", 1e-6)):
        assert np.abs(got[:, col] - exp[key][1:]).max() <= tol, key      # the file holds 6 decimals
    assert np.allclose(got[:, 5], 1.0 / exp["freq"][1:], atol=1e-5)
    # two halves of one noiseless data set agree at low resolution
    assert got[0, 1] > 0.9 and got[:4, 1].min() > 0.5
    # R-factor block (row format) and the cut-offs of writeFiles
    _, rf_rows = xmipp_io.read_xmd(str(tmp_path / "halves.frc"), block="rfactor")
    assert open(tmp_path / "halves.frc").read().rstrip().endswith("_resolutionRfactor -1.000000")
    r = _run([fsc, "--ref", str(tmp_path / "fsc_1_recons.vol"), "-i", str(tmp_path / "fsc_2_recons.vol"), "-s", "5.6", "--do_rfactor",
              "--max_sam", "40", "--min_sam", "100", "-o", str(tmp_path / "cut.frc")])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "cut.frc"))
    assert "resolutionDPR" not in labels
    cut = np.array(rows, float)
    c = {l: i for i, l in enumerate(labels)}
    res = cut[:, c["resolutionFreqReal"]]
    assert np.all(cut[(res < 40) | (res > 100), c["resolutionFRC"]] == 0.0)
    keep = (res >= 40) & (res <= 100)
    assert keep.any() and np.abs(cut[keep, c["resolutionFRC"]] - got[keep, 1]).max() <= 1e-6
    exp_rf = oracle.frc_dpr(v1, v2, 5.6, do_rfactor=True, min_freq=5.6 / 100, max_freq=5.6 / 40)["rfactor"]
    tail = open(tmp_path / "cut.frc").read().split()
    assert tail[-2] == "_resolutionRfactor" and abs(float(tail[-1]) - exp_rf) <= 1e-6
    # different shapes are an error; --set_of_images fails loudly
    xmipp_io.write_volume(str(tmp_path / "small.vol"), np.zeros((16, 16, 16), np.float32))
    r = _run([fsc, "--ref", str(tmp_path / "small.vol"), "-i", str(tmp_path / "fsc_2_recons.vol")])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "different shapes" in r.stderr
    r = _run([fsc, "--set_of_images", str(tmp_path / "all.xmd")])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "not available" in r.stderr


@pytest.mark.gpu
def test_cli_5d_search_and_number_orientations(bins, tmp_path, oracle):
    """`--search5d_shift 2 --search5d_step 2 --number_orientations 3`: rows per image are the
    valid ranks of the reference's running top-N, each with its own translational alignment."""
    import xmipp3_amd as xa
    refs, dirs, parts, ids, nbrs = _write_dataset(tmp_path)
    n, nrefs = len(parts), len(refs)
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp.xmd"), "-o", str(tmp_path / "out.xmd"),
              "--ref", str(tmp_path / "ref.stk"), "--max_shift", "6", "--batch", "4", "--search5d_shift", "2", "--search5d_step", "2",
              "--number_orientations", "3"])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"))
    c = {l: i for i, l in enumerate(labels)}
    pos = {ids[i]: i for i in range(nrefs)}
    lists = [[pos[int(v)] for v in s.split()] for s in nbrs]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    xo, yo = xa.search5d_offsets(2, 2)
    assert len(xo) == 5
    pm = oracle.PM(refs)
    er, ep, ef, _ = pm.match(parts, off, np.concatenate(lists).astype(np.int32), n_orient=3, xoff5d=xo, yoff5d=yo)
    assert (er >= 0).all() and len(rows) == 3 * n
    for i in range(n):
        for o in range(3):
            row = rows[3 * i + o]
            ex, ey, ec = pm.translate(parts[i:i + 1], er[i:i + 1, o], ep[i:i + 1, o], ef[i:i + 1, o], 6.0)
            assert int(row[c["itemId"]]) == 100 + i
            assert int(row[c["ref"]]) == ids[er[i, o]] and int(row[c["flip"]]) == ef[i, o]
            assert abs(float(row[c["anglePsi"]]) - ep[i, o] * 360.0 / pm.N) < 1e-5
            assert abs(float(row[c["shiftX"]]) - ex[0]) < 1e-3 and abs(float(row[c["shiftY"]]) - ey[0]) < 1e-3
            assert abs(float(row[c["maxCC"]]) - ec[0]) < 1e-5


@pytest.mark.gpu
def test_cli_symmetry_names_and_files(bins, tmp_path, oracle):
    """--sym d2 (built-in: 2-fold about Z and about X) == the same group given as a symmetry file of
    rot_axis lines == the oracle summing over {I, Rz(180), Rx(180), Ry(180)}."""
    D, n = 32, 7
    vol = synth.phantom(D, seed=9, nblobs=8)
    ang = synth.random_angles(n, np.random.default_rng(2))
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "p.stk"), imgs)
    xmipp_io.write_xmd(str(tmp_path / "in.xmd"), [("noname", ["image", "angleRot", "angleTilt", "anglePsi"],
                                                  [[f"{i + 1}@{tmp_path}/p.stk"] + [f"{v:.6f}" for v in ang[i]] for i in range(n)])])
    (tmp_path / "d2.sym").write_text("# dihedral, order 4\nrot_axis 2 0 0 1\nrot_axis 2 1 0 0\n")
    vols = {}
    for tag, sym in (("name", "d2"), ("file", str(tmp_path / "d2.sym"))):
        r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "in.xmd"), "-o", str(tmp_path / f"{tag}.vol"),
                  "--sym", sym])
        assert r.returncode == 0, r.stderr
        vols[tag] = xmipp_io.read_volume(str(tmp_path / f"{tag}.vol"))
    assert np.abs(vols["name"] - vols["file"]).max() <= 2e-6 * np.abs(vols["name"]).max()
    rf = oracle.RF(D)
    Rs = [np.eye(3), np.diag([-1.0, -1.0, 1.0]), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0])]
    ang6 = np.round(ang, 6)          # the metadata carries 6 decimals
    for i in range(n):
        f = rf.prepare_image(imgs[i])
        for R in Rs:
            rf.insert(f, synth.euler_matrix(*ang6[i]).T, R=R)
    rf.mirror_and_crop()
    exp = rf.finish()
    assert np.abs(vols["name"] - exp).max() <= 1e-4 * np.abs(exp).max()
    # octahedral group by name == its generators as written by the reference (sampling.cpp:1399-1404)
    (tmp_path / "o.sym").write_text("rot_axis 3  .5773502  .5773502 .5773502\nrot_axis 4 0 0 1\n")
    vo = {}
    for tag, sym in (("oname", "o"), ("ofile", str(tmp_path / "o.sym"))):
        r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "in.xmd"), "-o", str(tmp_path / f"{tag}.vol"),
                  "--sym", sym])
        assert r.returncode == 0, r.stderr
        vo[tag] = xmipp_io.read_volume(str(tmp_path / f"{tag}.vol"))
    assert np.abs(vo["oname"] - vo["ofile"]).max() <= 1e-4 * np.abs(vo["oname"]).max()   # the file has 7-digit axes
    assert np.abs(vo["oname"] - vols["name"]).max() > 1e-2 * np.abs(vo["oname"]).max()  # and it is not d2
    # a group with a mirror plane: the reference multiplies by R whatever its determinant (reconstruct_fourier_accel.cpp:252-254,
    # 953-956).  c2h by name == rot_axis + mirror_plane lines == the oracle over {I, Rz(180), mirror z, inversion}
    (tmp_path / "c2h.sym").write_text("rot_axis 2 0 0 1\nmirror_plane 0 0 -1\n")
    vm = {}
    for tag, sym in (("mname", "c2h"), ("mfile", str(tmp_path / "c2h.sym"))):
        r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "in.xmd"), "-o", str(tmp_path / f"{tag}.vol"),
                  "--sym", sym])
        assert r.returncode == 0, r.stderr
        vm[tag] = xmipp_io.read_volume(str(tmp_path / f"{tag}.vol"))
    assert np.abs(vm["mname"] - vm["mfile"]).max() <= 2e-6 * np.abs(vm["mname"]).max()
    rf = oracle.RF(D)
    for i in range(n):
        f = rf.prepare_image(imgs[i])
        for R in (np.eye(3), np.diag([-1.0, -1.0, 1.0]), np.diag([1.0, 1.0, -1.0]), -np.eye(3)):
            rf.insert(f, synth.euler_matrix(*ang6[i]).T, R=R)
    rf.mirror_and_crop()
    expm = rf.finish()
    assert np.abs(vm["mname"] - expm).max() <= 1e-4 * np.abs(expm).max()
    # an unknown name must fail loudly, not reconstruct without symmetry
    r = _run([os.path.join(bins, "xmipp_reconstruct_fourier_accel"), "-i", str(tmp_path / "in.xmd"), "-o", str(tmp_path / "x.vol"), "--sym", "i5h"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr


@pytest.mark.gpu
def test_cli_gallery_then_matching(bins, tmp_path, oracle):
    """volume -> xmipp_angular_project_library (gallery + sampling files) -> xmipp_angular_projection_matching:
    the stack holds the FourierProjector's projections at the sampled directions (oracle parity), and particles
    projected at gallery directions are assigned back to them."""
    D = 32
    vol = synth.phantom(D, seed=21, nblobs=12).astype(np.float32)
    xmipp_io.write_volume(str(tmp_path / "in.vol"), vol)
    # experimental images: oracle projections at three sampled directions, rotated in plane
    fp = oracle.FP(vol, 2.0, 0.5, 3)
    r = _run([os.path.join(bins, "xmipp_angular_project_library"), "-i", str(tmp_path / "in.vol"), "-o", str(tmp_path / "ref.stk"),
              "--sampling_rate", "20", "--sym", "c1", "--only_create_sampling"])
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "ref.doc"))
    c = {l: i for i, l in enumerate(labels)}
    dirs = np.array([[float(rw[c["angleRot"]]), float(rw[c["angleTilt"]])] for rw in rows])
    picks = [(5, 40.0), (17, 200.0), (len(dirs) - 3, 0.0)]
    parts = np.stack([fp.project(dirs[k, 0], dirs[k, 1], psi) for k, psi in picks]).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "exp.stk"), parts)
    xmipp_io.write_xmd(str(tmp_path / "exp.xmd"), [("noname", ["image", "angleRot", "angleTilt", "anglePsi"],
                                                   [[f"{i + 1}@{tmp_path}/exp.stk", "0.000000", "0.000000", "0.000000"] for i in range(len(picks))])])
    r = _run([os.path.join(bins, "xmipp_angular_project_library"), "-i", str(tmp_path / "in.vol"), "-o", str(tmp_path / "ref.stk"),
              "--sampling_rate", "20", "--sym", "c1", "--experimental_images", str(tmp_path / "exp.xmd"), "--angular_distance", "-1",
              "--compute_neighbors", "--method", "fourier", "2", "0.5", "bspline", "--batch", "7"])
    assert r.returncode == 0, r.stderr
    gallery = xmipp_io.read_stack(str(tmp_path / "ref.stk"))
    assert gallery.shape == (len(dirs), D, D)
    for k in (0, 1, 9, len(dirs) - 1):
        exp = fp.project(dirs[k, 0], dirs[k, 1], 0.0)
        assert np.abs(gallery[k] - exp).max() <= 3e-7 * np.abs(exp).max()
    r = _run([os.path.join(bins, "xmipp_angular_projection_matching"), "-i", str(tmp_path / "exp.xmd"), "-o", str(tmp_path / "out.xmd"),
              "--ref", str(tmp_path / "ref.stk"), "--max_shift", "3"])
    assert r.returncode == 0, r.stderr
    labels, orow = xmipp_io.read_xmd(str(tmp_path / "out.xmd"))
    oc = {l: i for i, l in enumerate(labels)}
    assert len(orow) == len(picks)
    def direction(rot, tilt):
        a, b = np.radians([rot, tilt])
        return np.array([np.sin(b) * np.cos(a), np.sin(b) * np.sin(a), np.cos(b)])

    for (k, psi), rw in zip(picks, orow):
        # a projection along -d is the mirror image of the one along d: with the whole sphere in the gallery
        # both (d, no flip) and (-d, flip) are exact matches and either may win the tie
        dot = float(direction(float(rw[oc["angleRot"]]), float(rw[oc["angleTilt"]])) @ direction(*dirs[k]))
        flip = int(rw[oc["flip"]])
        assert abs(dot) > 1 - 1e-9 and flip == (1 if dot < 0 else 0)
        if not flip:
            assert int(rw[oc["ref"]]) == k
            d = abs(float(rw[oc["anglePsi"]]) - psi) % 360
            assert min(d, 360 - d) <= 360.0 / 90 + 1e-6      # within one step of the in-plane angular grid
        assert float(rw[oc["maxCC"]]) > 0.98


_CTF_COLS = ["ctfSamplingRate", "ctfVoltage", "ctfDefocusU", "ctfDefocusV", "ctfDefocusAngle", "ctfSphericalAberration", "ctfQ0", "ctfK"]


@pytest.mark.gpu
def test_cli_ctf_phase_flip_and_wiener2d(bins, tmp_path, oracle):
    """xmipp_ctf_phase_flip (ctf_phase_flip.cpp:29-86) and xmipp_ctf_correct_wiener2d (ctf_correct_wiener2d.cpp:30-105) with the
    reference's flags, against the oracle on the same files' contents."""
    rng = np.random.default_rng(21)
    mic = rng.standard_normal((96, 128)).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "mic.stk"), mic[None])
    vals = [2.0, 300.0, 21000.0, 19500.0, 25.0, 2.7, 0.07, 1.0]
    xmipp_io.write_xmd(str(tmp_path / "mic.ctfparam"), [("noname", _CTF_COLS, [[f"{v:.6f}" for v in vals]])])
    r = _run([os.path.join(bins, "xmipp_ctf_phase_flip"), "-i", f"1@{tmp_path}/mic.stk", "-o", str(tmp_path / "flipped.spi"),
              "--ctf", str(tmp_path / "mic.ctfparam"), "--downsampling", "1.5"])
    assert r.returncode == 0, r.stderr
    got = xmipp_io.read_volume(str(tmp_path / "flipped.spi"))
    got = got.reshape(got.shape[-2:])
    kw = dict(zip(("Tm", "kV", "DeltafU", "DeltafV", "azimuthal_angle", "Cs", "Q0", "K"), vals))
    kw["Tm"] = 2.0 * 1.5                  # no --sampling: the ctfparam's rate times the downsampling (ctf_phase_flip.cpp:75-79)
    exp = oracle.ctf_phase_flip(mic, oracle.ctf_params(**kw))
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    r = _run([os.path.join(bins, "xmipp_ctf_phase_flip"), "-i", f"1@{tmp_path}/mic.stk", "-o", str(tmp_path / "x.spi")])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "--ctf" in r.stderr
    # Wiener correction of a particle stack, one CTF per row
    n, D = 7, 40
    parts = rng.standard_normal((n, D, D)).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "parts.stk"), parts)
    rows = []
    for i in range(n):
        rows.append([f"{i + 1}@{tmp_path}/parts.stk"] + [f"{v:.6f}" for v in (1.0, 300.0, 15000.0 + 900.0 * i, 15400.0 + 900.0 * i, 10.0 * i, 2.7, 0.07, 1.0)] + [f"{0.5 * i:.6f}"])
    xmipp_io.write_xmd(str(tmp_path / "parts.xmd"), [("noname", ["image"] + _CTF_COLS + ["shiftX"], rows)])
    r = _run([os.path.join(bins, "xmipp_ctf_correct_wiener2d"), "-i", str(tmp_path / "parts.xmd"), "-o", str(tmp_path / "corrected.stk"),
              "--sampling_rate", "1.2", "--pad", "2", "--phase_flipped", "--batch", "3"])
    assert r.returncode == 0, r.stderr
    got = xmipp_io.read_stack(str(tmp_path / "corrected.stk"))
    assert got.shape == parts.shape
    for i in range(n):
        c = oracle.ctf_params(Tm=1.0, kV=300.0, DeltafU=15000.0 + 900.0 * i, DeltafV=15400.0 + 900.0 * i, azimuthal_angle=10.0 * i, Cs=2.7, Q0=0.07, K=1.0)
        exp = oracle.ctf_wiener2d(parts[i], c, sampling_rate=1.2, pad=2.0, phase_flipped=True)
        assert np.abs(got[i] - exp).max() <= 1e-5 * np.abs(exp).max()
    labels, orows = xmipp_io.read_xmd(str(tmp_path / "corrected.xmd"))
    # postProcess (ctf_correct_wiener2d.cpp:58-93): the defoci and K are gone, the image column names the new stack
    assert "ctfDefocusU" not in labels and "ctfK" not in labels and "shiftX" in labels and "ctfVoltage" in labels
    assert len(orows) == n and orows[2][labels.index("image")] == f"3@{tmp_path}/corrected.stk"


@pytest.mark.gpu
def test_cli_reconstruct_fourier_is_the_double_precision_program(bins, tmp_path, oracle):
    """BASELINE config 1's binary: xmipp_reconstruct_fourier = ProgRecFourier (reconstruction/reconstruct_fourier.cpp) with its own
    arithmetic on the device. Against oracle.RF2 on the same files' contents: --iter 1 (default), --iter 0 (weights set to one,
    RF:1058-1064), --iter 3 (re-processing passes), symmetry and --weight; 1e-6 of the peak (the file holds floats).
    --prepare_fsc: the halves are finished as the reference finishes them (no correctWeight, RF:991-1045), the final volume
    is the plain one; several device slots sum to the same volume."""
    D, n = 32, 30
    vol = synth.phantom(D, seed=8, nblobs=10)
    rng = np.random.default_rng(5)
    ang = synth.random_angles(n, rng)
    imgs = np.stack([synth.project(vol, *a) for a in ang]).astype(np.float32)
    w = np.round(rng.uniform(0.2, 2.0, n), 3)
    xmipp_io.write_stack(str(tmp_path / "p.stk"), imgs)
    xmipp_io.write_xmd(str(tmp_path / "all.xmd"), [("noname", ["image", "angleRot", "angleTilt", "anglePsi", "weight"],
                       [[f"{i + 1}@{tmp_path}/p.stk"] + [f"{v:.6f}" for v in ang[i]] + [f"{w[i]:.6f}"] for i in range(n)])])
    rf = os.path.join(bins, "xmipp_reconstruct_fourier")
    h_ang = np.round(ang, 6)
    c2 = np.diag([-1.0, -1.0, 1.0])

    def expected(niter, use_w, sym):
        o = oracle.RF2(D, niter_weight=niter)
        for i in range(n):
            for R in sym:
                o.insert(imgs[i], synth.euler_matrix(*h_ang[i]).T, R=R, weight=float(w[i]) if use_w else 1.0)
        return o.finish()

    cases = [([], 1, False, [np.eye(3)]), (["--iter", "0"], 0, False, [np.eye(3)]), (["--iter", "3", "--weight", "--sym", "c2", "--batch", "7"], 3, True, [np.eye(3), c2])]
    for k, (flags, niter, use_w, sym) in enumerate(cases):
        r = _run([rf, "-i", str(tmp_path / "all.xmd"), "-o", str(tmp_path / f"v{k}.vol")] + flags)
        assert r.returncode == 0, r.stderr
        got = xmipp_io.read_volume(str(tmp_path / f"v{k}.vol"))
        exp = expected(niter, use_w, sym)
        assert np.abs(got - exp).max() <= 1e-6 * np.abs(exp).max(), (flags, np.abs(got - exp).max() / np.abs(exp).max())
    plain = xmipp_io.read_volume(str(tmp_path / "v0.vol"))
    r = _run([rf, "-i", str(tmp_path / "all.xmd"), "-o", str(tmp_path / "f.vol"), "--prepare_fsc", str(tmp_path / "fsc"), "--devices", "0,0"])
    assert r.returncode == 0, r.stderr
    assert np.abs(xmipp_io.read_volume(str(tmp_path / "f.vol")) - plain).max() <= 1e-6 * np.abs(plain).max()
    split = (n - 1) // 2
    for name, idx in (("fsc_1_recons.vol", range(0, split + 1)), ("fsc_2_recons.vol", range(split + 1, n))):
        o = oracle.RF2(D, niter_weight=1)
        for i in idx:
            o.insert(imgs[i], synth.euler_matrix(*h_ang[i]).T)
        exp = o.finish(correct_weight=False)
        got = xmipp_io.read_volume(str(tmp_path / name))
        assert np.abs(got - exp).max() <= 1e-6 * np.abs(exp).max()
    # the accel program refuses nothing it used to accept; --fast belongs to it alone
    r = _run([rf, "-i", str(tmp_path / "all.xmd"), "-o", str(tmp_path / "x.vol"), "--fast"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr


@pytest.mark.gpu
def test_cli_movie_global_alignment(bins, tmp_path, oracle):
    """xmipp_movie_alignment_correlation --skipLocalAlignment (movie_alignment_correlation_base.cpp:36-68,364-396,560-600): the
    shifts written to frameShifts@out.xmd are the negatives of the oracle's estimated shifts (storeGlobalShifts), the frames
    outside --frameRange are disabled, referenceFrame@out.xmd names the reference frame; dark and gain are applied."""
    from tests.test_gpu_flexalign import synthetic_movie
    N, Y, X = 7, 200, 260
    frames, drift = synthetic_movie(N, Y, X, seed=11)
    rng = np.random.default_rng(2)
    dark = (0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    gain = (1.0 + 0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    xmipp_io.write_stack(str(tmp_path / "dark.stk"), dark[None])
    xmipp_io.write_stack(str(tmp_path / "gain.stk"), gain[None])
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "out.xmd"), "--sampling", "1.25", "--maxShift", "25",
              "--maxResForCorrelation", "10", "--frameRange", "1", "5", "--frameRangeSum", "1", "5", "--dark", f"1@{tmp_path}/dark.stk", "--gain", f"1@{tmp_path}/gain.stk",
              "--skipLocalAlignment", "--oavgInitial", str(tmp_path / "initial.spi")])
    assert r.returncode == 0, r.stderr
    exp = oracle.fa_global_alignment(frames[1:6], Ts=1.25, max_shift_px=25.0 / 1.25, max_res=10.0, dark=dark, igain=gain)
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"), block="frameShifts")
    c = {l: i for i, l in enumerate(labels)}
    assert len(rows) == N
    for n, row in enumerate(rows):
        if 1 <= n <= 5:
            assert int(float(row[c["enabled"]])) == 1
            assert abs(float(row[c["shiftX"]]) + exp["shiftX"][n - 1]) < 3e-3 and abs(float(row[c["shiftY"]]) + exp["shiftY"][n - 1]) < 3e-3
        else:
            assert int(float(row[c["enabled"]])) == -1 and float(row[c["shiftX"]]) == 0.0
    _, ref_rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"), block="referenceFrame")
    assert int(float(ref_rows[0][0])) == 1 + exp["ref"]
    ini = xmipp_io.read_volume(str(tmp_path / "initial.spi"))[0]
    assert np.abs(ini - ((frames[1:6] - dark) * gain).mean(0)).max() < 1e-4
    # a movie too small for the default patches (500 A) and a scale factor >= 1 are refused loudly
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "x.xmd"), "--sampling", "1.25", "--maxResForCorrelation", "10"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "control points" in r.stderr
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "x.xmd"), "--skipLocalAlignment", "--sampling", "5", "--maxResForCorrelation", "10"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "scale factor" in r.stderr


@pytest.mark.gpu
def test_cli_movie_size_search_stand_in(bins, tmp_path, oracle):
    """--sizeSearch smooth: the deterministic stand-in for the CUDA program's search of FFT-friendly sizes (findGoodCropSize /
    findGoodPatchSize, movie_alignment_correlation_gpu.cpp:73-121 -- they time cuFFT plans on the installed card).  Frames of 200 x 262:
    262 = 2 x 131 -> the global alignment runs on the top-left 200 x 256 window (getCroppedFrame, :727-734), which is what the oracle gets;
    patches of 100 px stay (100 = 2^2 5^2), patches of 110 px grow to 112 = 2^4 7."""
    from tests.test_gpu_flexalign import synthetic_movie
    N, Y, X = 6, 200, 262
    frames, drift = synthetic_movie(N, Y, X, seed=23)
    rng = np.random.default_rng(4)
    dark = (0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    gain = (1.0 + 0.05 * rng.standard_normal((Y, X))).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    xmipp_io.write_stack(str(tmp_path / "dark.stk"), dark[None])
    xmipp_io.write_stack(str(tmp_path / "gain.stk"), gain[None])
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    common = [prog, "-i", str(tmp_path / "movie.stk"), "--sampling", "1.25", "--maxShift", "25", "--maxResForCorrelation", "10",
              "--dark", f"1@{tmp_path}/dark.stk", "--gain", f"1@{tmp_path}/gain.stk"]
    r = _run(common + ["-o", str(tmp_path / "out.xmd"), "--skipLocalAlignment", "--sizeSearch", "smooth"])
    assert r.returncode == 0, r.stderr
    assert "global alignment on the top-left 256 x 200 of 262 x 200" in r.stdout
    exp = oracle.fa_global_alignment(frames[:, :200, :256], Ts=1.25, max_shift_px=25.0 / 1.25, max_res=10.0, dark=dark[:200, :256], igain=gain[:200, :256])
    whole = oracle.fa_global_alignment(frames, Ts=1.25, max_shift_px=25.0 / 1.25, max_res=10.0, dark=dark, igain=gain)
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"), block="frameShifts")
    c = {l: i for i, l in enumerate(labels)}
    got = np.array([[float(row[c["shiftX"]]), float(row[c["shiftY"]])] for row in rows])
    assert np.abs(got[:, 0] + exp["shiftX"]).max() < 3e-3 and np.abs(got[:, 1] + exp["shiftY"]).max() < 3e-3
    # ... and that is not what the whole frames give (the flag did something), while both see the same drift
    assert max(np.abs(got[:, 0] + whole["shiftX"]).max(), np.abs(got[:, 1] + whole["shiftY"]).max()) > 1e-3
    assert np.abs(exp["shiftX"] - whole["shiftX"]).max() < 0.5
    # patch sizes: 100 px stays, 110 px grows to 112
    for res, size in ((125, 100), (138, 112)):
        r = _run(common + ["-o", str(tmp_path / f"loc{res}.xmd"), "--sizeSearch", "smooth", "--minLocalRes", str(res), "--patches", "4", "4", "--controlPoints", "3", "3", "3"])
        assert r.returncode == 0, r.stderr
        assert f"patches of {size} px" in r.stdout and f"of {size} x {size} px" in r.stdout
    r = _run(common + ["-o", str(tmp_path / "x.xmd"), "--sizeSearch", "fastest"])
    assert r.returncode != 0 and "--sizeSearch" in r.stderr


@pytest.mark.gpu
def test_cli_movie_sum_range_inside_the_alignment_range(bins, tmp_path, oracle):
    """--frameRangeSum inside --frameRange (movie_alignment_correlation_gpu.cpp:519-541): frame fi of the aligned movie lands in slot
    fi - nfirst + 1 of --oaligned, the stack is nlastSum - nfirst + 1 images long and the slots before the first summed frame stay
    empty; the metadata blocks are appended (MD_APPEND, movie_alignment_correlation_base.cpp:396,433): a block that is already in
    the -o file from an earlier run is replaced, other blocks stay."""
    from tests.test_gpu_flexalign import synthetic_movie
    N, Y, X = 7, 200, 260
    frames, drift = synthetic_movie(N, Y, X, seed=13)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    out = tmp_path / "out.xmd"
    out.write_text("# XMIPP_STAR_1 * \n# \ndata_other\nloop_\n _ref\n 7 \ndata_referenceFrame\nloop_\n _ref\n 99 \n")
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(out), "--maxShift", "25", "--maxResForCorrelation", "8", "--frameRange", "1", "5",
              "--frameRangeSum", "3", "4", "--skipLocalAlignment", "--patches", "4", "4", "--minLocalRes", "60", "--controlPoints", "3", "3", "3",
              "--oaligned", str(tmp_path / "aligned.stk"), "--oavg", str(tmp_path / "avg.spi")])
    assert r.returncode == 0, r.stderr
    aligned = xmipp_io.read_stack(str(tmp_path / "aligned.stk"))
    assert aligned.shape == (4, Y, X)                          # slots for frames 1..4, frames 3 and 4 written
    assert not aligned[0].any() and not aligned[1].any() and aligned[2].any() and aligned[3].any()
    avg = xmipp_io.read_volume(str(tmp_path / "avg.spi"))[0]
    assert np.abs(avg - (aligned[2] + aligned[3]) / 2).max() < 1e-4 * np.abs(avg).max()
    exp = oracle.fa_global_alignment(frames[1:6], max_shift_px=25.0, max_res=8.0)
    _, ref_rows = xmipp_io.read_xmd(str(out), block="referenceFrame")
    assert len(ref_rows) == 1 and int(float(ref_rows[0][0])) == 1 + exp["ref"]
    _, other = xmipp_io.read_xmd(str(out), block="other")
    assert int(float(other[0][0])) == 7
    assert out.read_text().count("data_referenceFrame") == 1


@pytest.mark.gpu
def test_cli_movie_binning(bins, tmp_path, oracle):
    """--bin 2 (movie_alignment_correlation_base.cpp:39-42,356-370,376; movie_alignment_correlation_gpu.cpp:667-691): frames are corrected
    and binned as they are loaded, the alignment runs on the binned movie at the binned sampling rate, the shifts go out multiplied
    by the binning factor and the micrograph has the binned size."""
    from tests.test_gpu_flexalign import synthetic_movie
    N, Y, X = 6, 400, 520
    frames, drift = synthetic_movie(N, Y, X, seed=17, max_step=2.5)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "out.xmd"), "--bin", "2", "--sampling", "1", "--maxShift", "40", "--maxResForCorrelation", "16",
              "--skipLocalAlignment", "--oavgInitial", str(tmp_path / "initial.spi")])
    assert r.returncode == 0, r.stderr
    binned = np.stack([oracle.fa_bin_frame(f, Y // 2, X // 2) for f in frames])
    exp = oracle.fa_global_alignment(binned.astype(np.float32), Ts=2.0, max_shift_px=40.0 / 2.0, max_res=16.0)
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"), block="frameShifts")
    c = {l: i for i, l in enumerate(labels)}
    for n, row in enumerate(rows):
        assert abs(float(row[c["shiftX"]]) + 2 * exp["shiftX"][n]) < 1e-2 and abs(float(row[c["shiftY"]]) + 2 * exp["shiftY"][n]) < 1e-2
    ini = xmipp_io.read_volume(str(tmp_path / "initial.spi"))[0]
    assert ini.shape == (Y // 2, X // 2)
    assert np.abs(ini - binned.mean(0)).max() < 1e-4 * np.abs(binned).max()
    # and the drift that was put in comes out in pixels of the RAW movie
    t = drift - drift[exp["ref"]]
    got = np.array([[float(r_[c["shiftX"]]), float(r_[c["shiftY"]])] for r_ in rows])
    assert np.abs(got - t).max() < 1.5


@pytest.mark.gpu
def test_cli_movie_local_alignment(bins, tmp_path, oracle):
    """xmipp_movie_alignment_correlation with the local alignment (run(), movie_alignment_correlation_base.cpp:531-586, steps of the
    CUDA program): localAlignment@out.xmd carries patches, control points and the spline coefficients; the aligned frames
    (--oaligned) and their average (--oavg) are the oracle's warp of the corrected frames by the oracle's spline; with
    --skipLocalAlignment the spline of localFromGlobal moves every frame by its global shift."""
    from tests import synth
    N, Y, X = 8, 384, 384
    frames, drift, field = synth.movie(N, Y, X, seed=3, local=4.0)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    common = ["-i", str(tmp_path / "movie.stk"), "--maxShift", "30", "--maxResForCorrelation", "8", "--patches", "5", "5", "--minLocalRes", "128",
              "--controlPoints", "3", "3", "3"]
    r = _run([prog] + common + ["-o", str(tmp_path / "out.xmd"), "--oavg", str(tmp_path / "avg.spi"), "--oaligned", str(tmp_path / "aligned.stk"),
                                "--oavgInitial", str(tmp_path / "initial.spi")])
    assert r.returncode == 0, r.stderr
    g = oracle.fa_global_alignment(frames, max_shift_px=30.0, max_res=8.0)
    cp = (3, 3, 3)
    loc = oracle.fa_local_alignment(frames, g["shiftX"], g["shiftY"], g["ref"], max_shift_px=30.0, max_res=8.0, patches=(5, 5), patch_size=(128, 128), control_points=cp)
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "out.xmd"), block="localAlignment")
    c = {l: i for i, l in enumerate(labels)}
    vec = lambda t: np.array([float(v) for v in t.strip("'[] ").split()])
    assert list(vec(rows[0][c["localAlignmentPatches"]])) == [5, 5] and list(vec(rows[0][c["localAlignmentControlPoints"]])) == [3, 3, 3]
    cx, cy = vec(rows[0][c["localAlignmentCoeffsX"]]), vec(rows[0][c["localAlignmentCoeffsY"]])
    assert cx.size == 27 and cy.size == 27
    # the spline of the written coefficients against the oracle's over the field (the coefficients themselves are ill-conditioned)
    worst = 0.0
    for n in range(N):
        for y in range(0, Y, 48):
            for x in range(0, X, 48):
                a, b = oracle.fa_bspline_shift(cx, cy, cp, X, Y, N, x, y, n), oracle.fa_bspline_shift(loc["coeffsX"], loc["coeffsY"], cp, X, Y, N, x, y, n)
                worst = max(worst, abs(a[0] - b[0]), abs(a[1] - b[1]))
    assert worst < 0.02, worst
    lo, hi = float(rows[0][c["localAlignmentConf2_5Perc"]]), float(rows[0][c["localAlignmentConf97_5Perc"]])
    assert 0 <= lo <= hi < 10
    # storeResults (movie_alignment_correlation_base.cpp:470-486) receives BSplineHelper::getShift's pair with X and Y exchanged
    # (bspline_helper.cpp:99 calls a callee declared (shiftY, shiftX), :112): distances are hypot(splineY - globalX, splineX - globalY)
    dist = []
    for (px, py) in np.asarray(loc["centers"]).reshape(-1, 2):
        for t in range(N):
            bx, by = oracle.fa_bspline_shift(cx, cy, cp, X, Y, N, int(px), int(py), t)
            dist.append(np.hypot(by - g["shiftX"][t], bx - g["shiftY"][t]))
    dist = np.sort(dist)
    assert abs(lo - dist[int(dist.size * 0.025)]) < 0.02 and abs(hi - dist[int(dist.size * 0.975)]) < 0.02
    aligned = xmipp_io.read_stack(str(tmp_path / "aligned.stk"))
    assert aligned.shape == (N, Y, X)
    exp = np.stack([oracle.fa_apply_bspline(frames[n], loc["coeffsX"].astype(np.float32), loc["coeffsY"].astype(np.float32), cp, N, n) for n in range(N)])
    scale = np.abs(exp).max()
    assert np.abs(aligned - exp).max() < 2e-3 * scale
    avg = xmipp_io.read_volume(str(tmp_path / "avg.spi"))[0]
    assert np.abs(avg - exp.mean(0)).max() < 2e-3 * scale
    ini = xmipp_io.read_volume(str(tmp_path / "initial.spi"))[0]
    assert np.abs(ini - frames.mean(0)).max() < 1e-4
    # global only: every frame moved by its global shift (order-3 interpolation of the prefiltered frame), sums of frames 2..6 only
    r = _run([prog] + common + ["-o", str(tmp_path / "out2.xmd"), "--skipLocalAlignment", "--oavg", str(tmp_path / "avg2.spi"), "--frameRangeSum", "2", "6"])
    assert r.returncode == 0, r.stderr
    avg2 = xmipp_io.read_volume(str(tmp_path / "avg2.spi"))[0]
    clean = synth.movie(N, Y, X, seed=3, local=4.0, noise=0.0)[0]
    inner = (slice(40, -40), slice(40, -40))
    cc = lambda a: np.corrcoef(a[inner].ravel(), clean[g["ref"]][inner].ravel())[0, 1]
    print("correlation with the clean reference frame: aligned", cc(avg2), "unaligned", cc(frames[2:7].mean(0)), "locally aligned", cc(avg))
    assert cc(avg2) > cc(frames[2:7].mean(0)) + 0.01 and cc(avg) > cc(avg2) - 0.01
    labels2, _ = xmipp_io.read_xmd(str(tmp_path / "out2.xmd"), block="frameShifts")
    assert "shiftX" in labels2


@pytest.mark.gpu
def test_cli_movie_filter_dose(bins, tmp_path, oracle):
    """xmipp_movie_filter_dose (reconstruction/movie_filter_dose.cpp:36-290): every frame of the range filtered with the doses of its
    place in the movie (frame n: n x dose + pre-exposure .. (n + 1) x dose + pre-exposure), written at its place in the output stack;
    the oracle's frames to 1e-5 of the frame's range (fp32 transforms); a voltage other than 200 / 300 kV is refused."""
    rng = np.random.default_rng(4)
    N, Y, X = 5, 96, 120
    frames = rng.standard_normal((N, Y, X)).astype(np.float32)
    xmipp_io.write_stack(str(tmp_path / "movie.stk"), frames)
    prog = os.path.join(bins, "xmipp_movie_filter_dose")
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "out.stk"), "--sampling", "1.3", "--dosePerFrame", "3.5", "--accVoltage", "200",
              "--preExposure", "1", "--frameRange", "1", "3"])
    assert r.returncode == 0, r.stderr
    out = xmipp_io.read_stack(str(tmp_path / "out.stk"))
    assert out.shape == (4, Y, X) and np.abs(out[0]).max() == 0
    for n in (1, 2, 3):
        exp = oracle.dose_filter_frame(frames[n], 1.3, 200, n * 3.5 + 1, (n + 1) * 3.5 + 1)
        assert np.abs(out[n] - exp).max() <= 1e-5 * np.abs(exp).max() + 1e-6
        assert np.abs(exp - frames[n]).max() > 0.1                    # the filter does something
    r = _run([prog, "-i", str(tmp_path / "movie.stk"), "-o", str(tmp_path / "x.stk"), "--accVoltage", "250"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr and "acceleration voltage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 6])
def test_cli_movie_of_integer_counts(bins, tmp_path, mode):
    """Movies as detectors write them: MRC stacks of int8 / int16 / uint16 counts (modes 0, 1, 6).  The program sends the counts to the
    device and casts there (xh_movie_frame_to_float) -- same shifts, same sums, bit for bit, as from the float32 stack of the same
    values, with dark and gain, with and without --bin."""
    from tests.test_gpu_flexalign import synthetic_movie
    N, Y, X = 5, 256, 328
    frames, drift = synthetic_movie(N, Y, X, seed=23, max_step=2.0)
    lo, hi = {0: (-100, 100), 1: (-20000, 20000), 6: (0, 60000)}[mode]
    f = frames - frames.min()
    counts = np.round(lo + f / f.max() * (hi - lo)).astype(np.float32)
    xmipp_io.write_mrcs(str(tmp_path / "counts.mrcs"), counts, mode)
    xmipp_io.write_mrcs(str(tmp_path / "floats.mrcs"), counts, 2)
    rng = np.random.default_rng(4)
    xmipp_io.write_volume(str(tmp_path / "dark.spi"), rng.uniform(0, 2, (1, Y, X)).astype(np.float32))
    xmipp_io.write_volume(str(tmp_path / "gain.spi"), rng.uniform(0.9, 1.1, (1, Y, X)).astype(np.float32))
    prog = os.path.join(bins, "xmipp_movie_alignment_correlation")
    for extra in ([], ["--bin", "2"]):
        out = {}
        for tag in ("counts", "floats"):
            r = _run([prog, "-i", str(tmp_path / f"{tag}.mrcs"), "-o", str(tmp_path / f"{tag}.xmd"), "--sampling", "1", "--maxShift", "30", "--maxResForCorrelation", "16",
                      "--skipLocalAlignment", "--dark", str(tmp_path / "dark.spi"), "--gain", str(tmp_path / "gain.spi"), "--oavgInitial", str(tmp_path / f"{tag}.spi")] + extra)
            assert r.returncode == 0, r.stderr
            labels, rows = xmipp_io.read_xmd(str(tmp_path / f"{tag}.xmd"), block="frameShifts")
            out[tag] = (rows, xmipp_io.read_volume(str(tmp_path / f"{tag}.spi")))
        c = {l: i for i, l in enumerate(labels)}
        num = lambda rows: [(r_[c["shiftX"]], r_[c["shiftY"]]) for r_ in rows]
        assert num(out["counts"][0]) == num(out["floats"][0]) and len(out["counts"][0]) == N
        assert np.array_equal(out["counts"][1], out["floats"][1])
    c = {l: i for i, l in enumerate(labels)}
    got = np.array([[float(r_[c["shiftX"]]), float(r_[c["shiftY"]])] for r_ in out["counts"][0]])
    assert np.abs(np.diff(got, axis=0) - np.diff(drift, axis=0)).max() < 1.5          # and they are the drift that was put in
    # a mode the reader does not know is refused, not misread
    bad = bytearray(open(tmp_path / "counts.mrcs", "rb").read())
    bad[12:16] = (4).to_bytes(4, "little")
    open(tmp_path / "bad.mrcs", "wb").write(bytes(bad))
    r = _run([prog, "-i", str(tmp_path / "bad.mrcs"), "-o", str(tmp_path / "bad.xmd"), "--skipLocalAlignment"])
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr
