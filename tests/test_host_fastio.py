"""CPU checks of the host side that feeds the two hot-path programs (xmipp3_amd/host/fastio.h): the in-place, multi-threaded
metadata reader against the plain reader of minicore.h (cell for cell) and against the reference's own sampling fixtures, the
neighbour lists with their sharing of identical rows, the stack reader for every pixel type.  No device."""
import os
import subprocess

import numpy as np
import pytest

from tests import xmipp_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "sampling")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "fastio_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Wno-unused-function",
                           os.path.join(ROOT, "tests", "host", "fastio_probe.cpp"), "-o", exe])
    return exe


def run(exe, *args, binary=False):
    r = subprocess.run([exe, *map(str, args)], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
    return r.stdout if binary else r.stdout.decode()


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("name", ["experimental_images.xmd", "projectionDirections@neigh_ref_c1_exp_sampling.xmd",
                                  "neighbors@neigh_ref_i3h_exp_sampling.xmd", "extra@ref_c1_sampling.xmd"])
def test_table_equals_the_plain_reader_on_the_reference_fixtures(probe, name, threads):
    path = name if "@" not in name else name.split("@")[0] + "@" + os.path.join(GOLD, name.split("@")[1])
    if "@" not in name:
        path = os.path.join(GOLD, name)
    out = run(probe, "table", path, threads)
    assert out.startswith("rows ")


def test_large_table_split_over_threads(probe, tmp_path):
    # > 1 MB of rows so that the reader really cuts the block into pieces; odd spacing, comments, a quoted cell, short rows
    rng = np.random.default_rng(3)
    n = 30000
    vals = rng.normal(0, 100, (n, 3))
    fn = tmp_path / "big.xmd"
    with open(fn, "w") as f:
        f.write("# XMIPP_STAR_1 * \n# \ndata_first\nloop_\n _a\n _b\n1 2\n3 4\ndata_noname\nloop_\n _itemId\n _image\n _angleRot #c\n _shiftX\n _text\n _last\n")
        for i in range(n):
            if i % 1000 == 7:
                f.write("# a comment line\n\n")
            short = i % 977 == 5
            f.write(f"  {i + 1}   {i + 1:06d}@/some/where/parts.stk\t{vals[i, 0]:.6f}    {vals[i, 1]:+.6e} 'two words {i}'" + ("" if short else f" {vals[i, 2]:.4f}") + " \n")
        f.write("data_after\nloop_\n _x\n 5\n")
    for thr in (1, 4):
        out = run(probe, "table", f"noname@{fn}", thr)
        assert out.startswith(f"rows {n} labels 6")
    got = np.array([float(x) for x in run(probe, "numbers", f"noname@{fn}", "shiftX").split()])
    exp = np.array([float(f"{v:+.6e}") for v in vals[:, 1]])
    assert np.array_equal(got, exp)                      # from_chars is correctly rounded, as float() is
    got = np.array([float(x) for x in run(probe, "numbers", f"noname@{fn}", "last").split()])
    exp = np.array([(-12345.0 if i % 977 == 5 else float(f"{vals[i, 2]:.4f}")) for i in range(n)])
    assert np.array_equal(got, exp)                      # a short row reads as "label absent" -> the caller's default
    assert run(probe, "table", f"first@{fn}", 2).startswith("rows 2 labels 2")
    assert run(probe, "table", f"after@{fn}", 2).startswith("rows 1 labels 1")
    assert run(probe, "table", str(fn), 2).startswith("rows 2 labels 2")          # no block name: the first block


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_neighbour_lists(probe, tmp_path, threads):
    rng = np.random.default_rng(5)
    nrefs, n = 300, 1500
    lists = []
    whole = list(range(nrefs))
    for i in range(n):
        if i % 4 != 3:
            lists.append(whole if i < 900 else lists[-1] if lists and i % 2 else sorted(rng.choice(nrefs, rng.integers(0, 40), replace=False).tolist()))
        else:
            lists.append(sorted(rng.choice(nrefs, rng.integers(0, 40), replace=False).tolist()))
    fn = tmp_path / "ref_sampling.xmd"
    xmipp_io.write_xmd(fn, [("extra", ["sampling_rate"], [[0.05]]),
                            ("neighbors", ["neighbor", "neighbors"], [[i + 1, " " + " ".join(map(str, l)) + " "] for i, l in enumerate(lists)]),
                            ("projectionDirections", ["neighbor"], [[i] for i in range(nrefs)])])
    out = run(probe, "neigh", fn, threads).splitlines()
    head = out[0].split()
    assert int(head[1]) == n
    got = [[int(x) for x in l.split()] for l in out[1:]]
    assert got == lists
    # identical consecutive rows are stored once (a piece boundary may cost one extra copy per thread)
    distinct = sum(1 for k, l in enumerate(lists) if k == 0 or l != lists[k - 1])
    assert distinct <= int(head[3]) <= distinct + threads and int(head[3]) < n


def test_neighbour_lists_of_the_reference_fixture(probe):
    out = run(probe, "neigh", os.path.join(GOLD, "neigh_ref_c1_exp_sampling.xmd"), 2).splitlines()
    labels, rows = xmipp_io.read_xmd(os.path.join(GOLD, "neigh_ref_c1_exp_sampling.xmd"), "neighbors")
    exp = [[int(x) for x in r[labels.index("neighbors")].split()] for r in rows]
    assert [[int(x) for x in l.split()] for l in out[1:]] == exp


@pytest.mark.parametrize("mode", [2, 0, 1, 6])
def test_stack_reader(probe, tmp_path, mode):
    rng = np.random.default_rng(mode)
    n, D = 37, 24
    if mode == 2:
        imgs = rng.standard_normal((n, D, D)).astype(np.float32)
    else:
        lo, hi = {0: (-128, 128), 1: (-3000, 3000), 6: (0, 60000)}[mode]
        imgs = rng.integers(lo, hi, (n, D, D)).astype(np.float32)
    order = rng.permutation(n)
    names = []
    if mode == 2:
        fn = tmp_path / "s.stk"
        xmipp_io.write_stack(fn, imgs)
    else:
        fn = tmp_path / "s.mrcs"
        xmipp_io.write_mrcs(fn, imgs, mode)
    names = [f"{i + 1}@{fn}" for i in order]
    for thr in (1, 4):
        raw = run(probe, "stack", D, thr, *names, binary=True)
        got = np.frombuffer(raw, np.float32).reshape(n, D, D)
        assert np.array_equal(got, imgs[order])
    r = subprocess.run([probe, "stack", str(D + 1), "1", names[0]], capture_output=True)
    assert r.returncode == 3 and b"different size" in r.stderr
    r = subprocess.run([probe, "stack", str(D), "1", f"{n + 1}@{fn}"], capture_output=True)
    assert r.returncode == 3


def test_metadata_write_round_trip(probe, tmp_path):
    src = os.path.join(GOLD, "experimental_images.xmd")
    dst = tmp_path / "out.xmd"
    run(probe, "write", src, dst)
    (la, ra), (lb, rb) = xmipp_io.read_xmd(src), xmipp_io.read_xmd(dst)
    # (the reference's fixture carries one value more per row than it has labels; every reader drops it)
    assert la == lb and [r[:len(la)] for r in ra] == rb
    # the fixed-width layout the programs have always written: cells right-aligned to 12, one blank in front, " \n" at the end
    line = [l for l in open(dst) if "proj_sh000001" in l][0]
    assert line.endswith(" \n") and line.startswith(" ")
