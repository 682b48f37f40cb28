"""Host-side sampling of xmipp_angular_project_library against the reference's own fixtures
(resources/test/sampling/*.xmd, test_sampling_main.cpp:128-179; copied as data under tests/golden/sampling/).
Needs no device: `--only_create_sampling` stops before the projections."""
import os
import subprocess

import numpy as np
import pytest

from tests import xmipp_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "sampling")


@pytest.fixture(scope="module")
def prog():
    import __graft_entry__ as ge
    ge.build()
    p = os.path.join(ROOT, "xmipp3_amd", "bin", "xmipp_angular_project_library")
    assert os.path.exists(p)
    return p


def _block(path, block):
    labels, rows = xmipp_io.read_xmd(path, block=block)
    return labels, rows


def _dirs(rows, c):
    idx = np.array([int(r[c["neighbor"]]) for r in rows])
    ang = np.array([[float(r[c["angleRot"]]), float(r[c["angleTilt"]]), float(r[c["anglePsi"]])] for r in rows])
    return idx, ang


def test_sampling_points_c1_match_the_reference_fixture(prog, tmp_path):
    """3 degrees, c1, whole sphere: 4412 directions, same order, same angles and unit vectors (6 decimals)."""
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", "c1",
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "g.doc"))
    c = {l: i for i, l in enumerate(labels)}
    gl, grows = _block(os.path.join(GOLD, "ref_c1_sampling.xmd"), "projectionDirections")
    gc = {l: i for i, l in enumerate(gl)}
    assert len(rows) == len(grows) == 4412
    got = np.array([[float(r[c[k]]) for k in ("angleRot", "angleTilt", "anglePsi", "X", "Y", "Z")] for r in rows])
    exp = np.array([[float(r[gc[k]]) for k in ("angleRot", "angleTilt", "anglePsi", "X", "Y", "Z")] for r in grows])
    assert [int(r[c["ref"]]) for r in rows] == [int(r[gc["neighbor"]]) for r in grows]
    # both sides are printed with 6 decimals; rot = atan2 of ~0/~0 at the poles is the only ill-conditioned entry
    assert np.abs(got[:, 1:] - exp[:, 1:]).max() <= 2e-6
    d = np.abs(got[:, 0] - exp[:, 0])
    d = np.minimum(d, 360 - d)
    assert d[2:].max() <= 2e-6
    assert [r[c["image"]] for r in rows][:2] == [f"000001@{tmp_path}/g.stk", f"000002@{tmp_path}/g.stk"]


@pytest.mark.parametrize("only_winner", [False, True])
def test_neighbourhoods_c1_match_the_reference_fixture(prog, tmp_path, only_winner):
    """--near_exp_data + --compute_neighbors at 5 degrees around three experimental images."""
    args = [prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", "c1",
            "--experimental_images", os.path.join(GOLD, "experimental_images.xmd"), "--angular_distance", "5",
            "--near_exp_data", "--compute_neighbors", "--only_create_sampling"]
    if only_winner:
        args.append("--only_winner")
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fn = str(tmp_path / "g_sampling.xmd")
    gold = os.path.join(GOLD, "neigh_ref_c1_exp_sampling.xmd")
    l, rows = _block(fn, "extra")
    c = {k: i for i, k in enumerate(l)}
    assert abs(float(rows[0][c["sampling_rate"]]) - 0.0523599) < 1e-6
    assert abs(float(rows[0][c["neighborhoodRadius"]]) - 0.996195) < 1e-6
    l, rows = _block(fn, "projectionDirections")
    c = {k: i for i, k in enumerate(l)}
    gl, grows = _block(gold, "projectionDirections")
    gc = {k: i for i, k in enumerate(gl)}
    gi, ga = _dirs(rows, c)
    ei, ea = _dirs(grows, gc)
    assert gi.tolist() == ei.tolist()                       # same survivors in the same (swap-delete) order
    assert np.abs(ga - ea).max() <= 2e-6
    l, rows = _block(fn, "neighbors")
    c = {k: i for i, k in enumerate(l)}
    gl, grows = _block(gold, "neighbors")
    gc = {k: i for i, k in enumerate(gl)}
    got = [[int(v) for v in r[c["neighbors"]].split()] for r in rows]
    exp = [[int(v) for v in r[gc["neighbors"]].split()] for r in grows]
    assert [int(r[c["neighbor"]]) for r in rows] == [1, 2, 3]
    if only_winner:
        # a single neighbour: the closest direction, which is one of the fixture's neighbours
        assert all(len(g) == 1 and g[0] in e for g, e in zip(got, exp))
    else:
        assert got == exp


def test_cn_dn_units_and_loud_failures(prog, tmp_path):
    n = {}
    for sym in ("c1", "c4", "d2"):
        r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / f"{sym}.stk"), "--sampling_rate", "6", "--sym", sym,
                            "--only_create_sampling"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        n[sym] = len(xmipp_io.read_xmd(str(tmp_path / f"{sym}.doc"))[1])
    assert 0.2 < n["c4"] / n["c1"] < 0.32 and 0.2 < n["d2"] / n["c1"] < 0.32
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "x.stk"), "--sym", "i3", "--only_create_sampling"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "x.stk"), "--method", "real_space", "--only_create_sampling"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr
