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


def test_groups_write_one_sampling_file_per_block(prog, tmp_path):
    """--groups (createGroupSamplingFiles, angular_project_library.cpp:405-470): for every block of the groups file the neighbours of
    the experimental images in the block of that name are written to <root>_groupNNNNNN_sampling.xmd.  A group holding all three
    images of the reference's fixture gives the neighbour lists of the ungrouped run; a group with the second image alone gives its
    list as the only one."""
    src = open(os.path.join(GOLD, "experimental_images.xmd")).read().splitlines()
    head = [l for l in src if l.startswith("#")]
    start = next(i for i, l in enumerate(src) if l.startswith("data_"))
    body = src[start + 1:]
    rows = [l for l in body if l.strip() and not l.strip().startswith(("_", "loop_"))]
    labels = [l for l in body if l.strip().startswith(("_", "loop_"))]
    exp = tmp_path / "exp.xmd"
    exp.write_text("\n".join(head + ["data_all"] + labels + rows + ["data_second"] + labels + [rows[1]]) + "\n")
    (tmp_path / "groups.xmd").write_text("# XMIPP_STAR_1 * \n# \ndata_all\nloop_\n _image\n x\ndata_second\nloop_\n _image\n x\n")
    args = [prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", "c1", "--experimental_images", "all@" + str(exp),
            "--angular_distance", "5", "--compute_neighbors", "--groups", str(tmp_path / "groups.xmd"), "--only_create_sampling"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    def lists(fn):
        l, rw = _block(fn, "neighbors")
        c = {k: i for i, k in enumerate(l)}
        return [[int(v) for v in q[c["neighbors"]].split()] for q in rw]
    whole = lists(str(tmp_path / "g_sampling.xmd"))
    assert lists(str(tmp_path / "g_group000001_sampling.xmd")) == whole and len(whole) == 3
    assert lists(str(tmp_path / "g_group000002_sampling.xmd")) == [whole[1]]
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "h.stk"), "--groups", str(tmp_path / "groups.xmd"), "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr


def test_cn_dn_units_and_loud_failures(prog, tmp_path):
    n = {}
    for sym in ("c1", "c4", "d2"):
        r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / f"{sym}.stk"), "--sampling_rate", "6", "--sym", sym,
                            "--only_create_sampling"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        n[sym] = len(xmipp_io.read_xmd(str(tmp_path / f"{sym}.doc"))[1])
    assert 0.2 < n["c4"] / n["c1"] < 0.32 and 0.2 < n["d2"] / n["c1"] < 0.32
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "x.stk"), "--sym", "i5h", "--only_create_sampling"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr         # not implemented in the reference either (sampling.cpp:1216)
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "x.stk"), "--method", "real_space", "--only_create_sampling"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "XMIPP_ERROR" in r.stderr


# ---- point groups beyond cN: generators (Sampling::createSymFile, sampling.cpp:1377-1416) and asymmetric
# units (Sampling::removeRedundantPoints, sampling.cpp:808-1069) are both in the reference tree; that they fit
# together -- and in which orientation i1, i3, i4 relate to i2 -- is checked here numerically.
def _rot_axis(fold, ax):
    ax = np.array(ax, float)
    ax /= np.linalg.norm(ax)
    a = 2 * np.pi / fold
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _closure(gens):
    G = [np.eye(3)]
    grew = True
    while grew:
        grew = False
        for a in list(G):
            for g in gens:
                P = a @ g
                if not any(np.abs(P - h).max() < 1e-6 for h in G):
                    G.append(P)
                    grew = True
                    assert len(G) <= 240
    return G


def _group(name):
    from tests import synth
    phi = (1 + 5 ** 0.5) / 2
    if name == "t":
        return _closure([_rot_axis(3, (0, 0, 1)), _rot_axis(2, (0, (2 / 3) ** 0.5, (1 / 3) ** 0.5))])
    if name == "o":
        return _closure([_rot_axis(3, (1, 1, 1)), _rot_axis(4, (0, 0, 1))])
    G2 = _closure([_rot_axis(2, (0, 0, 1)), _rot_axis(5, (-phi, -1, 0)), _rot_axis(3, (-1, -phi * phi, 0))])
    tilt = {"i1": 90.0, "i2": 0.0, "i3": 31.7174745559, "i4": -31.7174745559}[name]
    A = synth.euler_matrix(0, tilt, 0)
    return [A @ g @ A.T for g in G2]


def _in_unit(name, v, eps=1e-9):
    """the reference's asymmetric unit, open (strict) version for the uniqueness count"""
    from tests import synth
    u = lambda x: np.array(x, float) / np.linalg.norm(x)
    rot = np.degrees(np.arctan2(v[1], v[0]))
    tilt = np.degrees(np.arccos(np.clip(v[2], -1, 1)))
    if name == "t":
        n = [u((-0.942809, 0, 0)), u((0.471405, 0.272165, 0.7698)), u((0.471404, 0.816497, 0))]
        return all(v @ k > eps for k in n) and 90 <= rot <= 150
    if name == "o":
        n = [u((0, -1, 1)), u((1, 1, 0)), u((-1, 1, 0))]
        return all(v @ k > eps for k in n) and 45 <= rot <= 135 and tilt <= 90
    if name == "i4":
        A = synth.euler_matrix(0, -31.7174745559, 0)
        n = [A @ u((0, 0, 1)), A @ u((0.187592467856686, -0.303530987314591, -0.491123477863004)),
             A @ u((0.187592467856686, 0.303530987314591, -0.491123477863004))]
        return all(v @ k < -eps for k in n)
    A = synth.euler_matrix(0, {"i1": 90.0, "i2": 0.0, "i3": 31.7174745559}[name], 0)
    n = [A @ u((0, 1, 0)), A @ u((-0.4999999839058737, -0.8090170074556163, 0.3090169861701543)),
         A @ u((0.4999999839058737, -0.8090170074556163, 0.3090169861701543))]
    return all(v @ k > eps for k in n)


@pytest.mark.parametrize("name,order", [("t", 12), ("o", 24), ("i1", 60), ("i2", 60), ("i3", 60), ("i4", 60)])
def test_cubic_groups_generators_fit_the_asymmetric_units(prog, tmp_path, name, order):
    # (1) in-tree generators x in-tree asymmetric unit = a fundamental domain: every direction has exactly one image in it
    G = _group(name)
    assert len(G) == order
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((1500, 3))
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    assert all(sum(_in_unit(name, g @ p) for g in G) == 1 for p in pts)
    # (2) the host program uses exactly that group and that unit
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "4", "--sym", name,
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    M = np.loadtxt(str(tmp_path / "g_symmetry.txt")).reshape(-1, 3, 3)
    assert len(M) == order and np.allclose(M[0], np.eye(3))
    for m in M:
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-12) and np.linalg.det(m) > 0
        assert any(np.abs(m - g).max() < 1e-9 for g in G)
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "all.stk"), "--sampling_rate", "4", "--sym", "c1",
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    la, ra = xmipp_io.read_xmd(str(tmp_path / "all.doc"))
    lu, ru = xmipp_io.read_xmd(str(tmp_path / "g.doc"))
    ca, cu = {k: i for i, k in enumerate(la)}, {k: i for i, k in enumerate(lu)}
    allv = np.array([[float(x[ca[k]]) for k in ("X", "Y", "Z")] for x in ra])
    unit = np.array([[float(x[cu[k]]) for k in ("X", "Y", "Z")] for x in ru])
    assert abs(len(unit) / len(allv) - 1.0 / order) < 0.6 / order       # boundary points count for several units
    # (the docfile carries 6 decimals and i3's boundary planes pass through sampling points: keep a margin)
    inside = np.array([_in_unit(name, v, 1e-5) for v in allv])
    keys = {tuple(np.round(v, 6)) for v in unit}
    assert all(tuple(np.round(v, 6)) in keys for v in allv[inside])    # every interior direction was kept


def test_named_groups_in_symlist(prog, tmp_path):
    for name, order in (("c5", 5), ("d7", 14), ("d2", 4), ("i", 60)):
        r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "10", "--sym", name,
                            "--only_create_sampling"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        M = np.loadtxt(str(tmp_path / "g_symmetry.txt")).reshape(-1, 3, 3)
        assert len(M) == order
        prods = {tuple(np.round((a @ b).ravel(), 6) + 0.0) for a in M for b in M}
        assert len(prods) == order                                   # closed under multiplication


# ---- groups with mirror planes / inversion.  The reference's own typed test runs on i3h (test_sampling_main.cpp:58-76, 94-110,
# 144-162); its three fixtures are under tests/golden/sampling/ (the projectionDirectionsSphere block, a repeat of ref_c1's, cut).
def test_asymmetric_unit_i3h_matches_the_reference_fixture(prog, tmp_path):
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", "i3h",
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    labels, rows = xmipp_io.read_xmd(str(tmp_path / "g.doc"))
    c = {l: i for i, l in enumerate(labels)}
    gold = os.path.join(GOLD, "ref_i3h_sampling.xmd")
    gl, grows = _block(gold, "projectionDirections")
    gc = {l: i for i, l in enumerate(gl)}
    assert " _pointsAsymmetricUnit 30\n" in open(gold).read()
    assert len(rows) == len(grows) == 30
    keys = ("angleRot", "angleTilt", "anglePsi", "X", "Y", "Z")
    got = np.array([[float(r[c[k]]) for k in keys] for r in rows])
    exp = np.array([[float(r[gc[k]]) for k in keys] for r in grows])
    assert [int(r[c["ref"]]) for r in rows] == [int(r[gc["neighbor"]]) for r in grows]
    assert np.abs(got - exp).max() <= 2e-6


def test_neighbourhoods_i3h_match_the_reference_fixtures(prog, tmp_path):
    """removePointsFarAwayFromExperimentalData + computeNeighbors over the 120 elements of i3h: 15 of the 30 directions survive, in the
    fixture's (swap-delete) order, with the fixture's neighbour lists.  This is also what pins the action of the improper
    elements on a direction: d -> R^T d with R of determinant -1, no left matrix (with xmippCore's L = diag(1,1,-1) applied
    to the vector, 20 directions would survive)."""
    args = [prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", "i3h",
            "--experimental_images", os.path.join(GOLD, "experimental_images.xmd"), "--angular_distance", "5",
            "--near_exp_data", "--compute_neighbors", "--only_create_sampling"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fn = str(tmp_path / "g_sampling.xmd")
    for gold in ("ref_i3h_exp_sampling.xmd", "neigh_ref_i3h_exp_sampling.xmd"):
        l, rows = _block(fn, "projectionDirections")
        gl, grows = _block(os.path.join(GOLD, gold), "projectionDirections")
        gi, ga = _dirs(rows, {k: i for i, k in enumerate(l)})
        ei, ea = _dirs(grows, {k: i for i, k in enumerate(gl)})
        assert gi.tolist() == ei.tolist() and len(gi) == 15
        assert np.abs(ga - ea).max() <= 2e-6
    l, rows = _block(fn, "neighbors")
    gl, grows = _block(os.path.join(GOLD, "neigh_ref_i3h_exp_sampling.xmd"), "neighbors")
    got = [[int(v) for v in r[l.index("neighbors")].split()] for r in rows]
    exp = [[int(v) for v in r[gl.index("neighbors")].split()] for r in grows]
    assert got == exp
    M = np.loadtxt(str(tmp_path / "g_symmetry.txt")).reshape(-1, 3, 3)
    assert len(M) == 120 and sum(np.linalg.det(m) < 0 for m in M) == 60


def _mirror(n):
    n = np.array(n, float) / np.linalg.norm(n)
    return np.eye(3) - 2 * np.outer(n, n)


def _improper_group(name):
    """generators of Sampling::createSymFile (sampling.cpp:1328-1420), dihedral 2-fold on X"""
    from tests import synth
    inv = -np.eye(3)
    z = lambda n: [_rot_axis(n, (0, 0, 1))] if n > 1 else []
    if name == "ci": return _closure([inv])
    if name == "cs": return _closure([_mirror((0, 0, 1))])
    if name[0] in "cds" and name[1].isdigit():
        n = int("".join(ch for ch in name[1:] if ch.isdigit()))
        tail = name[-1]
        if name[0] == "s": return _closure(z(n // 2) + [inv])
        if name[0] == "c": return _closure(z(n) + [_mirror((0, 1, 0)) if tail == "v" else _mirror((0, 0, 1))])
        return _closure(z(n) + [_rot_axis(2, (1, 0, 0)), _mirror((1, 0, 0)) if tail == "v" else _mirror((0, 0, 1))])
    if name == "td": return _closure([_rot_axis(3, (0, 0, 1)), _rot_axis(2, (0, 0.816496, 0.577350)), _mirror((1.4142136, 2.4494897, 0))])
    if name == "th": return _closure([_rot_axis(3, (0, 0, 1)), _rot_axis(2, (0, -0.816496, -0.577350)), inv])
    if name == "oh": return _closure([_rot_axis(3, (1, 1, 1)), _rot_axis(4, (0, 0, 1)), _mirror((0, 1, 1))])
    G = _group("i" + name[1])
    return G + [-g for g in G]


def _in_improper_unit(name, v, eps):
    from tests import synth
    u = lambda x: np.array(x, float) / np.linalg.norm(x)
    rot = np.degrees(np.arctan2(v[1], v[0]))
    tilt = np.degrees(np.arccos(np.clip(v[2], -1, 1)))
    e = np.degrees(eps)
    wedge = lambda lo, hi, up=True: lo + e < rot < hi - e and (not up or tilt < 90 - e)
    if name in ("ci", "cs"): return tilt < 90 - e
    if name[0] in "cds" and name[1].isdigit():
        n = int("".join(ch for ch in name[1:] if ch.isdigit()))
        if name[0] == "s": return wedge(-360 / n, 360 / n)
        if name[0] == "c": return wedge(0, 180 / n, False) if name[-1] == "v" else wedge(-180 / n, 180 / n)
        return wedge(90, 180 / n + 90) if name[-1] == "v" else wedge(0, 180 / n)
    if name == "td": n = [u((-0.942809, 0, 0)), u((0.471405, 0.272165, 0.7698)), u((0, 0.471405, -0.666667))]
    elif name == "th": n = [u((-0.816496, 0, 0)), u((0.707107, 0.408248, -0.57735)), u((-0.408248, -0.707107, 0))]
    elif name == "oh":
        n = [u((0, -1, 1)), u((1, 1, 0)), u((-1, 1, 0))]
        if not wedge(90, 135): return False
    elif name in ("i1h", "i2h"):
        A = synth.euler_matrix(0, 90.0 if name == "i1h" else 0.0, 0)
        n = [A @ u((0, 1, 0)), A @ u((-0.4999999839058737, -0.8090170074556163, 0.3090169861701543)), A @ u((1, 0, 0))]
    else:
        sgn = 1.0 if name == "i3h" else -1.0
        A = synth.euler_matrix(0, sgn * 31.7174745559, 0)
        n = [sgn * (A @ u((0, 0, 1))), sgn * (A @ u((0.187592467856686, -0.303530987314591, -0.491123477863004))),
             sgn * (A @ u((0.187592467856686, 0.303530987314591, -0.491123477863004))), u((0, 1, 0))]
    return all(v @ k > eps for k in n)


@pytest.mark.parametrize("name,order", [("ci", 2), ("cs", 2), ("c1v", 2), ("c3v", 6), ("c4v", 8), ("c1h", 2), ("c5h", 10), ("s6", 6), ("s2", 2),
                                        ("d2v", 8), ("d3v", 12), ("d5v", 20), ("d2h", 8), ("d3h", 12), ("d6h", 24), ("td", 24), ("th", 24),
                                        ("oh", 48), ("i1h", 120), ("i2h", 120), ("i3h", 120), ("i4h", 120)])
def test_groups_with_improper_elements_fit_their_asymmetric_units(prog, tmp_path, name, order):
    """as for the cubic rotation groups above: (1) generators x asymmetric unit of the reference = a fundamental domain of the
    action d -> g d on directions; (2) the host program uses that group and that unit."""
    G = _improper_group(name)
    assert len(G) == order
    rng = np.random.default_rng(5)
    pts = rng.standard_normal((600, 3))
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    counts = [sum(_in_improper_unit(name, g @ p, 1e-9) for g in G) for p in pts]
    assert all(k == 1 for k in counts), (name, sorted(set(counts)))
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "5", "--sym", name,
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    M = np.loadtxt(str(tmp_path / "g_symmetry.txt")).reshape(-1, 3, 3)
    assert len(M) == order and np.allclose(M[0], np.eye(3))
    for m in M:
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-12)
        assert any(np.abs(m - g).max() < 2e-6 for g in G)          # (createSymFile's axes carry 6-7 digits)
    assert sum(np.linalg.det(m) < 0 for m in M) == order // 2
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "all.stk"), "--sampling_rate", "5", "--sym", "c1",
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    la, ra = xmipp_io.read_xmd(str(tmp_path / "all.doc"))
    lu, ru = xmipp_io.read_xmd(str(tmp_path / "g.doc"))
    allv = np.array([[float(x[la.index(k)]) for k in ("X", "Y", "Z")] for x in ra])
    unit = np.array([[float(x[lu.index(k)]) for k in ("X", "Y", "Z")] for x in ru])
    inside = np.array([_in_improper_unit(name, v, 1e-5) for v in allv])
    keys = {tuple(np.round(v, 6)) for v in unit}
    assert all(tuple(np.round(v, 6)) in keys for v in allv[inside])    # every interior direction was kept ...
    assert inside.sum() <= len(unit) <= inside.sum() + 2.2 * np.sqrt(len(allv))   # ... and only the boundary besides


@pytest.mark.parametrize("sym", ["c1", "d3", "i3h"])
def test_closer_sampling_points(prog, tmp_path, sym):
    """--closer_sampling_points (findClosestSamplingPoint, sampling.cpp:1991-2098): per experimental image the direction of the
    asymmetric unit closest to any of its symmetry mates.  Checked by brute force over the program's own direction list and group."""
    r = subprocess.run([prog, "-i", "none.vol", "-o", str(tmp_path / "g.stk"), "--sampling_rate", "3", "--sym", sym,
                        "--experimental_images", os.path.join(GOLD, "experimental_images.xmd"), "--closer_sampling_points",
                        "--only_create_sampling"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from tests import synth
    l, rows = xmipp_io.read_xmd(str(tmp_path / "g_closest_sampling_points.doc"))
    ld, rd = xmipp_io.read_xmd(str(tmp_path / "g.doc"))
    le, re_ = xmipp_io.read_xmd(os.path.join(GOLD, "experimental_images.xmd"))
    dirs = np.array([[float(x[ld.index(k)]) for k in ("X", "Y", "Z")] for x in rd])
    M = np.loadtxt(str(tmp_path / "g_symmetry.txt")).reshape(-1, 3, 3)
    assert len(rows) == len(re_) == 3
    for row, e in zip(rows, re_):
        assert row[l.index("image")] == e[le.index("image")]
        d = synth.euler_matrix(float(e[le.index("angleRot")]), float(e[le.index("angleTilt")]), 0.0)[2]
        mates = np.array([m.T @ d for m in M])
        dots = mates @ dirs.T                         # [mate, direction]
        w = int(row[l.index("ref")])
        assert dots[:, w].max() >= dots.max() - 1e-6   # (the docfile's unit vectors carry 6 decimals)
        assert int(row[l.index("neighbor")]) == w
        got = [float(row[l.index(k)]) for k in ("angleRot", "angleTilt", "anglePsi")]
        exp = [float(rd[w][ld.index(k)]) for k in ("angleRot", "angleTilt", "anglePsi")]
        assert np.allclose(got, exp, atol=2e-6)
        assert np.degrees(np.arccos(min(1.0, dots.max()))) < 2.2        # a 3-degree sampling: the closest one is within ~2
