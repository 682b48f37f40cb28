"""Seeded synthetic cryo-EM-like data shared by the tests, smoke() and bench.py.
(phantom = sum of 3-D Gaussians; projections by real-space rotation + sum.)"""
import numpy as np


def euler_matrix(rot, tilt, psi):
    """xmippCore Euler_angles2matrix (closed form: function_tests/test_geometry_main.cpp:46-65)."""
    a, b, g = np.radians([rot, tilt, psi])
    ca, cb, cg, sa, sb, sg = np.cos(a), np.cos(b), np.cos(g), np.sin(a), np.sin(b), np.sin(g)
    cc, cs, sc, ss = cb * ca, cb * sa, sb * ca, sb * sa
    return np.array([[cg * cc - sg * sa, cg * cs + sg * ca, -cg * sb],
                     [-sg * cc - cg * sa, -sg * cs + cg * ca, sg * sb],
                     [sc, ss, cb]])


def phantom(D, seed=1, nblobs=20):
    rng = np.random.default_rng(seed)
    z, y, x = np.mgrid[-(D // 2):D - D // 2, -(D // 2):D - D // 2, -(D // 2):D - D // 2].astype(np.float64)
    vol = np.zeros((D, D, D))
    for _ in range(nblobs):
        c = rng.uniform(-0.3 * D / 2 * 1.0, 0.3 * D / 2 * 1.0, 3) * 1.0
        s = rng.uniform(2.0, 5.0) * D / 64.0
        a = rng.uniform(0.5, 1.0)
        vol += a * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * s * s))
    return vol


def project(vol, rot, tilt, psi):
    from scipy import ndimage
    D = vol.shape[0]
    A = euler_matrix(rot, tilt, psi)
    c = np.array([D // 2] * 3, float)
    M = A.T[::-1, ::-1]
    r = ndimage.affine_transform(vol, M, offset=c - M @ c, order=3, mode="constant")
    return r.sum(0)


def fibonacci_directions(n):
    """(rot, tilt) in degrees, roughly uniform on the sphere."""
    i = np.arange(n) + 0.5
    tilt = np.degrees(np.arccos(1 - 2 * i / n))
    rot = np.degrees((np.pi * (1 + 5 ** 0.5) * i) % (2 * np.pi))
    return np.stack([rot, tilt], 1)


def random_angles(n, rng):
    rot = rng.uniform(0, 360, n)
    tilt = np.degrees(np.arccos(rng.uniform(-1, 1, n)))
    psi = rng.uniform(0, 360, n)
    return np.stack([rot, tilt, psi], 1)


def make_refs(vol, nrefs):
    dirs = fibonacci_directions(nrefs)
    return np.stack([project(vol, r, t, 0.0) for r, t in dirs]).astype(np.float32), dirs


def make_particles(refs, n, rng, snr=0.1, max_shift=3):
    """Random reference, in-plane rotation, mirror, integer shift, white noise."""
    from scipy import ndimage
    nrefs, D, _ = refs.shape
    out = np.empty((n, D, D), np.float32)
    truth = []
    for i in range(n):
        k = int(rng.integers(nrefs))
        ang = float(rng.uniform(0, 360))
        fl = int(rng.integers(2))
        img = ndimage.rotate(refs[k].astype(np.float64), ang, reshape=False, order=3, mode="constant")
        if fl:
            img = img[:, ::-1]
        sh = rng.integers(-max_shift, max_shift + 1, 2)
        img = np.roll(img, sh, (0, 1))
        sig = img.std()
        img = img + rng.standard_normal(img.shape) * sig / np.sqrt(snr)
        out[i] = img
        truth.append((k, ang, fl, int(sh[1]), int(sh[0])))
    return out, truth


def movie(N, Y, X, seed, max_step=1.5, noise=0.5, smooth=3.0, local=0.0):
    """A smooth random field drifting under the frame window plus white noise per frame. local > 0 adds a drift that grows
    with time and differs over the field (a dilation about the centre plus a shear, `local` px at the corners of the last frame):
    returns frames [N, Y, X] (float32), the global drift [N, 2] (x, y) and a function field(n, x, y) -> (dx, dy), the
    displacement of the content of frame n at movie position (x, y)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    base = ndimage.gaussian_filter(rng.standard_normal((Y + 64, X + 64)), smooth) * 10
    drift = np.cumsum(rng.uniform(-max_step, max_step, (N, 2)), 0)
    drift -= drift[0]

    def field(n, x, y):
        u, v = (np.asarray(x, float) / X - 0.5) * 2, (np.asarray(y, float) / Y - 0.5) * 2
        a = local * n / max(N - 1, 1)
        return drift[n, 0] + a * (0.7 * u + 0.3 * v), drift[n, 1] + a * (0.8 * v - 0.2 * u)

    yy, xx = np.mgrid[0:Y, 0:X].astype(float)
    frames = []
    for n in range(N):
        if local == 0.0:
            f = ndimage.shift(base, (-drift[n, 1], -drift[n, 0]), order=3, mode="wrap")[32:32 + Y, 32:32 + X]
        else:
            dx, dy = field(n, xx, yy)
            f = ndimage.map_coordinates(base, [yy + 32 + dy, xx + 32 + dx], order=3, mode="wrap")
        frames.append(f + noise * rng.standard_normal((Y, X)))
    return np.stack(frames).astype(np.float32), drift, field
