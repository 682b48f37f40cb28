"""Minimal writers/readers of the on-disk formats the two programs exchange with Scipion
(SURVEY.md Appendix A): Spider stacks/volumes and XMIPP_STAR_1 metadata."""
import numpy as np


def _spider_header(x, y, z, iform, istack, maxim, imgnum):
    lenbyt = x * 4
    labrec = (1024 + lenbyt - 1) // lenbyt
    labbyt = labrec * lenbyt
    h = np.zeros(labbyt // 4, np.float32)
    h[0], h[1], h[2], h[4], h[11] = z, y, labrec + y * z, iform, x
    h[12], h[21], h[22], h[23], h[25], h[26] = labrec, labbyt, lenbyt, istack, maxim, imgnum
    return h


def write_stack(path, imgs):
    imgs = np.ascontiguousarray(imgs, np.float32)
    n, y, x = imgs.shape
    with open(path, "wb") as f:
        f.write(_spider_header(x, y, 1, 1, 2, n, 0).tobytes())
        for i in range(n):
            f.write(_spider_header(x, y, 1, 1, 0, 0, i + 1).tobytes())
            f.write(imgs[i].tobytes())


def write_volume(path, vol):
    vol = np.ascontiguousarray(vol, np.float32)
    z, y, x = vol.shape
    with open(path, "wb") as f:
        f.write(_spider_header(x, y, z, 3, 0, 0, 0).tobytes())
        f.write(vol.tobytes())


def read_stack(path):
    raw = np.fromfile(path, np.float32)
    y, x, labbyt, n = int(raw[1]), int(raw[11]), int(raw[21]), int(raw[25])
    hw = labbyt // 4
    per = hw + x * y
    body = raw[hw:hw + n * per].reshape(n, per)
    return body[:, hw:].reshape(n, y, x).copy()


def read_volume(path):
    raw = np.fromfile(path, np.float32)
    z, y, x, labbyt = int(abs(raw[0])), int(raw[1]), int(raw[11]), int(raw[21])
    return raw[labbyt // 4:labbyt // 4 + x * y * z].reshape(z, y, x)


def write_xmd(path, blocks):
    """blocks: list of (name, labels, rows) ; rows of values (str/float/int)."""
    with open(path, "w") as f:
        f.write("# XMIPP_STAR_1 * \n# \n")
        for name, labels, rows in blocks:
            f.write(f"data_{name}\nloop_\n")
            for l in labels:
                f.write(f" _{l}\n")
            for r in rows:
                f.write(" " + " ".join((f"'{v}'" if isinstance(v, str) and " " in v else str(v)) for v in r) + " \n")


def read_xmd(path, block=None):
    labels, rows, inb = [], [], False
    for line in open(path):
        t = line.strip()
        if not t or t[0] == "#":
            continue
        if t.startswith("data_"):
            if inb:
                break
            inb = block is None or t[5:] == block
            continue
        if not inb or t == "loop_":
            continue
        if t[0] == "_" and not rows:
            labels.append(t.split()[0][1:])
            continue
        rows.append(_split_row(t))
    return labels, rows


def _split_row(t):
    """whitespace-separated values; a value in single quotes may contain blanks (e.g. neighbour lists)"""
    out, i = [], 0
    while i < len(t):
        if t[i].isspace():
            i += 1
        elif t[i] == "'":
            j = t.find("'", i + 1)
            j = len(t) if j < 0 else j
            out.append(t[i + 1:j])
            i = j + 1
        else:
            j = i
            while j < len(t) and not t[j].isspace():
                j += 1
            out.append(t[i:j])
            i = j
    return out


def write_mrcs(path, imgs, mode=2):
    """MRC2014 stack (.mrcs): mode 0 int8, 1 int16, 2 float32, 6 uint16; imgs [n, y, x]"""
    dt = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16}[mode]
    a = np.ascontiguousarray(imgs).astype(dt)
    n, y, x = a.shape
    h = np.zeros(256, np.int32)
    h[0], h[1], h[2], h[3] = x, y, n, mode
    h[7], h[8], h[9] = x, y, n
    h[16], h[17], h[18] = 1, 2, 3
    h[52] = int.from_bytes(b"MAP ", "little")
    h[53] = 0x00004444
    with open(path, "wb") as f:
        f.write(h.tobytes())
        f.write(a.tobytes())
