"""Shared by tests/test_pin_orb.py and tests/golden/make_orb_reference.py: the cases, and the ctypes binding of
oracle/_ref/libref_orb.so — the reference's OWN src/orbslam/ORBextractor.cc compiled by `make -C oracle ref` (OpenCV
primitives under it: the restatements of oracle/orb_oracle.c)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_orb.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "orb_reference.json")
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                     ("octave", np.int32), ("class_id", np.int32)])


def images():
    from conftest import synthetic_frame
    out = {"synthetic-7": synthetic_frame(7), "synthetic-21": synthetic_frame(21), "small-120x160": synthetic_frame(3, 120, 160),
           "odd-241x517": synthetic_frame(5, 241, 517)}
    p = os.path.join(ROOT, "tests", "golden", "frame_bgr_352x1024.npy")
    if os.path.exists(p):
        frame = np.load(p)
        out["kitti-crop"] = np.ascontiguousarray(frame[..., 0])
        out["kitti-crop-green"] = np.ascontiguousarray(frame[..., 1])
    rng = np.random.default_rng(11)
    out["noise-200x300"] = rng.integers(0, 256, (200, 300), dtype=np.uint8)          # corners everywhere: the octree has to cut
    out["flat-200x300"] = np.full((200, 300), 77, np.uint8)                             # no corner at all
    return out


# (nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST): config/*.yaml of the reference uses 2000 / 1.2 / 8 / 20 / 7
CONFIGS = [(2000, 1.2, 8, 20, 7), (500, 1.2, 3, 20, 7), (1000, 1.5, 4, 30, 5), (300, 1.2, 1, 12, 7)]


def cases():
    for name, img in images().items():
        for cfg in CONFIGS:
            yield f"{name}|{cfg[0]}/{cfg[1]}/{cfg[2]}/{cfg[3]}/{cfg[4]}", img, cfg


def reference_extract(gray, cfg, mode=0):
    """mode 0: heap addresses grow with creation order (the order the oracle and the device fix for the octree's
    size ties); mode 1: the process's malloc.  Returns keys, descriptors, pyramid levels, the four scale tables."""
    lib = C.CDLL(REF_LIB)
    n, s, l, ini, mn = cfg
    cap = 2 * n + 64
    k = np.zeros(cap, KP_DTYPE); d = np.zeros((cap, 32), np.uint8)
    g = np.ascontiguousarray(gray, np.uint8)
    lev = np.zeros(g.size * 4, np.uint8); lr = np.zeros(l, np.int32); lc = np.zeros(l, np.int32); tab = np.zeros(4 * l, np.float32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    cnt = lib.ref_orb_extract(n, C.c_float(s), l, ini, mn, mode, vp(g), g.shape[0], g.shape[1], g.strides[0], vp(k), vp(d), cap, vp(lev),
                              lev.size, vp(lr), vp(lc), vp(tab))
    assert 0 <= cnt <= cap
    levels, off = [], 0
    for i in range(l):
        levels.append(lev[off:off + lr[i] * lc[i]].reshape(lr[i], lc[i]).copy()); off += lr[i] * lc[i]
    return k[:cnt].copy(), d[:cnt].copy(), levels, tab.reshape(4, l)


def digest(keys, desc, levels=None):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(keys).tobytes()); h.update(np.ascontiguousarray(desc).tobytes())
    out = {"n": int(len(keys)), "sha256": h.hexdigest()}
    if levels is not None:
        hl = hashlib.sha256()
        for lv in levels:
            hl.update(np.ascontiguousarray(lv).tobytes())
        out["levels_sha256"] = hl.hexdigest()
    return out


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)
