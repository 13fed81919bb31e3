"""Shared by tests/test_pin_frame.py and tests/golden/make_frame_reference.py: the stereo scenes and the ctypes binding of
oracle/_ref/libref_frame.so — the reference's OWN src/orbslam/Frame.cc (+ ORBextractor.cc) compiled by
`make -C oracle ref`; see oracle/ref_frame_wrap.cpp."""
import ctypes as C
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_frame.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "frame_reference.json")
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                     ("octave", np.int32), ("class_id", np.int32)])
FX, FY, CX, CY, BF = 718.856, 718.856, 607.1928, 185.2157, 386.1448          # KITTI 00-02 (config/KITTI00-02.yaml)
TERRAIN = 8


class _In(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("classes", C.c_void_p),
                ("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("ini", C.c_int32), ("min", C.c_int32),
                ("scale", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("n_queries", C.c_int32), ("qx", C.c_void_p), ("qy", C.c_void_p), ("qr", C.c_void_p), ("qmin", C.c_void_p), ("qmax", C.c_void_p),
                ("Tcw", C.c_void_p), ("n_points", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p),
                ("max_dist", C.c_void_p), ("cos_limit", C.c_float)]


class _Out(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n_left", C.c_int32), ("n_right", C.c_int32), ("n_semantic", C.c_int32),
                ("keys", C.c_void_p), ("desc", C.c_void_p), ("right", C.c_void_p), ("depth", C.c_void_p), ("unprojected", C.c_void_p),
                ("query_capacity", C.c_int32), ("query_off", C.c_void_p), ("query_idx", C.c_void_p), ("in_view", C.c_void_p),
                ("track", C.c_void_p), ("track_level", C.c_void_p), ("bounds", C.c_float * 4), ("grid_inv", C.c_float * 2)]


def scenes():
    """name -> (left, right, classes, extractor config)."""
    from conftest import synthetic_stereo
    out = {}
    for seed, disp in ((7, 8), (21, 23)):
        left, right = synthetic_stereo(seed, disparity=disp)
        out[f"synthetic-{seed}-d{disp}"] = (left, right)
    p = os.path.join(ROOT, "tests", "golden", "frame_bgr_352x1024.npy")
    if os.path.exists(p):
        g = np.ascontiguousarray(np.load(p)[..., 0])
        rng = np.random.default_rng(3)
        right = np.concatenate([g[:, 12:], np.repeat(g[:, -1:], 12, axis=1)], axis=1).astype(np.float64) + rng.normal(0, 1.5, g.shape)
        out["kitti-crop-d12"] = (g, np.clip(np.rint(right), 0, 255).astype(np.uint8))
    res = {}
    for name, (left, right) in out.items():
        rng = np.random.default_rng(len(name))
        classes = np.full(left.shape, 2, np.uint8)                              # BUILDING everywhere ...
        for _ in range(14):                                                     # ... with blobs of every class incl. dynamic ones and VOID
            x0, x1 = sorted(rng.integers(0, left.shape[1], 2)); y0, y1 = sorted(rng.integers(0, left.shape[0], 2))
            classes[y0:y1 + 1, x0:x1 + 1] = rng.choice([0, 1, 4, 7, 8, 9, 10, 11, 12, 13, 255])
        for cfg in ((2000, 1.2, 8, 20, 7), (600, 1.2, 4, 20, 7)):
            res[f"{name}|{cfg[0]}/{cfg[2]}"] = (left, right, classes, cfg)
    return res


def probes(rows, cols, seed):
    """Window queries (incl. empty windows, windows over the border, level filters) and map points for isInFrustum."""
    rng = np.random.default_rng(seed)
    nq = 400
    qx = rng.uniform(-30, cols + 30, nq).astype(np.float32); qy = rng.uniform(-30, rows + 30, nq).astype(np.float32)
    qr = rng.choice([1.5, 4, 10, 25, 60], nq).astype(np.float32)
    qmin = rng.choice([-1, -1, 0, 1, 2, 3], nq).astype(np.int32)
    qmax = np.where(rng.random(nq) < 0.5, -1, qmin + rng.integers(0, 3, nq)).astype(np.int32)
    th = np.deg2rad(2.0)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 0] = Tcw[2, 2] = np.cos(th); Tcw[0, 2] = np.sin(th); Tcw[2, 0] = -np.sin(th); Tcw[:3, 3] = (0.2, -0.05, 0.4)
    npts = 600
    z = rng.uniform(-2, 60, npts); u = rng.uniform(-200, cols + 200, npts); v = rng.uniform(-100, rows + 100, npts)
    pos = np.stack([(u - CX) * z / FX, (v - CY) * z / FY, z], 1).astype(np.float32)
    normal = pos / np.maximum(np.linalg.norm(pos, axis=1, keepdims=True), 1e-6)
    normal[rng.random(npts) < 0.15] *= -1
    normal = (normal + rng.normal(0, 0.25, normal.shape)).astype(np.float32)
    dist = np.linalg.norm(pos, axis=1)
    max_dist = (dist * rng.uniform(0.5, 4.0, npts)).astype(np.float32); min_dist = (max_dist / 3.58).astype(np.float32)
    return dict(qx=qx, qy=qy, qr=qr, qmin=qmin, qmax=qmax, Tcw=Tcw, pos=pos, normal=np.ascontiguousarray(normal), min_dist=min_dist,
                max_dist=max_dist, cos_limit=0.5)


def reference_frame(left, right, classes, cfg, pr):
    lib = C.CDLL(REF_LIB)
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8); classes = np.ascontiguousarray(classes, np.uint8)
    n, s, l, ini, mn = cfg
    cap = 2 * n + 64
    vp = lambda a: C.c_void_p(a.ctypes.data)
    nq, npts = len(pr["qx"]), len(pr["pos"])
    Tcw = np.ascontiguousarray(pr["Tcw"], np.float32)
    i = _In(vp(left), vp(right), left.shape[0], left.shape[1], vp(classes), n, l, ini, mn, s, FX, FY, CX, CY, BF, nq, vp(pr["qx"]), vp(pr["qy"]),
            vp(pr["qr"]), vp(pr["qmin"]), vp(pr["qmax"]), vp(Tcw), npts, vp(pr["pos"]), vp(pr["normal"]), vp(pr["min_dist"]), vp(pr["max_dist"]),
            pr["cos_limit"])
    keys = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); uR = np.zeros(cap, np.float32); depth = np.zeros(cap, np.float32)
    unp = np.zeros((cap, 3), np.float32); qoff = np.zeros(nq + 1, np.int32); qidx = np.zeros(nq * 400, np.int32)
    inview = np.zeros(npts, np.uint8); track = np.zeros((npts, 4), np.float32); level = np.zeros(npts, np.int32)
    o = _Out(cap, 0, 0, 0, vp(keys), vp(desc), vp(uR), vp(depth), vp(unp), qidx.size, vp(qoff), vp(qidx), vp(inview), vp(track), vp(level))
    rc = lib.ref_frame_build(C.byref(i), C.byref(o))
    assert rc == 0, rc
    m = o.n_semantic
    return dict(n_left=o.n_left, n_right=o.n_right, keys=keys[:m].copy(), desc=desc[:m].copy(), right=uR[:m].copy(), depth=depth[:m].copy(),
                unprojected=unp[:m].copy(), query_off=qoff.copy(), query_idx=qidx[:qoff[-1]].copy(), in_view=inview, track=track, level=level,
                bounds=np.array(o.bounds[:], np.float32), grid_inv=np.array(o.grid_inv[:], np.float32))


def digest(d, fields):
    out = {"n_semantic": int(len(d["keys"]))}
    for f in fields:
        out[f] = hashlib.sha256(np.ascontiguousarray(d[f]).tobytes()).hexdigest()[:32]
    return out


FRAME_FIELDS = ("keys", "desc", "right", "depth", "query_off", "query_idx")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)
