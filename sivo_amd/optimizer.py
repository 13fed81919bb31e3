"""Python mirror of the per-edge linearisation behind SIVO::Optimizer (reference
src/orbslam/Optimizer.cc:318-409, 651-755) over the C ABI."""
import ctypes as C
import math

import numpy as np
import torch

from ._lib import Edge, check, lib

EDGE_DTYPE = np.dtype([("pose", np.int32), ("point", np.int32), ("stereo", np.int32), ("pad_", np.int32),
                       ("obs", np.float64, 3), ("inv_sigma2", np.float64)])
assert EDGE_DTYPE.itemsize == C.sizeof(Edge) == 48
TH_HUBER_MONO = math.sqrt(5.991)     # Optimizer.cc:647
TH_HUBER_STEREO = math.sqrt(7.815)   # Optimizer.cc:648


def linearize(poses, points, edges, intr, delta_mono=TH_HUBER_MONO, delta_stereo=TH_HUBER_STEREO):
    """Host arrays in/out (numpy)."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    intr = (C.c_double * 5)(*[float(v) for v in intr])
    nE = edges.shape[0]
    out = {"err": np.empty((nE, 3)), "Jx": np.empty((nE, 3, 3)), "Jp": np.empty((nE, 3, 6)), "chi2": np.empty(nE),
           "rho": np.empty(nE), "w": np.empty(nE), "depth_ok": np.empty(nE, np.uint8)}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().sivo_ba_linearize(p(poses), poses.shape[0], p(points), points.shape[0], p(edges), nE, intr,
                                  delta_mono, delta_stereo, p(out["err"]), p(out["Jx"]), p(out["Jp"]), p(out["chi2"]),
                                  p(out["rho"]), p(out["w"]), p(out["depth_ok"])))
    return out


def linearize_dev(d_poses, d_points, d_edges_u8, n_edges, intr, out, delta_mono=TH_HUBER_MONO,
                  delta_stereo=TH_HUBER_STEREO):
    """Device-resident: d_edges_u8 is a cuda uint8 tensor holding n_edges SivoEdge records; `out` a dict of
    preallocated cuda tensors (err, Jx, Jp, chi2, rho, w, depth_ok)."""
    intr = (C.c_double * 5)(*[float(v) for v in intr])
    g = lambda k: out[k].data_ptr() if out.get(k) is not None else None
    check(lib().sivo_ba_linearize_dev(d_poses.data_ptr(), d_points.data_ptr(), d_edges_u8.data_ptr(), n_edges, intr,
                                      delta_mono, delta_stereo, g("err"), g("Jx"), g("Jp"), g("chi2"), g("rho"), g("w"),
                                      g("depth_ok"), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def _intr(intr):
    return (C.c_double * 5)(*[float(v) for v in intr])


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def ba_optimize(poses, fixed, points, edges, intr, iterations, level=None, robust=None,
                delta_mono=float(np.sqrt(np.float32(5.991))), delta_stereo=float(np.sqrt(np.float32(7.815))), stop=None):
    """One g2o optimize(iterations) call (Optimizer::BundleAdjustment, reference Optimizer.cc:49-271) on arrays."""
    poses = np.array(poses, np.float64).reshape(-1, 12).copy(); points = np.array(points, np.float64).reshape(-1, 3).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8); edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    nE = edges.shape[0]
    level = None if level is None else np.ascontiguousarray(level, np.uint8)
    robust = None if robust is None else np.ascontiguousarray(robust, np.uint8)
    err = np.zeros((nE, 3)); hpp = np.zeros((int((fixed == 0).sum()), 6, 6)); n = C.c_int(0); tr = C.c_int(0)
    check(lib().sivo_ba_optimize(_vp(poses), _vp(fixed), poses.shape[0], _vp(points), points.shape[0], _vp(edges), nE,
                                 _intr(intr), delta_mono, delta_stereo, _vp(level), _vp(robust), iterations,
                                 C.addressof(stop) if stop is not None else None, _vp(err), _vp(hpp), C.byref(n), C.byref(tr)))
    return {"poses": poses, "points": points, "err": err, "hpp": hpp, "iterations": n.value, "trials": tr.value}


def local_ba(poses, fixed, points, edges, intr, cov_pose=-1, stop=None):
    """Optimizer::LocalBundleAdjustment (reference Optimizer.cc:757-926) on arrays; `stop` is a ctypes c_uint8 (the reference's bool *pbStopFlag) or None."""
    poses = np.array(poses, np.float64).reshape(-1, 12).copy(); points = np.array(points, np.float64).reshape(-1, 3).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8); edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    nE = edges.shape[0]
    outlier = np.zeros(nE, np.uint8); cov = np.zeros((6, 6)); ok = C.c_int(0); n = C.c_int(0); tr = C.c_int(0)
    check(lib().sivo_local_ba(_vp(poses), _vp(fixed), poses.shape[0], _vp(points), points.shape[0], _vp(edges), nE,
                              _intr(intr), C.addressof(stop) if stop is not None else None, _vp(outlier), cov_pose,
                              _vp(cov), C.byref(ok), C.byref(n), C.byref(tr)))
    return {"poses": poses, "points": points, "outlier": outlier, "cov": cov, "cov_ok": bool(ok.value),
            "iterations": n.value, "trials": tr.value}


def pose_optimize(pose0, points, edges, intr):
    """Optimizer::PoseOptimization (reference Optimizer.cc:273-491) on arrays."""
    pose0 = np.ascontiguousarray(pose0, np.float64).reshape(12); points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    nE = edges.shape[0]
    outlier = np.zeros(nE, np.uint8); pose = np.empty(12); cov = np.zeros((6, 6)); chi2 = np.zeros(nE)
    ok = C.c_int(0); inl = C.c_int(0); n = C.c_int(0); tr = C.c_int(0)
    check(lib().sivo_pose_optimize(_vp(pose0), _vp(points), points.shape[0], _vp(edges), nE, _intr(intr), _vp(outlier),
                                   _vp(pose), _vp(cov), C.byref(ok), _vp(chi2), C.byref(inl), C.byref(n), C.byref(tr)))
    return {"pose": pose, "outlier": outlier, "cov": cov, "cov_ok": bool(ok.value), "chi2": chi2, "inliers": inl.value,
            "iterations": n.value, "trials": tr.value}
