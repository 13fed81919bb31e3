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
