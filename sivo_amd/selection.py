"""Python mirror of SIVO's feature-selection helpers (reference include/sivo_helpers/sivo_helpers.hpp,
src/sivo_helpers/sivo_helpers.cpp:64-88,160-180,201-219) batched as Tracking.cc:934-1023 applies them."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .orb import KP_DTYPE


def entropy_gate(kps, depth, xyz, entropy, state_cov, fx, fy, bl, level_sigma2, th_entropy_reduction):
    """Host arrays (numpy).  Returns (mutual_information, entropy_reduction, accept)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float64); entropy = np.ascontiguousarray(entropy, np.float64)
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    cov = (C.c_double * 36)(*np.asarray(state_cov, np.float64).ravel())
    n = len(kps)
    mi = np.empty(n); red = np.empty(n); acc = np.empty(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().sivo_entropy_gate(n, p(kps), p(depth), p(xyz), p(entropy), entropy.shape[0], entropy.shape[1], cov, fx, fy, bl,
                                  p(ls2), len(ls2), th_entropy_reduction, p(mi), p(red), p(acc)))
    return mi, red, acc


def entropy_gate_map_dev(kps, depth, xyz, d_entropy, state_cov, fx, fy, bl, level_sigma2, th_entropy_reduction):
    """Host key arrays (numpy) against the entropy map the network left on the device (a cuda f64 tensor): the per-frame form
    (sivo_entropy_gate_map_dev).  Returns (mutual_information, entropy_reduction, accept)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float64)
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    cov = (C.c_double * 36)(*np.asarray(state_cov, np.float64).ravel())
    n = len(kps)
    mi = np.empty(n); red = np.empty(n); acc = np.empty(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().sivo_entropy_gate_map_dev(n, p(kps), p(depth), p(xyz), d_entropy.data_ptr(), d_entropy.shape[0], d_entropy.shape[1], cov,
                                          fx, fy, bl, p(ls2), len(ls2), th_entropy_reduction, p(mi), p(red), p(acc)))
    return mi, red, acc


def entropy_gate_dev(d_kps_u8, d_depth, d_xyz, d_entropy, state_cov, fx, fy, bl, level_sigma2, th, d_mi, d_red, d_acc):
    """Device-resident form: cuda tensors (kps as a uint8 view of SivoKeyPoint records); entropy is the f64 map
    BayesianSegNet.finalize() left in HBM."""
    import torch
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    cov = (C.c_double * 36)(*np.asarray(state_cov, np.float64).ravel())
    n = d_depth.shape[0]
    check(lib().sivo_entropy_gate_dev(n, d_kps_u8.data_ptr(), d_depth.data_ptr(), d_xyz.data_ptr(), d_entropy.data_ptr(),
                                      d_entropy.shape[0], d_entropy.shape[1], cov, fx, fy, bl, ls2.ctypes.data_as(C.c_void_p), len(ls2),
                                      th, d_mi.data_ptr(), d_red.data_ptr(), d_acc.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def check_semantics(kps, depth, xyz, entropy, confidence, classes, state_cov, fx, fy, bl, level_sigma2, th_entropy_reduction, th_confidence):
    """LocalMapping::CheckSemantics(..., compute_information=true) (reference LocalMapping.cc:474-538), batched.  Returns
    (mutual_information, entropy_reduction, detected_class) with detected_class = 255 (VOID) where a criterion fails."""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float64); entropy = np.ascontiguousarray(entropy, np.float64)
    confidence = np.ascontiguousarray(confidence, np.float64); classes = np.ascontiguousarray(classes, np.uint8)
    ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    cov = (C.c_double * 36)(*np.asarray(state_cov, np.float64).ravel())
    n = len(kps)
    mi = np.empty(n); red = np.empty(n); det = np.empty(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().sivo_check_semantics(n, p(kps), p(depth), p(xyz), p(entropy), p(confidence), p(classes), entropy.shape[0], entropy.shape[1],
                                     cov, fx, fy, bl, p(ls2), len(ls2), th_entropy_reduction, th_confidence, p(mi), p(red), p(det)))
    return mi, red, det
