"""Python mirror of SIVO::ORBmatcher's Hamming primitives (reference include/orbslam/ORBmatcher.h:36-142,
src/orbslam/ORBmatcher.cc:37-39,78-104,1582-1596) over the C ABI."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # ORBmatcher.cc:37-39


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def descriptor_distance_matrix(a, b):
    """Dense DescriptorDistance: torch cuda uint8 (nA,32) x (nB,32) -> int32 (nA,nB); or numpy in/out."""
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        out = np.empty((a.shape[0], b.shape[0]), np.int32)
        check(lib().sivo_hamming_matrix(a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                                        out.ctypes.data_as(C.c_void_p)))
        return out
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.int32, device=a.device)
    check(lib().sivo_hamming_matrix_dev(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream()))
    return out


def argmin2(a, b, cand_off, cand_idx):
    """Best / second-best over per-query candidate lists (numpy host arrays)."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32); cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    n = a.shape[0]
    bi, bd, sd = (np.empty(n, np.int32) for _ in range(3))
    check(lib().sivo_hamming_argmin2(a.ctypes.data_as(C.c_void_p), n, b.ctypes.data_as(C.c_void_p), b.shape[0],
                                     cand_off.ctypes.data_as(C.c_void_p), cand_idx.ctypes.data_as(C.c_void_p),
                                     bi.ctypes.data_as(C.c_void_p), bd.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p)))
    return bi, bd, sd


def bruteforce(a, b):
    """Best / second-best of every row of a over all rows of b (cuda tensors)."""
    n = a.shape[0]
    bi, bd, sd = (torch.empty(n, dtype=torch.int32, device=a.device) for _ in range(3))
    check(lib().sivo_hamming_bruteforce_dev(a.data_ptr(), n, b.data_ptr(), b.shape[0], bi.data_ptr(), bd.data_ptr(),
                                            sd.data_ptr(), _stream()))
    return bi, bd, sd
