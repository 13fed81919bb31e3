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


def argmin2(a, b, cand_off, cand_idx, want_second_idx=False):
    """Best / second-best over per-query candidate lists (numpy host arrays).  want_second_idx: also return the row that
    holds the second-best distance (for the same-octave ratio rule of ORBmatcher.cc:117-119)."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32); cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    n = a.shape[0]
    bi, bd, sd, si = (np.empty(n, np.int32) for _ in range(4))
    check(lib().sivo_hamming_argmin2(a.ctypes.data_as(C.c_void_p), n, b.ctypes.data_as(C.c_void_p), b.shape[0],
                                     cand_off.ctypes.data_as(C.c_void_p), cand_idx.ctypes.data_as(C.c_void_p),
                                     bi.ctypes.data_as(C.c_void_p), bd.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p),
                                     si.ctypes.data_as(C.c_void_p)))
    return (bi, bd, sd, si) if want_second_idx else (bi, bd, sd)


def bruteforce(a, b):
    """Best / second-best of every row of a over all rows of b (cuda tensors)."""
    n = a.shape[0]
    bi, bd, sd = (torch.empty(n, dtype=torch.int32, device=a.device) for _ in range(3))
    check(lib().sivo_hamming_bruteforce_dev(a.data_ptr(), n, b.data_ptr(), b.shape[0], bi.data_ptr(), bd.data_ptr(),
                                            sd.data_ptr(), _stream()))
    return bi, bd, sd


# ---------------------------------------------------------------------------------------------------------------------
# Guided matching (the Search* / Fuse members, reference ORBmatcher.h:44-120) on arrays, through the C ABI.
# Host numpy arrays in and out; the work runs on the GPU (sivo_amd/csrc/search.hip).
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])
QUERY_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("radius", np.float32), ("lvl_lo", np.int32), ("lvl_hi", np.int32),
                        ("ur", np.float32), ("gate", np.float32), ("angle", np.float32), ("flags", np.int32)])
Q_VALID, Q_BLOCKS, Q_STEREO = 1, 2, 4


def _a(x, dt):
    return np.ascontiguousarray(x, dt)


def _p(x):
    return x.ctypes.data_as(C.c_void_p) if x is not None else None


class MatchFrame:
    """What the matcher reads of a Frame / KeyFrame (mvKeysSemantic, mvRight, mDescriptorsSemantic, bounds, scale tables),
    resident on the GPU with its 64 x 48 grid (Frame.cc:205-221)."""

    def __init__(self, keys, u_right, desc, bounds, scale, sigma2, inv_sigma2, device=0):
        self.keys = _a(keys, KP_DTYPE); self.n = len(self.keys)
        self.u_right = None if u_right is None else _a(u_right, np.float32)
        self.desc = _a(desc, np.uint8).reshape(self.n, 32)
        self.scale, self.sigma2, self.inv_sigma2 = _a(scale, np.float32), _a(sigma2, np.float32), _a(inv_sigma2, np.float32)
        h = C.c_void_p()
        self._L = lib()          # the library this object lives in
        check(self._L.sivo_mframe_create(_p(self.keys), self.n, _p(self.u_right), _p(self.desc), *[float(b) for b in bounds],
                                       _p(self.scale), _p(self.sigma2), _p(self.inv_sigma2), len(self.scale), device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._L.sivo_mframe_destroy(h)
            except Exception:
                pass
            self._h = None

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        """Frame::GetFeaturesInArea (Frame.cc:326-390)."""
        out = np.empty(self.n + 1, np.int32); n = C.c_int32(0)
        check(self._L.sivo_mframe_features_in_area(self._h, x, y, r, min_level, max_level, _p(out), out.size, C.byref(n)))
        return out[:n.value].copy()


def search(train, queries, query_desc, rule, cand_begin=None, cand_end=None, cand_idx=None, blocked=None):
    """The generic engine (sivo_search).  rule: dict of SivoSearchRule fields."""
    from ._lib import SearchRule
    q = _a(queries, QUERY_DTYPE); nq = len(q)
    qd = _a(query_desc, np.uint8)
    r = SearchRule()
    for k, v in rule.items():
        if k == "F12":
            for i, x in enumerate(np.asarray(v, np.float32).ravel()):
                r.F12[i] = float(x)
        else:
            setattr(r, k, v)
    cb = ce = ci = None; nc = 0
    if cand_idx is not None:
        cb, ce, ci = _a(cand_begin, np.int32), _a(cand_end, np.int32), _a(cand_idx, np.int32); nc = len(ci)
    bl = None if blocked is None else _a(blocked, np.uint8)
    mq = np.empty(nq, np.int32); mt = np.empty(train.n, np.int32); bd = np.empty(nq, np.int32); sd = np.empty(nq, np.int32)
    nm, rounds = C.c_int32(0), C.c_int32(0)
    check(train._L.sivo_search(train._h, _p(q), _p(qd), nq, _p(cb), _p(ce), _p(ci), nc, C.byref(r), _p(bl), _p(mq), _p(mt), _p(bd), _p(sd),
                            C.byref(nm), C.byref(rounds)))
    return {"match_query": mq, "match_train": mt, "best_dist": bd, "second_dist": sd, "n_matches": nm.value, "rounds": rounds.value}


def search_by_projection_mappoints(F, track_in_view, px, py, pxr, level, view_cos, mp_desc, mp_obs, th, nn_ratio, occ_obs):
    """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.cc:44-127)."""
    occ = _a(occ_obs, np.int32).copy(); match = np.empty(F.n, np.int32); nm = C.c_int32(0)
    check(F._L.sivo_search_by_projection_mappoints(F._h, len(px), _p(_a(track_in_view, np.uint8)), _p(_a(px, np.float32)),
                                                    _p(_a(py, np.float32)), _p(_a(pxr, np.float32)), _p(_a(level, np.int32)),
                                                    _p(_a(view_cos, np.float32)), _p(_a(mp_desc, np.uint8)), _p(_a(mp_obs, np.int32)),
                                                    th, nn_ratio, _p(occ), _p(match), C.byref(nm)))
    return nm.value, match, occ


def search_by_projection_frame(Cur, valid, u, v, inv_z, last_octave, last_angle, mp_desc, mp_obs, th, forward, backward, bf,
                               check_ori, occ_obs):
    """ORBmatcher::SearchByProjection(Frame &Current, const Frame &Last, th, bMono) (ORBmatcher.cc:1278-1418)."""
    occ = _a(occ_obs, np.int32).copy(); match = np.empty(Cur.n, np.int32); nm = C.c_int32(0)
    check(Cur._L.sivo_search_by_projection_frame(Cur._h, len(u), _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                                _p(_a(inv_z, np.float32)), _p(_a(last_octave, np.int32)), _p(_a(last_angle, np.float32)),
                                                _p(_a(mp_desc, np.uint8)), _p(_a(mp_obs, np.int32)), th, int(forward), int(backward), bf,
                                                int(check_ori), _p(occ), _p(match), C.byref(nm)))
    return nm.value, match, occ


def search_by_projection_reloc(Cur, valid, u, v, pred_level, kf_angle, mp_desc, th, orb_dist, check_ori, occupied):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1420-1543)."""
    occ = _a(occupied, np.uint8).copy(); match = np.empty(Cur.n, np.int32); nm = C.c_int32(0)
    check(Cur._L.sivo_search_by_projection_reloc(Cur._h, len(u), _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                                _p(_a(pred_level, np.int32)), _p(_a(kf_angle, np.float32)), _p(_a(mp_desc, np.uint8)), th,
                                                int(orb_dist), int(check_ori), _p(occ), _p(match), C.byref(nm)))
    return nm.value, match, occ


def search_by_projection_kf(KF, valid, u, v, pred_level, mp_desc, th, matched):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:286-399)."""
    m = _a(matched, np.uint8).copy(); match = np.empty(KF.n, np.int32); nm = C.c_int32(0)
    check(KF._L.sivo_search_by_projection_kf(KF._h, len(u), _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                             _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)), int(th), _p(m), _p(match),
                                             C.byref(nm)))
    return nm.value, match, m


def fuse(KF, valid, u, v, ur, pred_level, mp_desc, th, scw_variant):
    """ORBmatcher::Fuse (ORBmatcher.cc:787-929 / :931-1053) up to the choice of the keypoint."""
    n = len(u)
    bi = np.empty(n, np.int32); bd = np.empty(n, np.int32); nf = C.c_int32(0)
    check(KF._L.sivo_fuse(KF._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                          _p(_a(ur, np.float32)) if ur is not None else None, _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)),
                          th, int(scw_variant), _p(bi), _p(bd), C.byref(nf)))
    return nf.value, bi, bd


def search_by_sim3_dir(KF, valid, u, v, pred_level, mp_desc, th):
    """One direction of ORBmatcher::SearchBySim3 (ORBmatcher.cc:1102-1176)."""
    out = np.empty(len(u), np.int32)
    check(KF._L.sivo_search_by_sim3_dir(KF._h, len(u), _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                        _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)), th, _p(out)))
    return out


def sim3_agree(m1, m2):
    """ORBmatcher.cc:1254-1273: keep i1 -> idx2 iff idx2 -> i1."""
    m1 = np.asarray(m1); m2 = np.asarray(m2)
    out = np.full(len(m1), -1, np.int32)
    ok = m1 >= 0
    ok[ok] = m2[m1[ok]] == np.nonzero(ok)[0]
    out[ok] = m1[ok]
    return int(ok.sum()), out


def search_by_bow_kf_frame(off1, idx1, off2, idx2, kf_valid, keys_kf, desc_kf, F, nn_ratio, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:161-284) on node lists."""
    keys_kf = _a(keys_kf, KP_DTYPE)
    match_f = np.empty(F.n, np.int32); nm = C.c_int32(0)
    off1 = _a(off1, np.int32)
    check(F._L.sivo_search_by_bow_kf_frame(len(off1) - 1, _p(off1), _p(_a(idx1, np.int32)), _p(_a(off2, np.int32)), _p(_a(idx2, np.int32)),
                                            _p(_a(kf_valid, np.uint8)), _p(keys_kf), _p(_a(desc_kf, np.uint8)), len(keys_kf), F._h,
                                            nn_ratio, int(check_ori), _p(match_f), C.byref(nm)))
    return nm.value, match_f


def search_by_bow_kf_kf(off1, idx1, off2, idx2, valid1, keys1, desc1, valid2, KF2, nn_ratio, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:508-629) on node lists."""
    keys1 = _a(keys1, KP_DTYPE)
    m12 = np.empty(len(keys1), np.int32); nm = C.c_int32(0)
    off1 = _a(off1, np.int32)
    check(KF2._L.sivo_search_by_bow_kf_kf(len(off1) - 1, _p(off1), _p(_a(idx1, np.int32)), _p(_a(off2, np.int32)), _p(_a(idx2, np.int32)),
                                         _p(_a(valid1, np.uint8)), _p(keys1), _p(_a(desc1, np.uint8)), len(keys1), _p(_a(valid2, np.uint8)),
                                         KF2._h, nn_ratio, int(check_ori), _p(m12), C.byref(nm)))
    return nm.value, m12


def search_for_triangulation(off1, idx1, off2, idx2, keys1, ur1, has_mp1, desc1, KF2, has_mp2, F12, ex, ey, only_stereo, check_ori):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:631-785) on node lists."""
    keys1 = _a(keys1, KP_DTYPE)
    m12 = np.empty(len(keys1), np.int32); nm = C.c_int32(0)
    off1 = _a(off1, np.int32)
    check(KF2._L.sivo_search_for_triangulation(len(off1) - 1, _p(off1), _p(_a(idx1, np.int32)), _p(_a(off2, np.int32)), _p(_a(idx2, np.int32)),
                                              _p(keys1), _p(_a(ur1, np.float32)) if ur1 is not None else None, _p(_a(has_mp1, np.uint8)),
                                              _p(_a(desc1, np.uint8)), len(keys1), KF2._h, _p(_a(has_mp2, np.uint8)),
                                              _p(_a(F12, np.float32)), ex, ey, int(only_stereo), int(check_ori), _p(m12), C.byref(nm)))
    return nm.value, m12
