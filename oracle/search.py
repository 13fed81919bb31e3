"""ctypes front-end of oracle/search_oracle.c — TEST INFRASTRUCTURE ONLY.

Sequential restatements of the ORBmatcher Search* / Fuse routines (reference src/orbslam/ORBmatcher.cc) and of the
frame grid (reference src/orbslam/Frame.cc:205-221, 326-390) on plain arrays.  Parity pinned against the reference's own ORBmatcher.cc compiled into oracle/_ref (tests/cpp/pin_matcher.cpp).
"""
import ctypes as C

import numpy as np

from .oracle import KP_DTYPE, lib

_vp = C.c_void_p


def _a(x, dt):
    return np.ascontiguousarray(x, dt)


def _p(x):
    return x.ctypes.data_as(_vp) if x is not None else None


class Frame:
    """The part of Frame / KeyFrame the matcher reads."""

    def __init__(self, keys, u_right, desc, bounds, scale, sigma2, inv_sigma2):
        self.keys = _a(keys, KP_DTYPE); self.n = len(self.keys)
        self.u_right = None if u_right is None else _a(u_right, np.float32)
        self.desc = _a(desc, np.uint8).reshape(self.n, 32)
        self.bounds = tuple(float(b) for b in bounds)
        self.scale, self.sigma2, self.inv_sigma2 = _a(scale, np.float32), _a(sigma2, np.float32), _a(inv_sigma2, np.float32)
        f = lib().orc_frame_create
        f.restype = _vp
        self._h = _vp(f(_p(self.keys), self.n, _p(self.u_right), _p(self.desc), *(C.c_float(b) for b in self.bounds),
                        _p(self.scale), _p(self.sigma2), _p(self.inv_sigma2), len(self.scale)))

    def __del__(self):
        try:
            lib().orc_frame_destroy(self._h)
        except Exception:
            pass

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.empty(self.n + 1, np.int32)
        n = lib().orc_frame_features_in_area(self._h, C.c_float(x), C.c_float(y), C.c_float(r), min_level, max_level, _p(out), out.size)
        return out[:n].copy()


def search_by_projection_mappoints(F, track_in_view, px, py, pxr, level, view_cos, mp_desc, mp_obs, th, nn_ratio, occ_obs):
    n = len(px)
    occ = _a(occ_obs, np.int32).copy(); match = np.empty(F.n, np.int32)
    nm = lib().orc_search_by_projection_mappoints(F._h, n, _p(_a(track_in_view, np.uint8)), _p(_a(px, np.float32)), _p(_a(py, np.float32)),
                                                  _p(_a(pxr, np.float32)), _p(_a(level, np.int32)), _p(_a(view_cos, np.float32)),
                                                  _p(_a(mp_desc, np.uint8)), _p(_a(mp_obs, np.int32)), C.c_float(th), C.c_float(nn_ratio),
                                                  _p(occ), _p(match))
    return nm, match, occ


def search_by_projection_frame(Cur, valid, u, v, inv_z, last_octave, last_angle, mp_desc, mp_obs, th, forward, backward, bf,
                               check_ori, occ_obs):
    n = len(u)
    occ = _a(occ_obs, np.int32).copy(); match = np.empty(Cur.n, np.int32)
    nm = lib().orc_search_by_projection_frame(Cur._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                              _p(_a(inv_z, np.float32)), _p(_a(last_octave, np.int32)), _p(_a(last_angle, np.float32)),
                                              _p(_a(mp_desc, np.uint8)), _p(_a(mp_obs, np.int32)), C.c_float(th), int(forward),
                                              int(backward), C.c_float(bf), int(check_ori), _p(occ), _p(match))
    return nm, match, occ


def search_by_projection_reloc(Cur, valid, u, v, pred_level, kf_angle, mp_desc, th, orb_dist, check_ori, occupied):
    n = len(u)
    occ = _a(occupied, np.uint8).copy(); match = np.empty(Cur.n, np.int32)
    nm = lib().orc_search_by_projection_reloc(Cur._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                              _p(_a(pred_level, np.int32)), _p(_a(kf_angle, np.float32)), _p(_a(mp_desc, np.uint8)),
                                              C.c_float(th), int(orb_dist), int(check_ori), _p(occ), _p(match))
    return nm, match, occ


def search_by_projection_kf(KF, valid, u, v, pred_level, mp_desc, th, matched):
    n = len(u)
    m = _a(matched, np.uint8).copy(); match = np.empty(KF.n, np.int32)
    nm = lib().orc_search_by_projection_kf(KF._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                           _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)), int(th), _p(m), _p(match))
    return nm, match, m


def fuse(KF, valid, u, v, ur, pred_level, mp_desc, th, scw_variant):
    n = len(u)
    bi = np.empty(n, np.int32); bd = np.empty(n, np.int32)
    nf = lib().orc_fuse(KF._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                        _p(_a(ur, np.float32)) if ur is not None else None, _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)),
                        C.c_float(th), int(scw_variant), _p(bi), _p(bd))
    return nf, bi, bd


def search_by_sim3_dir(KF, valid, u, v, pred_level, mp_desc, th):
    n = len(u)
    out = np.empty(n, np.int32)
    lib().orc_search_by_sim3_dir(KF._h, n, _p(_a(valid, np.uint8)), _p(_a(u, np.float32)), _p(_a(v, np.float32)),
                                 _p(_a(pred_level, np.int32)), _p(_a(mp_desc, np.uint8)), C.c_float(th), _p(out))
    return out


def sim3_agree(m1, m2):
    m1 = _a(m1, np.int32); m2 = _a(m2, np.int32)
    out = np.empty(len(m1), np.int32)
    n = lib().orc_sim3_agree(len(m1), _p(m1), _p(m2), _p(out))
    return n, out


def _nodes(off1, idx1, off2, idx2):
    return _a(off1, np.int32), _a(idx1, np.int32), _a(off2, np.int32), _a(idx2, np.int32)


def search_by_bow_kf_frame(off1, idx1, off2, idx2, kf_valid, keys_kf, desc_kf, F, nn_ratio, check_ori):
    off1, idx1, off2, idx2 = _nodes(off1, idx1, off2, idx2)
    match_f = np.empty(F.n, np.int32)
    nm = lib().orc_search_by_bow_kf_frame(len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), _p(_a(kf_valid, np.uint8)),
                                          _p(_a(keys_kf, KP_DTYPE)), _p(_a(desc_kf, np.uint8)), _p(F.keys), _p(F.desc), F.n,
                                          C.c_float(nn_ratio), int(check_ori), _p(match_f))
    return nm, match_f


def search_by_bow_kf_kf(off1, idx1, off2, idx2, valid1, keys1, desc1, valid2, keys2, desc2, nn_ratio, check_ori):
    off1, idx1, off2, idx2 = _nodes(off1, idx1, off2, idx2)
    keys1 = _a(keys1, KP_DTYPE); keys2 = _a(keys2, KP_DTYPE)
    m12 = np.empty(len(keys1), np.int32)
    nm = lib().orc_search_by_bow_kf_kf(len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), _p(_a(valid1, np.uint8)), _p(keys1),
                                       _p(_a(desc1, np.uint8)), len(keys1), _p(_a(valid2, np.uint8)), _p(keys2), _p(_a(desc2, np.uint8)),
                                       len(keys2), C.c_float(nn_ratio), int(check_ori), _p(m12))
    return nm, m12


def search_for_triangulation(off1, idx1, off2, idx2, keys1, ur1, has_mp1, desc1, keys2, ur2, has_mp2, desc2, F12, ex, ey, scale2,
                             sigma2_2, only_stereo, check_ori):
    off1, idx1, off2, idx2 = _nodes(off1, idx1, off2, idx2)
    keys1 = _a(keys1, KP_DTYPE); keys2 = _a(keys2, KP_DTYPE)
    m12 = np.empty(len(keys1), np.int32)
    nm = lib().orc_search_for_triangulation(len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), _p(keys1),
                                            _p(_a(ur1, np.float32)) if ur1 is not None else None, _p(_a(has_mp1, np.uint8)),
                                            _p(_a(desc1, np.uint8)), len(keys1), _p(keys2),
                                            _p(_a(ur2, np.float32)) if ur2 is not None else None, _p(_a(has_mp2, np.uint8)),
                                            _p(_a(desc2, np.uint8)), len(keys2), _p(_a(F12, np.float32)), C.c_float(ex), C.c_float(ey),
                                            _p(_a(scale2, np.float32)), _p(_a(sigma2_2, np.float32)), int(only_stereo), int(check_ori),
                                            _p(m12))
    return nm, m12


def search_for_initialization(keys1, desc1, F2, prev_xy, window, nn_ratio, check_ori):
    keys1 = _a(keys1, KP_DTYPE)
    prev = _a(prev_xy, np.float32).copy()
    m12 = np.empty(len(keys1), np.int32)
    nm = lib().orc_search_for_initialization(_p(keys1), _p(_a(desc1, np.uint8)), len(keys1), F2._h, _p(prev), int(window),
                                             C.c_float(nn_ratio), int(check_ori), _p(m12))
    return nm, m12, prev
