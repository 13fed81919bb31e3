"""Shared by tests/test_pin_helpers.py and tests/golden/make_helpers_reference.py: ctypes binding of
oracle/_ref/libref_helpers.so — the reference's OWN src/sivo_helpers/sivo_helpers.cpp compiled by `make -C oracle ref`
against the Eigen stand-in of oracle/ref_shims_eigen — and the seeded cases."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_helpers.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "helpers_reference.json")
FX, FY, BL = 718.856, 718.856, 386.1448 / 718.856
_d = C.c_double
_vp = lambda a: C.c_void_p(a.ctypes.data)


def _lib():
    lib = C.CDLL(REF_LIB)
    lib.ref_stereo_mutual_information.restype = C.c_double
    lib.ref_mono_mutual_information.restype = C.c_double
    return lib


def cases(n=512, seed=1):
    """(Sx 6x6 SPD over four decades, camera-frame point, measurement variance = mvLevelSigma2[octave])."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        A = rng.normal(size=(6, 6))
        Sx = A @ A.T * 10 ** rng.uniform(-6, -2) + np.eye(6) * 1e-9
        Sx = np.ascontiguousarray((Sx + Sx.T) / 2)
        xyz = (rng.uniform(-30, 30), rng.uniform(-5, 5), rng.uniform(0.5, 80))
        s2 = float(np.float32(1.2) ** (2 * int(rng.integers(0, 8))))
        out.append((Sx, xyz, s2))
    return out


def stereo_jacobian_pose(X, Y, Z):
    J = np.zeros((3, 6)); _lib().ref_stereo_jacobian_pose(_d(FX), _d(FY), _d(BL), _d(X), _d(Y), _d(Z), _vp(J)); return J


def mono_jacobian_pose(X, Y, Z):
    J = np.zeros((2, 6)); _lib().ref_mono_jacobian_pose(_d(FX), _d(FY), _d(X), _d(Y), _d(Z), _vp(J)); return J


def stereo_jacobian_point(X, Y, Z, Ccw):
    Ccw = np.ascontiguousarray(Ccw, np.float64)
    J = np.zeros((3, 3)); _lib().ref_stereo_jacobian_point(_d(FX), _d(FY), _d(BL), _d(X), _d(Y), _d(Z), _vp(Ccw), _vp(J)); return J


def stereo_mutual_information(Sx, xyz, s2):
    """computeStereoJacobianPose -> computeStereoCovariance -> computeStereoMutualInformation, as Tracking.cc:963-985 chains them."""
    lib = _lib()
    J = stereo_jacobian_pose(*xyz)
    N = np.ascontiguousarray(np.eye(3) * s2); cov = np.zeros((9, 9))
    lib.ref_stereo_covariance(_vp(Sx), _vp(J), _vp(N), _vp(cov))
    return lib.ref_stereo_mutual_information(_vp(cov)), cov


def mono_mutual_information(Sx, xyz, s2):
    lib = _lib()
    J = mono_jacobian_pose(*xyz)
    N = np.ascontiguousarray(np.eye(2) * s2); cov = np.zeros((8, 8))
    lib.ref_mono_covariance(_vp(Sx), _vp(J), _vp(N), _vp(cov))
    return lib.ref_mono_mutual_information(_vp(cov))


def update_stereo(Sx, xyz, s2):
    J = stereo_jacobian_pose(*xyz)
    N = np.ascontiguousarray(np.eye(3) * s2); out = np.zeros((6, 6))
    _lib().ref_update_covariance_stereo(_vp(Sx), _vp(J), _vp(N), _vp(out))
    return out


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)
