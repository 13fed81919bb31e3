"""The information-gain formulas of the feature-selection gate pinned against the reference's OWN code.

`make -C oracle ref` compiles /root/reference/src/sivo_helpers/sivo_helpers.cpp as it is into oracle/_ref/libref_helpers.so;
Eigen (absent here) is a stand-in that evaluates dense double matrices in Eigen's order (oracle/ref_shims_eigen).  So the
FORMULAS are the reference's — the 3 x 6 stereo projection Jacobian, the 9 x 9 joint covariance, 0.5 log2(det Sx det Sz /
det S) — and the linear algebra under them is restated.  The same Jacobians are the only in-tree statement of what g2o's
stereo edges compute (g2o itself is an empty submodule), so they also anchor the BA oracle."""
import os

import numpy as np
import pytest

import pin_helpers_common as P

HAVE_REF = os.path.exists(P.REF_LIB)
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/libref_helpers.so not built (no /root/reference here)")


def test_oracle_mutual_information_equals_the_reference_helpers():
    from oracle import oracle as O
    golden = P.load_golden()["stereo_mi_hex"]
    cases = P.cases()
    assert len(cases) == len(golden) == 512
    for (Sx, xyz, s2), want in zip(cases, golden):
        got = O.stereo_mutual_information(Sx, P.FX, P.FY, P.BL, *xyz, s2)
        assert float(got).hex() == want                                       # bit for bit
        if HAVE_REF:
            assert float(P.stereo_mutual_information(Sx, xyz, s2)[0]).hex() == want, "stale golden"


@needs_ref
def test_reference_helpers_are_self_consistent():
    """Schur identity det S9 = det Sx * det(R) (so MI = 0.5 log2(det Sz / det R)), and the Kalman update shrinks the covariance."""
    for Sx, xyz, s2 in P.cases(64, 3):
        mi, cov = P.stereo_mutual_information(Sx, xyz, s2)
        Sz = cov[6:, 6:]
        assert np.isclose(mi, 0.5 * np.log2(np.linalg.det(Sz) / s2 ** 3), rtol=1e-9, atol=1e-9)
        up = P.update_stereo(Sx, xyz, s2)
        J = P.stereo_jacobian_pose(*xyz)
        want = Sx - Sx @ J.T @ np.linalg.inv(J @ Sx @ J.T + np.eye(3) * s2) @ J @ Sx
        np.testing.assert_allclose(up, want, rtol=1e-8, atol=1e-16)
        assert np.isfinite(P.mono_mutual_information(Sx, xyz, s2))


@needs_ref
def test_ba_oracle_jacobians_agree_with_the_reference_helpers():
    """g2o's EdgeStereoSE3ProjectXYZ (source not in the tree) linearises e = obs - proj: its pose Jacobian is minus the
    helper's, with the rotation columns first, and its point Jacobian is minus projection-Jacobian x R — which is what
    oracle/ba_oracle.c restates.  Checked here against the reference's computeStereoJacobianPose / ...Point."""
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    cx, cy, bf = 607.1928, 185.2157, P.FX * P.BL
    for _ in range(200):
        th = rng.uniform(-0.3, 0.3, 3)
        K = np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]])
        a = np.linalg.norm(th)
        R = np.eye(3) + np.sin(a) / a * K + (1 - np.cos(a)) / a ** 2 * K @ K
        t = rng.uniform(-1, 1, 3)
        Xw = rng.uniform(-10, 10, 3); Xw[2] = rng.uniform(4, 50)
        Xc = R @ Xw + t
        pose = np.concatenate([R.ravel(), t])                                   # 12 doubles per pose: R row-major, t
        edge = np.zeros(1, O.EDGE_DTYPE)
        edge["stereo"] = 1; edge["obs"] = (100.0, 50.0, 90.0); edge["inv_sigma2"] = 1.0
        lin = O.ba_linearize(pose[None], Xw[None], edge, [P.FX, P.FY, cx, cy, bf])
        Jpose_ref = P.stereo_jacobian_pose(*Xc)                                 # columns: translation | rotation
        Jpoint_ref = P.stereo_jacobian_point(*Xc, R)
        np.testing.assert_allclose(lin["Jp"][0][:, :3], -Jpose_ref[:, 3:], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lin["Jp"][0][:, 3:], -Jpose_ref[:, :3], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lin["Jx"][0], -Jpoint_ref, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_device_gate_equals_the_reference_helpers():
    """sivo_entropy_gate on the GPU against the reference's chain (or its committed results)."""
    from sivo_amd import selection
    golden = [float.fromhex(h) for h in P.load_golden()["stereo_mi_hex"]]
    cases = P.cases()
    kp_dtype = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                         ("octave", np.int32), ("class_id", np.int32)])
    ls2 = np.array([np.float32(1.2) ** (2 * i) for i in range(8)], np.float32)
    for i in range(0, 512, 64):                                                 # the gate takes one state covariance per call
        Sx = cases[i][0]
        pts = [cases[i + k] for k in range(64)]
        kps = np.zeros(64, kp_dtype)
        kps["x"] = 5; kps["y"] = 5
        kps["octave"] = [int(np.argmin(np.abs(ls2 - np.float32(c[2])))) for c in pts]
        xyz = np.array([c[1] for c in pts])
        mi, _, _ = selection.entropy_gate(kps, np.ones(64, np.float32), xyz, np.zeros((16, 16)), Sx, P.FX, P.FY, P.BL, ls2, 0.0)
        want = np.array([P.stereo_mutual_information(Sx, c[1], float(ls2[o]))[0] if HAVE_REF else np.nan for c, o in zip(pts, kps["octave"])])
        if HAVE_REF:
            np.testing.assert_allclose(mi, want, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(mi[0], golden[i], rtol=1e-13, atol=1e-13)   # case i uses its own Sx and sigma2
