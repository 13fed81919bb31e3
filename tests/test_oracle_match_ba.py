"""CPU: the matching and BA-edge oracles against independent definitions.  The reference has no tests
for ORBmatcher / Frame / Optimizer; anchors are the popcount identity, the in-tree stereo Jacobians of
src/sivo_helpers/sivo_helpers.cpp:64-88,113-136 (algebraic relation) and finite differences."""
import numpy as np
import pytest

from conftest import synthetic_stereo


def test_descriptor_distance_is_popcount(oracle):
    rng = np.random.default_rng(0)
    A = rng.integers(0, 256, (40, 32), dtype=np.uint8); B = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    ref = np.unpackbits(A[:, None, :] ^ B[None, :, :], axis=2).sum(2)
    assert np.array_equal(oracle.hamming_matrix(A, B), ref)
    assert oracle.descriptor_distance(A[0], A[0]) == 0 and oracle.descriptor_distance(A[0], ~A[0]) == 256


def test_argmin2_sequential_semantics(oracle):
    rng = np.random.default_rng(1)
    A = rng.integers(0, 256, (30, 32), dtype=np.uint8); B = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    B[7] = B[3]
    off = np.arange(0, 31 * 12, 12, dtype=np.int32)[:31]; idx = rng.integers(0, 60, off[-1]).astype(np.int32)
    idx[:4] = [7, 3, 3, 7]
    bi, bd, sd = oracle.hamming_argmin2(A, B, off, idx)
    D = oracle.hamming_matrix(A, B)
    for i in range(30):
        c = idx[off[i]:off[i + 1]]; d = D[i, c]
        assert bd[i] == d.min() and bi[i] == c[np.argmin(d)] and sd[i] == np.sort(d)[1]
    # duplicates: make B[7] == B[3] the unique best for query 0 -> the EARLIER list entry (7) must win, second == best
    A[0] = B[3]
    bi, bd, sd = oracle.hamming_argmin2(A, B, off, idx)
    assert bi[0] == 7 and bd[0] == 0 and sd[0] == 0


def test_stereo_matches_recover_disparity(oracle):
    L, R = synthetic_stereo(21, disparity=8)
    el, er = oracle.OrbExtractor(), oracle.OrbExtractor()
    kl, dl = el(L); kr, dr = er(R)
    bf, b = 386.1448, 386.1448 / 718.856
    uR, depth, best, kept = oracle.stereo_matches(kl, dl, kr, dr, el.scale, el.inv_scale, [el.level(l) for l in range(8)],
                                                  [er.level(l) for l in range(8)], bf, b)
    ok = uR >= 0
    assert kept == ok.sum() and kept > 200
    assert abs(np.median(kl["x"][ok] - uR[ok]) - 8) < 0.5
    np.testing.assert_allclose(depth[ok], bf / (kl["x"][ok] - uR[ok]), rtol=1e-5)
    assert (depth[~ok] == -1).all()
    # no right keypoints at all -> nothing matches
    uR0, _, _, kept0 = oracle.stereo_matches(kl, dl, kr[:0], dr[:0], el.scale, el.inv_scale, [el.level(l) for l in range(8)],
                                             [er.level(l) for l in range(8)], bf, b)
    assert kept0 == 0 and (uR0 == -1).all()


def _scene(seed=0, n=200):
    rng = np.random.default_rng(seed)
    poses = np.zeros((3, 12))
    for k in range(3):
        a = rng.normal(0, 0.05, 3)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + K + K @ K / 2
        u, _, vt = np.linalg.svd(R); R = u @ vt
        poses[k, :9] = R.ravel(); poses[k, 9:] = rng.normal(0, 0.3, 3)
    pts = np.stack([rng.uniform(-10, 10, n), rng.uniform(-3, 3, n), rng.uniform(4, 40, n)], 1)
    from oracle.oracle import EDGE_DTYPE
    edges = np.zeros(n, EDGE_DTYPE)
    edges["pose"] = rng.integers(0, 3, n); edges["point"] = np.arange(n); edges["stereo"] = rng.integers(0, 2, n)
    edges["obs"] = rng.uniform(0, 300, (n, 3)); edges["inv_sigma2"] = 1 / 1.2 ** (2 * rng.integers(0, 8, n))
    return poses, pts, edges, (718.856, 718.856, 498.692, 173.215, 386.1448)


def test_ba_jacobians_by_finite_differences(oracle):
    poses, pts, edges, intr = _scene()
    o = oracle.ba_linearize(poses, pts, edges, intr)
    h = 1e-6
    for j in range(3):                                      # d err / d point
        p2 = pts.copy(); p2[:, j] += h; m2 = pts.copy(); m2[:, j] -= h
        fd = (oracle.ba_linearize(poses, p2, edges, intr)["err"] - oracle.ba_linearize(poses, m2, edges, intr)["err"]) / (2 * h)
        np.testing.assert_allclose(o["Jx"][:, :, j], fd, atol=1e-5)

    def perturb(sign, j):                                   # T <- exp([w, v]) T, rotation components first
        out = poses.copy()
        d = np.zeros(6); d[j] = sign * h
        w, v = d[:3], d[3:]
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        dR = np.eye(3) + K + K @ K / 2
        for k in range(len(out)):
            R = out[k, :9].reshape(3, 3); t = out[k, 9:]
            out[k, :9] = (dR @ R).ravel(); out[k, 9:] = dR @ t + v
        return out
    for j in range(6):
        fd = (oracle.ba_linearize(perturb(+1, j), pts, edges, intr)["err"] - oracle.ba_linearize(perturb(-1, j), pts, edges, intr)["err"]) / (2 * h)
        np.testing.assert_allclose(o["Jp"][:, :, j], fd, atol=2e-4)


def test_ba_matches_sivo_helpers_algebra(oracle):
    """J_pose_g2o[:, 0:3] = -J_sivo[:, 3:6], J_pose_g2o[:, 3:6] = -J_sivo[:, 0:3] with
    J_sivo = computeStereoJacobianPose(fx, fy, bl, X, Y, Z) (sivo_helpers.cpp:64-88), bf = fx * bl."""
    poses, pts, edges, intr = _scene(1)
    edges["stereo"] = 1
    o = oracle.ba_linearize(poses, pts, edges, intr)
    fx, fy, cx, cy, bf = intr; bl = bf / fx
    for e in range(len(edges)):
        R = poses[edges["pose"][e], :9].reshape(3, 3); t = poses[edges["pose"][e], 9:]
        X, Y, Z = R @ pts[edges["point"][e]] + t
        J = np.array([[fx / Z, 0, -fx * X / Z**2, -fx * X * Y / Z**2, fx * (1 + X * X / Z**2), -fx * Y / Z],
                      [0, fy / Z, -fy * Y / Z**2, -fy * (1 + Y * Y / Z**2), fy * X * Y / Z**2, fy * X / Z],
                      [fx / Z, 0, -fx * (X - bl) / Z**2, -fx * (X - bl) * Y / Z**2, fx * (1 + X * (X - bl) / Z**2), -fx * Y / Z]])
        np.testing.assert_allclose(o["Jp"][e][:, 0:3], -J[:, 3:6], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o["Jp"][e][:, 3:6], -J[:, 0:3], rtol=1e-9, atol=1e-9)
        Pj = np.array([[fx / Z, 0, -fx * X / Z**2], [0, fy / Z, -fy * Y / Z**2], [fx / Z, 0, -fx * (X - bl) / Z**2]])
        np.testing.assert_allclose(o["Jx"][e], -Pj @ R, rtol=1e-9, atol=1e-9)   # computeStereoJacobianPoint :113-136


def test_ba_chi2_and_huber(oracle):
    poses, pts, edges, intr = _scene(2)
    o = oracle.ba_linearize(poses, pts, edges, intr)
    np.testing.assert_allclose(o["chi2"], (o["err"] ** 2).sum(1) * edges["inv_sigma2"], rtol=1e-12)
    mono = edges["stereo"] == 0
    assert (o["err"][mono, 2] == 0).all() and (o["Jp"][mono, 2] == 0).all()
    delta = np.where(mono, np.sqrt(5.991), np.sqrt(7.815))
    inl = o["chi2"] <= delta**2
    assert (o["w"][inl] == 1).all() and (o["rho"][inl] == o["chi2"][inl]).all()
    np.testing.assert_allclose(o["rho"][~inl], 2 * np.sqrt(o["chi2"][~inl]) * delta[~inl] - delta[~inl] ** 2, rtol=1e-12)
    np.testing.assert_allclose(o["w"][~inl], delta[~inl] / np.sqrt(o["chi2"][~inl]), rtol=1e-12)
