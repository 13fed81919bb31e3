"""CPU checks of the LM / Schur restatement (oracle/ba_solve_oracle.c).  g2o is absent (PARITY UNPINNED), so
the anchors are: the Schur step equals a dense solve of the full robustified normal equations; noise-free
scenes converge to the generating state; the chi2 schedules flag exactly the planted outliers."""
import numpy as np

from conftest import make_ba_scene, perturb_pose

TH_M, TH_S = float(np.sqrt(np.float32(5.991))), float(np.sqrt(np.float32(7.815)))


def _noise_free(poses, pts, edges, intr):
    fx, fy, cx, cy, bf = intr
    for e in edges:
        R = poses[e["pose"], :9].reshape(3, 3); pc = R @ pts[e["point"]] + poses[e["pose"], 9:]
        e["obs"] = [fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy, fx * pc[0] / pc[2] + cx - bf / pc[2]]
    return edges


def test_schur_step_equals_dense_normal_equations(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=4, n_kf=6, n_pts=300)
    rng = np.random.default_rng(0)
    fixed = np.zeros(6, np.uint8); fixed[:2] = 1
    P0 = poses.copy(); P0[2:, 9:] += rng.normal(0, 0.02, (4, 3)); X0 = pts + rng.normal(0, 0.05, pts.shape)
    r = oracle.ba_optimize(P0, fixed, X0, edges, intr, 1)
    assert r["iterations"] == 1 and r["trials"] == 1
    lin = oracle.ba_linearize(P0, X0, edges, intr, TH_M, TH_S)
    nF, nX = 4, len(pts); n = 6 * nF + 3 * nX
    H = np.zeros((n, n)); b = np.zeros(n)
    for e, ed in enumerate(edges):
        J = np.zeros((3, n)); s = ed["pose"] - 2
        if s >= 0: J[:, 6 * s:6 * s + 6] = lin["Jp"][e]
        q = 6 * nF + 3 * ed["point"]; J[:, q:q + 3] = lin["Jx"][e]
        wo = lin["w"][e] * ed["inv_sigma2"]
        H += wo * J.T @ J; b -= wo * J.T @ lin["err"][e]
    lam = 1e-5 * np.abs(np.diag(H)).max()                       # computeLambdaInit: tau * max diagonal
    dx = np.linalg.solve(H + lam * np.eye(n), b)
    np.testing.assert_allclose(r["points"], X0 + dx[6 * nF:].reshape(-1, 3), atol=1e-11, rtol=0)
    # pose update T <- exp([omega, upsilon]) * T with the se(3) exponential (scipy's expm of the 4x4 twist)
    from scipy.linalg import expm
    for s in range(nF):
        w, u = dx[6 * s:6 * s + 3], dx[6 * s + 3:6 * s + 6]
        xi = np.zeros((4, 4)); xi[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]; xi[:3, 3] = u
        T = np.eye(4); T[:3, :3] = P0[2 + s, :9].reshape(3, 3); T[:3, 3] = P0[2 + s, 9:]
        Tn = expm(xi) @ T
        np.testing.assert_allclose(r["poses"][2 + s, :9].reshape(3, 3), Tn[:3, :3], atol=1e-12)
        np.testing.assert_allclose(r["poses"][2 + s, 9:], Tn[:3, 3], atol=1e-12)
    # the Hpp blocks g2o's computeMarginals would factorise
    np.testing.assert_allclose(r["hpp"][0], H[:6, :6], rtol=1e-12)


def test_local_ba_reduces_chi2_and_converges_noise_free(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=4, n_kf=6, n_pts=300)
    edges = _noise_free(poses, pts, edges, intr)
    rng = np.random.default_rng(0)
    fixed = np.zeros(6, np.uint8); fixed[:2] = 1
    P0 = poses.copy(); P0[2:, 9:] += rng.normal(0, 0.02, (4, 3)); X0 = pts + rng.normal(0, 0.05, pts.shape)
    chi0 = oracle.ba_linearize(P0, X0, edges, intr)["chi2"].sum()
    r = oracle.local_ba(P0, fixed, X0, edges, intr, cov_pose=5)
    chi1 = oracle.ba_linearize(r["poses"], r["points"], edges, intr)["chi2"].sum()
    assert chi1 < 1e-4 * chi0 and r["outlier"].sum() == 0 and r["iterations"] == 15
    assert np.abs(r["poses"] - poses).max() < 1e-3
    assert np.array_equal(r["poses"][:2], poses[:2])                       # fixed keyframes untouched
    assert r["cov_ok"] and np.allclose(r["cov"], r["cov"].T) and (np.linalg.eigvalsh(r["cov"]) > 0).all()
    r2 = oracle.local_ba(P0, fixed, X0, edges, intr, cov_pose=0)           # a fixed keyframe has no marginal
    assert not r2["cov_ok"]
    r3 = oracle.local_ba(P0, fixed, X0, edges, intr, stop=True)            # pbStopFlag set on entry (:757-761)
    assert np.array_equal(r3["poses"], P0) and r3["iterations"] == 0


def test_pose_optimization_recovers_pose_and_flags_planted_outliers(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=99, n_kf=8, n_pts=3000)
    k = 5
    ek = edges[edges["pose"] == k].copy()
    p0 = perturb_pose(poses[k], np.random.default_rng(0))
    r = oracle.pose_optimize(p0, pts, ek, intr)
    assert np.abs(r["pose"] - poses[k]).max() < 5e-3 < np.abs(p0 - poses[k]).max()
    truth = oracle.ba_linearize(poses, pts, ek, intr)["chi2"]
    planted = (truth > 7.815) & (ek["stereo"] == 1)
    assert (r["outlier"] == planted).mean() > 0.995            # borderline chi2 values may flip with the 2 mm pose error
    assert not r["outlier"][ek["stereo"] == 0].any()            # mono edges are never re-classified (:432-467)
    assert r["inliers"] == len(ek) - r["outlier"].sum()
    assert r["cov_ok"] and (np.linalg.eigvalsh(r["cov"]) > 0).all()
    # fewer than 3 correspondences: nothing happens (:409-411)
    r0 = oracle.pose_optimize(p0, pts, ek[:2], intr)
    assert r0["inliers"] == 0 and np.array_equal(r0["pose"], p0)
