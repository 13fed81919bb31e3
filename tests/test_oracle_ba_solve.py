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


def test_converged_bundle_adjustment_equals_minpack_on_the_same_residuals(oracle):
    """Second opinion for the unpinned g2o solve: the state the LM / Schur restatement converges to must minimise the same sum of
    squares as a solver that shares no code with it — MINPACK's Levenberg-Marquardt (scipy.optimize.least_squares, method
    'lm', finite-difference Jacobian) on whitened residuals written here from the projection equations, poses moved by the
    se(3) exponential (scipy's expm of the twist).  Kernels off (robust = 0: a smooth problem; the Huber weights have their own
    test, test_oracle_match_ba.py::test_ba_chi2_and_huber), noisy observations, two fixed and two free keyframes."""
    from scipy.linalg import expm
    from scipy.optimize import least_squares
    poses, pts, edges, intr = make_ba_scene(seed=5, n_kf=4, n_pts=40, stereo_frac=1.0)
    fx, fy, cx, cy, bf = intr
    rng = np.random.default_rng(1)
    fixed = np.zeros(4, np.uint8); fixed[:2] = 1
    P0 = poses.copy(); P0[2:, 9:] += rng.normal(0, 0.02, (2, 3)); X0 = pts + rng.normal(0, 0.05, pts.shape)
    r = oracle.ba_optimize(P0, fixed, X0, edges, intr, 200, robust=np.zeros(len(edges), np.uint8))
    assert r["iterations"] < 200                                   # stopped by its own criterion, not by the cap

    def unpack(x):
        P = P0.copy()
        for s in range(2):
            w, u = x[6 * s:6 * s + 3], x[6 * s + 3:6 * s + 6]
            xi = np.zeros((4, 4)); xi[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]; xi[:3, 3] = u
            T = np.eye(4); T[:3, :3] = P0[2 + s, :9].reshape(3, 3); T[:3, 3] = P0[2 + s, 9:]
            Tn = expm(xi) @ T
            P[2 + s, :9] = Tn[:3, :3].ravel(); P[2 + s, 9:] = Tn[:3, 3]
        return P, X0 + x[12:].reshape(-1, 3)

    def residuals(P, X):
        R = P[edges["pose"], :9].reshape(-1, 3, 3); t = P[edges["pose"], 9:]
        pc = np.einsum("eij,ej->ei", R, X[edges["point"]]) + t
        u = fx * pc[:, 0] / pc[:, 2] + cx
        proj = np.stack([u, fy * pc[:, 1] / pc[:, 2] + cy, u - bf / pc[:, 2]], 1)
        return ((edges["obs"] - proj) * np.sqrt(edges["inv_sigma2"])[:, None]).ravel()

    sol = least_squares(lambda x: residuals(*unpack(x)), np.zeros(12 + 3 * len(pts)), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                        max_nfev=200000)
    Ps, Xs = unpack(sol.x)
    c_oracle, c_minpack = float((residuals(r["poses"], r["points"]) ** 2).sum()), float((residuals(Ps, Xs) ** 2).sum())
    start = float((residuals(P0, X0) ** 2).sum())
    assert c_oracle < 0.9 * start                                  # there was something to optimise
    assert abs(c_oracle - c_minpack) <= 1e-7 * c_minpack, (c_oracle, c_minpack)
    assert np.abs(r["poses"] - Ps).max() < 1e-3                      # (far points' depth is a flat direction: compare what is observable)
    assert np.abs(residuals(r["poses"], r["points"]) - residuals(Ps, Xs)).max() < 1e-2
    # and the oracle's own chi2 is the sum of squares written here
    np.testing.assert_allclose(oracle.ba_linearize(r["poses"], r["points"], edges, intr)["chi2"],
                               (residuals(r["poses"], r["points"]).reshape(-1, 3) ** 2).sum(1), rtol=1e-10, atol=1e-12)
