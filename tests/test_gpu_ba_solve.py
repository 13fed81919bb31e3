"""GPU parity of the LM / Schur loops (sivo_ba_optimize, sivo_local_ba, sivo_pose_optimize) vs the oracle.
fp64 with a different (but fixed) summation order than the oracle's sequential loops: tolerances are stated per
quantity; inlier/outlier decisions must be identical."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_ba_scene, perturb_pose
from sivo_amd import optimizer

pytestmark = pytest.mark.gpu


def _start(poses, pts, n_fixed, seed, rot=0.002, trans=0.02, dx=0.05):
    rng = np.random.default_rng(seed)
    fixed = np.zeros(len(poses), np.uint8); fixed[:n_fixed] = 1
    P0 = poses.copy()
    for i in range(n_fixed, len(poses)):
        P0[i] = perturb_pose(poses[i], rng, rot, trans)
    return P0, fixed, pts + rng.normal(0, dx, pts.shape)


@pytest.mark.parametrize("iters", [1, 4])
def test_ba_optimize_matches_oracle(oracle, iters):
    poses, pts, edges, intr = make_ba_scene(seed=4, n_kf=6, n_pts=300)
    P0, fixed, X0 = _start(poses, pts, 2, 0)
    g = optimizer.ba_optimize(P0, fixed, X0, edges, intr, iters)
    o = oracle.ba_optimize(P0, fixed, X0, edges, intr, iters)
    assert (g["iterations"], g["trials"]) == (o["iterations"], o["trials"])
    np.testing.assert_allclose(g["poses"], o["poses"], atol=1e-10, rtol=0)
    np.testing.assert_allclose(g["points"], o["points"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(g["err"], o["err"], atol=1e-8, rtol=0)
    np.testing.assert_allclose(g["hpp"], o["hpp"], rtol=1e-9, atol=1e-6)
    assert np.array_equal(g["poses"][:2], P0[:2])


def test_ba_optimize_levels_kernels_and_degenerate_graphs(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=7, n_kf=5, n_pts=200)
    P0, fixed, X0 = _start(poses, pts, 1, 1)
    rng = np.random.default_rng(2)
    level = (rng.random(len(edges)) < 0.2).astype(np.uint8); robust = (rng.random(len(edges)) < 0.5).astype(np.uint8)
    g = optimizer.ba_optimize(P0, fixed, X0, edges, intr, 3, level, robust)
    o = oracle.ba_optimize(P0, fixed, X0, edges, intr, 3, level, robust)
    assert (g["iterations"], g["trials"]) == (o["iterations"], o["trials"])
    np.testing.assert_allclose(g["poses"], o["poses"], atol=1e-10, rtol=0)
    np.testing.assert_allclose(g["points"], o["points"], atol=1e-9, rtol=0)
    assert np.array_equal(g["err"][level == 1], np.zeros((int(level.sum()), 3)))       # inactive edges are never evaluated
    # every keyframe fixed: structure-only refinement
    allfixed = np.ones(len(poses), np.uint8)
    g = optimizer.ba_optimize(poses, allfixed, X0, edges, intr, 2)
    o = oracle.ba_optimize(poses, allfixed, X0, edges, intr, 2)
    np.testing.assert_allclose(g["points"], o["points"], atol=1e-9, rtol=0)
    assert np.array_equal(g["poses"], poses)
    # no edges: nothing to do, nothing crashes
    g = optimizer.ba_optimize(P0, fixed, X0, edges[:0], intr, 2)
    assert np.array_equal(g["poses"], P0) and np.array_equal(g["points"], X0)
    # duplicate (keyframe, map point) observation is refused
    dup = np.concatenate([edges, edges[edges["pose"] == 2][:1]])           # a free keyframe's observation, twice
    with pytest.raises(ValueError, match="same"):
        optimizer.ba_optimize(P0, fixed, X0, dup, intr, 1)


def test_ba_optimize_more_than_21_free_keyframes(oracle):
    """28 free keyframes: the reduced system (168 x 168) no longer fits the LDS-resident Cholesky and takes the
    global-memory path of the same blocked solver."""
    poses, pts, edges, intr = make_ba_scene(seed=12, n_kf=30, n_pts=400)
    P0, fixed, X0 = _start(poses, pts, 2, 5)
    g = optimizer.ba_optimize(P0, fixed, X0, edges, intr, 3)
    o = oracle.ba_optimize(P0, fixed, X0, edges, intr, 3)
    assert (g["iterations"], g["trials"]) == (o["iterations"], o["trials"])
    np.testing.assert_allclose(g["poses"], o["poses"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(g["points"], o["points"], atol=1e-8, rtol=0)


def test_local_ba_config5_matches_oracle(oracle):
    """20 keyframes x 3000 map points (~36 k edges), the first two keyframes fixed."""
    poses, pts, edges, intr = make_ba_scene()
    P0, fixed, X0 = _start(poses, pts, 2, 3)
    g = optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)
    o = oracle.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)
    assert (g["iterations"], g["trials"]) == (o["iterations"], o["trials"])
    np.testing.assert_allclose(g["poses"], o["poses"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(g["points"], o["points"], atol=1e-7, rtol=0)
    assert (g["outlier"] != o["outlier"]).sum() <= 2            # chi2 within 1e-9 of 5.991 / 7.815 may flip
    assert g["cov_ok"] and o["cov_ok"]
    np.testing.assert_allclose(g["cov"], o["cov"], rtol=1e-7, atol=1e-16)
    assert np.abs(g["poses"] - poses).max() < np.abs(P0 - poses).max()
    # reproducible run to run (fixed summation order, no atomics)
    g2 = optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)
    assert np.array_equal(g2["poses"], g["poses"]) and np.array_equal(g2["points"], g["points"])
    # pbStopFlag raised before the call: untouched (Optimizer.cc:757-761)
    stop = C.c_uint8(1)
    s = optimizer.local_ba(P0, fixed, X0, edges, intr, stop=stop)
    assert np.array_equal(s["poses"], P0) and s["iterations"] == 0 and not s["outlier"].any()


@pytest.mark.parametrize("kf,seed", [(5, 0), (0, 1), (19, 2)])
def test_pose_optimization_matches_oracle(oracle, kf, seed):
    poses, pts, edges, intr = make_ba_scene()
    ek = edges[edges["pose"] == kf].copy()
    p0 = perturb_pose(poses[kf], np.random.default_rng(seed))
    g = optimizer.pose_optimize(p0, pts, ek, intr)
    o = oracle.pose_optimize(p0, pts, ek, intr)
    # Once converged, chi_new - chi is rounding noise, so the accept/reject sign of the last trials (and with it
    # the trial COUNT) depends on the summation order; the state they converge to does not.
    assert 4 <= g["iterations"] <= 40 and g["iterations"] <= g["trials"] <= 400
    np.testing.assert_allclose(g["pose"], o["pose"], atol=1e-7, rtol=0)
    assert np.array_equal(g["outlier"], o["outlier"]) and g["inliers"] == o["inliers"]
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-4, atol=1e-4)
    assert g["cov_ok"] and o["cov_ok"]
    np.testing.assert_allclose(g["cov"], o["cov"], rtol=1e-6, atol=1e-18)
    g2 = optimizer.pose_optimize(p0, pts, ek, intr)                       # fixed summation order: reproducible
    assert np.array_equal(g2["pose"], g["pose"]) and g2["trials"] == g["trials"]
    assert np.abs(g["pose"] - poses[kf]).max() < 2e-2 < np.abs(p0 - poses[kf]).max()


def test_pose_optimization_more_edges_than_the_lds_copy_holds(oracle):
    """pose_optimize_kernel keeps 2560 edges in LDS and reads the rest from memory in every pass: one keyframe seeing ~5000 points
    (with planted outliers, so that flags beyond the LDS copy change), then the same call timed at a tracking-sized problem."""
    import time
    poses, pts, edges, intr = make_ba_scene(seed=11, n_kf=2, n_pts=9000)
    ek = edges[edges["pose"] == 1].copy()
    assert len(ek) > 3000
    rng = np.random.default_rng(5)
    planted = rng.choice(len(ek), 150, replace=False)
    ek["obs"][planted, 0] += rng.choice([-1, 1], 150) * rng.uniform(8, 30, 150)
    p0 = perturb_pose(poses[1], np.random.default_rng(6))
    g = optimizer.pose_optimize(p0, pts, ek, intr); o = oracle.pose_optimize(p0, pts, ek, intr)
    np.testing.assert_allclose(g["pose"], o["pose"], atol=1e-7, rtol=0)
    assert np.array_equal(g["outlier"], o["outlier"]) and g["inliers"] == o["inliers"] and g["outlier"][2560:].any()
    np.testing.assert_allclose(g["chi2"], o["chi2"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g["cov"], o["cov"], rtol=1e-6, atol=1e-18)
    for n in (300, 1000, 2000, len(ek)):
        sub = ek[:n]
        optimizer.pose_optimize(p0, pts, sub, intr)
        t0 = time.perf_counter()
        for _ in range(20):
            r = optimizer.pose_optimize(p0, pts, sub, intr)
        print(f"[pose_optimize {n} edges] {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call (through ctypes), {r['iterations']} iterations, {r['trials']} trials")


def test_pose_optimization_edge_cases(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=3, n_kf=2, n_pts=60)
    ek = edges[edges["pose"] == 1].copy()
    p0 = perturb_pose(poses[1], np.random.default_rng(4), 0.005, 0.05)
    r = optimizer.pose_optimize(p0, pts, ek[:2], intr)                    # < 3 correspondences (:409-411)
    assert r["inliers"] == 0 and np.array_equal(r["pose"], p0) and not r["cov_ok"]
    for n in (3, 9, 10, len(ek)):                                          # < 10 edges: a single round (:469-471)
        g = optimizer.pose_optimize(p0, pts, ek[:n], intr); o = oracle.pose_optimize(p0, pts, ek[:n], intr)
        assert (g["iterations"] <= 10) == (n < 10)                         # one round only when < 10 edges
        np.testing.assert_allclose(g["pose"], o["pose"], atol=1e-6, rtol=0)
        assert np.array_equal(g["outlier"], o["outlier"])
    mono = ek.copy(); mono["stereo"] = 0                                   # mono only: never re-classified
    g = optimizer.pose_optimize(p0, pts, mono, intr); o = oracle.pose_optimize(p0, pts, mono, intr)
    assert not g["outlier"].any() and g["inliers"] == len(mono) == o["inliers"]
    np.testing.assert_allclose(g["pose"], o["pose"], atol=1e-7, rtol=0)
    bad = ek.copy(); bad["point"][0] = 10 ** 6
    with pytest.raises(ValueError):
        optimizer.pose_optimize(p0, pts, bad, intr)
