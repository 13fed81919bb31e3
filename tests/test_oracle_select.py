"""CPU: the feature-selection oracle (sivo_helpers restatement) against numpy and the Schur identity."""
import numpy as np


def _cov(rng, scale=1e-4):
    A = rng.standard_normal((6, 6))
    return A @ A.T * scale + np.eye(6) * scale * 0.1


def test_mutual_information_against_numpy(oracle):
    rng = np.random.default_rng(0)
    fx = fy = 718.856; bl = 386.1448 / 718.856
    for _ in range(50):
        Sx = _cov(rng, 10 ** rng.uniform(-6, -2))
        X, Y, Z = rng.uniform(-20, 20), rng.uniform(-3, 3), rng.uniform(1, 60)
        s2 = 1.2 ** (2 * rng.integers(0, 8))
        mi = oracle.stereo_mutual_information(Sx, fx, fy, bl, X, Y, Z, s2)
        J = np.array([[fx / Z, 0, -fx * X / Z**2, -fx * X * Y / Z**2, fx * (1 + X * X / Z**2), -fx * Y / Z],
                      [0, fy / Z, -fy * Y / Z**2, -fy * (1 + Y * Y / Z**2), fy * X * Y / Z**2, fy * X / Z],
                      [fx / Z, 0, -fx * (X - bl) / Z**2, -fx * (X - bl) * Y / Z**2, fx * (1 + X * (X - bl) / Z**2), -fx * Y / Z]])
        Sz = J @ Sx @ J.T + np.eye(3) * s2
        S9 = np.block([[Sx, Sx @ J.T], [J @ Sx, Sz]])
        ref = 0.5 * (np.linalg.slogdet(Sx)[1] + np.linalg.slogdet(Sz)[1] - np.linalg.slogdet(S9)[1]) / np.log(2)
        assert abs(mi - ref) < 1e-6 * max(1, abs(ref))
        assert abs(mi - 0.5 * np.log2(np.linalg.det(Sz) / s2**3)) < 1e-6 * max(1, abs(ref))   # Schur: det S9 = det Sx det R
        assert mi > 0


def test_gate_semantics(oracle):
    rng = np.random.default_rng(1)
    n, H, W = 300, 60, 80
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(0, W - 1, n); kps["y"] = rng.uniform(0, H - 1, n); kps["octave"] = rng.integers(0, 8, n)
    depth = rng.uniform(-1, 40, n).astype(np.float32)
    xyz = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(2, 50, n)], 1)
    ent = rng.uniform(0, 3.9, (H, W))
    ls2 = (1.2 ** (2 * np.arange(8))).astype(np.float32)
    Sx = _cov(rng, 1e-3)
    mi, red, acc = oracle.entropy_gate(kps, depth, xyz, ent, Sx, 718.856, 718.856, 0.537, ls2, 4.0)
    neg = ~(depth > 0)
    assert (acc[neg] == 0).all() and (mi[neg] == 0).all()
    e = ent[kps["y"].astype(int), kps["x"].astype(int)]            # static_cast<int> truncation
    pos = ~neg
    np.testing.assert_allclose(red[pos], mi[pos] - e[pos], rtol=0, atol=1e-12)
    assert np.array_equal(acc[pos], (red[pos] > 4.0).astype(np.uint8))
    assert 0 < acc.sum() < pos.sum()


def test_check_semantics_criteria(oracle):
    """LocalMapping::CheckSemantics: depth, static class, confidence, and rejection only BELOW the threshold (equality passes)."""
    rng = np.random.default_rng(2)
    n, H, W = 400, 60, 80
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(0, W - 1, n); kps["y"] = rng.uniform(0, H - 1, n); kps["octave"] = rng.integers(0, 8, n)
    depth = rng.uniform(-1, 40, n).astype(np.float32)
    xyz = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(2, 50, n)], 1)
    ent = rng.uniform(0, 3.9, (H, W)); conf = rng.uniform(0.3, 1.0, (H, W)); cls = rng.integers(0, 15, (H, W)).astype(np.uint8)
    ls2 = (1.2 ** (2 * np.arange(8))).astype(np.float32)
    Sx = _cov(rng, 1e-3)
    args = (kps, depth, xyz, ent, conf, cls, Sx, 718.856, 718.856, 0.537, ls2)
    r, c = kps["y"].astype(int), kps["x"].astype(int)
    ok = (depth > 0) & (cls[r, c] <= 8) & (conf[r, c] >= 0.7)
    th = float(np.median(oracle.check_semantics(*args, -1e9, 0.7)[1][ok])) + 1e-3     # about half of the candidates pass
    mi, red, det = oracle.check_semantics(*args, th, 0.7)
    assert (det[~ok] == 255).all() and (mi[~ok] == 0).all()
    assert np.array_equal(det[ok], np.where(red[ok] < th, 255, cls[r, c][ok]))
    assert 0 < (det != 255).sum() < ok.sum()
    # the Tracking gate on the same points agrees wherever the extra criteria hold and the reduction is off the threshold
    _, red_t, acc_t = oracle.entropy_gate(kps, depth, xyz, ent, Sx, 718.856, 718.856, 0.537, ls2, th)
    np.testing.assert_array_equal(red_t[ok], red[ok])
    assert np.array_equal(acc_t[ok] == 1, det[ok] != 255)
    # at equality the two rules differ: CheckSemantics keeps the point, CreateNewKeyFrame does not
    i = int(np.nonzero(ok)[0][0])
    _, _, det_eq = oracle.check_semantics(*args, float(red[i]), 0.7)
    _, _, acc_eq = oracle.entropy_gate(kps, depth, xyz, ent, Sx, 718.856, 718.856, 0.537, ls2, float(red[i]))
    assert det_eq[i] != 255 and acc_eq[i] == 0
