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
