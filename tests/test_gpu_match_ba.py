"""GPU parity (bit-exact) of Hamming matching and BA edge linearisation vs the oracle."""
import numpy as np
import pytest
import torch

from sivo_amd import matcher, optimizer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nA,nB", [(1, 1), (3, 300), (257, 255), (2000, 2000), (33, 1)])
def test_hamming_matrix_bit_exact(oracle, nA, nB):
    rng = np.random.default_rng(nA + nB)
    A = rng.integers(0, 256, (nA, 32), dtype=np.uint8); B = rng.integers(0, 256, (nB, 32), dtype=np.uint8)
    B[0] = A[0]                       # a zero distance
    if nB > 1: B[1] = ~A[0]           # a 256 distance
    ref = oracle.hamming_matrix(A, B)
    assert np.array_equal(matcher.descriptor_distance_matrix(A, B), ref)
    out = matcher.descriptor_distance_matrix(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())
    assert np.array_equal(out.cpu().numpy(), ref)
    assert ref[0, 0] == 0 and (nB < 2 or ref[0, 1] == 256)


def test_hamming_empty():
    out = matcher.descriptor_distance_matrix(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8))
    assert out.shape == (0, 5)


def test_argmin2_candidate_lists(oracle):
    rng = np.random.default_rng(5)
    nA, nB = 700, 900
    A = rng.integers(0, 256, (nA, 32), dtype=np.uint8); B = rng.integers(0, 256, (nB, 32), dtype=np.uint8)
    B[10:40] = B[5]                     # duplicates: ties must keep the earlier candidate
    lens = rng.integers(0, 200, nA); lens[::7] = 0; lens[3] = 1
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, nB, off[-1]).astype(np.int32)
    bi, bd, sd = matcher.argmin2(A, B, off, idx)
    obi, obd, osd = oracle.hamming_argmin2(A, B, off, idx)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert (bi[lens == 0] == -1).all() and (bd[lens == 0] == 256).all()


def test_bruteforce_matches_matrix(oracle):
    rng = np.random.default_rng(6)
    A = rng.integers(0, 256, (500, 32), dtype=np.uint8); B = rng.integers(0, 256, (333, 32), dtype=np.uint8)
    bi, bd, sd = matcher.bruteforce(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())
    D = oracle.hamming_matrix(A, B)
    assert np.array_equal(bd.cpu().numpy(), D.min(1)) and np.array_equal(bi.cpu().numpy(), D.argmin(1))
    assert np.array_equal(sd.cpu().numpy(), np.sort(D, 1)[:, 1])


from conftest import make_ba_scene  # noqa: E402


def test_ba_linearize_bit_exact(oracle):
    poses, pts, edges, intr = make_ba_scene()
    assert 20000 < len(edges) < 70000
    g = optimizer.linearize(poses, pts, edges, intr)
    o = oracle.ba_linearize(poses, pts, edges, intr)
    for k in ("err", "Jx", "Jp", "chi2", "rho", "w", "depth_ok"):
        assert np.array_equal(g[k], o[k]), k
    assert (o["w"] < 1).any() and (o["w"] == 1).any()
    assert (edges["stereo"] == 0).any() and (edges["stereo"] == 1).any()


def test_ba_edge_cases(oracle):
    poses, pts, edges, intr = make_ba_scene(seed=3, n_kf=2, n_pts=50)
    # a point behind the camera (depth_ok = 0) and an empty batch
    pts[0] = [0, 0, -5]
    g = optimizer.linearize(poses, pts, edges, intr); o = oracle.ba_linearize(poses, pts, edges, intr)
    assert np.array_equal(g["depth_ok"], o["depth_ok"]) and (o["depth_ok"] == 0).any()
    assert np.array_equal(g["Jp"], o["Jp"])
    e0 = optimizer.linearize(poses, pts, edges[:0], intr)
    assert e0["err"].shape == (0, 3)
    from sivo_amd._lib import SivoError
    bad = edges[:1].copy(); bad["point"] = 10 ** 6
    with pytest.raises(SivoError):
        optimizer.linearize(poses, pts, bad, intr)


def test_entropy_gate_matches_oracle(oracle):
    from sivo_amd import selection
    rng = np.random.default_rng(4)
    n, H, W = 2000, 352, 1024
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(19, W - 19, n); kps["y"] = rng.uniform(19, H - 19, n); kps["octave"] = rng.integers(0, 8, n)
    depth = rng.uniform(-2, 60, n).astype(np.float32)
    xyz = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(1, 60, n)], 1)
    ent = rng.uniform(0, 3.9, (H, W))
    A = rng.standard_normal((6, 6)); Sx = A @ A.T * 1e-3 + np.eye(6) * 1e-4
    ls2 = (np.float32(1.2) ** (2 * np.arange(8))).astype(np.float32)
    o = oracle.entropy_gate(kps, depth, xyz, ent, Sx, 718.856, 718.856, 0.537, ls2, 4.0)
    g = selection.entropy_gate(kps, depth, xyz, ent, Sx, 718.856, 718.856, 0.537, ls2, 4.0)
    np.testing.assert_allclose(g[0], o[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g[1], o[1], rtol=1e-12, atol=1e-11)
    far = np.abs(o[1] - 4.0) > 1e-9
    assert np.array_equal(g[2][far], o[2][far]) and 0 < o[2].sum() < n
    assert (g[2][~(depth > 0)] == 0).all()
    e = selection.entropy_gate(kps[:0], depth[:0], xyz[:0], ent, Sx, 718.856, 718.856, 0.537, ls2, 4.0)
    assert len(e[0]) == 0
    # the per-frame form: host keys against the device-resident map (same kernel, read through pinned memory): identical
    import torch
    d_ent = torch.from_numpy(ent).cuda()
    for m in (n, 700, 1):                   # (shrinking calls re-use the thread's staging buffer)
        gm = selection.entropy_gate_map_dev(kps[:m], depth[:m], xyz[:m], d_ent, Sx, 718.856, 718.856, 0.537, ls2, 4.0)
        assert np.array_equal(gm[0], g[0][:m]) and np.array_equal(gm[1], g[1][:m]) and np.array_equal(gm[2], g[2][:m])


def test_check_semantics_matches_oracle(oracle):
    from sivo_amd import selection
    rng = np.random.default_rng(6)
    n, H, W = 1500, 352, 1024
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(19, W - 19, n); kps["y"] = rng.uniform(19, H - 19, n); kps["octave"] = rng.integers(0, 8, n)
    depth = rng.uniform(-2, 60, n).astype(np.float32)
    xyz = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(1, 60, n)], 1)
    ent = rng.uniform(0, 3.9, (H, W)); conf = rng.uniform(0.3, 1.0, (H, W)); cls = rng.integers(0, 15, (H, W)).astype(np.uint8)
    A = rng.standard_normal((6, 6)); Sx = A @ A.T * 1e-3 + np.eye(6) * 1e-4
    ls2 = (np.float32(1.2) ** (2 * np.arange(8))).astype(np.float32)
    red0 = oracle.check_semantics(kps, depth, xyz, ent, conf, cls, Sx, 718.856, 718.856, 0.537, ls2, -1e9, 0.7)
    th = float(np.median(red0[1][red0[2] != 255])) + 1e-3
    o = oracle.check_semantics(kps, depth, xyz, ent, conf, cls, Sx, 718.856, 718.856, 0.537, ls2, th, 0.7)
    g = selection.check_semantics(kps, depth, xyz, ent, conf, cls, Sx, 718.856, 718.856, 0.537, ls2, th, 0.7)
    np.testing.assert_allclose(g[0], o[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g[1], o[1], rtol=1e-12, atol=1e-11)
    far = np.abs(o[1] - th) > 1e-9
    assert np.array_equal(g[2][far], o[2][far]) and 0 < (o[2] != 255).sum() < n
