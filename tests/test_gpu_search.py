"""GPU parity of the guided-matching routines (SIVO::ORBmatcher Search* / Fuse from the projected point on,
sivo_amd/csrc/search.hip) against the sequential CPU restatement (oracle/search_oracle.c), through the C ABI.
Everything is integer / index work: the bar is bit-exact equality of every output array and of the match count."""
import numpy as np
import pytest

from oracle import search as OS
from search_scene import node_lists, orb_frames, stereo_right, tie_descriptors
from sivo_amd import matcher as M

pytestmark = pytest.mark.gpu
BF = 386.1448


@pytest.fixture(scope="module")
def scene(oracle, kitti_like_bgr):
    return build_scene(oracle, kitti_like_bgr)


def build_scene(oracle, kitti_like_bgr):
    ex, (k1, d1), (k2, d2), (H, W) = orb_frames(oracle, kitti_like_bgr)
    rng = np.random.default_rng(5)
    d1 = tie_descriptors(d1, rng); d2 = tie_descriptors(d2, rng)
    ur1, ur2 = stereo_right(k1, rng), stereo_right(k2, rng)
    bounds = (0.0, float(W), 0.0, float(H))
    tabs = (ex.scale, ex.sigma2, ex.inv_sigma2)
    F1o, F2o = OS.Frame(k1, ur1, d1, bounds, *tabs), OS.Frame(k2, ur2, d2, bounds, *tabs)
    F1g, F2g = M.MatchFrame(k1, ur1, d1, bounds, *tabs), M.MatchFrame(k2, ur2, d2, bounds, *tabs)
    return dict(ex=ex, k1=k1, d1=d1, k2=k2, d2=d2, ur1=ur1, ur2=ur2, F1o=F1o, F2o=F2o, F1g=F1g, F2g=F2g, H=H, W=W, bounds=bounds,
                tabs=tabs)


def _points(s, rng, dup=2, jitter=3.0, shift=(6.0, 2.0)):
    """Map points hanging on the keys of frame 1, projected into frame 2: `dup` points per key (collisions)."""
    k1 = s["k1"]; n = len(k1)
    src = np.concatenate([rng.permutation(n) for _ in range(dup)])
    u = (k1["x"][src] + shift[0] + rng.normal(0, jitter, len(src))).astype(np.float32)
    v = (k1["y"][src] + shift[1] + rng.normal(0, jitter, len(src))).astype(np.float32)
    u[::37] = -50.0; v[::41] = 1e4                       # outside the image
    inv_z = rng.uniform(0.02, 0.5, len(src)).astype(np.float32)
    inv_z[::29] *= -1                                     # behind the camera
    obs = rng.choice([0, 0, 1, 3, 5], len(src)).astype(np.int32)
    desc = s["d1"][src].copy()
    flip = rng.random(len(src)) < 0.5                     # a few flipped bits: near but not identical
    desc[flip, rng.integers(0, 32, int(flip.sum()))] ^= np.uint8(1) << rng.integers(0, 8, int(flip.sum())).astype(np.uint8)
    octave = k1["octave"][src].astype(np.int32)
    angle = k1["angle"][src].astype(np.float32)
    return src, u, v, inv_z, obs, desc, octave, angle


def test_get_features_in_area_matches_the_oracle(scene):
    rng = np.random.default_rng(0)
    for _ in range(300):
        x, y = rng.uniform(-40, scene["W"] + 40), rng.uniform(-40, scene["H"] + 40)
        r = float(rng.choice([0.5, 3.0, 12.5, 40.0, 400.0]))
        lo, hi = rng.choice([(-1, -1), (0, 0), (2, -1), (0, 3), (1, 2), (-1, 0), (7, 9)])
        a = scene["F2o"].features_in_area(x, y, r, int(lo), int(hi))
        b = scene["F2g"].features_in_area(x, y, r, int(lo), int(hi))
        assert np.array_equal(a, b), (x, y, r, lo, hi)


@pytest.mark.parametrize("th", [1.0, 3.0, 7.0])
@pytest.mark.parametrize("seed", [0, 1])
def test_search_by_projection_mappoints(scene, th, seed):
    rng = np.random.default_rng(seed)
    src, u, v, inv_z, obs, desc, octave, _ = _points(scene, rng, dup=3)
    n = len(u)
    tiv = (rng.random(n) < 0.9).astype(np.uint8)
    level = np.clip(octave + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    view_cos = rng.choice([0.9, 0.9985, 0.9999], n).astype(np.float32)
    pxr = (u - BF * np.abs(inv_z)).astype(np.float32)
    occ0 = rng.choice([-1, -1, -1, 0, 2], scene["F2o"].n).astype(np.int32)
    a = OS.search_by_projection_mappoints(scene["F2o"], tiv, u, v, pxr, level, view_cos, desc, obs, th, 0.8, occ0)
    b = M.search_by_projection_mappoints(scene["F2g"], tiv, u, v, pxr, level, view_cos, desc, obs, th, 0.8, occ0)
    assert a[0] == b[0] and a[0] > 100
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("mode", ["forward", "backward", "lateral"])
@pytest.mark.parametrize("th,check_ori", [(7.0, True), (15.0, True), (15.0, False), (2.0, True)])
def test_search_by_projection_frame(scene, mode, th, check_ori):
    rng = np.random.default_rng(int(th) + len(mode))
    src, u, v, inv_z, obs, desc, octave, angle = _points(scene, rng, dup=2)
    valid = (rng.random(len(u)) < 0.85).astype(np.uint8)
    occ0 = rng.choice([-1, -1, -1, 0, 4], scene["F2o"].n).astype(np.int32)
    fwd, bwd = mode == "forward", mode == "backward"
    a = OS.search_by_projection_frame(scene["F2o"], valid, u, v, inv_z, octave, angle, desc, obs, th, fwd, bwd, BF, check_ori, occ0)
    b = M.search_by_projection_frame(scene["F2g"], valid, u, v, inv_z, octave, angle, desc, obs, th, fwd, bwd, BF, check_ori, occ0)
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if th >= 7:
        assert a[0] > 100
        if check_ori:
            assert (a[1] == -2).any()                # the rotation check removed something


@pytest.mark.parametrize("th,orb_dist", [(10.0, 100), (3.0, 64)])
def test_search_by_projection_reloc_and_kf(scene, th, orb_dist):
    rng = np.random.default_rng(int(th))
    src, u, v, inv_z, obs, desc, octave, angle = _points(scene, rng, dup=2)
    valid = (rng.random(len(u)) < 0.9).astype(np.uint8)
    pred = np.clip(octave + rng.integers(-1, 2, len(u)), 0, 7).astype(np.int32)
    occ0 = (rng.random(scene["F2o"].n) < 0.2).astype(np.uint8)
    a = OS.search_by_projection_reloc(scene["F2o"], valid, u, v, pred, angle, desc, th, orb_dist, True, occ0)
    b = M.search_by_projection_reloc(scene["F2g"], valid, u, v, pred, angle, desc, th, orb_dist, True, occ0)
    assert a[0] == b[0] and a[0] > 50
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    a = OS.search_by_projection_kf(scene["F2o"], valid, u, v, pred, desc, int(th), occ0)
    b = M.search_by_projection_kf(scene["F2g"], valid, u, v, pred, desc, int(th), occ0)
    assert a[0] == b[0] and a[0] > 20
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("scw", [0, 1])
def test_fuse_and_sim3(scene, scw):
    rng = np.random.default_rng(scw)
    src, u, v, inv_z, obs, desc, octave, angle = _points(scene, rng, dup=2, jitter=1.0)
    valid = (rng.random(len(u)) < 0.9).astype(np.uint8)
    pred = np.clip(octave + rng.integers(0, 2, len(u)), 0, 7).astype(np.int32)
    # stereo keys of frame 2: make most right coordinates consistent with the projected ones so that the chi2 gate passes
    ur = (u - BF * np.abs(inv_z)).astype(np.float32)
    a = OS.fuse(scene["F2o"], valid, u, v, ur, pred, desc, 3.0, scw)
    b = M.fuse(scene["F2g"], valid, u, v, ur, pred, desc, 3.0, scw)
    assert a[0] == b[0] and a[0] > 20
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    m1o = OS.search_by_sim3_dir(scene["F2o"], valid, u, v, pred, desc, 7.5)
    m1g = M.search_by_sim3_dir(scene["F2g"], valid, u, v, pred, desc, 7.5)
    assert np.array_equal(m1o, m1g) and (m1o >= 0).sum() > 100


def test_fuse_chi2_gate_separates_stereo_and_mono(scene):
    """Points projected exactly onto keys of frame 2 with a right coordinate 2.6 sigma off: stereo keys fail 7.8, mono pass."""
    k2, ur2 = scene["k2"], scene["ur2"]
    n = len(k2)
    u, v = k2["x"].copy(), k2["y"].copy()
    sig = np.sqrt(scene["ex"].sigma2[k2["octave"]])
    ur = np.where(ur2 >= 0, ur2 + 2.9 * sig, 0).astype(np.float32)
    valid = np.ones(n, np.uint8)
    a = OS.fuse(scene["F2o"], valid, u, v, ur, k2["octave"], scene["d2"], 3.0, 0)
    b = M.fuse(scene["F2g"], valid, u, v, ur, k2["octave"], scene["d2"], 3.0, 0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    mono = ur2 < 0
    assert (a[1][mono] >= 0).mean() > 0.9 and (a[1][~mono] >= 0).mean() < 0.6


@pytest.mark.parametrize("check_ori", [True, False])
def test_bow_guided_routines(scene, check_ori):
    rng = np.random.default_rng(11)
    n1, n2 = len(scene["k1"]), len(scene["k2"])
    # frame 2's keys shifted back by the image shift ARE frame 1's content: a real matching problem inside the nodes
    off1, idx1, off2, idx2 = node_lists(n1, n2, rng, n_nodes=25)
    valid1 = (rng.random(n1) < 0.8).astype(np.uint8)
    valid2 = (rng.random(n2) < 0.8).astype(np.uint8)
    a = OS.search_by_bow_kf_frame(off1, idx1, off2, idx2, valid1, scene["k1"], scene["d1"], scene["F2o"], 0.9, check_ori)
    b = M.search_by_bow_kf_frame(off1, idx1, off2, idx2, valid1, scene["k1"], scene["d1"], scene["F2g"], 0.9, check_ori)
    assert a[0] == b[0] and a[0] > 10 and np.array_equal(a[1], b[1])
    a = OS.search_by_bow_kf_kf(off1, idx1, off2, idx2, valid1, scene["k1"], scene["d1"], valid2, scene["k2"], scene["d2"], 0.9, check_ori)
    b = M.search_by_bow_kf_kf(off1, idx1, off2, idx2, valid1, scene["k1"], scene["d1"], valid2, scene["F2g"], 0.9, check_ori)
    assert a[0] == b[0] and a[0] > 5 and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("only_stereo", [False, True])
def test_search_for_triangulation(scene, only_stereo):
    rng = np.random.default_rng(21)
    n1, n2 = len(scene["k1"]), len(scene["k2"])
    off1, idx1, off2, idx2 = node_lists(n1, n2, rng, n_nodes=12)
    has1 = (rng.random(n1) < 0.3).astype(np.uint8)
    has2 = (rng.random(n2) < 0.3).astype(np.uint8)
    # image 2 = image 1 translated by (6, 2): x2' F12 x1 = 0 with F12 = [t]x for a pure image translation t = (6, 2, 0)
    F12 = np.array([[0, 0, 2.0], [0, 0, -6.0], [-2.0, 6.0, 0]], np.float32).T.copy()
    ex, ey = 500.0, 170.0
    a = OS.search_for_triangulation(off1, idx1, off2, idx2, scene["k1"], scene["ur1"], has1, scene["d1"], scene["k2"], scene["ur2"], has2,
                                    scene["d2"], F12, ex, ey, scene["ex"].scale, scene["ex"].sigma2, only_stereo, True)
    b = M.search_for_triangulation(off1, idx1, off2, idx2, scene["k1"], scene["ur1"], has1, scene["d1"], scene["F2g"], has2, F12, ex, ey,
                                   only_stereo, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert (a[1] >= 0).sum() > 5


def test_repair_rounds_reproduce_the_sequential_result(scene):
    """Worst case for the speculation: every query wants the same few keypoints (identical descriptors, one window)."""
    k2, d2 = scene["k2"], scene["d2"]
    centre = int(np.argmin((k2["x"] - 500) ** 2 + (k2["y"] - 170) ** 2))
    nq = 64
    u = np.full(nq, k2["x"][centre], np.float32); v = np.full(nq, k2["y"][centre], np.float32)
    desc = np.repeat(d2[centre][None], nq, 0)
    lvl = np.full(nq, int(k2["octave"][centre]), np.int32)
    valid = np.ones(nq, np.uint8)
    occ0 = np.zeros(scene["F2o"].n, np.uint8)
    a = OS.search_by_projection_reloc(scene["F2o"], valid, u, v, lvl, np.zeros(nq, np.float32), desc, 40.0, 255, False, occ0)
    b = M.search_by_projection_reloc(scene["F2g"], valid, u, v, lvl, np.zeros(nq, np.float32), desc, 40.0, 255, False, occ0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[0] >= 8                                         # many keypoints handed out one after the other
    # the generic entry point reports how many rounds that took
    q = np.zeros(nq, M.QUERY_DTYPE)
    q["u"], q["v"], q["radius"] = u, v, 40.0 * scene["ex"].scale[lvl[0]]
    q["lvl_lo"], q["lvl_hi"], q["flags"] = lvl - 1, lvl + 1, M.Q_VALID | M.Q_BLOCKS
    r = M.search(scene["F2g"], q, desc, dict(th_dist=255, dynamic=1))
    assert r["n_matches"] == a[0] and r["rounds"] >= a[0]
    assert np.array_equal(r["match_train"], a[1])


def test_empty_inputs(scene):
    z = np.zeros(0, np.float32); zi = np.zeros(0, np.int32); zb = np.zeros(0, np.uint8); zd = np.zeros((0, 32), np.uint8)
    occ = np.full(scene["F2g"].n, -1, np.int32)
    nm, match, _ = M.search_by_projection_frame(scene["F2g"], zb, z, z, z, zi, z, zd, zi, 7.0, False, False, BF, True, occ)
    assert nm == 0 and (match == -1).all()
    empty = M.MatchFrame(np.zeros(0, M.KP_DTYPE), None, zd, scene["bounds"], *scene["tabs"])
    src, u, v, inv_z, obs, desc, octave, angle = _points(scene, np.random.default_rng(0), dup=1)
    nm, match, _ = M.search_by_projection_frame(empty, np.ones(len(u), np.uint8), u, v, inv_z, octave, angle, desc, obs, 7.0, False, False,
                                                BF, True, np.zeros(0, np.int32))
    assert nm == 0 and len(match) == 0
