"""CPU: the oracle's restatement of the ORBmatcher search routines and the frame grid (oracle/search_oracle.c).
The reference has no tests for them; tests/test_pin_matcher.py pins them against the reference's own ORBmatcher.cc. Checked here: the grid query against a
brute-force statement of Frame::GetFeaturesInArea's predicate, the sequential semantics on hand-made cases that have one
possible answer, and the rotation-histogram rule."""
import numpy as np
import pytest

from oracle import search as OS
from oracle.oracle import KP_DTYPE
from search_scene import orb_frames, stereo_right


@pytest.fixture(scope="module")
def frame(oracle, kitti_like_bgr):
    ex, (k1, d1), _, (H, W) = orb_frames(oracle, kitti_like_bgr, nfeatures=600)
    ur = stereo_right(k1, np.random.default_rng(1))
    return ex, k1, d1, ur, OS.Frame(k1, ur, d1, (0.0, float(W), 0.0, float(H)), ex.scale, ex.sigma2, ex.inv_sigma2), H, W


def test_features_in_area_is_the_window_predicate(frame):
    """Every key inside the open square |dx| < r, |dy| < r (and the level range) that sits in a grid cell is returned,
    nothing else, in grid-column-major order (Frame.cc:359-387)."""
    ex, k, d, ur, F, H, W = frame
    rng = np.random.default_rng(0)
    # PosInGrid rounds half away from zero (C round), numpy rounds half to even: x = 8 + 16 k sits exactly on .5
    gx = np.floor((k["x"] - np.float32(0)) * np.float32(64.0 / W) + np.float32(0.5)).astype(int)
    gy = np.floor((k["y"] - np.float32(0)) * np.float32(48.0 / H) + np.float32(0.5)).astype(int)
    in_grid = (gx >= 0) & (gx < 64) & (gy >= 0) & (gy < 48)
    for _ in range(200):
        x, y = np.float32(rng.uniform(0, W)), np.float32(rng.uniform(0, H))
        r = np.float32(rng.choice([2.0, 10.0, 33.3, 90.0]))
        lo, hi = rng.choice([(-1, -1), (0, 0), (1, -1), (0, 2), (3, 5)])
        got = F.features_in_area(x, y, r, int(lo), int(hi))
        want = in_grid & (np.abs(k["x"] - x) < r) & (np.abs(k["y"] - y) < r)
        if lo > 0 or hi >= 0:
            want &= k["octave"] >= lo
            if hi >= 0:
                want &= k["octave"] <= hi
        assert sorted(got) == list(np.nonzero(want)[0])
        # order: cell column, then cell row, then insertion (= key index) order
        key = list(zip(gx[got], gy[got], got))
        assert key == sorted(key)


def _tiny_frame(xs, ys, descs, ur=None, octave=0, angles=None):
    k = np.zeros(len(xs), KP_DTYPE)
    k["x"], k["y"], k["octave"] = xs, ys, octave
    if angles is not None:
        k["angle"] = angles
    sc = np.array([1.0, 1.2], np.float32)
    return OS.Frame(k, ur, np.asarray(descs, np.uint8), (0.0, 640.0, 0.0, 480.0), sc, sc * sc, 1 / (sc * sc))


def _desc(*bits):
    d = np.zeros(32, np.uint8)
    for b in bits:
        d[b // 8] |= 1 << (b % 8)
    return d


def test_sequential_semantics_on_hand_cases():
    # two keys next to each other; three map points with the same descriptor project onto them
    F = _tiny_frame([100, 102], [100, 100], [_desc(), _desc(0)])
    u = np.full(3, 101, np.float32); v = np.full(3, 100, np.float32)
    desc = np.stack([_desc()] * 3)
    one = np.ones(3, np.uint8)
    # relocalisation variant: every match blocks -> keys handed out in order of distance, third point gets nothing
    nm, match, occ = OS.search_by_projection_reloc(F, one, u, v, np.zeros(3, np.int32), np.zeros(3, np.float32), desc, 5.0, 100, False,
                                                   np.zeros(2, np.uint8))
    assert nm == 2 and list(match) == [0, 1] and list(occ) == [1, 1]
    # frame-to-frame variant with Observations() == 0: nobody blocks, every point takes key 0, the last one keeps it,
    # and all three count (ORBmatcher.cc:1372-1374)
    nm, match, occ = OS.search_by_projection_frame(F, one, u, v, np.full(3, 0.1, np.float32), np.zeros(3, np.int32), np.zeros(3, np.float32),
                                                   desc, np.zeros(3, np.int32), 5.0, False, False, 386.0, False, np.full(2, -1, np.int32))
    assert nm == 3 and list(match) == [2, -1]
    # same with Observations() > 0 for the first point: it locks key 0, the others fall back to key 1, the last keeps it
    nm, match, occ = OS.search_by_projection_frame(F, one, u, v, np.full(3, 0.1, np.float32), np.zeros(3, np.int32), np.zeros(3, np.float32),
                                                   desc, np.array([2, 0, 0], np.int32), 5.0, False, False, 386.0, False,
                                                   np.full(2, -1, np.int32))
    assert nm == 3 and list(match) == [0, 2] and list(occ) == [2, 0]


def test_ratio_rule_applies_only_within_one_level():
    """ORBmatcher.cc:117-119: best 10, second 11 -> rejected at nnratio 0.8 only when both sit on the same octave."""
    d_best, d_second = _desc(*range(10)), _desc(*range(11))
    for oct2, expect in ((0, 0), (1, 1)):
        k = np.zeros(2, KP_DTYPE)
        k["x"], k["y"] = [100, 101], [100, 100]
        k["octave"] = [0, oct2]
        sc = np.array([1.0, 1.2], np.float32)
        F = OS.Frame(k, None, np.stack([d_best, d_second]), (0.0, 640.0, 0.0, 480.0), sc, sc * sc, 1 / (sc * sc))
        nm, match, _ = OS.search_by_projection_mappoints(F, [1], [100.5], [100.0], [90.0], [1], [0.9], _desc()[None], [1], 1.0, 0.8,
                                                         np.full(2, -1, np.int32))
        assert nm == expect and match[0] == (0 if expect else -1)


def test_rotation_histogram_keeps_three_dominant_bins():
    """60 matches rotate by ~0 deg, 30 by ~90, 20 by ~180, 3 by ~270: the fourth bin is culled (ORBmatcher.cc:1545-1577)."""
    rots = [0.0] * 60 + [90.0] * 30 + [180.0] * 20 + [270.0] * 3
    n = len(rots)
    xs = 20 + 5 * np.arange(n, dtype=np.float32)
    descs = np.stack([_desc(i % 200, (i * 7) % 200 + 1) for i in range(n)])
    F = _tiny_frame(xs % 600, 20 + 10 * (np.arange(n) // 100), descs, angles=np.zeros(n, np.float32))
    k = F.keys
    nm, match, _ = OS.search_by_projection_reloc(F, np.ones(n, np.uint8), k["x"], k["y"], np.zeros(n, np.int32),
                                                 np.array(rots, np.float32), descs, 1.0, 0, True, np.zeros(n, np.uint8))
    assert nm == 110
    assert list(np.nonzero(match == -2)[0]) == [110, 111, 112] and (match[:110] == np.arange(110)).all()


def test_triangulation_takes_the_last_of_equal_candidates():
    """ORBmatcher.cc:703 `dist > bestDist -> continue`: an equal distance REPLACES the current best."""
    k1 = np.zeros(1, KP_DTYPE); k1["x"], k1["y"] = 100, 100
    k2 = np.zeros(3, KP_DTYPE); k2["x"], k2["y"] = [110, 120, 130], [100, 100, 100]
    d = np.stack([_desc(1), _desc(2), _desc(1, 2, 3)])               # distances to _desc(): 1, 1, 3
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)   # pure x translation: epipolar lines are the rows
    sc = np.array([1.0], np.float32)
    nm, m12 = OS.search_for_triangulation([0, 1], [0], [0, 3], [0, 1, 2], k1, np.array([5.0], np.float32), [0], _desc()[None], k2,
                                          np.array([5.0, 5.0, 5.0], np.float32), [0, 0, 0], d, F12, 0.0, 0.0, sc, sc, False, False)
    assert nm == 1 and m12[0] == 1


def test_search_for_initialization_rematches_to_the_closer_key():
    """ORBmatcher.cc:455-471: a key of frame 2 already matched at distance 2 is taken over by a later key at distance 1."""
    F2 = _tiny_frame([100], [100], [_desc()])
    k1 = np.zeros(2, KP_DTYPE); k1["x"], k1["y"] = [100, 101], [100, 100]
    d1 = np.stack([_desc(0, 1), _desc(0)])
    prev = np.array([[100, 100], [101, 100]], np.float32)
    nm, m12, prev2 = OS.search_for_initialization(k1, d1, F2, prev, 10, 0.9, False)
    assert nm == 1 and list(m12) == [-1, 0] and tuple(prev2[1]) == (100.0, 100.0)
