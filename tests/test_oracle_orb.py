"""CPU: the ORB oracle against the constants the reference's source fixes (SURVEY.md 8c "constants that
act as KATs"), against the frozen golden output, and against independent float re-derivations of the
restated OpenCV primitives."""
import os

import numpy as np
import pytest
from scipy import ndimage

from conftest import ROOT, synthetic_frame


def test_constructor_tables(oracle):
    ex = oracle.OrbExtractor(2000, 1.2, 8, 20, 7)
    assert list(ex.features_per_level) == [434, 362, 302, 251, 209, 175, 145, 122]      # ORBextractor.cc:440-452
    assert list(ex.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]   # :460-474
    want = np.array([1, 1.2, 1.44, 1.7280002, 2.0736003, 2.4883204, 2.9859846, 3.5831816], np.float32)
    assert np.array_equal(ex.scale, want)                                                # f32 chain :424-429
    assert np.array_equal(ex.sigma2, ex.scale * ex.scale) and np.array_equal(ex.inv_scale, np.float32(1) / ex.scale)


def test_pyramid_sizes_and_border(oracle, kitti_like_bgr):
    gray = oracle.bgr2gray(kitti_like_bgr)
    ex = oracle.OrbExtractor()
    ex(gray)
    sizes = [ex.level(l).shape for l in range(8)]
    assert sizes == [(352, 1024), (293, 853), (244, 711), (204, 593), (170, 494), (141, 412), (118, 343), (98, 286)]
    assert sum(h * w for h, w in sizes) == 1115407
    assert np.array_equal(ex.level(0), gray)
    full = ex.level(3, with_border=True); inner = ex.level(3)
    assert np.array_equal(full, np.pad(inner, 19, mode="reflect"))       # BORDER_REFLECT_101 == numpy 'reflect'


def test_golden_kitti_output_is_frozen(oracle, kitti_like_bgr):
    g = np.load(os.path.join(ROOT, "tests", "golden", "orb_kitti_golden.npz"))
    kps, desc = oracle.OrbExtractor()(oracle.bgr2gray(kitti_like_bgr))
    assert kps.tobytes() == g["keypoints"].tobytes() and np.array_equal(desc, g["descriptors"])
    assert len(kps) == 2006
    assert list(np.bincount(kps["octave"])) == [434, 362, 303, 253, 211, 175, 145, 123]
    assert kps["x"].min() >= 19 and (kps["size"] == (31 * oracle.OrbExtractor().scale[kps["octave"]]).astype(np.int32)).all()


def test_resize_is_bilinear_within_one_lsb(oracle):
    src = synthetic_frame(3, 120, 200)
    dst = oracle.resize_linear_u8(src, 100, 167)
    sy, sx = 120 / 100, 200 / 167
    yy = np.clip((np.arange(100) + 0.5) * sy - 0.5, 0, 119); xx = np.clip((np.arange(167) + 0.5) * sx - 0.5, 0, 199)
    ref = ndimage.map_coordinates(src.astype(np.float64), np.meshgrid(yy, xx, indexing="ij"), order=1, mode="nearest")
    assert np.abs(dst.astype(np.float64) - ref).max() <= 1.0
    assert np.array_equal(oracle.resize_linear_u8(src, 120, 200), src)      # identity scale


def test_gaussian_is_the_8bit_fixed_point_kernel(oracle):
    src = synthetic_frame(4, 64, 80)
    dst = oracle.gaussian7_u8(src)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)                   # round(gauss(sigma 2) * 256), sum 257
    pad = np.pad(src.astype(np.int64), 3, mode="reflect")
    rows = sum(k[i] * pad[3:-3, i:i + 80] for i in range(7))
    rows = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    ref = (sum(k[i] * rows[i:i + 64] for i in range(7)) + (1 << 15)) >> 16
    assert np.array_equal(dst, np.clip(ref, 0, 255).astype(np.uint8))
    flat = oracle.gaussian7_u8(np.full((20, 20), 100, np.uint8))
    assert (flat == (100 * 257 * 257 + 32768) // 65536).all()              # the 257/256 gain of that OpenCV path


def test_gaussian_variant_of_newer_opencv(oracle):
    """OpenCV >= 3.4.13 / >= 4.5.1 (getGaussianKernelFixedPoint_ED): the rounding error of each tap is carried into the next from the
    outside in and the centre takes the rest of 256 — taps 18 34 48 56 48 34 18, no brightness gain; the arithmetic is the same 8.8
    fixed point.  (The default "rounded" variant is OpenCV 3.2 - 3.4.12 / 4.0 - 4.5.0.)"""
    assert oracle.gaussian7_taps("rounded") == [18, 34, 49, 55, 49, 34, 18]
    assert oracle.gaussian7_taps("ed") == [18, 34, 48, 56, 48, 34, 18] and sum(oracle.gaussian7_taps("ed")) == 256
    src = synthetic_frame(4, 64, 80)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    pad = np.pad(src.astype(np.int64), 3, mode="reflect")
    rows = sum(k[i] * pad[3:-3, i:i + 80] for i in range(7))
    rows = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    ref = (sum(k[i] * rows[i:i + 64] for i in range(7)) + (1 << 15)) >> 16
    dst = oracle.gaussian7_u8(src, "ed")
    assert np.array_equal(dst, np.clip(ref, 0, 255).astype(np.uint8))
    assert (oracle.gaussian7_u8(np.full((20, 20), 100, np.uint8), "ed") == 100).all()
    d = dst.astype(int) - oracle.gaussian7_u8(src, "rounded").astype(int)
    assert np.abs(d).max() <= 2 and (d != 0).mean() > 0.2                  # the two OpenCV generations differ by an LSB or two on many pixels
    kp_r, de_r = oracle.OrbExtractor(nfeatures=300, nlevels=4)(synthetic_frame(5, 120, 160))
    kp_e, de_e = oracle.OrbExtractor(nfeatures=300, nlevels=4, gaussian="ed")(synthetic_frame(5, 120, 160))
    assert kp_r.tobytes() == kp_e.tobytes()                                # keypoints come from the unblurred pyramid ...
    assert not np.array_equal(de_r, de_e)                                  # ... descriptors from the blurred one


def test_fast_against_bruteforce_definition(oracle):
    img = synthetic_frame(5, 60, 70)
    xy, sc = oracle.fast9_16(img, 20, nonmax=False)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    found = set(map(tuple, xy))
    score = {}
    for y in range(3, 57):
        for x in range(3, 67):
            v = int(img[y, x]); r = [int(img[y + dy, x + dx]) for dx, dy in ring]
            best = -1
            for s in range(16):
                arc = [r[(s + i) % 16] for i in range(9)]
                best = max(best, min(v - a for a in arc) - 1, min(a - v for a in arc) - 1)
            if best >= 20: score[(x, y)] = best
    assert found == set(score)
    assert all(score[tuple(p)] == s for p, s in zip(xy, sc))
    xy_n, sc_n = oracle.fast9_16(img, 20, nonmax=True)
    smap = np.zeros(img.shape, int)
    for (x, y), s in score.items(): smap[y, x] = s
    keep = [(x, y) for (x, y), s in score.items() if all(s > smap[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy or dx))]
    assert sorted(map(tuple, xy_n)) == sorted(keep)
    assert list(map(tuple, xy_n)) == sorted(map(tuple, xy_n), key=lambda p: (p[1], p[0]))    # raster emission order


def test_fast_atan2_and_cvround(oracle):
    rng = np.random.default_rng(0)
    for _ in range(500):
        y, x = rng.normal(0, 1000, 2)
        a = oracle.fast_atan2(y, x)
        assert abs(((a - np.degrees(np.arctan2(y, x))) + 180) % 360 - 180) < 0.3
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    lib = oracle.lib()
    assert [lib.orc_cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999)] == [0, 2, 2, 0, -2, 2]   # half to even


def test_quadtree_invariants(oracle):
    rng = np.random.default_rng(1)
    n = 3000
    keys = np.zeros(n, oracle.KP_DTYPE)
    keys["x"] = rng.integers(0, 992, n); keys["y"] = rng.integers(0, 320, n); keys["response"] = rng.integers(7, 200, n)
    out = oracle.distribute_octtree(keys, 16, 1008, 16, 336, 434)
    assert 434 <= len(out) <= 434 + 3
    pts = set(zip(out["x"], out["y"], out["response"]))
    assert pts <= set(zip(keys["x"], keys["y"], keys["response"]))
    few = oracle.distribute_octtree(keys[:10], 16, 1008, 16, 336, 434)
    assert len(few) == len(set(zip(keys["x"][:10], keys["y"][:10]))) or len(few) <= 10


def test_extractor_edge_cases(oracle):
    ex = oracle.OrbExtractor()
    kps, desc = ex(np.full((352, 1024), 77, np.uint8))                     # no corners at all
    assert len(kps) == 0
    kps, desc = oracle.OrbExtractor(nfeatures=100, nlevels=2)(synthetic_frame(8, 100, 120))
    assert 0 < len(kps) <= 110 and desc.shape == (len(kps), 32)
    assert kps["angle"].min() >= 0 and kps["angle"].max() < 360
