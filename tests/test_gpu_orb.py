"""GPU parity (bit-exact) of the ORB extractor and stereo matching vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import synthetic_frame, synthetic_stereo
from sivo_amd import orb

pytestmark = pytest.mark.gpu


def _compare(oracle, gray, **kw):
    ex_o = oracle.OrbExtractor(**{k: v for k, v in kw.items()})
    names = {"nfeatures": "nfeatures", "scale_factor": "scale_factor", "nlevels": "nlevels", "ini_th": "ini_th_fast", "min_th": "min_th_fast", "gaussian": "gaussian"}
    ex_g = orb.ORBextractor(**{names[k]: v for k, v in kw.items()})
    kp_o, d_o = ex_o(gray)
    kp_g, d_g = ex_g(gray)
    for l in range(ex_o.nlevels):
        assert np.array_equal(ex_g.image_pyramid(l, with_border=True), ex_o.level(l, with_border=True)), f"pyramid level {l}"
        co, cg = ex_o.candidates(l), ex_g.candidates(l)
        assert len(co) == len(cg) and co.tobytes() == cg.tobytes(), f"FAST candidates level {l}"
    assert len(kp_o) == len(kp_g)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kp_o[f], kp_g[f]), f
    assert np.array_equal(kp_o["angle"].view(np.uint32), kp_g["angle"].view(np.uint32)), "angle bits"
    assert np.array_equal(d_o, d_g), "descriptors"
    return ex_o, ex_g, kp_g, d_g


def test_orb_kitti_frame_bit_exact(oracle, kitti_like_bgr):
    """BASELINE config 1: 2000 features / 8 levels on the KITTI frame."""
    gray = oracle.bgr2gray(kitti_like_bgr)
    ex_o, ex_g, kp, d = _compare(oracle, gray)
    assert 1900 <= len(kp) <= 2100
    assert list(ex_g.features_per_level) == [434, 362, 302, 251, 209, 175, 145, 122]


def test_orb_synthetic_bit_exact(oracle):
    _compare(oracle, synthetic_frame(1234))


@pytest.mark.parametrize("mode", [1, 2, 4, 8, 14, 15])
def test_orb_launch_modes_are_bit_identical(oracle, kitti_like_bgr, mode):
    """sivo_orb_set_launch_mode: the pyramid in one launch (bit 0: footprints recomputed in LDS), FAST + scan + emission in one launch
    (bit 1: a cell waits for the cells before it), blur + borders (bit 2), angle + descriptor (bit 3) against the oracle on two frames and
    a geometry with few levels, and every level, candidate list, key and descriptor against mode 0 (the fifteen launches of rounds 1 - 5)
    on the same extractor parameters."""
    for gray, kw in ((oracle.bgr2gray(kitti_like_bgr), {}), (synthetic_frame(9), {}), (synthetic_frame(5, 97, 131), dict(nfeatures=50, nlevels=3, scale_factor=1.5))):
        ex_o = oracle.OrbExtractor(**kw)
        names = {"nfeatures": "nfeatures", "scale_factor": "scale_factor", "nlevels": "nlevels"}
        a = orb.ORBextractor(launch_mode=0, **{names[k]: v for k, v in kw.items()})
        b = orb.ORBextractor(launch_mode=mode, **{names[k]: v for k, v in kw.items()})
        kp_o, d_o = ex_o(gray)
        (kp_a, d_a), (kp_b, d_b) = a(gray), b(gray)
        for _ in range(3):                      # (the one-launch FAST keeps an epoch across calls)
            kp_b2, d_b2 = b(gray)
            assert kp_b2.tobytes() == kp_b.tobytes() and np.array_equal(d_b2, d_b)
        for l in range(ex_o.nlevels):
            assert np.array_equal(b.image_pyramid(l, with_border=True), a.image_pyramid(l, with_border=True)), f"pyramid level {l}"
            assert b.candidates(l).tobytes() == a.candidates(l).tobytes() == ex_o.candidates(l).tobytes(), f"FAST candidates level {l}"
        assert kp_a.tobytes() == kp_b.tobytes() == kp_o.tobytes() and np.array_equal(d_a, d_b) and np.array_equal(d_a, d_o)


def test_orb_gaussian_taps_of_newer_opencv_bit_exact(oracle, kitti_like_bgr):
    """sivo_orb_set_gaussian(1): the descriptor image blurred with the error-diffused taps of OpenCV >= 3.4.13 / >= 4.5.1 (18 34 48 56 ...):
    bit-exact against the oracle's restatement of that variant, same keypoints as the default variant, other descriptors."""
    gray = oracle.bgr2gray(kitti_like_bgr)
    _, _, kp_e, d_e = _compare(oracle, gray, gaussian="ed")
    _, _, kp_r, d_r = _compare(oracle, gray, gaussian="rounded")
    assert kp_e.tobytes() == kp_r.tobytes() and not np.array_equal(d_e, d_r)
    _compare(oracle, synthetic_frame(5, 97, 131), nfeatures=50, nlevels=3, scale_factor=1.5, gaussian="ed")


def test_extract_pair_equals_two_extractions():
    """sivo_orb_extract_pair_dev (the two ExtractORB threads of Frame::Frame, Frame.cc:126-131): the same keys and descriptors as two calls."""
    left, right = synthetic_stereo(21, disparity=8)
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    a, b = orb.ORBextractor(), orb.ORBextractor()
    for _ in range(3):
        (kl, el), (kr, er) = orb.extract_pair(a, b, dl, dr)
        kl1, el1 = a(dl); kr1, er1 = b(dr)
        assert kl.tobytes() == kl1.tobytes() and np.array_equal(el, el1) and kr.tobytes() == kr1.tobytes() and np.array_equal(er, er1)
    with pytest.raises(Exception):
        orb.extract_pair(a, a, dl, dr)           # one extractor holds one image


def test_orb_device_resident_input(oracle):
    gray = synthetic_frame(77)
    ex = orb.ORBextractor()
    kp_h, d_h = ex(gray)
    kp_d, d_d = ex(torch.from_numpy(gray).cuda())
    assert kp_h.tobytes() == kp_d.tobytes() and np.array_equal(d_h, d_d)


@pytest.mark.parametrize("shape,kw", [((120, 160), dict(nfeatures=300, nlevels=4)),
                                       ((375, 1242), dict(nfeatures=1000, ini_th=12)),
                                       ((97, 131), dict(nfeatures=50, nlevels=3, scale_factor=1.5))])
def test_orb_other_geometries(oracle, shape, kw):
    _compare(oracle, synthetic_frame(5, *shape), **kw)


def test_orb_flat_and_low_contrast_images(oracle):
    """No corners anywhere -> zero keypoints; low contrast -> the minThFAST fallback path."""
    ex = orb.ORBextractor()
    kp, d = ex(np.full((352, 1024), 128, np.uint8))
    assert len(kp) == 0 and d.shape == (0, 32)
    rng = np.random.default_rng(0)
    low = (128 + rng.integers(-9, 10, (352, 1024))).astype(np.uint8)
    _compare(oracle, low)
    kp, d = ex(np.zeros((0, 0), np.uint8))
    assert len(kp) == 0


def test_stereo_matches(oracle):
    L, R = synthetic_stereo(21, disparity=8)
    eo_l, eo_r = oracle.OrbExtractor(), oracle.OrbExtractor()
    kl, dl = eo_l(L); kr, dr = eo_r(R)
    bf, b = 386.1448, 386.1448 / 718.856
    pyrL = [eo_l.level(l) for l in range(8)]; pyrR = [eo_r.level(l) for l in range(8)]
    uR_o, depth_o, best_o, kept = oracle.stereo_matches(kl, dl, kr, dr, eo_l.scale, eo_l.inv_scale, pyrL, pyrR, bf, b)
    eg_l, eg_r = orb.ORBextractor(), orb.ORBextractor()
    kgl, dgl = eg_l(L); kgr, dgr = eg_r(R)
    assert kgl.tobytes() == kl.tobytes() and kgr.tobytes() == kr.tobytes()
    uR, depth, best = orb.stereo_match(eg_l, eg_r, kgl, dgl, kgr, dgr, bf, b)
    assert np.array_equal(best, best_o)
    assert np.array_equal(uR.view(np.uint32), uR_o.view(np.uint32))
    assert np.array_equal(depth.view(np.uint32), depth_o.view(np.uint32))
    assert kept > 200 and abs(np.median((kl["x"] - uR)[uR >= 0]) - 8) < 0.5
    # two-step form: matching every left keypoint first and culling over a kept subset afterwards gives, bit for bit,
    # what the one-call form (and the oracle) give on that subset alone — the property bench.py relies on to run the
    # matching beside the network and only the median cull after SelectSemanticKeys
    keep = np.random.default_rng(1).random(len(kgl)) < 0.4
    uR_a, depth_a, best_a, sad = orb.stereo_match_begin(eg_l, eg_r, kgl, dgl, kgr, dgr, bf, b)
    assert np.array_equal(best_a, best)
    orb.stereo_match_cull(keep, sad, uR_a, depth_a)
    uR_s, depth_s, best_s = orb.stereo_match(eg_l, eg_r, kgl[keep], dgl[keep], kgr, dgr, bf, b)
    uR_so, depth_so, _, _ = oracle.stereo_matches(kl[keep], dl[keep], kr, dr, eo_l.scale, eo_l.inv_scale, pyrL, pyrR, bf, b)
    assert np.array_equal(uR_a[keep].view(np.uint32), uR_s.view(np.uint32)) and np.array_equal(uR_s.view(np.uint32), uR_so.view(np.uint32))
    assert np.array_equal(depth_a[keep].view(np.uint32), depth_s.view(np.uint32)) and (uR_a[~keep] == -1).all()
    uR_n, depth_n, _, sad_n = orb.stereo_match_begin(eg_l, eg_r, kgl, dgl, kgr, dgr, bf, b)
    orb.stereo_match_cull(None, sad_n, uR_n, depth_n)
    assert np.array_equal(uR_n.view(np.uint32), uR.view(np.uint32))
