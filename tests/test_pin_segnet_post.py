"""BayesianSegNet around the forward pass, pinned against the reference's OWN code.

`make -C oracle ref` compiles /root/reference/src/bayesian_segnet/bayesian_segnet.cpp as it is into
oracle/_ref/libref_segnet.so, with a stand-in for Caffe's network whose Forward() copies in the softmax probabilities the
test supplies (Caffe cannot be built here) and stand-ins for Eigen's Tensor module and OpenCV.  Pinned bit for bit: what the
input layer is fed (centre crop, u8 -> f32, channel split into every batch slot), the f64 mean over the T samples, argmax
(first maximum wins), max, the classification entropy, computeVariance, and the constructor's exceptions.  The forward pass
itself — the network — stays unpinned."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_segnet.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "segnet_post_reference.json")
HAVE_REF = os.path.exists(REF_LIB)
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/libref_segnet.so not built (no /root/reference here)")
_vp = lambda a: C.c_void_p(a.ctypes.data)


def probabilities(T, K, H, W, seed):
    """Softmax outputs with the awkward cases in: exact two-way and K-way ties of the mean, classes at exactly 0."""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    prob = O.softmax(rng.normal(0, 3, (T, K, H, W)).astype(np.float32))
    prob[:, :, 0, :8] = 0; prob[:, 3, 0, :8] = 0.5; prob[:, 7, 0, :8] = 0.5         # tie between classes 3 and 7, the others 0
    prob[:, :, 1, :4] = np.float32(1.0 / K)                                          # all classes equal
    prob[:, :, 2, :4] = 0; prob[:, K - 1, 2, :4] = 1.0                               # one-hot: entropy 0
    return np.ascontiguousarray(prob, np.float32)


def reference(prob, image, want_variance=True):
    T, K, H, W = prob.shape
    lib = C.CDLL(REF_LIB)
    data = np.zeros((T, 3, H, W), np.float32); cls = np.zeros((H, W), np.uint8)
    conf = np.zeros((H, W)); ent = np.zeros((H, W)); var = np.zeros((H, W))
    rc = lib.ref_segnet_segment(T, K, H, W, _vp(prob), _vp(image), image.shape[0], image.shape[1], _vp(data), _vp(cls), _vp(conf), _vp(ent),
                                _vp(var) if want_variance else None)
    assert rc == 0
    return dict(data=data, classes=cls, confidence=conf, entropy=ent, variance=var)


CASES = [(12, 15, 40, 72, 0, (51, 79)), (6, 15, 33, 64, 1, (33, 64)), (2, 15, 16, 24, 2, (40, 25)), (12, 15, 352, 1024, 3, (376, 1241))]


def _digest(d):
    return {k: hashlib.sha256(np.ascontiguousarray(d[k]).tobytes()).hexdigest()[:32] for k in ("data", "classes", "confidence", "entropy", "variance")}


def _oracle(prob, image):
    from oracle import oracle as O
    T, K, H, W = prob.shape
    mean = O.mc_mean(prob)
    cls, conf, ent = O.mc_finalize(mean)
    return dict(data=O.preprocess(image, T, H, W), classes=cls, confidence=conf, entropy=ent, variance=O.mc_variance(prob, cls))


def test_oracle_pre_and_post_processing_equal_the_reference_class():
    golden = json.load(open(GOLDEN))
    for T, K, H, W, seed, (rows, cols) in CASES:
        prob = probabilities(T, K, H, W, seed)
        image = np.random.default_rng(100 + seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        got = _oracle(prob, image)
        key = f"T{T}_{H}x{W}_seed{seed}"
        assert _digest(got) == golden[key], key
        assert (got["classes"][0, :8] == 3).all() and (got["entropy"][0, :8] == 1.0).all()          # first maximum wins; 2 x 0.5 -> 1 bit
        assert (got["classes"][1, :4] == 0).all() and (got["entropy"][2, :4] == 0.0).all()
        if HAVE_REF:
            ref = reference(prob, image)
            for k in got:
                assert np.array_equal(ref[k], got[k]), (key, k)
            assert _digest(ref) == golden[key], "stale golden"


@needs_ref
def test_reference_constructor_exceptions_match_this_library():
    """bayesian_segnet.cpp:65-70, 80-89 — the messages this library's constructor reproduces (tests/cpp/test_api.cpp)."""
    lib = C.CDLL(REF_LIB)
    what = C.create_string_buffer(200)
    expect = [((b"", b"w", 12, 3), b"model_file (.prototxt file) is empty!"), ((b"m", b"", 12, 3), b"weights_file (.caffemodel file) is empty!"),
              ((b"m", b"w", 12, 4), b"Input layer must have 3 channels!"), ((b"m", b"w", 1, 3), b"Input layer must have a batch size greater than 1!")]
    for args, msg in expect:
        assert lib.ref_segnet_construct(*args, what, 200) == 1 and what.value == msg
    assert lib.ref_segnet_construct(b"m", b"w", 2, 3, what, 200) == 0
    src = "".join(open(os.path.join(ROOT, "sivo_amd", "csrc", f)).read() for f in ("segnet.cpp", "segnet_plan.cpp"))      # (C ABI checks / plan construction)
    for _, msg in expect:
        assert msg.decode() in src, msg


@pytest.mark.gpu
def test_device_maps_agree_with_the_reference_post_processing(oracle, kitti_like_bgr):
    """The device's classes / confidence / entropy against the reference's post-processing fed with the device's own logits
    (softmax by the oracle): the MC reduction on the GPU vs extractMeanConfidence / computeClasses / computeMaxConfidence /
    computeClassificationEntropy of the reference."""
    import torch
    from oracle import oracle as O, prototxt as oproto
    from sivo_amd import netspec, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    T, H, W = 6, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net = oproto.parse(text)
    w = wts.synth_weights(net["layers"], 42)
    sn = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T)
    img = np.ascontiguousarray(kitti_like_bgr[:H, :W])
    prob_sum, logits, _ = sn.forward(torch.from_numpy(img).cuda(), 11, want_logits=True)
    cls, conf, ent = (t.cpu().numpy() for t in sn.finalize(prob_sum))
    prob = O.softmax(logits.cpu().numpy())
    post = reference(prob, img, want_variance=False) if HAVE_REF else _oracle(prob, img)
    np.testing.assert_allclose(conf, post["confidence"], atol=2e-7, rtol=0)
    np.testing.assert_allclose(ent, post["entropy"], atol=5e-6, rtol=0)
    top2 = np.sort(O.mc_mean(prob), axis=0)[-2:]
    clear = (top2[1] - top2[0]) > 1e-6
    assert np.array_equal(cls[clear], post["classes"][clear]) and clear.mean() > 0.99
