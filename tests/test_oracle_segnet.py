"""CPU: the SegNet oracle against an independent second opinion (PyTorch-CPU ops) and known answers.
The reference pins nothing numerically for this path (tests/test_bayesian_segnet.cpp:152-168 asserts
sizes only), so these are the checks that pin the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import prototxt as oproto
from sivo_amd import netspec, weights as wts


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert [hex(v) for v in oracle.philox4x32_10([0] * 4, [0] * 2)] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in oracle.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_dropout_mask_definition(oracle):
    """keep bit of element e = bit (e & 31) of word (e >> 5) & 3 of Philox(ctr={e >> 7, site, sample, 0}, key=seed)."""
    x = np.ones((2, 1, 8, 64), np.float32)
    seed, site, s0 = 0x1234567890abcdef, 3, 9
    y = oracle.dropout(x, site, s0, seed)
    for n in range(2):
        for e in (0, 1, 31, 32, 127, 128, 300, 511):
            w = oracle.philox4x32_10([e >> 7, site, s0 + n, 0], [seed & 0xffffffff, seed >> 32])
            bit = (int(w[(e >> 5) & 3]) >> (e & 31)) & 1
            assert y[n].ravel()[e] == (2.0 if bit else 0.0)
    big = oracle.dropout(np.ones((4, 8, 64, 64), np.float32), 0, 0, 1)
    assert abs((big > 0).mean() - 0.5) < 0.01 and set(np.unique(big)) == {0.0, 2.0}
    assert not np.array_equal(big[0], big[1])            # every MC sample draws its own mask


@pytest.mark.parametrize("k,pad", [(3, 1), (7, 3), (1, 0)])
def test_conv_matches_torch(oracle, k, pad):
    rng = np.random.default_rng(k)
    x = rng.standard_normal((2, 5, 13, 20)).astype(np.float32)
    w = rng.standard_normal((7, 5, k, k)).astype(np.float32); b = rng.standard_normal(7).astype(np.float32)
    ref = F.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), padding=pad).numpy()
    np.testing.assert_allclose(oracle.conv2d(x, w, b, pad), ref, atol=1e-4)
    np.testing.assert_allclose(oracle.conv2d(x, w, b, pad, acc64=True), ref, atol=1e-4)


def test_pool_unpool_lrn_softmax_bn_match_torch(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 12, 20)).astype(np.float32)
    x[0, 0, 0:2, :4] = 9.0                                 # ties: the first maximum in scan order wins
    p, m = oracle.maxpool(x)
    tp, tm = F.max_pool2d(torch.tensor(x), 2, 2, return_indices=True)
    assert np.array_equal(p, tp.numpy()) and np.array_equal(m, tm.numpy())
    assert m[0, 0, 0, 0] == 0 and m[0, 0, 0, 1] == 2
    u = oracle.unpool(p, m, 12, 20)
    assert np.array_equal(u, F.max_unpool2d(tp, tm, 2, 2, output_size=(12, 20)).numpy())
    assert (u != 0).sum() <= p.size
    x3 = (rng.random((2, 3, 8, 8)) * 255).astype(np.float32)
    np.testing.assert_allclose(oracle.lrn(x3, 5, 1e-4, 0.75), F.local_response_norm(torch.tensor(x3), 5, 1e-4, 0.75, 1.0).numpy(), rtol=1e-6)
    np.testing.assert_allclose(oracle.softmax(x), F.softmax(torch.tensor(x), 1).numpy(), atol=1e-7)
    s, sh = rng.random(6).astype(np.float32), rng.random(6).astype(np.float32)
    np.testing.assert_allclose(oracle.bn_inference(x, s, sh), x * s[None, :, None, None] + sh[None, :, None, None], atol=1e-6)
    assert np.array_equal(oracle.relu(x), np.maximum(x, 0))


def test_odd_sized_pooling_is_ceil_mode(oracle):
    x = np.arange(2 * 5 * 7, dtype=np.float32).reshape(1, 2, 5, 7)
    p, m = oracle.maxpool(x)
    tp, tm = F.max_pool2d(torch.tensor(x), 2, 2, ceil_mode=True, return_indices=True)
    assert p.shape == (1, 2, 3, 4) and np.array_equal(p, tp.numpy()) and np.array_equal(m, tm.numpy())


def test_mc_known_answers(oracle):
    """bayesian_segnet.cpp:38-44,180-203,262-276: entropy in bits, exact-zero guard, first-wins argmax."""
    K, H, W = 15, 4, 6
    uni = np.full((3, K, H, W), 1 / 15, np.float32)
    cls, conf, ent = oracle.mc_finalize(oracle.mc_mean(uni))
    assert (cls == 0).all()                                # tie -> first index (tests/test_bayesian_segnet.cpp:43-136)
    np.testing.assert_allclose(ent, np.log2(15), atol=1e-6)    # 3.9069 bits
    onehot = np.zeros((2, K, H, W), np.float32); onehot[:, 7] = 1
    cls, conf, ent = oracle.mc_finalize(oracle.mc_mean(onehot))
    assert (cls == 7).all() and (conf == 1).all() and (ent == 0).all()
    # two-class coin flip across samples: mean (0.5, 0.5) -> 1 bit, class = the lower index
    coin = np.zeros((2, K, H, W), np.float32); coin[0, 3] = 1; coin[1, 9] = 1
    cls, conf, ent = oracle.mc_finalize(oracle.mc_mean(coin))
    assert (cls == 3).all() and (conf == 0.5).all() and np.allclose(ent, 1.0)
    var = oracle.mc_variance(coin, cls)
    assert np.allclose(var, 0.5)                           # sample variance of {1, 0} with n-1


def test_argmax_semantics_of_the_reference_eigen_test(oracle):
    """EigenTests.ArgmaxTest (tests/test_bayesian_segnet.cpp:43-136), the part computeClasses relies on (argmax over
    dimension 0 of a (15, 352, 1024) tensor): values log(u + 0.5) with u in [0, 1) stay below 0.41, so a planted 10.0 in
    the first class gives index 0 everywhere, a planted 20.0 in the last class then gives 14 everywhere, and the result has
    352 * 1024 entries."""
    rng = np.random.default_rng(5)
    mean = np.log(rng.random((15, 352, 1024)) + 0.5)
    mean[0] = 10.0
    cls, conf, _ = oracle.mc_finalize(mean)
    assert cls.size == 352 * 1024 and cls.dtype == np.uint8 and (cls == 0).all() and (conf == 10.0).all()
    mean[14] = 20.0
    cls, conf, _ = oracle.mc_finalize(mean)
    assert (cls == 14).all() and (conf == 20.0).all()


def test_mean_is_taken_in_f64(oracle):
    p = np.zeros((3, 2, 1, 1), np.float32); p[:, 0] = np.float32(0.1); p[:, 1] = np.float32(0.9)
    mean = oracle.mc_mean(p)
    assert mean.dtype == np.float64 and mean[0, 0, 0] == (3 * float(np.float32(0.1))) / 3


def test_preprocess_centre_crop_rule(oracle, kitti_like_bgr):
    """resizeImage (bayesian_segnet.cpp:142-162) / System.cc:161-163: x_tl = cols/2 - W/2."""
    img = np.arange(10 * 12 * 3, dtype=np.uint8).reshape(10, 12, 3)
    blob = oracle.preprocess(img, 2, 4, 6)
    y0, x0 = 10 // 2 - 2, 12 // 2 - 3
    assert blob.shape == (2, 3, 4, 6)
    for c in range(3):
        assert np.array_equal(blob[0, c], img[y0:y0 + 4, x0:x0 + 6, c].astype(np.float32))
    assert np.array_equal(blob[0], blob[1])                # the same image in every MC slot
    assert oracle.preprocess(img, 2, 11, 6) is None        # smaller than the net: empty Mat in the reference
    assert oracle.preprocess(kitti_like_bgr, 2, 352, 1024).max() <= 255.0


def test_tiny_net_end_to_end_against_torch(oracle):
    """The whole oracle forward (layer order, BN/ReLU placement, mask routing) vs a torch re-implementation
    with the oracle's dropout masks injected."""
    T, H, W = 2, 16, 32
    net = oproto.parse(netspec.tiny_prototxt(T, H, W))
    w = wts.synth_weights(net["layers"], 3)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    blob = oracle.preprocess(img, T, H, W)
    ob = oracle.run_net(net, w, blob, seed=5)
    t = lambda a: torch.tensor(np.asarray(a))
    x = F.local_response_norm(t(blob), 5, 1e-4, 0.75, 1.0)
    x = F.relu(F.conv2d(x, t(w["c1"][0]), t(w["c1"][1]), padding=1) * t(w["c1_bn"][0])[None, :, None, None] + t(w["c1_bn"][1])[None, :, None, None])
    x, m1 = F.max_pool2d(x, 2, 2, return_indices=True)
    x = F.relu(F.conv2d(x, t(w["c2"][0]), t(w["c2"][1]), padding=1) * t(w["c2_bn"][0])[None, :, None, None] + t(w["c2_bn"][1])[None, :, None, None])
    x, m2 = F.max_pool2d(x, 2, 2, return_indices=True)
    x = x * t((ob["p2"] != 0) | (x.numpy() == 0)).float() * 2     # inject the oracle's site-0 mask
    np.testing.assert_allclose(x.numpy(), ob["p2"], atol=1e-4)
    x = F.max_unpool2d(x, m2, 2, 2)
    x = F.relu(F.conv2d(x, t(w["d2"][0]), t(w["d2"][1]), padding=3))
    keep = t(ob["d2"] != 0).float()
    x = x * keep * 2
    x = F.max_unpool2d(x, m1, 2, 2)
    x = F.relu(F.conv2d(x, t(w["d1"][0]), t(w["d1"][1]), padding=1) * t(w["d1_bn"][0])[None, :, None, None] + t(w["d1_bn"][1])[None, :, None, None])
    x = F.conv2d(x, t(w["cls"][0]), t(w["cls"][1]))
    np.testing.assert_allclose(x.numpy(), ob["cls"], atol=2e-4)
    np.testing.assert_allclose(F.softmax(x, 1).numpy(), ob["__last__"], atol=1e-5)


@pytest.mark.parametrize("N,Cin,Cout,H,W,k", [(2, 3, 64, 13, 21, 3), (1, 64, 15, 9, 12, 3), (2, 5, 7, 10, 33, 7), (1, 128, 130, 16, 32, 3), (1, 8, 8, 8, 64, 3)])
def test_caffe_style_convolution_of_the_cpu_baseline_agrees_with_the_oracle(oracle, N, Cin, Cout, H, W, k):
    """oracle/caffe_cpu.c (im2col + blocked SGEMM per image: bench.py's reference-equivalent CPU timing baseline) against the oracle's
    chain and against the f64-accumulated convolution: the same numbers up to float round-off of another summation order (ragged
    strips, row-crossing strips, cout not a multiple of the register tile, 7x7 taps)."""
    rng = np.random.default_rng(N * 1000 + Cin)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = oracle.conv2d(x, w, b, k // 2, acc64=True)
    got = oracle.caffe_conv2d(x, w, b, k // 2)
    assert np.abs(got - ref).max() < 2e-5 and np.abs(got - ref).max() <= 4 * max(np.abs(oracle.conv2d(x, w, b, k // 2) - ref).max(), 1e-6)
    assert np.array_equal(oracle.caffe_conv2d(x, w, None, k // 2) + b[None, :, None, None], got) or np.abs(oracle.caffe_conv2d(x, w, None, k // 2) + b[None, :, None, None] - got).max() < 1e-6
