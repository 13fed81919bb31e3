"""GPU parity of the Bayesian SegNet path against the CPU oracle (through the C ABI).

Tolerances: logits max-abs 1e-3 (BASELINE.json north_star: "segmentation logits within
1e-3 fp32"); probabilities / confidence / entropy 1e-5 (fp32 softmax, expf 1-ulp
differences between libm and the device); classes exact except where the top-2 mean
probabilities are closer than 1e-5."""
import contextlib
import os

import numpy as np
import pytest
import torch

from oracle import prototxt as oproto
from sivo_amd import netspec, weights as wts
from sivo_amd.segnet import BayesianSegNet, mc_finalize, mc_reduce, mc_segment, mc_variance

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3



# The product library reads eight environment switches; every A/B or fault-injection switch exists in the
# diagnostic build only (libsivo_hip_diag.so: the same sources compiled with -DSIVO_DIAG).  Objects created inside the block live there.
# (handle options the Python wrapper takes from the environment: sivo_amd/segnet.py segnet_options; anything else is a diagnostic-build switch)
_PRODUCT_SWITCHES = {"SIVO_LANES", "SIVO_GEMM", "SIVO_D3", "SIVO_D3_PK", "SIVO_CONV7", "SIVO_WINO4_MB", "SIVO_DEBUG_SYNC"}


@contextlib.contextmanager
def _diag(**env):
    from sivo_amd import _lib
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        with _lib.use("diag"):
            yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

def _make(text, T, seed=42):
    net = oproto.parse(text)
    w = wts.synth_weights(net["layers"], seed)
    flat = wts.pack(net["layers"], w)
    return net, w, BayesianSegNet(prototxt=text, weights=flat, T=T)


def _image(rng, H, W):
    return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)


def _check_outputs(oracle, res, cls, conf, ent, prob_sum, T, prob_tol=1e-5, ent_tol=2e-4):
    """prob_tol follows from the logit error: |dp| <= p (1 - p) * 2 |dlogit| <= |dlogit| / 2."""
    mean = res["mean"]
    np.testing.assert_allclose(prob_sum / T, mean, atol=prob_tol, rtol=0)
    np.testing.assert_allclose(conf, res["confidence"], atol=prob_tol, rtol=0)
    np.testing.assert_allclose(ent, res["entropy"], atol=ent_tol, rtol=0)
    srt = np.sort(mean, axis=0)
    decided = (srt[-1] - srt[-2]) > 2 * prob_tol
    assert (cls[decided] == res["classes"][decided]).all()
    assert decided.mean() > 0.99


def test_tiny_net_every_blob(oracle):
    T, H, W = 3, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    rng = np.random.default_rng(0)
    img = _image(rng, H, W)
    seed = 7
    blob = oracle.preprocess(img, T, H, W)
    ob = oracle.run_net(net, w, blob, seed)
    d_img = torch.from_numpy(img).cuda()
    prob_sum, logits, prob = sn.forward(d_img, seed, want_logits=True, want_prob=True)
    torch.cuda.synchronize()
    # intermediate blobs (shared ones are stored once on the device)
    for name in ["norm", "c1", "p1", "p1_mask", "c2", "p2", "p2_mask", "p2_D", "d2", "p1_D", "d1", "cls"]:
        g = sn.blob(name)
        o = ob[name].astype(np.float32)
        if g.shape[0] == 1 and o.shape[0] > 1:
            assert np.array_equal(o[0], o[1]) or name in ("p2",), name
            o = o[:1]
        if name.endswith("_mask"):
            assert np.array_equal(g, o), name
        else:
            np.testing.assert_allclose(g, o, atol=LOGIT_TOL, rtol=0, err_msg=name)
    np.testing.assert_allclose(logits.cpu().numpy(), ob["cls"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(prob.cpu().numpy(), ob["__last__"], atol=1e-5, rtol=0)
    cls, conf, ent = sn.finalize(prob_sum)
    res = {"mean": oracle.mc_mean(ob["__last__"])}
    res["classes"], res["confidence"], res["entropy"] = oracle.mc_finalize(res["mean"])
    _check_outputs(oracle, res, cls.cpu().numpy(), conf.cpu().numpy(), ent.cpu().numpy(), prob_sum.cpu().numpy(), T)


def test_dropout_masks_bit_exact(oracle):
    """Every dropout site: the zero pattern of the device blob equals the oracle's Philox mask."""
    T, H, W = 4, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    img = _image(np.random.default_rng(1), H, W)
    ob = oracle.run_net(net, w, oracle.preprocess(img, T, H, W), 1234567890123, sample0=5)
    sn.forward(torch.from_numpy(img).cuda(), 1234567890123, sample0=5)
    torch.cuda.synchronize()
    for name in ["p2", "d2"]:
        g, o = sn.blob(name), ob[name]
        assert np.array_equal(g == 0, o == 0), name
        assert 0.3 < (o == 0).mean() < 0.9


@pytest.mark.parametrize("kind", ["basic", "standard"])
def test_reference_nets_reduced_geometry(oracle, kind):
    """The two reference architectures at full width, reduced image size (oracle runs in seconds)."""
    T, H, W = 2, 32, 64
    text = netspec.basic_prototxt(T, H, W) if kind == "basic" else netspec.standard_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    img = _image(np.random.default_rng(2), H, W)
    seed = 99
    logits_name = "dense_softmax_inner_prod" if kind == "basic" else "conv1_1_D"
    res = oracle.segment(net, w, img, seed, logits_name=logits_name)
    prob_sum, logits, _ = sn.forward(torch.from_numpy(img).cuda(), seed, want_logits=True)
    cls, conf, ent = sn.finalize(prob_sum)
    torch.cuda.synchronize()
    lg = logits.cpu().numpy()
    assert np.abs(res["logits"]).max() > 0.1            # the comparison is not vacuous
    np.testing.assert_allclose(lg, res["logits"], atol=LOGIT_TOL, rtol=0)
    _check_outputs(oracle, res, cls.cpu().numpy(), conf.cpu().numpy(), ent.cpu().numpy(), prob_sum.cpu().numpy(), T)


@pytest.mark.parametrize("kind", ["basic", "standard"])
def test_full_size_frame_logits_within_tolerance(oracle, kind, kitti_like_bgr):
    """BASELINE configs[1] / configs[2] geometry (352 x 1024, full channel widths), two MC samples: every logit within
    the north-star tolerance (1e-3, fp32) of the CPU oracle, and the finalized maps consistent.  The oracle needs
    ~10 s on the GPU box's host cores for this."""
    T, H, W = 2, 352, 1024
    text = netspec.basic_prototxt(T, H, W) if kind == "basic" else netspec.standard_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    img = np.ascontiguousarray(kitti_like_bgr[:H, :W])
    assert img.shape == (H, W, 3)
    seed = 2024
    logits_name = "dense_softmax_inner_prod" if kind == "basic" else "conv1_1_D"
    prob_sum, logits, _ = sn.forward(torch.from_numpy(img).cuda(), seed, want_logits=True)
    cls, conf, ent = sn.finalize(prob_sum)
    torch.cuda.synchronize()
    # Max pooling is discontinuous: at ~12 M pooling windows a few dozen hold two elements equal to the last ulps and two
    # correct fp32 evaluations pick different ones, after which every logit in that switch's receptive field differs by
    # O(1) (measured free-running: 165 of 11.9 M switches differ, 1.2 % of the logits move by > 1e-3).  So the oracle is
    # run with the device's switches; each differing switch must be a genuine near-tie, and THEN every logit has to agree.
    masks = {L["top"][1]: sn.blob(L["top"][1]) for L in net["layers"] if L["type"] == "Pooling"}
    flips = {}
    res = oracle.segment(net, w, img, seed, logits_name=logits_name, force_masks=masks, flips=flips)
    total = 0
    for name, (count, gap, mag) in flips.items():
        total += count
        assert count <= 1e-4 * masks[name].size and gap <= 1e-4 * max(mag, 1.0), (name, count, gap, mag)
    err = np.abs(logits.cpu().numpy() - res["logits"])
    print(f"{kind}: {total} pooling switches differ (near-ties); max |dlogit| = {err.max():.3e}, mean = {err.mean():.3e}, "
          f"max |logit| = {np.abs(res['logits']).max():.2f}")
    assert np.abs(res["logits"]).max() > 0.5
    assert err.max() < LOGIT_TOL
    _check_outputs(oracle, res, cls.cpu().numpy(), conf.cpu().numpy(), ent.cpu().numpy(), prob_sum.cpu().numpy(), T,
                   prob_tol=LOGIT_TOL / 2, ent_tol=5e-3)


@pytest.mark.parametrize("H,W,width", [(22, 64, 256), (9, 12, 128), (44, 128, 128)])
def test_bridged_convolutions_are_bit_identical(H, W, width, monkeypatch):
    """conv -> conv at >= 128 channels: output transform + epilogue (BN, ReLU, dropout) + next input transform in one
    kernel through an LDS image of the channel plane, vs the three-kernel path that writes the activation to HBM.
    Same arithmetic in the same order -> identical logits; ragged tile rows (22, 9) included."""
    T = 3
    text = _conv_stack_prototxt(T, H, W, width)
    monkeypatch.setenv("SIVO_D3", "0")          # (128-channel layers would otherwise run the direct f16x3 kernel, which has no transforms to bridge)
    net, w, sn = _make(text, T, seed=5)
    img = torch.from_numpy(_image(np.random.default_rng(H + W), H, W)).cuda()
    _, lg_fused, _ = sn.forward(img, 77, sample0=1, want_logits=True)
    with pytest.raises(ValueError, match="not materialised"):
        sn.blob("c1")
    with _diag(SIVO_NO_FUSE_BRIDGE="1"):
        _, _, sn2 = _make(text, T, seed=5)
    _, lg_plain, _ = sn2.forward(img, 77, sample0=1, want_logits=True)
    torch.cuda.synchronize()
    assert sn2.blob("c1").shape == (T, width, H, W)
    assert torch.equal(lg_fused, lg_plain)


def test_pooling_fused_into_the_output_transform_is_bit_identical():
    """conv4_3 -> pool4 and conv5_3 -> pool5: the F(4x4) output transform writes pooled values, window codes and the
    pooling dropout itself; SIVO_NO_FUSE_POOL=1 runs the separate pooling kernel.  Identical logits and identical masks."""
    T, H, W = 3, 64, 128
    text = netspec.standard_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    img = torch.from_numpy(_image(np.random.default_rng(9), H, W)).cuda()
    _, lg_fused, _ = sn.forward(img, 321, sample0=2, want_logits=True)
    with pytest.raises(ValueError, match="not materialised"):
        sn.blob("conv4_3")
    with _diag(SIVO_NO_FUSE_POOL="1"):
        _, _, sn2 = _make(text, T)
    _, lg_plain, _ = sn2.forward(img, 321, sample0=2, want_logits=True)
    torch.cuda.synchronize()
    assert sn2.blob("conv4_3").shape == (T, 512, H // 8, W // 8)
    for name in ("pool4", "pool4_mask", "pool5", "pool5_mask"):
        assert np.array_equal(sn.blob(name), sn2.blob(name)), name
    assert torch.equal(lg_fused, lg_plain)


def test_fused_upsample_is_bit_identical_to_the_materialised_one():
    """Upsample -> F(4x4,3x3) convolution reads the pooled tensor + window codes inside the input transform; the same
    net built with SIVO_NO_FUSE_UNPOOL=1 runs the unpool kernel first.  Same arithmetic -> identical logits, and the
    fused-away blob is reported as such."""
    from sivo_amd._lib import SivoError
    T, H, W = 3, 64, 96
    text = netspec.standard_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    img = torch.from_numpy(_image(np.random.default_rng(8), H, W)).cuda()
    _, lg_fused, _ = sn.forward(img, 123, want_logits=True)
    with pytest.raises(ValueError, match="not materialised"):
        sn.blob("pool4_D")
    with _diag(SIVO_NO_FUSE_UNPOOL="1"):
        _, _, sn2 = _make(text, T)
    _, lg_plain, _ = sn2.forward(img, 123, want_logits=True)
    torch.cuda.synchronize()
    assert sn2.blob("pool4_D").shape == (T, 512, H // 8, W // 8)
    assert torch.equal(lg_fused, lg_plain)


def test_sample_sharding_matches_single_pass(oracle):
    """Samples {0,1} + {2,3} computed separately sum to the 4-sample pass (multi-GPU partitioning)."""
    T, H, W = 4, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    d_img = torch.from_numpy(_image(np.random.default_rng(3), H, W)).cuda()
    full, lg_full, _ = sn.forward(d_img, 11, want_logits=True)
    a, lg_a, _ = sn.forward(d_img, 11, n_samples=2, sample0=0, want_logits=True)
    b, lg_b, _ = sn.forward(d_img, 11, n_samples=2, sample0=2, want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(lg_full[:2], lg_a) and torch.equal(lg_full[2:], lg_b)
    np.testing.assert_allclose((a + b).cpu().numpy(), full.cpu().numpy(), atol=1e-6, rtol=0)


def test_segment_image_host_entry_and_crop(oracle, kitti_like_bgr):
    """segmentImage on a larger frame: centre crop (resizeImage) then the full path."""
    T, H, W = 2, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    big = np.ascontiguousarray(kitti_like_bgr[:100, :200])
    cls, conf, ent = sn.segment_image(big, seed=5)
    res = oracle.segment(net, w, big, 5)
    np.testing.assert_allclose(conf, res["confidence"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(ent, res["entropy"], atol=2e-4, rtol=0)
    assert (cls == res["classes"]).mean() > 0.99
    from sivo_amd._lib import SivoError
    with pytest.raises(SivoError):
        sn.segment_image(big[:16, :16], seed=5)          # smaller than the net geometry


def test_mc_reduce_finalize_known_answers(oracle):
    K, H, W = 15, 8, 16
    # uniform logits -> p = 1/15 everywhere -> entropy log2(15) bits, class 0 (first index wins ties)
    lg = torch.zeros((2, K, H, W), device="cuda")
    ps, _ = mc_reduce(lg)
    cls, conf, ent = mc_finalize(ps, 2)
    assert (cls == 0).all()
    np.testing.assert_allclose(conf.cpu().numpy(), 1 / 15, atol=1e-7)
    np.testing.assert_allclose(ent.cpu().numpy(), np.log2(15), atol=1e-6)
    # one-hot (huge margin) -> entropy 0 with the exact-zero guard, confidence 1
    lg = torch.full((3, K, H, W), -1000.0, device="cuda"); lg[:, 4] = 1000.0
    ps, _ = mc_reduce(lg)
    cls, conf, ent = mc_finalize(ps, 3)
    assert (cls == 4).all() and (conf == 1).all() and (ent == 0).all()


@pytest.mark.parametrize("T,K,H,W", [(2, 15, 5, 7), (6, 15, 16, 32), (12, 15, 44, 128), (3, 11, 9, 10)])
def test_mc_reduce_random(oracle, T, K, H, W):
    rng = np.random.default_rng(T * 100 + K)
    lg = (rng.standard_normal((T, K, H, W)) * 4).astype(np.float32)
    prob_o = oracle.softmax(lg)
    mean_o = oracle.mc_mean(prob_o)
    cls_o, conf_o, ent_o = oracle.mc_finalize(mean_o)
    d = torch.from_numpy(lg).cuda()
    ps, prob = mc_reduce(d, want_prob=True)
    cls, conf, ent = mc_finalize(ps, T)
    # (device Softmax: v_exp_f32 of (x - max) log2 e and one reciprocal per pixel, sivo_amd/csrc/softmax.hpp: the exponent's rounding
    # adds |x - max| 2^-24 relative to a probability — at most ~4e-7 absolute here — where libm's expf and a division per class gave 2e-7)
    np.testing.assert_allclose(prob.cpu().numpy(), prob_o, atol=1e-6, rtol=0)
    np.testing.assert_allclose(ps.cpu().numpy() / T, mean_o, atol=1e-6, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy(), conf_o, atol=1e-6, rtol=0)
    np.testing.assert_allclose(ent.cpu().numpy(), ent_o, atol=2e-5, rtol=0)
    srt = np.sort(mean_o, axis=0)
    decided = (srt[-1] - srt[-2]) > 1e-6
    assert (cls.cpu().numpy()[decided] == cls_o[decided]).all()
    # accumulate flag: two halves add up
    if T % 2 == 0:
        ps2, _ = mc_reduce(d[:T // 2])
        mc_reduce(d[T // 2:], prob_sum=ps2, accumulate=True)
        np.testing.assert_allclose(ps2.cpu().numpy(), ps.cpu().numpy(), atol=1e-6, rtol=0)
    var = mc_variance(prob, cls)
    # the variance kernel on the SAME probabilities (the device's), then end to end with the Softmax difference above in its input
    np.testing.assert_allclose(var.cpu().numpy(), oracle.mc_variance(prob.cpu().numpy(), cls.cpu().numpy()), atol=1e-7, rtol=0)
    np.testing.assert_allclose(var.cpu().numpy(), oracle.mc_variance(prob_o, cls.cpu().numpy()), atol=1e-6, rtol=0)


def _conv_stack_prototxt(T, H, W, width):
    """data -> conv3x3(3->width) -> ReLU -> Dropout -> conv3x3(width->width)+BN+ReLU+Dropout -> conv3x3(width->width) ->
    conv1x1(width->15) -> Softmax: exercises the direct kernel (Cin = 3), the Winograd kernel with every
    epilogue option (BN, ReLU, dropout) and without, and the 1x1 kernel."""
    from sivo_amd.netspec import _bn, _conv, _drop, _relu, _softmax
    out = [f'name: "conv_stack"\ninput: "data"\ninput_dim: {T}\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n']
    out += [_conv("c0", "data", "c0", width, 3, 1), _relu("r0", "c0"), _drop("d0", "c0")]
    out += [_conv("c1", "c0", "c1", width, 3, 1), _bn("c1_bn", "c1"), _relu("r1", "c1"), _drop("d1", "c1")]
    out += [_conv("c2", "c1", "c2", width, 3, 1)]
    out += [_conv("cls", "c2", "cls", 15, 1, 0), _softmax("cls")]
    return "".join(out)


@pytest.mark.parametrize("H,W,width", [(22, 64, 64), (44, 136, 128), (6, 8, 64), (32, 64, 192), (10, 20, 64),
                                       (22, 64, 256), (9, 12, 256), (4, 4, 256)])
def test_winograd_and_direct_conv_shapes(oracle, H, W, width):
    """Ragged tile rows (H = 22 = 5.5 tiles of 4), widths that are multiples of 8 but not of 32, a Cout that is not a
    multiple of 128, and a width (20) that falls back to the direct kernel: every blob against the oracle.
    width 256 takes the F(4x4,3x3) path (conv_wino4.hip): ragged and odd tile rows, W a multiple of 4 only, one tile."""
    T = 3
    text = _conv_stack_prototxt(T, H, W, width)
    with _diag(SIVO_NO_FUSE_BRIDGE="1"):          # every intermediate blob is inspected below
        net, w, sn = _make(text, T, seed=11)
    img = _image(np.random.default_rng(H * W), H, W)
    ob = oracle.run_net(net, w, oracle.preprocess(img, T, H, W), 31, sample0=2)
    _, logits, _ = sn.forward(torch.from_numpy(img).cuda(), 31, sample0=2, want_logits=True)
    torch.cuda.synchronize()
    for name in ["c0", "c1", "c2", "cls"]:
        g, o = sn.blob(name), ob[name]
        np.testing.assert_allclose(g, o, atol=LOGIT_TOL, rtol=0, err_msg=name)
        if name in ("c0", "c1"):
            # dropout zero pattern is bit-exact; the only admissible differences are ReLU outputs within rounding of 0
            mism = (g == 0) != (o == 0)
            assert mism.mean() < 1e-4 and (np.abs(g[mism]) < 1e-4).all() and (np.abs(o[mism]) < 1e-4).all(), name
    np.testing.assert_allclose(logits.cpu().numpy(), ob["cls"], atol=LOGIT_TOL, rtol=0)


@pytest.mark.parametrize("H,W", [(22, 64), (44, 136), (10, 24), (64, 192)])
def test_persistent_fused_f4x4_kernel_is_bit_identical(H, W):
    """conv_wino4f_pp_kernel (one workgroup per CU walking a list of pixel tiles; used where every workgroup gets at least 16
    tiles, i.e. conv1_2_D at full size) against the one-workgroup-per-tile form, forced on small geometries: ragged tile rows
    and columns, more tiles than workgroups and fewer, with and without the dropout epilogue.  Every blob bit for bit."""
    T = 5
    text = _conv_stack_prototxt(T, H, W, 64)
    with _diag(SIVO_NO_FUSE_BRIDGE="1"):
        net, w, sn = _make(text, T, seed=3)
    d_img = torch.from_numpy(_image(np.random.default_rng(H + W), H, W)).cuda()
    got = {}
    for mode in ("0", "2"):
        with _diag(SIVO_W4F_PERSIST=mode):
            _, logits, _ = sn.forward(d_img, 9, want_logits=True)
            torch.cuda.synchronize()
            got[mode] = [sn.blob(n) for n in ("c1", "c2")] + [logits.cpu().numpy()]
    for a, b in zip(got["0"], got["2"]):
        assert np.array_equal(a, b)
    assert np.abs(got["2"][1]).max() > 0


def _conv7_stack_prototxt(T, H, W, width):
    """data -> conv3x3(3->width)+ReLU+Dropout -> conv7x7(width->64)+BN+ReLU+Dropout -> conv7x7(64->64) -> conv1x1(64->15) -> Softmax:
    SegNet-Basic's 7x7 layers behind a dropout (per-sample), with every epilogue option."""
    from sivo_amd.netspec import _bn, _conv, _drop, _relu, _softmax
    out = [f'name: "conv7_stack"\ninput: "data"\ninput_dim: {T}\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n']
    out += [_conv("c0", "data", "c0", width, 3, 1), _relu("r0", "c0"), _drop("d0", "c0")]
    out += [_conv("c1", "c0", "c1", 64, 7, 3), _bn("c1_bn", "c1"), _relu("r1", "c1"), _drop("d1", "c1")]
    out += [_conv("c2", "c1", "c2", 64, 7, 3)]
    out += [_conv("cls", "c2", "cls", 15, 1, 0), _softmax("cls")]
    return "".join(out)


@pytest.mark.parametrize("H,W,width", [(22, 36, 64), (9, 12, 32), (40, 100, 96), (16, 64, 64)])
def test_conv7_bf16x6_shapes(oracle, H, W, width):
    """conv7_x6.hip (direct 7x7 on the bf16 matrix cores, fp32 operands as three bf16 planes): ragged tile rows and columns
    (H not a multiple of 8, W a multiple of 4 only), images smaller than one workgroup tile, 32 / 64 / 96 input channels
    (one to three 32-channel halves), BN + ReLU + dropout epilogue and none: every blob against the oracle, and against the
    fp32-MFMA direct kernel (SIVO_CONV7=f32) to the rounding of two fp32 evaluations."""
    T = 3
    text = _conv7_stack_prototxt(T, H, W, width)
    net, w, sn = _make(text, T, seed=21)
    os.environ["SIVO_CONV7"] = "f32"
    try:
        _, _, sn32 = _make(text, T, seed=21)
    finally:
        del os.environ["SIVO_CONV7"]
    img = _image(np.random.default_rng(H * W + width), H, W)
    d_img = torch.from_numpy(img).cuda()
    ob = oracle.run_net(net, w, oracle.preprocess(img, T, H, W), 13, sample0=1)
    _, logits, _ = sn.forward(d_img, 13, sample0=1, want_logits=True)
    torch.cuda.synchronize()
    blobs = {n: sn.blob(n) for n in ("c1", "c2", "cls")}
    _, logits32, _ = sn32.forward(d_img, 13, sample0=1, want_logits=True)
    torch.cuda.synchronize()
    for name in ("c1", "c2", "cls"):
        g, o, g32 = blobs[name], ob[name], sn32.blob(name)
        np.testing.assert_allclose(g, o, atol=LOGIT_TOL, rtol=0, err_msg=name)
        np.testing.assert_allclose(g, g32, atol=2e-4 * max(1.0, float(np.abs(o).max())), rtol=0, err_msg=name)
        if name == "c1":
            mism = (g == 0) != (o == 0)        # dropout zero pattern: only ReLU outputs within rounding of 0 may differ
            assert mism.mean() < 1e-4 and (np.abs(g[mism]) < 1e-4).all() and (np.abs(o[mism]) < 1e-4).all()
    np.testing.assert_allclose(logits.cpu().numpy(), ob["cls"], atol=LOGIT_TOL, rtol=0)
    assert np.abs(ob["c2"]).max() > 0.1


def _classifier_prototxt(T, H, W, width, classes):
    """data -> conv3x3(3->width)+ReLU+Dropout -> conv3x3(width->width)+BN+ReLU -> conv3x3(width->classes) -> Softmax: the tail
    of SegNet-Standard (conv1_2_D, conv1_1_D, prob) behind a dropout, so that the classifier is per-sample."""
    from sivo_amd.netspec import _bn, _conv, _drop, _relu, _softmax
    out = [f'name: "classifier_tail"\ninput: "data"\ninput_dim: {T}\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n']
    out += [_conv("c0", "data", "c0", width, 3, 1), _relu("r0", "c0"), _drop("d0", "c0")]
    out += [_conv("c1", "c0", "c1", width, 3, 1), _bn("c1_bn", "c1"), _relu("r1", "c1")]
    out += [_conv("cls", "c1", "cls", classes, 3, 1), _softmax("cls")]
    return "".join(out)


@pytest.mark.parametrize("T,H,W,width,classes", [(4, 32, 64, 64, 15), (3, 40, 72, 64, 15), (2, 10, 24, 64, 11), (5, 8, 32, 66, 16),
                                                 (12, 64, 128, 64, 15)])
def test_classifier_fused_with_the_mc_postprocessing(oracle, T, H, W, width, classes):
    """conv_cls_mc.hip: the 3x3 classifier convolution, the Softmax layer, the f64 mean over the samples and the maps in
    one kernel (what segment / segment_dev / forward_dev without logits run).  Ragged tiles (H, W not multiples of the
    8 x 32 workgroup tile), a channel count that is not a multiple of the K-chunk, 11 / 15 / 16 classes.
      * its logits agree with the oracle's (1e-3) and with the separate classifier kernel's;
      * its maps equal the post-processing kernel's on exactly these logits, bit for bit;
      * its probability sums equal sivo_mc_reduce_dev of exactly these logits, bit for bit;
      * maps against the oracle."""
    text = _classifier_prototxt(T, H, W, width, classes)
    net, w, sn = _make(text, T, seed=5)
    img = _image(np.random.default_rng(T * H + W), H, W)
    seed = 77
    d_img = torch.from_numpy(img).cuda()
    maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
            torch.empty((H, W), dtype=torch.float64, device="cuda"))
    fl = torch.full((T, classes, H, W), float("nan"), dtype=torch.float32, device="cuda")
    sn.segment_into(d_img, seed, maps, logits=fl)
    plain = tuple(torch.empty_like(m) for m in maps)
    sn.segment_into(d_img, seed, plain)                                   # the production entry point (no logits stored)
    ps_fused = torch.full((classes, H, W), float("nan"), dtype=torch.float32, device="cuda")
    sn.forward_into(d_img, seed, ps_fused)                                # fused, probability-sum form
    ps_u, lg_u, _ = sn.forward(d_img, seed, want_logits=True)             # separate classifier kernel + mc_reduce
    c2, f2, e2 = mc_segment(fl)
    ps2, _ = mc_reduce(fl)
    torch.cuda.synchronize()
    assert torch.isfinite(fl).all()
    for a, b in zip(maps, (c2, f2, e2)):
        assert torch.equal(a, b)
    for a, b in zip(maps, plain):
        assert torch.equal(a, b)
    assert torch.equal(ps_fused, ps2)
    res = oracle.segment(net, w, img, seed, logits_name="cls")
    np.testing.assert_allclose(fl.cpu().numpy(), res["logits"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(fl.cpu().numpy(), lg_u.cpu().numpy(), atol=2e-4, rtol=0)
    _check_outputs(oracle, res, maps[0].cpu().numpy(), maps[1].cpu().numpy(), maps[2].cpu().numpy(), ps_fused.cpu().numpy(), T)


def test_multi_device_handle_on_one_device(oracle, kitti_like_bgr):
    """sivo_segnet_create_multi with device_ids = [0]: the whole multi-device path (chunk-major sums, RCCL communicator,
    reduce-scatter, per-chunk finalize, all-gather, hand-over) runs, on one rank.  The probability sums travel in f64 (the
    accumulators themselves), so the maps equal the single-device handle's f64 mean."""
    T, H, W = 4, 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    multi = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T, devices=[0])
    assert (multi.T, multi.H, multi.W, multi.classes) == (sn.T, sn.H, sn.W, sn.classes)
    big = np.ascontiguousarray(kitti_like_bgr[:80, :150])
    cls_m, conf_m, ent_m = multi.segment_image(big, seed=5)
    cls_s, conf_s, ent_s = sn.segment_image(big, seed=5)
    assert np.array_equal(conf_m, conf_s) and np.array_equal(ent_m, ent_s) and np.array_equal(cls_m, cls_s)
    res = oracle.segment(net, w, big, 5)
    np.testing.assert_allclose(conf_m, res["confidence"], atol=1e-5, rtol=0)
    with pytest.raises(ValueError):
        multi.forward(torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"), 1)
    with pytest.raises(ValueError):
        BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T, devices=[0, 0])


@pytest.mark.parametrize("ndev,T", [(2, 4), (4, 6), (8, 12), (8, 9)])
def test_multi_device_handle_emulated_on_one_gpu(ndev, T, kitti_like_bgr):
    """The index arithmetic of the in-handle multi-GPU path with ndev > 1, without a second GPU: SIVO_MULTI_EMULATE=1 lets
    device_ids name GPU 0 several times and carries the two collectives out as copies + f64 adds in device order; the rest
    is the code the RCCL path runs — contiguous sample shards incl. uneven ones (T = 12 over 8 -> 1,1,1,1,2,2,2,2; T = 9 over
    8), dropout keyed by the global sample, chunk-major f64 sums, reduce-scatter by pixel chunk, per-chunk f64 finalize,
    gathered maps.  Against the single-device handle: confidence / entropy to 1e-12 (f64 sums added in another order),
    classes identical except at ties of the mean."""
    H, W = 32, 64
    text = netspec.tiny_prototxt(T, H, W)
    net, w, sn = _make(text, T)
    with _diag(SIVO_MULTI_EMULATE="1"):
        multi = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T, devices=[0] * ndev)
    big = np.ascontiguousarray(kitti_like_bgr[:80, :150])
    for seed in (5, 6):
        cls_m, conf_m, ent_m = multi.segment_image(big, seed=seed)
        cls_s, conf_s, ent_s = sn.segment_image(big, seed=seed)
        np.testing.assert_allclose(conf_m, conf_s, atol=1e-12, rtol=0)
        np.testing.assert_allclose(ent_m, ent_s, atol=1e-11, rtol=0)
        assert (cls_m != cls_s).sum() <= 2
    assert torch.cuda.current_device() == 0


def _make_env(text, T, seed, **env):
    if set(env) - _PRODUCT_SWITCHES:
        with _diag(**env):
            return _make(text, T, seed=seed)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return _make(text, T, seed=seed)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("H,W,width", [(44, 128, 128), (22, 64, 256), (9, 12, 512)])
def test_f4x4_gemm_forms_agree(oracle, H, W, width):
    """The three GEMMs of the three-kernel F(4x4,3x3) path on the same layers: f16x3 (default: fp16 hi + lo planes, three
    products), bf16x6 (SIVO_GEMM=x6) and the fp32 MFMA kernel (SIVO_GEMM=f32).  Each against the oracle within the logit
    tolerance; the split forms are at least as close to the oracle as the fp32 FMA chain of the MFMA kernel (measured:
    2.1e-4 / 2.7e-4 / 2.8e-4 on the 512-channel case — fewer roundings of the accumulator), and the three forms agree with
    each other within half the tolerance; gemm_status names the form and reports the calibrated scales."""
    T = 3
    text = _conv_stack_prototxt(T, H, W, width)
    img_np = _image(np.random.default_rng(H * W + width), H, W)
    img = torch.from_numpy(img_np).cuda()
    lg = {}
    for form, env in (("h3", {}), ("x6", {"SIVO_GEMM": "x6"}), ("f32", {"SIVO_GEMM": "f32"})):
        net, w, sn = _make_env(text, T, 5, **env)
        _, l, _ = sn.forward(img, 77, sample0=1, want_logits=True)
        torch.cuda.synchronize()
        lg[form] = l.cpu().numpy()
        mode, overflow, layers = sn.gemm_status()
        assert (mode, overflow) == ({"h3": 2, "x6": 1, "f32": 0}[form], 0)
        if form == "h3":
            assert [n for n, *_ in layers] == ["c1", "c2"]
            for _, vmax, vscale, uscale in layers:
                assert vmax > 0 and 128 <= vmax * vscale < 256 and np.log2(vscale) % 1 == 0 and np.log2(uscale) % 1 == 0
    res = oracle.segment(net, w, img_np, 77, sample0=1, logits_name="cls")
    for form in lg:
        err = np.abs(lg[form] - res["logits"]).max()
        print(f"[{H}x{W}x{width} {form}] max|dlogit| vs oracle {err:.2e}; vs fp32 GEMM {np.abs(lg[form] - lg['f32']).max():.2e}")
        assert err < LOGIT_TOL
    assert np.abs(lg["h3"] - lg["f32"]).max() < LOGIT_TOL / 2 and np.abs(lg["x6"] - lg["h3"]).max() < LOGIT_TOL / 2
    assert np.abs(lg["h3"] - res["logits"]).max() <= 1.1 * np.abs(lg["f32"] - res["logits"]).max()


def test_fp16_overflow_recomputes_the_frame_and_backs_the_scales_off(oracle):
    """A value that leaves the fp16 range raises the flag of the f16x3 kernels: sivo_segnet_segment recomputes the frame without
    f16x3 (here: the bf16x6 GEMM) before it returns, the handle lowers its scales by 2^2 and stays on f16x3; the fourth such frame
    switches it to bf16x6 for good.  Forced with SIVO_H3_BOOST (the calibrated scales times 2^k: k = 9 puts the calibration
    maximum itself at >= 65536).  A recomputed frame equals that of a handle built with SIVO_GEMM=x6 bit for bit."""
    T, H, W, width = 3, 22, 64, 256
    text = _conv_stack_prototxt(T, H, W, width)
    img = _image(np.random.default_rng(4), H, W)
    _, _, boosted = _make_env(text, T, 5, SIVO_H3_BOOST=9)
    _, _, x6 = _make_env(text, T, 5, SIVO_GEMM="x6")
    _, _, plain = _make_env(text, T, 5)
    assert boosted.gemm_status()[:2] == (2, 0)
    scales0 = [r[2] for r in boosted.gemm_status()[2]]
    got = boosted.segment_image(img, seed=3)
    want = x6.segment_image(img, seed=3)
    assert boosted.gemm_status()[:2] == (2, 1)                       # still f16x3, one frame recomputed ...
    assert [r[2] for r in boosted.gemm_status()[2]] == [v / 4 for v in scales0]        # ... and two more bits of headroom
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert np.isfinite(got[1]).all() and np.isfinite(got[2]).all()
    # the next frame runs f16x3 at 2^7 times the calibrated scales (the calibration maximum at < 2^15): no flag, and the maps are
    # those of an unboosted handle up to the split's rounding
    got2 = boosted.segment_image(img, seed=4)
    ref2 = plain.segment_image(img, seed=4)
    assert boosted.gemm_status()[:2] == (2, 1)
    assert (got2[0] != ref2[0]).mean() < 2e-3
    np.testing.assert_allclose(got2[1], ref2[1], atol=1e-4, rtol=0)
    # a handle whose scales are hopeless (2^20): four recomputed frames, then bf16x6 for good
    _, _, hopeless = _make_env(text, T, 5, SIVO_H3_BOOST=20)
    for i in range(4):
        g = hopeless.segment_image(img, seed=3)
        assert hopeless.gemm_status()[:2] == ((2, i + 1) if i < 3 else (1, 4))
        for a, b in zip(g, want):
            assert np.array_equal(a, b)
    g = hopeless.segment_image(img, seed=3)
    assert hopeless.gemm_status()[:2] == (1, 4) and all(np.array_equal(a, b) for a, b in zip(g, want))
    # asynchronous entry point: the frame that overflowed is not recomputed; the caller asks once it has synchronised and issues it again
    _, _, boosted2 = _make_env(text, T, 5, SIVO_H3_BOOST=9)
    d = torch.from_numpy(img).cuda()
    boosted2.forward(d, 3)
    torch.cuda.synchronize()
    assert boosted2.take_overflow() and not boosted2.take_overflow()
    assert boosted2.gemm_status()[:2] == (2, 1)
    ps, _, _ = boosted2.forward(d, 3)                                 # this one runs without f16x3
    ps_x6, _, _ = x6.forward(d, 3)
    torch.cuda.synchronize()
    assert torch.equal(ps, ps_x6) and not boosted2.take_overflow()
    ps2, _, _ = boosted2.forward(d, 3)                                # and this one on f16x3 again
    torch.cuda.synchronize()
    assert not boosted2.take_overflow() and not torch.equal(ps2, ps_x6)
    assert (ps2 - ps_x6).abs().max() < 1e-3


def test_fp16_overflow_is_reported_with_two_frames_in_flight(oracle):
    """Two frames in flight (bench.py's loop: issue(k) runs before complete(k - 1)): frame k - 1 raises the flag, forward(k) finds it
    first.  forward(k) reacts (one back-off, frame k itself runs without f16x3) but must not swallow the report:
    sivo_segnet_take_overflow still says 1 when the caller asks about frame k - 1, exactly once, and a flag that goes up again
    before the back-off's pause is consumed does not lower the scales a second time."""
    T, H, W, width = 3, 22, 64, 256
    text = _conv_stack_prototxt(T, H, W, width)
    d = torch.from_numpy(_image(np.random.default_rng(4), H, W)).cuda()
    _, _, boosted = _make_env(text, T, 5, SIVO_H3_BOOST=9)
    _, _, x6 = _make_env(text, T, 5, SIVO_GEMM="x6")
    boosted.forward(d, 3)                                  # frame k - 1: overflows
    torch.cuda.synchronize()                               # (the deterministic form of the race: the flag is up before forward(k) is enqueued)
    ps_k, _, _ = boosted.forward(d, 4)                     # forward(k) consumes the flag and the pause ...
    ps_k_x6, _, _ = x6.forward(d, 4)
    torch.cuda.synchronize()
    assert torch.equal(ps_k, ps_k_x6)                      # ... so frame k ran without f16x3 and is right as it is
    assert boosted.take_overflow()                         # the report about frame k - 1 is still owed
    assert not boosted.take_overflow()                     # once
    mode, frames, layers = boosted.gemm_status()
    assert (mode, frames) == (2, 1)
    # the flag going up twice in ONE event: frame B, issued with the old scales, is still running when frame A's overflow is noticed
    _, _, b2 = _make_env(text, T, 5, SIVO_H3_BOOST=9)
    s0 = [r[2] for r in b2.gemm_status()[2]]
    ev_a = torch.cuda.Event()
    torch.cuda._sleep(200_000_000)                         # (spin kernels: both forwards are enqueued before A starts, and B starts
    b2.forward(d, 3); ev_a.record()                        #  long after the host has asked about A)
    torch.cuda._sleep(400_000_000)
    b2.forward(d, 5)
    ev_a.synchronize()
    assert b2.take_overflow()                              # A noticed: back-off, pause pending
    torch.cuda.synchronize()                               # B (old scales) raises the flag again
    assert b2.take_overflow()                              # the same event: reported, no second back-off
    b2.forward(d, 3); b2.forward(d, 5)                     # both again: the first without f16x3, the second on the lowered scales
    torch.cuda.synchronize()
    assert not b2.take_overflow()
    assert [r[2] for r in b2.gemm_status()[2]] == [v / 4 for v in s0]       # lowered once
    assert b2.gemm_status()[:2] == (2, 2)                  # (two frames raised the flag)


def test_accuracy_guard_measures_every_layer_and_reroutes_an_inaccurate_plan(oracle):
    """The load-time accuracy guard (segnet.cpp accuracy_guard, include/sivo_hip.h sivo_segnet_guard_report).
    (i) A default handle: one row per F(4x4) / f16x3 layer with its own error against the direct fp32 kernel on the same input,
    the predicted logit error within the budget, nothing rerouted, one plan, the cost reported.
    (ii) A plan that really is inaccurate — the f16x3 scales forced 2^16 too SMALL (diagnostic build, SIVO_H3_BOOST=-16: the
    operands sit in fp16's subnormal range and lose most of their lo plane; no weight family of the full-size sweep pushes the
    F(4x4) / f16x3 arithmetic itself over the budget): UNGUARDED (SIVO_GUARD=0) its logits miss the 1e-3 tolerance
    against the oracle; GUARDED the same configuration measures the damage layer by layer, takes those layers off F(4x4) and
    then off f16x3, plans again, and meets the tolerance."""
    T, H, W = 2, 96, 192
    text = netspec.standard_prototxt(T, H, W)
    img = _image(np.random.default_rng(8), H, W)
    d = torch.from_numpy(img).cuda()

    def logit_error(net, w, sn):
        _, lg, _ = sn.forward(d, 31, want_logits=True)
        torch.cuda.synchronize()
        masks = {L["top"][1]: sn.blob(L["top"][1]) for L in net["layers"] if L["type"] == "Pooling"}
        res = oracle.segment(net, w, img, 31, logits_name="conv1_1_D", force_masks=masks, flips={})
        return float(np.abs(lg.cpu().numpy() - res["logits"]).max()), float(np.abs(res["logits"]).max())

    net, w, sn = _make(text, T)
    g = sn.guard_report()
    assert g["builds"] == 1 and g["layers"] and all(r["level"] == 0 for r in g["layers"])
    assert 0 < g["predicted"] <= g["budget"] == pytest.approx(1e-3 / 30)
    assert {r["kernel"] for r in g["layers"]} <= {"F(4x4) f16x3 GEMM", "direct f16x3", "F(4x4) fp32 fused", "classifier f16x3"}
    assert "classifier f16x3" in {r["kernel"] for r in g["layers"]}
    assert all(0 < r["rel_err"] < 3e-5 and r["rel_rms"] < r["rel_err"] for r in g["layers"])
    err, mag = logit_error(net, w, sn)
    print(f"[guard] default plan: {len(g['layers'])} layers guarded in {g['ms']:.0f} ms, predicted {g['predicted']:.2e} of the logit scale "
          f"(budget {g['budget']:.2e}), largest layer error {max(r['rel_err'] for r in g['layers']):.2e}; max|dlogit| vs oracle {err:.2e} at max|logit| {mag:.1f}")
    assert err < LOGIT_TOL

    net_u, w_u, bad = _make_env(text, T, 42, SIVO_H3_BOOST=-16, SIVO_GUARD=0)
    assert bad.guard_report()["layers"] == []
    err_u, _ = logit_error(net_u, w_u, bad)
    net_g, w_g, good = _make_env(text, T, 42, SIVO_H3_BOOST=-16)
    gg = good.guard_report()
    err_g, _ = logit_error(net_g, w_g, good)
    moved = [(r["layer"], r["level"], r["kernel"], r["first_rel_err"]) for r in gg["layers"] if r["level"]]
    print(f"[guard] scales 2^-16: unguarded max|dlogit| {err_u:.2e}; guarded {err_g:.2e} after {gg['builds']} plans ({gg['ms']:.0f} ms), "
          f"predicted {gg['predicted']:.2e}; moved {len(moved)} layers, largest first-plan layer error {max(m[3] for m in moved):.2e}")
    assert err_u > LOGIT_TOL                      # the hazard is real ...
    assert gg["builds"] >= 2 and moved and err_g < LOGIT_TOL and gg["predicted"] <= gg["budget"] / 3   # ... and the guard removes it


def test_fork_dropout_in_the_input_transform_is_bit_identical():
    """pool3's in-place Dropout (the fork: everything behind it is per-sample) moved into conv4_1's F(4x4) input transform: the pooling
    writes its values once instead of T dropped copies, the transform applies the same counter-based dropout word as it reads.  Against
    the same build with the fusion off (diagnostic build, SIVO_NO_FUSE_INDROP): logits bit for bit; and sivo_segnet_blob still hands
    out Caffe's per-sample dropped blob (re-created from the stored values and the last forward's seed)."""
    T, H, W = 3, 96, 192
    text = netspec.standard_prototxt(T, H, W)
    _, _, fused = _make(text, T)
    _, _, plain = _make_env(text, T, 42, SIVO_NO_FUSE_INDROP=1)
    img = torch.from_numpy(_image(np.random.default_rng(12), H, W)).cuda()
    for seed, n, s0 in ((5, 3, 0), (6, 2, 1)):
        pa, la, _ = fused.forward(img, seed, n_samples=n, sample0=s0, want_logits=True)
        pb, lb, _ = plain.forward(img, seed, n_samples=n, sample0=s0, want_logits=True)
        torch.cuda.synchronize()
        assert torch.equal(la, lb) and torch.equal(pa, pb)
        a, b = fused.blob("pool3"), plain.blob("pool3")
        assert a.shape == b.shape == (T, 256, H // 8, W // 8)
        assert np.array_equal(a[:n], b[:n]) and (a[:n] == 0).mean() > 0.3


def test_profiling_events_survive_unprofiled_forwards_and_lanes(kitti_like_bgr):
    """sivo_segnet_profile as bench.py uses it inside its timed loop (round 6): a forward is profiled (MFMA kernels only — in one lane, or with
    its sample groups kept on their streams), profiling is switched off WITHOUT waiting, further forwards run, and only then the rows are
    read.  The rows must be the profiled forward's: every F(4x4) layer's GEMM and every direct convolution, one kernel launch per layer in
    one lane and one per layer and lane with the lanes kept, the same FLOPs either way.  (The deferred read had once dropped the GEMM rows:
    the counts of a profiled forward were cleared by the next unprofiled one.)"""
    T, H, W = 4, 64, 128
    text = netspec.standard_prototxt(T, H, W)
    net = oproto.parse(text)
    sn = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], wts.synth_weights(net["layers"], 42)), T=T)
    d_img = torch.from_numpy(np.ascontiguousarray(kitti_like_bgr[:H, :W])).cuda()
    out = {}
    for keep in (False, True):
        sn.profile(True, mfma_only=True, reset=True, keep_lanes=keep)
        sn.forward(d_img, 1)                      # the profiled forward
        sn.profile(False)                         # (does not wait)
        for s in (2, 3):
            sn.forward(d_img, s)                  # unprofiled forwards behind it
        rows = [r for r in sn.profile_read() if r["launches"]]
        torch.cuda.synchronize()
        gemm = [r for r in rows if r["kernel"].startswith("wino4_gemm")]
        direct = [r for r in rows if r["kernel"].startswith("conv3_h3")]
        assert len(gemm) == 15 and len(direct) >= 6, sorted({r["kernel"] for r in rows})
        assert all(r["launches"] == 1 and r["ms_total"] > 0 for r in gemm + direct)
        out[keep] = (sum(r["kernel_launches"] for r in gemm), sum(r["flops_per_sample"] * r["samples"] for r in gemm))
    assert out[False][0] == 15 and out[True][0] == 30           # one launch per layer / one per layer and lane (two lanes of two samples)
    assert out[False][1] == out[True][1]
