"""Parity of the configurations bench.py TIMES, at BASELINE.json's full geometry (352 x 1024, full widths):
SegNet-Standard T = 12 and SegNet-Basic T = 6, default build (the per-sample part of the forward in sample groups on
separate streams — two lanes since round 5, three before —, bridge / pooling / upsample fusions on), through the C ABI, against the CPU oracle.

Three statements per (net, image, dropout seed):
  1. lanes: the two-lane (default) and the three-lane forward are bit-identical to a one-lane forward of the same handle configuration;
  2. teacher-forced: with the device's pooling switches imposed on the oracle, EVERY logit of EVERY sample agrees
     within 1e-3 (north star), and every imposed switch that differs from the oracle's own choice is a near-tie;
  3. free-running: the oracle decides its own switches.  Max pooling is discontinuous, so where the two sides
     pick different elements of a near-tied window the unpooled value lands on a different pixel and the logits in
     that switch's decoder receptive field move by O(1).  Asserted: every differing switch is a near-tie of the
     oracle's own pre-pooling activations, every logit that differs by more than 1e-3 lies inside the receptive
     field of such a switch (computed by propagating the switch positions through the decoder: Upsample = 2x2
     block, k x k convolution = k x k dilation), and the class maps disagree on at most 0.3 % of the pixels.

The oracle runs the sample-invariant prefix once (oracle.run_net expand_to) — same values, 11 redundant prefixes less.
"""
import os

import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import prototxt as oproto
from sivo_amd import netspec, weights as wts
from sivo_amd.segnet import BayesianSegNet, mc_segment

pytestmark = pytest.mark.gpu
H, W = 352, 1024
LOGIT_TOL = 1e-3
# A pooling switch may differ between two fp32 evaluations only where the two window elements are closer than the
# evaluations' own error: 1e-4 relative to the activation magnitude (pre-pooling activations of the deep layers reach
# +-13 and carry the same ~2e-4 absolute error as the logits; measured gaps: <= 3.4e-5 relative).
NEAR_TIE = 1e-4
LOGITS = {"standard": "conv1_1_D", "basic": "dense_softmax_inner_prod"}
_cache = {}


def _text(kind, T):
    return (netspec.standard_prototxt if kind == "standard" else netspec.basic_prototxt)(T, H, W)


def _handle(kind, T, lanes=None, mutate=None, tag=""):
    """One device handle per configuration for the whole module (Standard T = 12 holds ~22 GB)."""
    key = (kind, T, lanes, tag)
    if key not in _cache:
        text = _text(kind, T)
        net = oproto.parse(text)
        w = wts.synth_weights(net["layers"], 42)
        if mutate:
            mutate(net, w)
        if lanes is not None:
            os.environ["SIVO_LANES"] = str(lanes)
        try:
            sn = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T)
        finally:
            os.environ.pop("SIVO_LANES", None)
        _cache[key] = (net, w, sn)
    return _cache[key]


def _images(kitti_like_bgr):
    from bench import make_inputs
    return {"kitti": np.ascontiguousarray(kitti_like_bgr[:H, :W]), "synthetic": make_inputs(H, W)[0]}


def _pool_layers(net):
    return [L for L in net["layers"] if L["type"] == "Pooling"]


def _device_run(sn, net, img, seed):
    prob_sum, logits, _ = sn.forward(torch.from_numpy(img).cuda(), seed, want_logits=True)
    cls, conf, ent = sn.finalize(prob_sum)
    torch.cuda.synchronize()
    masks = {L["top"][1]: sn.blob(L["top"][1]) for L in _pool_layers(net)}
    return logits.cpu().numpy(), masks, cls.cpu().numpy(), conf.cpu().numpy(), ent.cpu().numpy()


@pytest.mark.parametrize("kind,T", [("standard", 12), ("basic", 6)])
def test_three_lanes_equal_one_lane_at_full_size(kind, T, kitti_like_bgr):
    """(and the default, two lanes: SIVO_LANES unset)"""
    net, _, sn2 = _handle(kind, T)
    _, _, sn1 = _handle(kind, T, lanes=1)
    _, _, sn3 = _handle(kind, T, lanes=3)
    img = torch.from_numpy(_images(kitti_like_bgr)["kitti"]).cuda()
    for seed in (2024, 5):
        ps1, lg1, _ = sn1.forward(img, seed, want_logits=True)
        for snx in (sn2, sn3):
            psx, lgx, _ = snx.forward(img, seed, want_logits=True)
            torch.cuda.synchronize()
            assert torch.equal(lgx, lg1) and torch.equal(psx, ps1)
            for L in _pool_layers(net):
                assert np.array_equal(snx.blob(L["top"][1]), sn1.blob(L["top"][1])), L["name"]
    _cache.pop((kind, T, 1, ""))            # free the one-lane and the three-lane handle
    _cache.pop((kind, T, 3, ""))


def _decoder_influence(net, mask_name, dirty_pooled, logits_name):
    """dirty_pooled: bool (N, Ho, Wo) — pooling windows (any channel) whose switch differs.  Returns bool (N, H, W):
    the logits pixels that can see the displaced unpooled value."""
    layers = net["layers"]
    start = next(i for i, L in enumerate(layers) if L["type"] == "Upsample" and L["bottom"][1] == mask_name)
    cur = layers[start]["top"][0]
    d = np.kron(dirty_pooled, np.ones((1, 2, 2), bool))
    for L in layers[start + 1:]:
        if not L["bottom"] or L["bottom"][0] != cur:
            continue
        if L["type"] == "Convolution":
            k = L["kernel_size"]
            if k > 1:
                d = ndimage.maximum_filter(d, size=(1, k, k), mode="constant", cval=False)
        elif L["type"] == "Upsample":
            d = np.kron(d, np.ones((1, 2, 2), bool))
        elif L["type"] == "Softmax":
            break
        cur = L["top"][0]
        if cur == logits_name and L["type"] == "Convolution":
            break
    return d[:, :H, :W]


# one camera-like and one synthetic frame per net by default (54 s per Standard case: the oracle's 12 samples on the host); the further
# seeds run with --runslow (profiles/r05_fullsize_tests.log holds all nine)
_SLOW = pytest.mark.slow
CASES = [pytest.param("standard", 12, "kitti", 2024, marks=_SLOW), ("standard", 12, "kitti", 7), pytest.param("standard", 12, "kitti", 99, marks=_SLOW),
         pytest.param("standard", 12, "synthetic", 2024, marks=_SLOW), pytest.param("standard", 12, "synthetic", 7, marks=_SLOW), ("standard", 12, "synthetic", 99),
         pytest.param("basic", 6, "kitti", 2024, marks=_SLOW), ("basic", 6, "kitti", 7), ("basic", 6, "synthetic", 99)]


@pytest.mark.parametrize("kind,T,image,seed", CASES)
def test_timed_configuration_against_the_oracle(oracle, kind, T, image, seed, kitti_like_bgr):
    net, w, sn = _handle(kind, T)
    img = _images(kitti_like_bgr)[image]
    lname = LOGITS[kind]
    lg, masks, cls, conf, ent = _device_run(sn, net, img, seed)
    pools = _pool_layers(net)

    # ---- teacher-forced: the device's switches imposed on the oracle; every logit within tolerance
    flips = {}
    res = oracle.segment(net, w, img, seed, logits_name=lname, force_masks=masks, flips=flips, shared_prefix=True)
    n_forced = 0
    for name, (count, gap, mag) in flips.items():
        n_forced += count
        assert count <= 1e-4 * masks[name].size, (name, count)
        assert gap <= NEAR_TIE * max(mag, 1.0), (name, count, gap, mag)
    err = np.abs(lg - res["logits"])
    assert np.abs(res["logits"]).max() > 0.5
    print(f"[{kind} T={T} {image} seed={seed}] forced: {n_forced} near-tie switches, max|dlogit| {err.max():.3e} "
          f"(mean {err.mean():.2e}, max|logit| {np.abs(res['logits']).max():.1f})")
    assert err.max() < LOGIT_TOL
    np.testing.assert_allclose(conf, res["confidence"], atol=LOGIT_TOL / 2, rtol=0)
    np.testing.assert_allclose(ent, res["entropy"], atol=5e-3, rtol=0)

    # ---- the entry point bench.py times (segment_dev): classifier convolution + Softmax + mean over the samples in one
    # kernel (conv_cls_mc.hip; SegNet-Basic's 1x1 classifier keeps the separate kernels).  Same seed, same switches, so the
    # teacher-forced oracle run above is its reference too: every logit, and the maps == post-processing of these logits.
    maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
            torch.empty((H, W), dtype=torch.float64, device="cuda"))
    fl = torch.empty((T, sn.classes, H, W), dtype=torch.float32, device="cuda")
    d_img = torch.from_numpy(img).cuda()
    sn.segment_into(d_img, seed, maps, logits=fl)
    plain = tuple(torch.empty_like(m) for m in maps)
    sn.segment_into(d_img, seed, plain)
    post = mc_segment(fl)
    torch.cuda.synchronize()
    for a, b, c in zip(maps, post, plain):
        assert torch.equal(a, b) and torch.equal(a, c)
    ferr = np.abs(fl.cpu().numpy() - res["logits"])
    print(f"[{kind} T={T} {image} seed={seed}] segment entry point: max|dlogit| {ferr.max():.3e} (mean {ferr.mean():.2e}), "
          f"vs the separate classifier kernel {np.abs(fl.cpu().numpy() - lg).max():.2e}")
    assert ferr.max() < LOGIT_TOL
    np.testing.assert_allclose(maps[1].cpu().numpy(), res["confidence"], atol=LOGIT_TOL / 2, rtol=0)
    np.testing.assert_allclose(maps[2].cpu().numpy(), res["entropy"], atol=5e-3, rtol=0)
    # class flips against the oracle (same switches): bounded, and only where the oracle's own two best mean probabilities are closer
    # than the probability error the logit tolerance allows (|dp| <= |dlogit| / 2) — so that a later change of the device Softmax
    # (v_exp_f32 + one reciprocal per pixel, softmax.hpp) cannot widen the gap unnoticed
    flip = maps[0].cpu().numpy() != res["classes"]
    srt = np.sort(res["mean"], axis=0)
    print(f"[{kind} T={T} {image} seed={seed}] class map vs the oracle at the same switches: {int(flip.sum())} of {flip.size} pixels differ"
          + (f", largest top-2 gap of the oracle's mean among them {float((srt[-1] - srt[-2])[flip].max()):.2e}" if flip.any() else ""))
    assert flip.mean() <= 1e-4
    assert not flip.any() or float((srt[-1] - srt[-2])[flip].max()) <= LOGIT_TOL / 2
    del res, err, ferr, fl, maps, plain, post, srt, flip

    # ---- free-running: the oracle's own switches
    pre = [L["bottom"][0] for L in pools]
    free = oracle.segment(net, w, img, seed, logits_name=lname, shared_prefix=True,
                          keep=pre + [L["top"][1] for L in pools])
    dirty = np.zeros((T, H, W), bool)
    n_free = 0
    for L in pools:
        mname = L["top"][1]
        dm, om = masks[mname].astype(np.int64), free["blobs"][mname].astype(np.int64)
        x = free["blobs"][L["bottom"][0]]
        assert dm.shape == om.shape, (mname, dm.shape, om.shape)
        diff = dm != om
        if not diff.any():
            continue
        n_free += int(diff.sum())
        assert diff.sum() <= 2e-4 * diff.size, (mname, int(diff.sum()))
        flat = x.reshape(x.shape[0], x.shape[1], -1)
        v_o = np.take_along_axis(flat, om.reshape(om.shape[0], om.shape[1], -1), 2).reshape(om.shape)[diff]
        v_d = np.take_along_axis(flat, dm.reshape(dm.shape[0], dm.shape[1], -1), 2).reshape(dm.shape)[diff]
        gap = float((v_o - v_d).max())
        assert gap >= 0 and gap <= NEAR_TIE * max(1.0, float(np.abs(v_o).max())), (mname, gap)
        d = _decoder_influence(net, mname, diff.any(axis=1), lname)
        dirty |= d if d.shape[0] == T else np.broadcast_to(d, dirty.shape)
    err = np.abs(lg - free["logits"]).max(axis=1)                 # (T, H, W)
    moved = err > LOGIT_TOL
    outside = moved & ~dirty
    mism = float((cls != free["classes"]).mean())
    print(f"[{kind} T={T} {image} seed={seed}] free: {n_free} switches differ, {moved.mean():.3%} of the pixels "
          f"move by > 1e-3, receptive fields cover {dirty.mean():.2%}, class map differs at {mism:.3%}, "
          f"max|dlogit| outside the receptive fields {err[~dirty].max():.2e}")
    assert not outside.any(), f"{int(outside.sum())} pixels differ by > 1e-3 outside every flipped switch's receptive field"
    assert dirty.mean() < 0.6
    assert mism <= 3e-3
    clean = ~dirty.any(axis=0)
    np.testing.assert_allclose(ent[clean], free["entropy"][clean], atol=5e-3, rtol=0)
    np.testing.assert_allclose(conf[clean], free["confidence"][clean], atol=LOGIT_TOL / 2, rtol=0)


def _oracle_one_sample(oracle, net, w, img, seed, s, masks, lname):
    """Sample s of the frame on the oracle (the dropout masks are keyed by the global sample index), with the device's
    switches of that sample imposed.  Returns (logits (1, K, H, W), flips)."""
    one = dict(net, shape=[1] + list(net["shape"][1:]))
    fm = {k: (v if v.shape[0] == 1 else v[s:s + 1]) for k, v in masks.items()}
    flips = {}
    res = oracle.segment(one, w, img, seed, sample0=s, logits_name=lname, force_masks=fm, flips=flips, shared_prefix=True)
    return res["logits"], flips


def test_t48_and_its_shards_at_full_size(oracle, kitti_like_bgr):
    """BASELINE configs[3] as bench.py times it on one GPU (T = 48 in one handle) and as it shards (6 samples per rank,
    sample0 = 6 r).  T is the prototxt's batch dimension (bayesian_segnet.cpp:67-70,174-177): at T = 48 a V / M slot of
    conv2_2_D holds 5 GB — where a 32-bit offset would hide.
      * samples 0-11 of the T = 48 frame are bit-identical to the T = 12 handle's (masks are keyed by the global sample);
      * samples 0, 23 and 47: every logit within 1e-3 of the oracle (switches teacher-forced, each a near-tie);
      * a 6-sample shard forward(n_samples = 6, sample0 = 18): logits bit-identical to samples 18-23 of the full pass;
      * the probability sums of the 8 shards + finalize == the single pass: classes identical except at ties of the mean,
        confidence and entropy to 1e-6 (fp32 sums added in another order)."""
    net12, w, sn12 = _handle("standard", 12)
    img = _images(kitti_like_bgr)["kitti"]
    d_img = torch.from_numpy(img).cuda()
    seed = 2024
    _, lg12, _ = sn12.forward(d_img, seed, want_logits=True)
    lg12 = lg12.cpu()
    _cache.clear()                                   # the T = 12 handles: 22 GB each
    torch.cuda.empty_cache()
    net48, _, sn48 = _handle("standard", 48)
    ps48, lg48, _ = sn48.forward(d_img, seed, want_logits=True)
    maps48 = sn48.finalize(ps48)
    torch.cuda.synchronize()
    assert torch.equal(lg48[:12].cpu(), lg12)
    masks = {L["top"][1]: sn48.blob(L["top"][1]) for L in _pool_layers(net48)}
    for s in (0, 23, 47):
        ref, flips = _oracle_one_sample(oracle, net48, w, img, seed, s, masks, "conv1_1_D")
        for name, (count, gap, mag) in flips.items():
            assert gap <= NEAR_TIE * max(mag, 1.0), (s, name, count, gap, mag)
        err = float(np.abs(lg48[s:s + 1].cpu().numpy() - ref).max())
        print(f"[standard T=48 sample {s}] max|dlogit| {err:.3e}, switches forced {sum(c for c, _, _ in flips.values())}")
        assert err < LOGIT_TOL
    lg48_host = lg48.cpu()
    cls48, conf48, ent48 = (m.cpu().numpy() for m in maps48)
    del lg48
    # shards, in the handle that holds the T = 48 blobs (rank r of 8 runs samples 6 r .. 6 r + 5)
    total = torch.zeros_like(ps48)
    for r in range(8):
        ps, lg, _ = sn48.forward(d_img, seed, n_samples=6, sample0=6 * r, want_logits=(r == 3))
        total += ps
        if r == 3:
            torch.cuda.synchronize()
            assert torch.equal(lg.cpu(), lg48_host[18:24])
    cls, conf, ent = (m.cpu().numpy() for m in sn48.finalize(total, t_total=48))
    np.testing.assert_allclose(conf, conf48, atol=1e-6, rtol=0)
    np.testing.assert_allclose(ent, ent48, atol=1e-5, rtol=0)
    assert (cls != cls48).mean() < 1e-5
    assert sn48.gemm_status()[:2] == (2, 0)
    _cache.clear()
    torch.cuda.empty_cache()


def _scale_weights(factor):
    """Convolution weights x factor, the BN that follows / factor: the same function with weights (and Winograd-domain
    products) of another magnitude.  factor is not a power of two, so the roundings differ."""
    def f(net, w):
        layers = net["layers"]
        for i, L in enumerate(layers):
            if L["type"] == "Convolution" and i + 1 < len(layers) and layers[i + 1]["type"] == "BN":
                w[L["name"]][0] *= np.float32(factor)
                w[L["name"]][1] *= np.float32(factor)
                w[layers[i + 1]["name"]][0] /= np.float32(factor)
    return f


def _bn_spread(net, w):
    """BN scale log-uniform in [0.1, 10] per channel (trained SegNet BN scales span far more than the default
    [0.5, 1.5]): channels of very different magnitude meet in every contraction.  E[s^2] = 10.9, so the convolutions
    are scaled by 10.9^-0.5 to keep activations O(1)."""
    rng = np.random.default_rng(3)
    for L in net["layers"]:
        if L["type"] == "BN":
            w[L["name"]][0] = np.exp(rng.uniform(np.log(0.1), np.log(10.0), w[L["name"]][0].shape)).astype(np.float32)
    for L in [L for L in net["layers"] if L["type"] == "Convolution"][1:]:
        w[L["name"]][0] *= np.float32(0.303)


def _bn_offset(net, w):
    """BN shift +3: every activation carries a large common-mode component, which is what the F(4x4) input transform
    (4 d0 - 5 d2 + d4) has to cancel — its worst case.  The weights are centred per filter (zero mean over the taps of
    each input channel), so the offset itself contributes nothing and the logits stay O(1..10)."""
    for L in net["layers"]:
        if L["type"] == "BN":
            w[L["name"]][1] = (w[L["name"]][1] + 3.0).astype(np.float32)
    for L in [L for L in net["layers"] if L["type"] == "Convolution"][1:]:
        Wt = w[L["name"]][0]
        w[L["name"]][0] = (Wt - Wt.mean(axis=(2, 3), keepdims=True)).astype(np.float32) if Wt.shape[2] > 1 else Wt


SWEEP = [("w_x0.3", _scale_weights(0.3), "kitti"), ("w_x3", _scale_weights(3.0), "kitti"), ("bn_0.1_10", _bn_spread, "kitti"),
         ("bn_offset", _bn_offset, "kitti"), ("white", None, "white"), ("black", None, "black")]


@pytest.mark.parametrize("tag,mutate,image", [pytest.param(*sw, marks=() if sw[0] in ("bn_offset", "w_x3") else _SLOW) for sw in SWEEP], ids=[s[0] for s in SWEEP])
def test_f4x4_margin_robustness_sweep(oracle, tag, mutate, image, kitti_like_bgr):
    """How much of the 1e-3 logit budget Winograd F(4x4,3x3) uses depends on the dynamic range of weights and
    activations: weight scale x0.25 / x4, BN scales over two decades, saturated all-255 and all-0 frames (every pooling
    window an exact tie).  SegNet-Standard, full geometry, T = 2, switches teacher-forced; the error is asserted relative
    to the logit magnitude of each variant (1e-3 at |logit| <= 30, the range of the reference configuration)."""
    T = 2
    net, w, sn = _handle("standard", T, mutate=mutate, tag=tag if mutate else "")
    img = {"white": np.full((H, W, 3), 255, np.uint8), "black": np.zeros((H, W, 3), np.uint8)}.get(image)
    if img is None:
        img = _images(kitti_like_bgr)[image]
    lg, masks, *_ = _device_run(sn, net, img, 11)
    flips = {}
    res = oracle.segment(net, w, img, 11, logits_name="conv1_1_D", force_masks=masks, flips=flips, shared_prefix=True)
    for name, (count, gap, mag) in flips.items():
        assert gap <= NEAR_TIE * max(mag, 1.0), (name, count, gap, mag)
    mag = float(np.abs(res["logits"]).max())
    err = float(np.abs(lg - res["logits"]).max())
    print(f"[sweep {tag}/{image}] max|logit| {mag:.2f}, max|dlogit| {err:.3e}, switches forced "
          f"{sum(c for c, _, _ in flips.values())}")
    assert np.isfinite(lg).all()
    assert err < LOGIT_TOL * max(1.0, mag / 30.0)
    if mutate:
        _cache.pop(("standard", T, None, tag))


def test_trained_like_weights_through_a_caffemodel_file(oracle, kitti_like_bgr, tmp_path):
    """Real trained weights never passed through the path (the reference's .caffemodel files are LFS pointers).  This is the closest
    stand-in: tests/trained_like.py — VGG16's per-layer weight magnitudes, heavy-tailed filters of very different norm, dead filters,
    BN gains over two decades on the layer's ACTUAL statistics, ~90 % of the activations zero behind the ReLUs — written as a binary
    .caffemodel of the reference's size (117.8 MB) and loaded with the .prototxt through sivo_segnet_create_from_files, as
    BayesianSegNet's constructor does (bayesian_segnet.cpp:62-64).  SegNet-Standard, full geometry, T = 2, switches teacher-forced;
    prints the accuracy guard's table, the f16x3 scales and the used part of the 1e-3 logit budget."""
    from trained_like import trained_like_weights
    from sivo_amd.segnet import BayesianSegNetParams
    T = 2
    text = _text("standard", T)
    net = oproto.parse(text)
    frame = np.ascontiguousarray(kitti_like_bgr[:H, :W])
    w = trained_like_weights(net, frame)
    proto, model = tmp_path / "trained_like.prototxt", tmp_path / "trained_like.caffemodel"
    proto.write_text(text)
    model.write_bytes(wts.to_caffemodel(net["layers"], w))
    assert abs(model.stat().st_size - 117.8e6) < 0.3e6                      # the size of the reference's bayesian_segnet_kitti.caffemodel
    sn = BayesianSegNet(BayesianSegNetParams(str(proto), str(model)), T=T)
    rep = sn.guard_report()
    mode, overflowed, scales = sn.gemm_status()
    print(f"[trained-like] guard: predicted {rep['predicted']:.3e} of budget {rep['budget']:.3e}, plans built {rep['builds']}, gemm mode {mode}")
    for r in rep["layers"]:
        print(f"[trained-like]   {r['layer']:12s} {r['kernel']:16s} rel_err {r['rel_err']:.2e} rms {r['rel_rms']:.2e} ref_max {r['ref_max']:.3g} level {r['level']}")
    for name, vmax, vs, us in scales:
        print(f"[trained-like]   scale {name:12s} |V|max {vmax:.4g} vscale 2^{int(np.log2(vs))} uscale 2^{int(np.log2(us))}")
    for image, seed in (("kitti", 11), ("synthetic", 3)):
        img = _images(kitti_like_bgr)[image]
        lg, masks, cls, conf, ent = _device_run(sn, net, img, seed)
        assert not sn.take_overflow()
        flips = {}
        res = oracle.segment(net, w, img, seed, logits_name="conv1_1_D", force_masks=masks, flips=flips, shared_prefix=True)
        for name, (count, gap, mag) in flips.items():
            assert gap <= NEAR_TIE * max(mag, 1.0), (name, count, gap, mag)
        mag = float(np.abs(res["logits"]).max())
        err = float(np.abs(lg - res["logits"]).max())
        top2 = np.sort(res["logits"], axis=1)[:, -2:]
        margin = float(np.median(top2[:, 1] - top2[:, 0]))
        print(f"[trained-like {image}] max|logit| {mag:.2f}, max|dlogit| {err:.3e} = {err / (LOGIT_TOL * max(1.0, mag / 30.0)):.2f} of the budget, "
              f"median top-1 margin {margin:.3f}, switches forced {sum(c for c, _, _ in flips.values())}, "
              f"class map differs on {(cls != res['classes']).mean():.2e}, entropy max diff {np.abs(ent - res['entropy']).max():.2e}")
        assert np.isfinite(lg).all()
        assert err < LOGIT_TOL * max(1.0, mag / 30.0)
        assert (cls != res["classes"]).mean() < 3e-3
    assert sn.gemm_status()[0] == 2                                        # the matrix-core layers stayed on f16x3

    # What the guard's verdict costs and what it buys on this family: the same files through the diagnostic build with the guard off (the plan
    # as first made: every wide layer on F(4x4)) — its error against the oracle, and the forward time of both plans (T = 2, mean of 5).
    def ms_per_forward(net_handle):
        d_img = torch.from_numpy(frame).cuda()
        for i in range(4):                       # (the oracle ran on the CPU meanwhile: the idle GPU has dropped its clocks)
            net_handle.forward(d_img, 1 + i)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(5):
            net_handle.forward(d_img, 6 + i)
        ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / 5
    from sivo_amd import _lib
    t_guarded = ms_per_forward(sn)
    os.environ["SIVO_GUARD"] = "0"
    try:
        with _lib.use("diag"):
            sn0 = BayesianSegNet(BayesianSegNetParams(str(proto), str(model)), T=T)
            lg0, masks0, *_ = _device_run(sn0, net, frame, 11)
            t_plain = ms_per_forward(sn0)
    finally:
        os.environ.pop("SIVO_GUARD", None)
    res0 = oracle.segment(net, w, frame, 11, logits_name="conv1_1_D", force_masks=masks0, shared_prefix=True)
    err0 = float(np.abs(lg0 - res0["logits"]).max())
    mag0 = float(np.abs(res0["logits"]).max())
    print(f"[trained-like] plan as first made (guard off, diagnostic build): max|dlogit| {err0:.3e} = {err0 / (LOGIT_TOL * max(1.0, mag0 / 30.0)):.2f} of the budget, "
          f"{t_plain:.2f} ms per T = 2 forward; guarded plan (the product's): {t_guarded:.2f} ms — the guard moved "
          f"{sum(1 for r in rep['layers'] if r['level'] > 0)} of {len(rep['layers'])} guarded layers off F(4x4)")
    del sn0
    torch.cuda.empty_cache()
