"""CPU: the two REAL SegNet graphs end to end on the oracle against an independent evaluation in PyTorch (float64).

Caffe-SegNet is an empty submodule of the reference (README.md:49,94, .gitmodules:1-3), so the SegNet oracle cannot be pinned
against the reference's own arithmetic; tests/test_oracle_segnet.py checks its layers one at a time and a tiny net.  This
file closes the gap between "every layer type" and "the graphs the reference ships": the layer graphs of
config/bayesian_segnet/{standard,basic}/kitti/*.prototxt (sivo_amd.netspec reproduces them layer for layer —
tests/test_host_logic.py::test_netspec_reproduces_reference_graph compares with the reference files) at FULL channel widths,
64 x 128 input, T = 2, are executed
  (a) by oracle.run_net (fp32, the restatement every GPU parity test is measured against), and
  (b) by torch.nn.functional in float64, walking the same parsed prototxt: conv2d, the BN affine, relu,
      max_pool2d(return_indices) / max_unpool2d, local_response_norm, softmax, and the test-time Dropout with the SAME
      Philox masks (taken from the oracle's dropout applied to ones — the mask generator itself has known-answer tests).
Every blob of the graph is compared.  Max pooling is discontinuous: where torch's f64 activations and the oracle's fp32
activations order two window elements differently the decoder would diverge by O(1), so (b) pools with the oracle's
switches and the test asserts separately that torch's own switches differ from them only at near-ties."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import prototxt as oproto
from sivo_amd import netspec, weights as wts

H, W, T = 64, 128, 2
SEED = 77


def _torch_graph(net, w, blob, oracle, masks_from):
    blobs = {net["input"]: torch.from_numpy(blob.astype(np.float64))}
    site = 0
    flips = []
    for L in net["layers"]:
        t = L["type"]; bot = [blobs[b] for b in L["bottom"]]
        if t == "Convolution":
            Wt, b = w[L["name"]]
            out = F.conv2d(bot[0], torch.from_numpy(Wt.astype(np.float64)), torch.from_numpy(b.astype(np.float64)), padding=L["pad"])
        elif t == "BN":
            s, sh = w[L["name"]]
            out = bot[0] * torch.from_numpy(s.astype(np.float64)).view(1, -1, 1, 1) + torch.from_numpy(sh.astype(np.float64)).view(1, -1, 1, 1)
        elif t == "ReLU":
            out = F.relu(bot[0])
        elif t == "Pooling":
            x = bot[0]
            own, idx = F.max_pool2d(x, L["kernel_size"], L["stride"], ceil_mode=True, return_indices=True)
            om = torch.from_numpy(np.asarray(masks_from[L["top"][1]]).astype(np.int64))
            om = om.expand(idx.shape) if om.shape != idx.shape else om
            forced = torch.gather(x.flatten(2), 2, om.flatten(2)).view(own.shape)
            diff = idx != om
            flips.append((L["name"], int(diff.sum()), float((own - forced)[diff].max()) if diff.any() else 0.0, float(x.abs().max())))
            out = forced
            blobs[L["top"][1]] = om
        elif t == "Upsample":
            m = bot[1]
            if m.shape[0] != bot[0].shape[0]:
                m = m.expand(bot[0].shape)
            out = F.max_unpool2d(bot[0], m.contiguous(), 2, 2, output_size=(bot[0].shape[2] * 2, bot[0].shape[3] * 2))
        elif t == "Dropout":
            x = bot[0]
            if L["sample_weights_test"]:
                if x.shape[0] == 1:
                    x = x.expand(T, -1, -1, -1)
                m2 = oracle.dropout(np.ones(tuple(x.shape), np.float32), site, 0, SEED, L["dropout_ratio"])      # 0 or 1 / (1 - ratio)
                out = x * torch.from_numpy(m2.astype(np.float64))
            else:
                out = x
            site += 1
        elif t == "LRN":
            out = F.local_response_norm(bot[0], L["local_size"], L["alpha"], L["beta"], 1.0)
        elif t == "Softmax":
            out = F.softmax(bot[0], 1)
        else:
            raise ValueError(t)
        blobs[L["top"][0]] = out
    return blobs, flips


@pytest.mark.parametrize("kind", ["standard", "basic"])
def test_reference_graph_on_the_oracle_equals_torch_f64(oracle, kind, kitti_like_bgr):
    text = (netspec.standard_prototxt if kind == "standard" else netspec.basic_prototxt)(T, H, W)
    net = oproto.parse(text)
    w = wts.synth_weights(net["layers"], 42)
    img = np.ascontiguousarray(kitti_like_bgr[100:100 + H, 300:300 + W])
    blob = oracle.preprocess(img, 1, H, W)
    names = [L["top"][j] for L in net["layers"] for j in range(len(L["top"]))]
    ob = oracle.run_net(net, w, blob, SEED, keep=names, expand_to=T)
    tb, flips = _torch_graph(net, w, blob, oracle, ob)
    # torch's own switches: identical, or near-ties of its own f64 activations
    for name, count, gap, mag in flips:
        assert gap <= 1e-5 * max(mag, 1.0), (name, count, gap)
    assert sum(c for _, c, _, _ in flips) <= 50
    worst = ("", 0.0)
    n_checked = 0
    for L in net["layers"]:
        top = L["top"][0]
        if top not in ob:
            continue
        a, b = ob[top], tb[top].numpy()
        if a.shape != b.shape:
            b = np.broadcast_to(b, a.shape) if b.shape[0] == 1 else b
        assert a.shape == b.shape, (L["name"], a.shape, b.shape)
        err = float(np.abs(a - b).max())
        tol = 1e-4 * max(1.0, float(np.abs(b).max()))
        assert err <= tol, (L["name"], L["type"], err, tol)
        if err / tol > worst[1]:
            worst = (L["name"], err / tol)
        n_checked += 1
    assert n_checked >= (20 if kind == "basic" else 60)
    logits = tb[net["layers"][-1]["bottom"][0]]
    assert float(logits.abs().max()) > 0.5
    print(f"[{kind}] {n_checked} layer outputs compared, worst at {worst[0]}: {worst[1]:.3f} of the tolerance; "
          f"{sum(c for _, c, _, _ in flips)} switches of torch's own pooling differ (near-ties)")
