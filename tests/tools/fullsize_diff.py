#!/usr/bin/env python3
"""Where do full-size logits differ from the oracle?  Attributes the outliers to max-pool argmax flips."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle, prototxt as oproto
from sivo_amd import netspec, weights as wts
from sivo_amd.segnet import BayesianSegNet
kind = sys.argv[1] if len(sys.argv) > 1 else "standard"
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
T, H, W = 2, 352, 1024
text = netspec.basic_prototxt(T, H, W) if kind == "basic" else netspec.standard_prototxt(T, H, W)
net = oproto.parse(text); w = wts.synth_weights(net["layers"], 42)
sn = BayesianSegNet(prototxt=text, weights=wts.pack(net["layers"], w), T=T)
img = np.load(os.path.join(ROOT, "tests/golden/frame_bgr_352x1024.npy"))[:H, :W].copy()
ob = oracle.run_net(net, w, oracle.preprocess(img, T, H, W), SEED)
_, logits, _ = sn.forward(torch.from_numpy(img).cuda(), SEED, want_logits=True)
torch.cuda.synchronize()
name = "dense_softmax_inner_prod" if kind == "basic" else "conv1_1_D"
prob_sum, _, _ = sn.forward(torch.from_numpy(img).cuda(), SEED)
cls, conf, ent = sn.finalize(prob_sum)
mean = oracle.mc_mean(ob["__last__"]); cls_o, conf_o, ent_o = oracle.mc_finalize(mean)
cls = cls.cpu().numpy(); ent = ent.cpu().numpy()
print("free-running: class map differs at %.4f %% of the pixels; |d entropy| mean %.2e, 99.9th pct %.2e, max %.2e" % (
    100.0 * (cls != cls_o).mean(), np.abs(ent - ent_o).mean(), np.quantile(np.abs(ent - ent_o), 0.999), np.abs(ent - ent_o).max()))
err = np.abs(logits.cpu().numpy() - ob[name])
print("max", err.max(), "frac>1e-3", (err > 1e-3).mean(), "frac>1e-4", (err > 1e-4).mean(), "median", np.median(err))
for L in net["layers"]:
    if L["type"] == "Pooling":
        try:
            g = sn.blob(L["top"][1]); o = ob[L["top"][1]]
            if g.shape[0] == 1: o = o[:1]
            print(L["name"], "mask flips", int((g != o).sum()), "of", g.size)
        except Exception as e:
            print(L["name"], e)
