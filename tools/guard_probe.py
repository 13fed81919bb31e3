#!/usr/bin/env python3
"""GPU box: the load-time accuracy guard on SegNet-Standard at full geometry (T = 2) for the synthetic weights and for BN-offset
variants of growing common mode (tests/test_gpu_segnet_fullsize.py::_bn_offset: +3 is the sweep's case) — per-layer relative
errors, budget, reroutes, guard time — and, against the CPU oracle with the device's pooling switches imposed, the largest logit
error of the guarded handle and of an UNGUARDED one (diagnostic build, SIVO_GUARD=0).
    python tools/guard_probe.py [offsets ...]      default: 0 3 30 100"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from oracle import oracle as O, prototxt as oproto
    from sivo_amd import _lib, netspec, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    from bench import make_inputs
    H, W, T = 352, 1024, 2
    args = sys.argv[1:] or ["0", "3", "30", "100"]
    boosts = [int(a.split(":")[1]) for a in args if a.startswith("boost:")]         # boost:-16 = the f16x3 scales forced 2^-16 (diagnostic build)
    offsets = [float(x) for x in args if not x.startswith("boost:")]
    text = netspec.standard_prototxt(T, H, W)
    img = make_inputs(H, W)[0]
    d = torch.from_numpy(img).cuda()
    for off in offsets + [("boost", b) for b in boosts]:
        boost = None
        if isinstance(off, tuple):
            boost, off = off[1], 0.0
        net = oproto.parse(text)
        w = wts.synth_weights(net["layers"], 42)
        if off:
            for L in net["layers"]:
                if L["type"] == "BN":
                    w[L["name"]][1] = (w[L["name"]][1] + off).astype(np.float32)
            for L in [L for L in net["layers"] if L["type"] == "Convolution"][1:]:
                Wt = w[L["name"]][0]
                w[L["name"]][0] = (Wt - Wt.mean(axis=(2, 3), keepdims=True)).astype(np.float32) if Wt.shape[2] > 1 else Wt
        flat = wts.pack(net["layers"], w)
        res = None
        for guarded in (True, False):
            t0 = time.perf_counter()
            if boost is not None:
                os.environ["SIVO_H3_BOOST"] = str(boost)
            if guarded and boost is None:
                sn = BayesianSegNet(prototxt=text, weights=flat, T=T)
            else:
                if not guarded:
                    os.environ["SIVO_GUARD"] = "0"
                with _lib.use("diag"):
                    sn = BayesianSegNet(prototxt=text, weights=flat, T=T)
                os.environ.pop("SIVO_GUARD", None)
            os.environ.pop("SIVO_H3_BOOST", None)
            t_build = time.perf_counter() - t0
            g = sn.guard_report()
            _, lg, _ = sn.forward(d, 11, want_logits=True)
            torch.cuda.synchronize()
            lg = lg.cpu().numpy()
            masks = {L["top"][1]: sn.blob(L["top"][1]) for L in net["layers"] if L["type"] == "Pooling"}
            flips = {}
            res = O.segment(net, w, img, 11, logits_name="conv1_1_D", force_masks=masks, flips=flips, shared_prefix=True)
            err = float(np.abs(lg - res["logits"]).max())
            mag = float(np.abs(res["logits"]).max())
            print(f"[{'bn_offset %g' % off if boost is None else 'scales 2^%d' % boost}] {'guarded' if guarded else 'UNGUARDED'}: construction {t_build:.2f} s (guard {g['ms']:.0f} ms, plans {g['builds']}), "
                  f"predicted {g['predicted']:.2e} / budget {g['budget']:.2e} of the logit scale, max|logit| {g['logit_max']:.1f} (oracle {mag:.1f}); max|dlogit| vs oracle {err:.3e}; "
                  f"rerouted {[(r['layer'], r['level'], r['kernel']) for r in g['layers'] if r['level']]}", flush=True)
            if guarded:
                for r in g["layers"]:
                    print(f"    {r['layer']:12s} {r['kernel']:20s} level {r['level']} rel_err {r['rel_err']:.2e} (first plan {r['first_rel_err']:.2e}) rel_rms {r['rel_rms']:.2e} max|ref| {r['ref_max']:.2f}")
            del sn
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
