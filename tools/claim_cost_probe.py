#!/usr/bin/env python3
"""Diagnostic build (bridge as shipped), GPU box: what the 160 KB LDS claim of the f16x3 GEMM costs.  SegNet-Standard T = 12 at
352 x 1024, default lanes, 40 timed frames through segment_into, once with the claim (as shipped) and once with the GEMM asking for
its exact LDS (SIVO_H3_LDS_ALL=0: bridge workgroups of the other lane then share CUs with it) — each in its own process; every
frame's maps are compared with the first frame's (same seed) bit for bit.
    python tools/claim_cost_probe.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def body(name):
    import torch
    from sivo_amd import _lib, netspec, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    from bench import make_inputs
    H, W, T = 352, 1024, 12
    text = netspec.standard_prototxt(T, H, W)
    layers = netspec.parse_layers(text)
    flat = wts.pack(layers, wts.synth_weights(layers, 42))
    img = torch.from_numpy(make_inputs(H, W)[0]).cuda()
    with _lib.use("diag"):
        sn = BayesianSegNet(prototxt=text, weights=flat, T=T)

        def maps():
            return (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
                    torch.empty((H, W), dtype=torch.float64, device="cuda"))
        ref = maps()
        sn.segment_into(img, 2000, ref)
        torch.cuda.synchronize()
        for _ in range(5):
            sn.segment_into(img, 2000, maps())
        torch.cuda.synchronize()
        res = []
        for rep in range(3):
            outs = [maps() for _ in range(40)]
            t0 = time.perf_counter()
            for m in outs:
                sn.segment_into(img, 2000, m)
            torch.cuda.synchronize()
            res.append(1e3 * (time.perf_counter() - t0) / 40)
            bad = sum(1 for m in outs if not all(torch.equal(a, b) for a, b in zip(m, ref)))
        print(f"[{name}] SegNet-Standard T = 12 alone, ms per frame over 40 frames, three times: {', '.join(f'{r:.3f}' for r in res)}; frames of the last 40 that differ from the first: {bad}; "
              f"overflow {sn.take_overflow()}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        body(sys.argv[2])
    else:
        for name, env in (("GEMM claims 160 KB (as shipped)", {}), ("GEMM with its exact LDS", {"SIVO_H3_LDS_ALL": "0"}), ("GEMM claims 160 KB, again", {}), ("GEMM with its exact LDS, again", {"SIVO_H3_LDS_ALL": "0"})):
            e = dict(os.environ); e.update(env)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, timeout=300)
