#!/usr/bin/env python3
"""Debug aid (GPU box): SegNet-Standard T = 12 at 352 x 1024 — three lanes against one lane (must be bit-identical) under
environment variants, to find which piece breaks the identity."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sivo_amd import netspec, weights as wts          # noqa: E402
from sivo_amd.segnet import BayesianSegNet             # noqa: E402

H, W, T = 352, 1024, 12
text = netspec.standard_prototxt(T, H, W)
layers = netspec.parse_layers(text)
flat = wts.pack(layers, wts.synth_weights(layers, 42))
from bench import make_inputs                          # noqa: E402
img = torch.from_numpy(make_inputs(H, W)[0]).cuda()
VARIANTS = [("default", {}), ("x6", {"SIVO_GEMM": "x6"}), ("nobridge", {"SIVO_NO_FUSE_BRIDGE": "1"}), ("nopool", {"SIVO_NO_FUSE_POOL": "1"}),
            ("nostagger", {"SIVO_H3_STAGGER": "0"}), ("tile1", {"SIVO_H3_TILE": "1"}), ("tile0", {"SIVO_H3_TILE": "0"})]
only = sys.argv[1:] or [v[0] for v in VARIANTS]


def make(lanes, env):
    old = {k: os.environ.get(k) for k in list(env) + ["SIVO_LANES"]}
    os.environ.update(env); os.environ["SIVO_LANES"] = str(lanes)
    try:
        return BayesianSegNet(prototxt=text, weights=flat, T=T)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


for name, env in VARIANTS:
    if name not in only:
        continue
    if name in ("nostagger", "tile1", "tile0"):
        print(name, "needs its own process (static switch): run as  SIVO_H3_...=x python tools/h3_debug2.py default"); continue
    a, b = make(int(os.environ.get("DBG_LANES_A", "3")), env), make(1, env)
    for seed in (99, 5, 7):
        _, la, _ = a.forward(img, seed, want_logits=True)
        _, lb, _ = b.forward(img, seed, want_logits=True)
        _, la2, _ = a.forward(img, seed, want_logits=True)
        _, lb2, _ = b.forward(img, seed, want_logits=True)
        torch.cuda.synchronize()
        print("   one-lane handle repeatable:", bool(torch.equal(lb, lb2)), " scales equal:", a.gemm_status()[2] == b.gemm_status()[2])
        d = (la - lb).abs()
        rep = torch.equal(la, la2)
        nz = (d > 0)
        info = ""
        if nz.any():
            idx = nz.nonzero()
            info = f" differing {int(nz.sum())} samples {sorted(set(idx[:, 0].tolist()))} y {int(idx[:, 2].min())}-{int(idx[:, 2].max())} x {int(idx[:, 3].min())}-{int(idx[:, 3].max())} max {float(d.max()):.3e}"
        print(f"{name:10s} seed {seed}: 3 lanes == 1 lane: {bool(torch.equal(la, lb))}; repeatable {rep}; status {a.gemm_status()[:2]} {b.gemm_status()[:2]}{info}", flush=True)
    del a, b
    torch.cuda.empty_cache()
