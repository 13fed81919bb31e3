#!/usr/bin/env python3
"""Per-layer kernel times of a SegNet-Standard frame (T = 12, 352 x 1024, one lane, every kernel bracketed by HIP events on its
launch stream: sivo_segnet_profile), with the GEMM / scale status of the f16x3 layers.  GPU box only.
Usage: python tools/layer_times.py [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
os.environ.setdefault("SIVO_LANES", "1")
from sivo_amd import netspec, weights as wts
from sivo_amd.segnet import BayesianSegNet

T, H, W = 12, 352, 1024
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
text = netspec.standard_prototxt(T, H, W)
layers = netspec.parse_layers(text)
if os.environ.get("LT_DIAG"):          # the diagnostic build (A/B switches): LT_DIAG=1 SIVO_...=x python tools/layer_times.py
    from sivo_amd import _lib
    _lib.use("diag").__enter__()
sn = BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=T)
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
        torch.empty((H, W), dtype=torch.float64, device="cuda"))
for i in range(3):
    sn.segment_into(img, i, maps)
torch.cuda.synchronize()
sn.profile(True, reset=True)
for i in range(frames):
    sn.segment_into(img, 100 + i, maps)
torch.cuda.synchronize()
rows = sn.profile_read()
sn.profile(False)
total = 0.0
print(f"{'layer':14s} {'kernel':28s} {'ms':>8s} {'TFLOP/s alg':>12s} {'GB/s alg':>9s}")
for r in rows:
    if not r["launches"]:
        continue
    ms = r["ms_total"] / r["launches"]
    total += ms
    fl = r["flops_per_sample"] * r["samples"]
    by = r["bytes_per_sample"] * r["samples"]
    print(f"{r['layer']:14s} {r['kernel'][:28]:28s} {ms:8.3f} {fl / ms / 1e9 if fl else 0:12.1f} {by / ms / 1e6 if by else 0:9.0f}")
print(f"sum {total:.3f} ms per frame")
mode, ov, layers_st = sn.gemm_status()
print("gemm mode", mode, "overflow frames", ov)
for L in layers_st:
    print("  ", L)
