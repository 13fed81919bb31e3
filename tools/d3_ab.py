#!/usr/bin/env python3
"""Output hashes of the direct f16x3 convolution on fixed inputs with a given build of the library (bit-identity of a kernel change
across builds).  Usage: python tools/d3_ab.py <path to libsivo_hip.so>"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from sivo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from sivo_amd import segnet
rng = np.random.default_rng(0)
g = torch.Generator(device="cuda").manual_seed(3)
for N, Cin, Cout, H, W, unpool, relu in [(3, 64, 128, 44, 136, True, True), (2, 128, 64, 30, 100, False, False), (12, 64, 64, 352, 1024, True, True)]:
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = torch.randn((N, Cin, h, w), generator=g, device="cuda")
    mask = torch.randint(0, 4, (N, Cin, h, w), generator=g, device="cuda", dtype=torch.uint8) if unpool else None
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.05).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32); sh = rng.uniform(-0.2, 0.2, Cout).astype(np.float32)
    out, _, ov = segnet.conv3_h3(x, wt, sc, sh, relu=relu, mask=mask)
    print(N, Cin, Cout, H, W, unpool, relu, ov, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
