#!/usr/bin/env python3
"""One launch set of the direct f16x3 convolution for a rocprofv3 --pmc pass (production library): conv1_2_D's shape through its
Upsample, conv2_1_D's plain.  GPU box only.   rocprofv3 --pmc <counters> -- python tools/d3_pmc.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from sivo_amd import segnet
rng = np.random.default_rng(0)
for N, Cin, Cout, H, W, unpool in [(12, 64, 64, 352, 1024, True), (12, 128, 64, 176, 512, False)]:
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = (torch.randn((N, Cin, h, w), device="cuda").clamp_min(0) * 3).contiguous()
    mask = torch.randint(0, 4, (N, Cin, h, w), device="cuda", dtype=torch.uint8) if unpool else None
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.05).astype(np.float32)
    one = np.ones(Cout, np.float32)
    segnet.conv3_h3(x, wt, one, one * 0, relu=True, mask=mask, iters=3)
torch.cuda.synchronize()
