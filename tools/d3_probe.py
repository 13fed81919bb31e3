#!/usr/bin/env python3
"""Where the time of the direct f16x3 convolution goes: the production kernel and its compile-time ablations
(libsivo_hip_diag.so, `make -C sivo_amd/csrc diag`; SIVO_D3_ABL bits: 1 no patch loads, 2 no weight DMA, 4 no output stores,
8 no MFMAs, 16 no patch split / LDS writes) on the shapes it runs at in SegNet-Standard T = 12.  GPU box only.
Usage: python tools/d3_probe.py [iters]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sivo_amd", "libsivo_hip_diag.so"))
vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
lib.sivo_debug_conv3_h3_dev.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp, i, f, vp, i, C.POINTER(d), C.POINTER(i)]
lib.sivo_last_error.restype = C.c_char_p
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [("conv2_1_D 128->64 176x512", 12, 128, 64, 176, 512), ("conv1_2_D* 64->64 352x1024", 12, 64, 64, 352, 1024),
          ("conv2_2_D* 128->128 176x512", 12, 128, 128, 176, 512)]      # (* = without the Upsample in front: ablations exist for the plain form only)
VARIANTS = [("as built", None), ("no patch loads", 1), ("no weight DMA", 2), ("no loads at all", 3), ("no stores", 4), ("no split", 16),
            ("no loads, no split", 19), ("MFMA + LDS reads only", 23), ("no MFMA", 8)]
FORMS = sys.argv[2] if len(sys.argv) > 2 else "10"      # stage loops to run: "1" interleaved (default), "0" phased
rng = np.random.default_rng(0)
for name, N, Cin, Cout, H, W in SHAPES:
    x = (torch.randn((N, Cin, H, W), device="cuda").clamp_min(0) * 3).contiguous()
    out = torch.empty((N, Cout, H, W), device="cuda")
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    one = np.ones(Cout, np.float32)
    flop = 2.0 * 9 * Cin * Cout * H * W * N * 3
    for vname, abl in [(f"{v} [form {form}]", ab) for form in FORMS for v, ab in VARIANTS]:
        os.environ["SIVO_D3_FORM"] = vname[-2]
        os.environ.pop("SIVO_D3_ABL", None)
        if abl is not None:
            os.environ["SIVO_D3_ABL"] = str(abl)
        ms, ov = d(0), i(0)
        rc = lib.sivo_debug_conv3_h3_dev(N, Cin, Cout, H, W, x.data_ptr(), None, wt.ctypes.data, one.ctypes.data, one.ctypes.data, 1, f(8.0),
                                         out.data_ptr(), iters, C.byref(ms), C.byref(ov))
        if rc:
            print(name, vname, "error", lib.sivo_last_error().decode()); continue
        print(f"{name:30s} {vname:34s} {ms.value:8.4f} ms   {flop / ms.value / 1e9:8.1f} TFLOP/s executed", flush=True)
