#!/usr/bin/env python3
"""Where the time of the PACKED forms of the direct f16x3 convolution goes (conv3_h3.hip IN_PK / IN_PK_UNPOOL, OUT_PK): the
production kernel, its compile-time ablations and its cycle stamps (libsivo_hip_diag.so, `make -C sivo_amd/csrc diag`;
SIVO_D3_ABL bits: 1 no patch DMA / pooled loads, 2 no weight DMA, 4 no output stores, 8 no MFMAs, 16 no Upsample expansion,
64 s_memtime stamps) on the three decoder layers of SegNet-Standard, T = 12.  GPU box only.
Usage: python tools/d3pk_probe.py [iters]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sivo_amd", "libsivo_hip_diag.so"))
vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
lib.sivo_debug_conv3_h3_pk_dev.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp, i, f, f, i, vp, i, C.POINTER(d), C.POINTER(i)]
lib.sivo_last_error.restype = C.c_char_p
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# name, N, Cin, Cout, H, W, unpool, packed output
SHAPES = [("conv2_1_D 128->64 176x512 (PK -> PK)", 12, 128, 64, 176, 512, False, True),
          ("conv1_2_D 64->64 352x1024 (PK unpool -> fp32)", 12, 64, 64, 352, 1024, True, False),
          ("conv2_2_D 128->128 176x512 (PK unpool -> PK)", 12, 128, 128, 176, 512, True, True)]
VARIANTS = [("as built", None), ("no patch DMA / pooled loads", 1), ("no weight DMA", 2), ("no loads at all", 3), ("no stores", 4),
            ("no Upsample expansion", 16), ("no loads, no stores", 7), ("MFMA + LDS reads only", 23), ("no MFMA", 8), ("stamps", 64)]
rng = np.random.default_rng(0)
for name, N, Cin, Cout, H, W, unpool, pk_out in SHAPES:
    h, w = (H // 2, W // 2) if unpool else (H, W)
    if os.environ.get("SIVO_PROBE_SHAPE") and os.environ["SIVO_PROBE_SHAPE"] not in name:
        continue
    x = (torch.randn((N, Cin, h, w), device="cuda").clamp_min(0) * 3).contiguous()
    mask = torch.randint(0, 4, (N, Cin, h, w), device="cuda", dtype=torch.uint8) if unpool else None
    out = torch.empty((N, Cout, H, W), device="cuda")
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    if os.environ.get("SIVO_PROBE_ZEROS"):          # all-zero operands: the same instructions, no clock throttling (tools/h3_pmc.sh)
        x.zero_(); wt[:] = 0
    one = np.ones(Cout, np.float32)
    flop = 2.0 * 9 * Cin * Cout * H * W * N * 3
    for vname, abl in VARIANTS:
        if abl == 16 and not unpool:
            continue
        if os.environ.get("SIVO_PROBE_ONLY") and not any(k == vname for k in os.environ["SIVO_PROBE_ONLY"].split(",")):
            continue
        os.environ.pop("SIVO_D3_ABL", None)
        os.environ.pop("SIVO_D3_STAMPS", None)
        if abl is not None:
            os.environ["SIVO_D3_ABL"] = str(abl)
        if abl == 64:
            os.environ["SIVO_D3_STAMPS"] = "1"
        ms, ov = d(0), i(0)
        rc = lib.sivo_debug_conv3_h3_pk_dev(N, Cin, Cout, H, W, x.data_ptr(), mask.data_ptr() if unpool else None, wt.ctypes.data, one.ctypes.data,
                                            one.ctypes.data, 1, f(8.0), f(4.0), 1 | (2 if pk_out else 0), out.data_ptr(), iters, C.byref(ms), C.byref(ov))
        if rc:
            print(name, vname, "error", lib.sivo_last_error().decode()); continue
        print(f"{name:48s} {vname:30s} {ms.value:8.4f} ms   {flop / ms.value / 1e9:8.1f} TFLOP/s executed", flush=True)
