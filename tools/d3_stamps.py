#!/usr/bin/env python3
"""Cycle stamps of the interleaved direct f16x3 convolution (diag build, SIVO_D3_ABL & 64): where an iteration's time goes."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sivo_amd", "libsivo_hip_diag.so"))
vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
lib.sivo_debug_conv3_h3_dev.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp, i, f, vp, i, C.POINTER(d), C.POINTER(i)]
os.environ["SIVO_D3_STAMPS"] = "1"
rng = np.random.default_rng(0)
for name, N, Cin, Cout, H, W in [("conv1_2_D* 64->64 352x1024", 12, 64, 64, 352, 1024), ("conv2_1_D 128->64 176x512", 12, 128, 64, 176, 512)]:
    x = (torch.randn((N, Cin, H, W), device="cuda").clamp_min(0) * 3).contiguous()
    out = torch.empty((N, Cout, H, W), device="cuda")
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.05).astype(np.float32)
    one = np.ones(Cout, np.float32)
    for abl in sys.argv[1:] or ["64", "65", "68"]:
        os.environ["SIVO_D3_ABL"] = abl
        ms, ov = d(0), i(0)
        print(name, "ABL", abl, flush=True)
        lib.sivo_debug_conv3_h3_dev(N, Cin, Cout, H, W, x.data_ptr(), None, wt.ctypes.data, one.ctypes.data, one.ctypes.data, 1, f(8.0), out.data_ptr(), 5, C.byref(ms), C.byref(ov))
        print("   ", round(ms.value, 4), "ms", flush=True)
