#!/usr/bin/env python3
"""Where the time of the f16x3 classifier + MC kernel goes (conv_cls_h3.hip): the production form and the
compile-time ablations of libsivo_hip_diag.so (SIVO_CLS_ABL bits: 1 no MFMAs, 2 no Softmax / sum, 4 no patch DMA after the first stage,
8 no fragment reads after the first tap) at the network's shape (T = 12, 64 -> 15, 352 x 1024).  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sivo_amd", "libsivo_hip_diag.so"))
vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
lib.sivo_debug_conv_cls_h3_dev.argtypes = [i, i, i, i, i, vp, vp, vp, vp, i, f, vp, vp, vp, vp, i, C.POINTER(d)]
lib.sivo_last_error.restype = C.c_char_p
T, Cin, K, H, W = 12, 64, 15, 352, 1024
x = (torch.randn((T, Cin, H, W), device="cuda").clamp_min(0) * 2).contiguous()
rng = np.random.default_rng(0)
wt = (rng.standard_normal((K, Cin, 3, 3)) * 0.06).astype(np.float32)
one = np.ones(K, np.float32)
logits = torch.empty((T, K, H, W), device="cuda")
cls = torch.empty((H, W), dtype=torch.uint8, device="cuda")
conf = torch.empty((H, W), dtype=torch.float64, device="cuda")
ent = torch.empty((H, W), dtype=torch.float64, device="cuda")
for name, env in [("as built", {}), ("no MFMA", {"SIVO_CLS_ABL": "1"}), ("no softmax", {"SIVO_CLS_ABL": "2"}),
                  ("no patch DMA", {"SIVO_CLS_ABL": "4"}), ("no fragment reads", {"SIVO_CLS_ABL": "8"}), ("no MFMA, no reads", {"SIVO_CLS_ABL": "9"}),
                  ("DMA + barriers only", {"SIVO_CLS_ABL": "11"}), ("barriers only", {"SIVO_CLS_ABL": "15"}), ("MFMA + reads only", {"SIVO_CLS_ABL": "6"})]:
    for k in ("SIVO_CLS_ABL",):
        os.environ.pop(k, None)
    os.environ.update(env)
    ms = d(0)
    rc = lib.sivo_debug_conv_cls_h3_dev(T, Cin, K, H, W, x.data_ptr(), wt.ctypes.data, one.ctypes.data, one.ctypes.data, 0, f(16.0), logits.data_ptr(),
                                        cls.data_ptr(), conf.data_ptr(), ent.data_ptr(), 10, C.byref(ms))
    print(f"{name:24s} {ms.value:8.4f} ms   {T * Cin * H * W * 4 / ms.value / 1e9:6.2f} TB/s of input" if rc == 0 else f"{name}: {lib.sivo_last_error().decode()}", flush=True)
