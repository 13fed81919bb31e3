#!/usr/bin/env python3
"""Wave priorities of the two roles of wino4_gemm_x6p_kernel (SIVO_X6_PRIO: bits 0-1 consumers, bits 2-3 producers) and, with
the argument `bg`, the U fragments taken from global memory instead of LDS (SIVO_X6_BGLOBAL=1; 2: and two stages per barrier; 3: and the second stage's V fragments read early): whole-layer
times (input transform + GEMM + output transform, sivo_debug_conv) on the GEMM shapes of SegNet-Standard.  GPU box only."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sivo_amd._lib import lib, check

SHAPES = {"conv4_2": (12, 512, 512, 44, 128), "conv5_2": (12, 512, 512, 22, 64), "conv3_2_D": (12, 256, 256, 88, 256),
          "conv2_2_D": (12, 128, 128, 176, 512)}


def run(shape, iters=20):
    N, ci, co, H, W = shape
    ms = C.c_double()
    check(lib().sivo_debug_conv(N, ci, co, H, W, 3, iters, 512 | 2048, C.byref(ms)))
    return ms.value


for s in SHAPES.values():
    run(s, 3)
for label, prio, bg in (("base", 0, 0), ("bg3", 0, 3), ("bg2", 0, 2), ("bg", 0, 1), ("base", 0, 0), ("bg3", 0, 3), ("bg2", 0, 2), ("bg", 0, 1)) if "bg" in sys.argv[1:] else \
        (("base", 0, 0), ("c1", 1, 0), ("c3", 3, 0), ("p1", 4, 0), ("p2", 8, 0), ("p3", 12, 0), ("c1p2", 9, 0), ("base", 0, 0)):
    os.environ["SIVO_X6_PRIO"] = str(prio)
    os.environ["SIVO_X6_BGLOBAL"] = str(bg)
    print(f"{label:5s}", " ".join(f"{n} {run(s):.4f}" for n, s in SHAPES.items()), flush=True)
