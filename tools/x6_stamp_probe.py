#!/usr/bin/env python3
"""Where the waves of wino4_gemm_x6p_kernel spend their cycles: per role (consumers = MFMA waves, producers = DMA + split
waves) the share of shader-clock cycles between hand-overs (work) and inside them (waiting for the slowest wave), from the TS
form of the kernel (SIVO_X6_STAMPS=1, sivo_debug_x6_stamps).  GPU box only; prepared at the end of round 2, not yet run."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIVO_X6_STAMPS"] = "1"
from sivo_amd._lib import lib, check

SHAPES = {"conv4_2": (12, 512, 512, 44, 128), "conv5_2": (12, 512, 512, 22, 64), "conv3_2_D": (12, 256, 256, 88, 256),
          "conv2_2_D": (12, 128, 128, 176, 512)}
out = (C.c_uint64 * 8)()
for name, (N, ci, co, H, W) in SHAPES.items():
    ms = C.c_double()
    check(lib().sivo_debug_conv(N, ci, co, H, W, 3, 2, 512 | 2048, C.byref(ms)))        # warm-up (also counted: reset below)
    check(lib().sivo_debug_x6_stamps(out, 1))
    check(lib().sivo_debug_conv(N, ci, co, H, W, 3, 10, 512 | 2048, C.byref(ms)))
    check(lib().sivo_debug_x6_stamps(out, 1))
    cw, cq, pw, pq, n, pi, pv, ps = (int(v) for v in out)
    per = max(n // 8, 1)          # hand-overs per wave pair: 8 waves count each one
    print(f"{name:10s} layer {ms.value:.4f} ms | consumers work {cw / (cw + cq):.3f} wait {cq / (cw + cq):.3f} | "
          f"producers work {pw / (pw + pq):.3f} wait {pq / (pw + pq):.3f} | cycles per hand-over: consumer {(cw + cq) / 4 / per:.0f} "
          f"(work {cw / 4 / per:.0f}), producer {(pw + pq) / 4 / per:.0f} (work {pw / 4 / per:.0f} = DMA issue {pi / 4 / per:.0f} + vmcnt wait "
          f"{pv / 4 / per:.0f} + split {ps / 4 / per:.0f} + rest)", flush=True)
