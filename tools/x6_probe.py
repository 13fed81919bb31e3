#!/usr/bin/env python3
"""bf16x6 vs fp32 batched GEMM of the three-kernel F(4x4,3x3) path on the SegNet-Standard GEMM shapes
(sivo_debug_conv: input transform + GEMM + output transform, random data), and ablations of the bf16x6 kernel
(variant bits 12..16: 1 no V loads, 2 no split/LDS stores, 4 no U DMA, 8 no MFMAs, 16 no epilogue stores).  GPU box only."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sivo_amd._lib import lib, check

SHAPES = {"conv4_2 512->512 44x128": (12, 512, 512, 44, 128), "conv5_2 512->512 22x64": (12, 512, 512, 22, 64),
          "conv3_2_D 256->256 88x256": (12, 256, 256, 88, 256), "conv2_2_D 128->128 176x512": (12, 128, 128, 176, 512),
          "conv4_1 256->512 44x128": (12, 256, 512, 44, 128), "conv3_1_D 256->128 88x256": (12, 256, 128, 88, 256),
          "conv4_2 N=4": (4, 512, 512, 44, 128), "conv5_2 N=4": (4, 512, 512, 22, 64)}


def run(shape, variant, iters=10):
    N, ci, co, H, W = shape
    ms = C.c_double()
    check(lib().sivo_debug_conv(N, ci, co, H, W, 3, iters, variant, C.byref(ms)))
    return ms.value


if len(sys.argv) > 1 and sys.argv[1] == "ablate":
    for name in sys.argv[2:] or ["conv4_2 512->512 44x128", "conv3_2_D 256->256 88x256"]:
        base = 512 | 2048
        print(name, "f32 %.3f" % run(SHAPES[name], 512), " ".join(f"abl{a}={run(SHAPES[name], base | (a << 12)):.3f}" for a in ((0, 1, 2, 3, 4, 7, 8, 16, 23, 32) if os.environ.get("SIVO_X6") != "flat" else (0, 1, 2, 4, 7, 8, 16))), flush=True)
    sys.exit(0)
for name, shape in SHAPES.items():
    N, ci, co, H, W = shape
    row = []
    for v, label in ((512, "f32"), (512 | 2048, "bf16x6")):
        ms = run(shape, v)
        row.append(f"{label} {ms:.3f} ms ({2.0 * 9 * ci * co * H * W * N / ms / 1e9:.0f} TF alg)")
    print(f"{name:30s}", " | ".join(row), flush=True)
