#!/usr/bin/env python3
"""Attribute time inside conv_mfma_kernel: runs sivo_debug_conv on the SegNet-Standard layer shapes
with parts of the kernel switched off (variant bits).  GPU box only."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sivo_amd._lib import dbg, check

SHAPES = {"conv4_2": (12, 512, 512, 44, 128), "conv5_2": (12, 512, 512, 22, 64), "conv3_2_D": (12, 256, 256, 88, 256),
          "conv2_2_D": (12, 128, 128, 176, 512), "conv1_2_D": (12, 64, 64, 352, 1024), "conv3_2": (1, 256, 256, 88, 256),
          "conv4_1": (12, 256, 512, 44, 128), "conv4_1_D": (12, 512, 256, 44, 128),
          "conv3_1_D": (12, 256, 128, 88, 256), "conv2_1_D": (12, 128, 64, 176, 512), "conv3_1": (1, 128, 256, 88, 256)}
VARIANTS = {64: "wino F(2x2)", 512: "wino4 F(4x4)", 1024 + 4096 + 8192: "fused F(4x4), one chunk ahead", 1024 + 4096 + 16384: "fused F(4x4), in flight, one workgroup per tile", 1024 + 4096: "fused F(4x4), in flight, persistent"}

def run(name, variant, iters=10):
    N, ci, co, H, W = SHAPES[name]
    ms = C.c_double()
    check(dbg().sivo_debug_conv(N, ci, co, H, W, 3, iters, variant, C.byref(ms)))
    fl = 2.0 * 9 * ci * co * H * W * N
    return ms.value, fl / ms.value / 1e9

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "w4fp":
    # ablations of the in-flight fused F(4x4) kernel (conv_wino4f_p_kernel PABL bits << 16); every variant keeps the output stage
    for n in sys.argv[2:] or ["conv1_2_D", "conv2_1_D"]:
        row = []
        for abl, label in ((0, "full"), (1, "-staging"), (2, "-patchreads/transform"), (4, "-Breads"), (6, "-all LDS reads"), (8, "-barrier"),
                           (9, "-staging-barrier"), (7, "-staging-LDS reads"), (15, "MFMA + output only"), (31, "MFMA + output, no prologue traffic")):
            ms, _ = run(n, 1024 | 4096 | (abl << 16))
            row.append(f"{label}={ms:.3f}")
        print(n, " ".join(row), flush=True)
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "w4f":
    # ablations of the fused F(4x4) kernel (conv_wino4f.hip ABL bits << 16)
    for n in sys.argv[2:] or ["conv1_2_D", "conv2_1_D"]:
        row = []
        # (variants that drop the output stage are not listed: without it the compiler removes most MFMAs as dead code)
        for abl, label in ((0, "full"), (1, "-patch"), (2, "-wdma"), (3, "-patch-wdma"), (7, "-staging-barrier"), (8, "-transform"), (32, "-MFMA"),
                           (64, "-stores"), (128, "-exchwrites"), (192, "-stores-exchwrites"), (256, "-outbarriers"), (448, "-stores-exchwrites-outbarriers")):
            ms, _ = run(n, 1024 | 4096 | 8192 | (abl << 16))      # 8192: the one-chunk-ahead kernel the ablation switches live in
            row.append(f"{label}={ms:.3f}")
        print(n, " ".join(row), flush=True)
    sys.exit(0)
if __name__ == "__main__":
    names = sys.argv[1:] or list(SHAPES)
    for n in names:
        row = []
        for v, label in VARIANTS.items():
            ms, tf = run(n, v)
            row.append(f"{label}={ms:.3f}ms({tf:.0f}TF)")
        print(n, " ".join(row))
