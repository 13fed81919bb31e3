#!/usr/bin/env python3
"""Where the time of the f16x3 GEMM goes: the production kernel and its compile-time ablations (libsivo_hip_diag.so, `make -C
sivo_amd/csrc diag`; SIVO_H3_ABL bits: 1 no V' loads, 2 no U' DMA, 4 no M stores, 8 no MFMAs, 16 V' by LDS-DMA from a fragment-order slab — timing only) on the GEMM shapes of
SegNet-Standard T = 12.  GPU box only.  Usage: python tools/h3_probe.py [iters]
H3_PROBE_SHAPE / H3_PROBE_ONLY (comma-separated) / H3_PROBE_SKIP select shapes / variants by substring; H3_PROBE_ZEROS=1 runs on all-zero operands
(tools/power_probe.py wraps this script to read board power and shader clock per variant)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "sivo_amd", "libsivo_hip_diag.so"))
lib.sivo_debug_h3_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
lib.sivo_last_error.restype = C.c_char_p
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [("conv4_2  512->512 44x128", 512, 512, 4224), ("conv5_2  512->512 22x64", 512, 512, 1152),
          ("conv3_3D 256->256 88x256", 256, 256, 16896), ("conv4_1D 512->256 44x128", 512, 256, 4224)]
VARIANTS = [("as built", {}), ("FORM 2 (fragment reads left to the compiler)", {"SIVO_H3_FORM": "2"}), ("FORM 1 (4-byte V' loads)", {"SIVO_H3_FORM": "1"}), ("phased form (round 5)", {"SIVO_H3_FORM": "0"}), ("no V loads", {"SIVO_H3_ABL": "1"}), ("no U DMA", {"SIVO_H3_ABL": "2"}),
            ("no loads at all", {"SIVO_H3_ABL": "3"}), ("no M stores", {"SIVO_H3_ABL": "4"}), ("MFMA + LDS only", {"SIVO_H3_ABL": "7"}),
            ("no MFMA", {"SIVO_H3_ABL": "8"}), ("no MFMA, no stores", {"SIVO_H3_ABL": "12"}),
            ("V' by LDS-DMA (timing only)", {"SIVO_H3_ABL": "16"}), ("V' by LDS-DMA, no MFMA", {"SIVO_H3_ABL": "24"}),
            ("start skew (4 phases)", {"SIVO_H3_ABL": "32"}), ("start skew, no M stores", {"SIVO_H3_ABL": "36"}),
            ("MFMA + LDS, no barrier", {"SIVO_H3_ABL": "519"}), ("MFMA + barrier, no fragment reads", {"SIVO_H3_ABL": "1031"}), ("MFMA alone", {"SIVO_H3_ABL": "1543"}), ("whole-line 16-byte M stores, counted wait (timing only)", {"SIVO_H3_ABL": "4096"}), ("M stores nt", {"SIVO_H3_ABL": "64"}), ("V' loads nt", {"SIVO_H3_ABL": "128"}), ("M stores + V' loads nt", {"SIVO_H3_ABL": "192"})]
rng = np.random.default_rng(0)
for name, Cc, Kp, P in SHAPES:
    if os.environ.get("H3_PROBE_SHAPE") and os.environ["H3_PROBE_SHAPE"] not in name:
        continue
    Pp = (P + 127) // 128 * 128
    V = rng.standard_normal((36, Cc, Pp), dtype=np.float32)
    U = (rng.standard_normal((36, Cc, Kp), dtype=np.float32) * 0.05).astype(np.float32)
    if os.environ.get("H3_PROBE_ZEROS"):       # all-zero operands: the same instructions with no toggling in the matrix cores (power probe)
        V[:] = 0; U[:] = 0
    M = np.empty((36, Kp, Pp), np.float32)
    flop = 2.0 * 36 * Cc * Kp * P * 3          # executed fp16 products
    for vname, env in VARIANTS:
        if os.environ.get("H3_PROBE_ONLY") and not any(k in vname for k in os.environ["H3_PROBE_ONLY"].split(",")):
            continue
        if os.environ.get("H3_PROBE_SKIP") and os.environ["H3_PROBE_SKIP"] in vname:
            continue
        if "SIVO_H3_TILE" in env:
            continue                              # (static in the launcher: run the script again with SIVO_H3_TILE=2 for that column)
        for k in ("SIVO_H3_ABL", "SIVO_H3_FORM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = C.c_double(0)
        rc = lib.sivo_debug_h3_gemm(Cc, Kp, P, V.ctypes.data, U.ctypes.data, C.c_float(16.0), M.ctypes.data, iters, C.byref(ms))
        if rc:
            print(name, vname, "error", lib.sivo_last_error().decode()); continue
        print(f"{name:26s} {vname:44s} {ms.value:8.4f} ms   {flop / ms.value / 1e9:8.1f} TFLOP/s executed", flush=True)
