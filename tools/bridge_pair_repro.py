#!/usr/bin/env python3
"""The two-kernel reproducer of DESIGN 3.3 (diagnostic builds, GPU box): no network — `lanes` streams each run one bridged F(4x4) layer
(f16x3 GEMM, then wino4_bridge_kernel; 4 samples, 512 -> 512 channels, 44 x 128: conv4_x of SegNet-Standard) over and over on random
data, every GEMM + bridge run twice and compared word for word (sivo_debug_bridge_pair, SIVO_W4_VERIFY).  Each variant in its own
process: the bridge in its packed-FP32 form (libsivo_hip_diag_pkbridge.so) or as shipped (libsivo_hip_diag.so: no packed-FP32 instruction in any kernel); the GEMM with its exact
LDS (bridge workgroups of the other lane share CUs with it) or claiming 160 KB; two lanes or one.
    python tools/bridge_pair_repro.py"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUNDS = int(os.environ.get("REPRO_ROUNDS", "600"))

VARIANTS = [
    ("packed bridge, GEMM with its exact LDS, 2 lanes", "libsivo_hip_diag_pkbridge.so", {"SIVO_H3_LDS_ALL": "0"}, 2),
    ("bridge as shipped, GEMM with its exact LDS, 2 lanes", "libsivo_hip_diag.so", {"SIVO_H3_LDS_ALL": "0"}, 2),
    ("packed bridge, GEMM claiming 160 KB, 2 lanes", "libsivo_hip_diag_pkbridge.so", {}, 2),
    ("packed bridge, GEMM with its exact LDS, 1 lane", "libsivo_hip_diag_pkbridge.so", {"SIVO_H3_LDS_ALL": "0"}, 1),
    ("packed bridge, GEMM with its exact LDS, 3 lanes", "libsivo_hip_diag_pkbridge.so", {"SIVO_H3_LDS_ALL": "0"}, 3),
]


def body(name):
    import torch  # noqa: F401  (one HIP runtime per process)
    from sivo_amd import _lib
    _, lib, _, lanes = next(v for v in VARIANTS if v[0] == name)
    _lib.DIAG_PATH = os.path.join(os.path.dirname(_lib.DIAG_PATH), lib)
    with _lib.use("diag") as L:
        L.sivo_debug_bridge_pair.argtypes = [C.c_int] * 6 + [C.c_void_p]
        L.sivo_debug_words.argtypes = [C.c_void_p, C.c_int]
        out = (C.c_uint32 * 2)()
        t0 = time.perf_counter()
        rc = L.sivo_debug_bridge_pair(lanes, 4, 512, 44, 128, ROUNDS, out)
        assert rc == 0, L.sivo_last_error()
        w = (C.c_uint32 * 64)()
        L.sivo_debug_words(w, 1)
        print(f"[{name}] {w[6]} layer runs compared (each GEMM + bridge run twice): M words that differ {w[4]}, V' words that differ {w[5]}"
              + (f" ({w[20]} recorded)" if w[20] else "") + f"; overflow flag {out[1]}  [{time.perf_counter() - t0:.1f} s]", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        body(sys.argv[2])
    else:
        sel = os.environ.get("PROBE_ONLY")
        for name, _, env, _ in VARIANTS:
            if sel and sel not in name:
                continue
            e = dict(os.environ); e.update(env); e["SIVO_W4_VERIFY"] = "1"
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, timeout=200)
