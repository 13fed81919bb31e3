#!/usr/bin/env python3
"""Diagnostic build, GPU box: the LDS access pattern of wino4_bridge_kernel on synthetic SELF-CHECKING data (lds_victim_kernel,
diag_kernels.hip) beside (a) nothing, (b) the synthetic occupant (idle / ds traffic / LDS-DMA traffic) and (c) the REAL f16x3
GEMM of the F(4x4) path with its exact LDS size (the pair that corrupted frames) or claiming the whole LDS (the
mitigation).  The victim reports from the device how many of its workgroups ran BESIDE another LDS user on their CU (their LDS
allocation does not start at 0), so "it did not fail" can be told from "it never shared a CU".
    python tools/coresident_repro.py            -> every variant in its own process (the switches are read once per process)
    python tools/coresident_repro.py --one NAME"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEMM_ITERS = 3000
SECONDS = 2.5                          # of victim launches per plane geometry
GEOMS = [(44, 128), (22, 64)]          # the bridged layers' planes that fit beside a 112-128 KB GEMM workgroup (24 KB / 7 KB)

VARIANTS = [
    ("alone", {}, None),
    ("beside an idle occupant holding 128K", {}, (131072, 0)),
    ("beside an occupant with ds traffic, 128K", {}, (131072, 1)),
    ("beside an occupant with LDS-DMA traffic, 128K", {}, (131072, 2)),
    ("beside an occupant with LDS-DMA traffic, 112K", {}, (114688, 2)),
    ("beside a GEMM-like occupant (LDS-DMA + ds_read_b128 / ds_write_b128 + barriers), 128K", {}, (131072, 3)),
    ("beside a GEMM-like occupant with MFMAs, 128K", {}, (131072, 4)),
    ("beside the f16x3 GEMM, exact LDS", {"SIVO_H3_LDS_ALL": "0"}, "gemm"),
    ("beside the f16x3 GEMM, exact LDS, again", {"SIVO_H3_LDS_ALL": "0"}, "gemm"),
    ("beside the f16x3 GEMM claiming 160K (as shipped)", {}, "gemm"),
    # PK: the victim also runs the bridge's arithmetic (packed-FP32 VALU instructions) twice on every window and compares
    ("PK alone", {"REPRO_PK": "1"}, None),
    ("PK beside a GEMM-like occupant with MFMAs, 128K", {"REPRO_PK": "1"}, (131072, 4)),
    ("PK beside a GEMM-like occupant without MFMAs, 128K", {"REPRO_PK": "1"}, (131072, 3)),
    ("PK beside the f16x3 GEMM, exact LDS", {"REPRO_PK": "1", "SIVO_H3_LDS_ALL": "0"}, "gemm"),
    ("PK beside the f16x3 GEMM claiming 160K", {"REPRO_PK": "1"}, "gemm"),
]


def body(name):
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process)
    from sivo_amd import _lib
    from sivo_amd.segnet import h3_gemm
    spec = dict((n, o) for n, _, o in VARIANTS)[name]
    pk = bool(os.environ.get("REPRO_PK"))
    with _lib.use("diag") as L:
        L.sivo_debug_lds_victim.argtypes = [C.c_int] * 6 + [C.c_void_p]
        L.sivo_debug_occupy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        stop = threading.Event()
        gemm_ms = []

        def gemm_loop(shape):
            Cc, Kp, P = shape
            rng = np.random.default_rng(1)
            V = np.tile(rng.standard_normal((36, Cc, 128), dtype=np.float32), (1, 1, (P + 127) // 128))
            U = rng.standard_normal((36, Cc, Kp), dtype=np.float32)
            while not stop.is_set():
                gemm_ms.append(h3_gemm(V, U, P, vscale=16.0, iters=GEMM_ITERS)[1])

        th = None
        if spec == "gemm":
            # conv4_x of SegNet-Standard, one lane's four samples: 512 -> 512 channels, 1408 tiles (the 128 x 256 tile: 128 KB of LDS),
            th = threading.Thread(target=gemm_loop, args=((512, 512, 1408),))
            th.start()
            time.sleep(1.5)            # (operand upload + first launches)
        for H, W in GEOMS:
            tot = [0] * 64
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < SECONDS:
                if isinstance(spec, tuple):
                    L.sivo_debug_occupy(spec[0], spec[1], 3000, 4)       # 4 x 3 ms on the occupant's stream
                    time.sleep(0.0005)
                rep = (C.c_uint32 * 64)()
                rc = L.sivo_debug_lds_victim(2048, H, W, 24, 6, -6 if pk else 6, rep)
                assert rc == 0, L.sivo_last_error()
                if isinstance(spec, tuple):
                    L.sivo_debug_occupy_wait()
                first = tot[2] == 0 and rep[2] != 0
                for i in range(64):
                    if i in (0, 1, 2, 3, 50, 51) or 12 <= i < 48:
                        tot[i] += rep[i]
                    elif first or (i in (48, 49) and rep[i]) or (52 <= i <= 56 and rep[51] and not tot[53] and not tot[54]):
                        tot[i] = rep[i]
            words = {k: tot[12 + k] for k in range(36) if tot[12 + k]}
            print(f"[{name}] plane {H}x{W}: workgroups {tot[0]}, of which above >= 112 KB of other LDS {tot[1]} ({tot[3]} rounds; base exactly 112 / 128 KB: {tot[50]}); window words that differed {tot[2]}"
                  + (f"; first: round {tot[4]} window word (row {tot[5] // 6}, col {tot[5] % 6}) expected {tot[6]:08x} read {tot[7]:08x} re-read {tot[11]:08x} "
                     f"LDS_ALLOC {tot[8]:08x} workgroup {tot[9]} tile {tot[10]}; by window word {words}" if tot[2] else "")
                  + (f"; PK: rows of six packed words whose two computations differ {tot[51]}" + (f" (first: row {tot[52]} hashes {tot[53]:08x} / {tot[54]:08x} thread {tot[55]} LDS_ALLOC {tot[56]:08x})" if tot[51] else "") if pk else "")
                  + f"; LDS_ALLOC of a co-resident / a lone workgroup {tot[48]:08x} / {tot[49]:08x}  [{time.perf_counter() - t0:.1f} s]", flush=True)
        stop.set()
        if th:
            th.join()
            print(f"  (GEMM beside: {len(gemm_ms)} x {GEMM_ITERS} launches, mean {np.mean(gemm_ms):.3f} ms per launch)")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        body(sys.argv[2])
    else:
        sel = os.environ.get("PROBE_ONLY")
        for name, env, _ in VARIANTS:
            if sel and sel not in name:
                continue
            e = dict(os.environ); e.update(env)
            print(f"== {name}: {env}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, timeout=300)
