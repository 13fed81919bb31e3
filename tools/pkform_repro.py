#!/usr/bin/env python3
"""Diagnostic build, GPU box: ONE packed-FP32 instruction form in a loop on known operands (pkform_victim_kernel, diag_kernels.hip)
beside the real f16x3 GEMM at its exact LDS (its workgroups share CUs with the victim's) and alone.  Forms: 0 = the in-place
cross-half subtraction that only wino4_bridge_kernel's packed build contains (v_pk_add_f32 p, p, p op_sel_hi:[0,1] neg_lo:[0,1]
neg_hi:[0,1]), 1 = the same into another register pair, 2 = a plain in-place packed add.   python tools/pkform_repro.py"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECONDS = float(os.environ.get("PKFORM_SECONDS", "1.0"))


def body(name):
    import numpy as np
    import torch  # noqa: F401
    from sivo_amd import _lib
    from sivo_amd.segnet import h3_gemm
    with _lib.use("diag") as L:
        L.sivo_debug_pkform.argtypes = [C.c_int] * 4 + [C.c_void_p]
        stop = threading.Event()
        n_gemm = [0]

        def gemm_loop():
            rng = np.random.default_rng(1)
            V = np.tile(rng.standard_normal((36, 512, 128), dtype=np.float32), (1, 1, 11))
            U = rng.standard_normal((36, 512, 512), dtype=np.float32)
            while not stop.is_set():
                h3_gemm(V, U, 1408, vscale=16.0, iters=3000)
                n_gemm[0] += 3000
        th = None
        if name == "beside":
            th = threading.Thread(target=gemm_loop)
            th.start()
            time.sleep(1.5)
        for form in (0, 1, 2):
            tot = [0] * 16
            t0 = time.perf_counter()
            calls = 0
            while time.perf_counter() - t0 < SECONDS:
                rep = (C.c_uint32 * 16)()
                rc = L.sivo_debug_pkform(2048, form, 20000, 4, rep)
                assert rc == 0, L.sivo_last_error()
                calls += 1
                for i in (0, 2, 3, 4):
                    tot[i] += rep[i]
                if rep[2] and not tot[5]:
                    for i in range(5, 10):
                        tot[i] = rep[i]
            n_ins = tot[0] * 384 * 20000
            first = (f"; first wrong high half: a {tot[5]:08x} b {tot[6]:08x} found {tot[7]:08x} lane {tot[8]}" if tot[2] else "")
            print(f"[{name} the GEMM at its exact LDS, form {form}] {n_ins:.3e} instructions checked in {tot[0]} workgroups: wrong high halves {tot[2]} (of them equal to -b: {tot[4]}), "
                  f"wrong low halves {tot[3]}{first}  [{time.perf_counter() - t0:.1f} s]", flush=True)
        stop.set()
        if th:
            th.join()
            print(f"  (GEMM launches beside: {n_gemm[0]})")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        body(sys.argv[2])
    else:
        for name in ("beside", "without"):
            e = dict(os.environ); e["SIVO_H3_LDS_ALL"] = "0"
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, timeout=120)
