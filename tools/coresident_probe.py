#!/usr/bin/env python3
"""The co-residency reproducer on the whole network (docs/HW_NOTE_packed_fp32.md).  GPU box, diagnostic builds: SegNet-Standard
T = 12 at 352 x 1024, three lanes against one lane (must be bit identical) with the f16x3 GEMM at its EXACT LDS size, so that another
lane's bridge workgroups share CUs with it.  libsivo_hip_diag.so = the library as shipped (no packed-FP32 instruction in any kernel);
libsivo_hip_diag_pkbridge.so = the same with conv_wino4.hip compiled WITH packed-FP32 instructions, the form that fails.
    python tools/coresident_probe.py            -> runs every variant below in its own process (the switches are read once)
    python tools/coresident_probe.py --one NAME -> the body, under the caller's environment"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("HZ8 exact LDS, the bridge as shipped (no packed-FP32 instructions)", {"SIVO_H3_LDS_ALL": "0", "PROBE_DIAG_LIB": "libsivo_hip_diag.so"}),
    ("HZ8 exact LDS, the bridge as shipped, GEMM + bridge run twice and compared", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1", "PROBE_DIAG_LIB": "libsivo_hip_diag.so"}),
    ("HZ8 exact LDS, the bridge with packed-FP32 instructions (the reproducer)", {"SIVO_H3_LDS_ALL": "0"}),
    ("HZ8 exact LDS, the packed bridge, GEMM + bridge run twice and compared (which V' words differ)", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1"}),
    ("HZ8 claim-160K, the packed bridge (the other mitigation alone)", {}),
    ("HZ8 exact LDS, the packed bridge, one lane (no other lane's bridge beside a GEMM)", {"SIVO_H3_LDS_ALL": "0", "DBG_LANES_A": "1"}),
]


def body(name):
    import numpy as np
    import torch
    from sivo_amd import _lib, netspec, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    from bench import make_inputs
    H, W, T = 352, 1024, 12
    text = netspec.standard_prototxt(T, H, W)
    layers = netspec.parse_layers(text)
    flat = wts.pack(layers, wts.synth_weights(layers, 42))
    img = torch.from_numpy(make_inputs(H, W)[0]).cuda()
    # the probe is about the fault: by default it loads the diagnostic build whose bridge still has its packed-FP32 instructions
    # (sivo_amd/csrc/Makefile diag_pkbridge); PROBE_DIAG_LIB=libsivo_hip_diag.so is the diagnostic build of the bridge as shipped
    _lib.DIAG_PATH = os.path.join(os.path.dirname(_lib.DIAG_PATH), os.environ.get("PROBE_DIAG_LIB", "libsivo_hip_diag_pkbridge.so"))
    with _lib.use("diag") as L:
        L.sivo_debug_words.argtypes = [C.c_void_p, C.c_int]

        def words(reset=1):
            w = (C.c_uint32 * 64)()
            L.sivo_debug_words(w, reset)
            return list(w[:64])

        def make(lanes):
            os.environ["SIVO_LANES"] = str(lanes)
            return BayesianSegNet(prototxt=text, weights=flat, T=T)
        b = make(1)
        _, lb0, _ = b.forward(img, 99, want_logits=True)
        torch.cuda.synchronize()
        a = make(int(os.environ.get("DBG_LANES_A", "3")))
        words()
        bad = 0
        for seed in ((99, 5, 7, 11, 13, 17, 19, 23) if os.environ.get('PROBE_SEEDS8') else (99, 5, 7, 11)):
            _, la, _ = a.forward(img, seed, want_logits=True)
            _, la2, _ = a.forward(img, seed, want_logits=True)
            _, lb, _ = b.forward(img, seed, want_logits=True)
            torch.cuda.synchronize()
            d = (la - lb).abs()
            nz = (d > 0) | torch.isnan(la)
            info = ""
            if nz.any():
                idx = nz.nonzero()
                info = (f" differing {int(nz.sum())} (NaN {int(torch.isnan(la).sum())}) samples {sorted(set(idx[:, 0].tolist()))} y {int(idx[:, 2].min())}-{int(idx[:, 2].max())} "
                        f"x {int(idx[:, 3].min())}-{int(idx[:, 3].max())} max {float(torch.nan_to_num(d).max()):.3e}")
                bad += 1
            print(f"  seed {seed}: equal to the one-lane handle {bool(torch.equal(la, lb))}; repeatable {bool(torch.equal(la, la2))}{info}", flush=True)
        w = words()
        if w[20]:
            print(f"  first differing V' words ({w[20]} recorded; geometry of a layer seen after the first difference: K {w[16]} Pp {w[17]} tiles per sample {w[18]} tiles per row {w[19]}):")
            for k in range(min(w[20], 12)):
                idx, xa, xb = w[21 + 3 * k], w[22 + 3 * k], w[23 + 3 * k]
                K, Pp, nt, tw = max(w[16], 1), max(w[17], 1), max(w[18], 1), max(w[19], 1)
                xi, co, pp = idx // (K * Pp), (idx // Pp) % K, idx % Pp
                print(f"    word {idx}: xi {xi} cout {co} sample {pp // nt} tile {pp % nt} (row {pp % nt // tw} col {pp % nt % tw}): first run {xa:08x} second run {xb:08x}")
        print(f"[{name}] frames that differ: {bad} of {8 if os.environ.get('PROBE_SEEDS8') else 4}; "
              f"re-run compare over {w[6]} layers: M words differing {w[4]}, V' words differing {w[5]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        body(sys.argv[2])
    else:
        sel = os.environ.get("PROBE_ONLY")
        for name, env in VARIANTS:
            if sel and sel not in name:
                continue
            e = dict(os.environ); e.update(env)
            print(f"== {name}: {env}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, timeout=600)
