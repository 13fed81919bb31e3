#!/usr/bin/env python3
"""Diagnostic build (libsivo_hip_diag_pkbridge.so: the bridge WITH packed-FP32 instructions, the form that fails — DESIGN 3.3), GPU box: SegNet-Standard T = 12 at 352 x 1024, three lanes against one lane (must be bit
identical) under the environment of the call, plus the report words of the diagnostic hooks (bridge border check, GEMM canary).
    python tools/coresident_probe.py            -> runs every variant below in its own process (the switches are read once)
    python tools/coresident_probe.py --one NAME -> the body, under the caller's environment"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("claim-160K (as shipped)", {"SIVO_BRIDGE_CHECK": "1", "SIVO_H3_CANARY": "1"}),
    ("exact LDS", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_CHECK": "1"}),
    ("exact LDS again", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_CHECK": "1"}),
    ("exact LDS, every bridged layer's GEMM and bridge run twice and compared", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1"}),
    ("claim-160K, every bridged layer's GEMM and bridge run twice and compared", {"SIVO_W4_VERIFY": "1"}),
    ("exact LDS, coherent: the bridge reads M past the CU's L1 (agent-scope loads)", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_M_COHERENT": "1", "SIVO_W4_VERIFY": "1"}),
    ("exact LDS, bridge claims 160K", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_LDS_ALL": "1", "SIVO_BRIDGE_CHECK": "1"}),
    ("exact LDS, packed chain off", {"SIVO_H3_LDS_ALL": "0", "SIVO_D3_PK": "0", "SIVO_BRIDGE_CHECK": "1"}),
    ("exact LDS, x6 GEMM", {"SIVO_H3_LDS_ALL": "0", "SIVO_GEMM": "x6"}),
    ("one lane, claim-160K GEMM; the bridge's second run beside an IDLE occupant holding 128K of every CU", {"DBG_LANES_A": "1", "SIVO_W4_VERIFY": "1", "SIVO_W4_VERIFY_OCC": "0,131072"}),
    ("one lane, claim-160K GEMM; the bridge's second run beside an occupant with ds traffic", {"DBG_LANES_A": "1", "SIVO_W4_VERIFY": "1", "SIVO_W4_VERIFY_OCC": "1,131072"}),
    ("one lane, claim-160K GEMM; the bridge's second run beside an occupant with LDS-DMA traffic", {"DBG_LANES_A": "1", "SIVO_W4_VERIFY": "1", "SIVO_W4_VERIFY_OCC": "2,131072"}),
    ("occupant idle 128K beside a ONE-lane handle (bridge workgroups at LDS bases >= 128K, nobody else using LDS)", {"DBG_LANES_A": "1", "SIVO_H3_LDS_ALL": "0", "PROBE_OCCUPY": "131072,0"}),
    ("occupant with ds traffic 128K beside a one-lane handle", {"DBG_LANES_A": "1", "SIVO_H3_LDS_ALL": "0", "PROBE_OCCUPY": "131072,1"}),
    ("occupant with LDS-DMA traffic 128K beside a one-lane handle", {"DBG_LANES_A": "1", "SIVO_H3_LDS_ALL": "0", "PROBE_OCCUPY": "131072,2"}),
    ("occupant with LDS-DMA traffic 112K beside a one-lane handle", {"DBG_LANES_A": "1", "SIVO_H3_LDS_ALL": "0", "PROBE_OCCUPY": "114688,2"}),
    ("HZ exact LDS, baseline (does the round-3 corruption still reproduce on this build?)", {"SIVO_H3_LDS_ALL": "0"}),
    ("HZ exact LDS, baseline again", {"SIVO_H3_LDS_ALL": "0"}),
    ("HZ exact LDS, a second barrier between the plane's writes and its reads", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "1"}),
    ("HZ exact LDS, s_sleep behind the barrier", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "2"}),
    ("HZ exact LDS, the window's 8-byte reads as two 4-byte reads", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "4"}),
    ("HZ exact LDS, lgkmcnt(0) + s_sleep in front of the barrier", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "8"}),
    ("HZ2 exact LDS, the bridge's plane 16 KB into its allocation", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "16"}),
    ("HZ2 exact LDS, 256-tile GEMM items wherever the layer has 256 couts (160 KB: only the 112 KB 256 x 128 GEMM can share a CU)", {"SIVO_H3_LDS_ALL": "0", "SIVO_H3_TILE": "0"}),
    ("HZ2 exact LDS, 128-tile GEMM items everywhere (128 KB)", {"SIVO_H3_LDS_ALL": "0", "SIVO_H3_TILE": "1"}),
    ("HZ2 exact LDS, one lane (no other lane's bridge beside a GEMM; its own stream is ordered)", {"SIVO_H3_LDS_ALL": "0", "DBG_LANES_A": "1"}),
    ("HZ3 exact LDS, GEMM + bridge run twice and compared (which V' element differs)", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1"}),
    ("HZ3 exact LDS, run twice and compared, the bridge reads its window bottom-up", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1", "SIVO_BRIDGE_HAZARD": "32"}),
    ("HZ4 exact LDS, the plane written with one ds_write_b32 per word instead of ds_write2_b32 pairs", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "64"}),
    ("HZ4 exact LDS, baseline once more", {"SIVO_H3_LDS_ALL": "0"}),
    ("HZ5 exact LDS, every bridge workgroup recomputes its plane from M once more at its END", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "128"}),
    ("HZ5 claim-160K, the same late recomputation (control)", {"SIVO_BRIDGE_HAZARD": "128"}),
    ("HZ8 exact LDS, the bridge as shipped (no packed-FP32 instructions)", {"SIVO_H3_LDS_ALL": "0", "PROBE_DIAG_LIB": "libsivo_hip_diag.so"}),
    ("HZ8 exact LDS, the bridge as shipped, GEMM + bridge run twice and compared", {"SIVO_H3_LDS_ALL": "0", "SIVO_W4_VERIFY": "1", "PROBE_DIAG_LIB": "libsivo_hip_diag.so"}),
    ("HZ8 exact LDS, the bridge with packed-FP32 instructions (the reproducer)", {"SIVO_H3_LDS_ALL": "0"}),
    ("HZ7 exact LDS, every thread reads its window once more at the END of the workgroup and compares hashes", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "1024"}),
    ("HZ7 claim-160K, the same (control)", {"SIVO_BRIDGE_HAZARD": "1024"}),
    ("HZ6 exact LDS, a 50 us do-nothing kernel between every GEMM and its bridge (same stream)", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "256"}),
    ("HZ6 exact LDS, the same kernel in front of every GEMM instead (control)", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "512"}),
    ("HZ exact LDS, second barrier + sleep + split reads", {"SIVO_H3_LDS_ALL": "0", "SIVO_BRIDGE_HAZARD": "7"}),
    ("one lane both, LDS poisoned in front of every kernel", {"SIVO_POISON_LDS": "1", "DBG_LANES_A": "1", "SIVO_H3_LDS_ALL": "0"}),
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
        L.sivo_debug_occupy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]

        def words(reset=1):
            w = (C.c_uint32 * 64)()
            L.sivo_debug_words(w, reset)
            return list(w[:64])

        def make(lanes):
            os.environ["SIVO_LANES"] = str(lanes)
            return BayesianSegNet(prototxt=text, weights=flat, T=T)
        poison = os.environ.pop("SIVO_POISON_LDS", None)       # the reference handle is never poisoned
        b = make(1)
        _, lb0, _ = b.forward(img, 99, want_logits=True)
        torch.cuda.synchronize()
        if poison:
            os.environ["SIVO_POISON_LDS"] = poison
        a = make(int(os.environ.get("DBG_LANES_A", "3")))
        words()
        bad = 0
        occ = os.environ.get("PROBE_OCCUPY")
        for seed in ((99, 5, 7, 11, 13, 17, 19, 23) if os.environ.get('PROBE_SEEDS8') else (99, 5, 7, 11)):
            if occ:      # ~40 ms of occupant launches (150 us each) on their own stream, then the two frames beside them
                L.sivo_debug_occupy(int(occ.split(",")[0]), int(occ.split(",")[1]), 150, 400)
                import time
                time.sleep(0.002)
            _, la, _ = a.forward(img, seed, want_logits=True)
            _, la2, _ = a.forward(img, seed, want_logits=True)
            if occ:
                torch.cuda.synchronize()
                t_w = time.perf_counter()
                L.sivo_debug_occupy_wait()
                print(f"  (occupant launches outlasted the two frames by {1e3 * (time.perf_counter() - t_w):.1f} ms)")
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
        if w[13]:
            print(f"  plane words whose recomputation from M at the END of the workgroup differs from what the plane held: {w[13]}; first: held {w[14]:08x} recomputed {w[15]:08x} "
                  f"sample {w[40] >> 16} cout {w[40] & 0xffff} tile {w[41] >> 8} output row {(w[41] >> 2) & 3} column {w[41] & 3}")
        elif os.environ.get("SIVO_BRIDGE_HAZARD") and int(os.environ["SIVO_BRIDGE_HAZARD"]) & 128:
            print("  plane words whose recomputation from M at the END of the workgroup differs: 0")
        if os.environ.get("SIVO_BRIDGE_HAZARD") and int(os.environ["SIVO_BRIDGE_HAZARD"]) & 1024:
            print(f"  threads whose 6 x 6 window read at the END of the workgroup differs from their first read: {w[58]} (of them with another word (0, 5): {w[59]}; "
                  f"first: sample {w[60] >> 16} cout {w[60] & 0xffff} thread {w[61]})")
        print(f"[{name}] frames that differ: {bad} of {8 if os.environ.get('PROBE_SEEDS8') else 4}; bridge border cells dirty {w[0]} in {w[1]} workgroups checked; GEMM canary words changed {w[2]} in {w[3]} workgroups; "
              f"re-run compare over {w[6]} layers: M words differing {w[4]}, V' words differing {w[5]}; "
              f"plane words changed after they were written {w[7]} (first: index {w[8]} of a {w[12] >> 16} x {w[12] & 0xffff} plane, wrote {w[9]:08x} found {w[10]:08x}, n {w[11] >> 16} cout {w[11] & 0xffff})", flush=True)


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
