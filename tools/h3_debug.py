#!/usr/bin/env python3
"""Debug aid (GPU box): (1) the f16x3 GEMM alone at the exact GEMM shapes of SegNet-Standard T = 12 (one lane and three
lanes) against fp64, twice (determinism); (2) the net at T = 12 with every F(4x4) activation materialised: f16x3 handle against
a bf16x6 handle, layer by layer, with the pattern of the largest differences."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sivo_amd import netspec, weights as wts          # noqa: E402
from sivo_amd.segnet import BayesianSegNet, h3_gemm    # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("gemm", "all"):
    rng = np.random.default_rng(1)
    for C, Kp, P in [(256, 512, 1408), (512, 512, 1408), (512, 512, 4224), (512, 256, 4224), (256, 256, 5632), (256, 128, 5632), (128, 128, 22528)]:
        Pp = (P + 127) // 128 * 128
        V = rng.standard_normal((36, C, Pp)).astype(np.float32); V[:, :, P:] = 0
        U = (rng.standard_normal((36, C, Kp)) * 0.05).astype(np.float32)
        M1, _ = h3_gemm(V, U, P)
        M2, _ = h3_gemm(V, U, P)
        same = np.array_equal(M1, M2)
        worst, where = 0.0, None
        for xi in range(0, 36, 5):
            ref = U[xi].astype(np.float64).T @ V[xi].astype(np.float64)
            bound = np.abs(U[xi]).astype(np.float64).T @ np.abs(V[xi]).astype(np.float64)
            rel = np.abs(M1[xi][:, :P] - ref[:, :P]) / np.maximum(bound[:, :P], 1e-30)
            if rel.max() > worst:
                worst = float(rel.max()); k, p = np.unravel_index(rel.argmax(), rel.shape); where = (xi, int(k), int(p))
        bad = None
        if worst > 1e-6:
            xi = where[0]
            ref = U[xi].astype(np.float64).T @ V[xi].astype(np.float64)
            bound = np.abs(U[xi]).astype(np.float64).T @ np.abs(V[xi]).astype(np.float64)
            rel = np.abs(M1[xi][:, :P] - ref[:, :P]) / np.maximum(bound[:, :P], 1e-30)
            ks, ps = np.nonzero(rel > 1e-6)
            bad = dict(n=len(ks), k_range=(int(ks.min()), int(ks.max())), p_range=(int(ps.min()), int(ps.max())), p_blocks=sorted(set((ps // 32).tolist()))[:20], k_blocks=sorted(set((ks // 32).tolist()))[:20])
        print(f"gemm C={C} Kp={Kp} P={P}: deterministic={same} worst rel {worst:.2e} at {where} {bad}", flush=True)

if what in ("net", "all"):
    H, W, T = 352, 1024, 12
    text = netspec.standard_prototxt(T, H, W)
    layers = netspec.parse_layers(text)
    flat = wts.pack(layers, wts.synth_weights(layers, 42))
    from bench import make_inputs
    img = torch.from_numpy(make_inputs(H, W)[0]).cuda()
    os.environ["SIVO_NO_FUSE_BRIDGE"] = "1"; os.environ["SIVO_NO_FUSE_POOL"] = "1"
    lanes = sys.argv[2] if len(sys.argv) > 2 else "3"
    os.environ["SIVO_LANES"] = lanes
    h3 = BayesianSegNet(prototxt=text, weights=flat, T=T)
    os.environ["SIVO_GEMM"] = "x6"
    x6 = BayesianSegNet(prototxt=text, weights=flat, T=T)
    del os.environ["SIVO_GEMM"]
    print("status", h3.gemm_status()[:2], x6.gemm_status()[:2], "lanes", lanes)
    for seed in (99,):
        h3.forward(img, seed); x6.forward(img, seed)
        torch.cuda.synchronize()
        for name in ["conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "conv5_3_D", "conv5_2_D", "conv5_1_D", "conv4_3_D", "conv4_2_D", "conv4_1_D",
                     "conv3_3_D", "conv3_2_D", "conv3_1_D", "conv2_2_D", "conv2_1_D", "conv1_2_D", "conv1_1_D"]:
            try:
                a, b = h3.blob(name), x6.blob(name)
            except Exception as e:
                print(name, "n/a", str(e)[:60]); continue
            d = np.abs(a - b)
            n, c, y, x = np.unravel_index(d.argmax(), d.shape)
            big = d > 1e-3 * max(1.0, float(np.abs(b).max()))
            info = ""
            if big.any():
                ns, cs, ys, xs = np.nonzero(big)
                info = f" BIG {big.sum()} samples {sorted(set(ns.tolist()))} ch {cs.min()}-{cs.max()} ({len(set(cs.tolist()))}) y {ys.min()}-{ys.max()} x {xs.min()}-{xs.max()} tiles(y//4,x//4) {sorted(set(zip((ys // 4).tolist(), (xs // 4).tolist())))[:12]}"
            print(f"{name:10s} max|d| {d.max():.3e} at n{n} c{c} y{y} x{x}; max|x6| {np.abs(b).max():.2f} finite {np.isfinite(a).all()}{info}", flush=True)
        print("status after", h3.gemm_status()[:2])

if what == "coresident":
    # the GEMM alone, while a second thread keeps small-LDS workgroups (wino4_bridge_kernel of a narrow bridged stack) on the CUs
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_segnet import _conv_stack_prototxt
    from oracle import prototxt as oproto
    T2, H2, W2, width = 3, 22, 64, 256
    t2 = _conv_stack_prototxt(T2, H2, W2, width)
    n2 = oproto.parse(t2)
    small = BayesianSegNet(prototxt=t2, weights=wts.pack(n2["layers"], wts.synth_weights(n2["layers"], 5)), T=T2)
    img2 = torch.randint(0, 256, (H2, W2, 3), dtype=torch.uint8, device="cuda")
    stop = [False]

    def spam():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                for _ in range(20):
                    small.forward(img2, 1)
                st.synchronize()
    th = threading.Thread(target=spam); th.start()
    rng = np.random.default_rng(1)
    for C, Kp, P in [(256, 512, 1408), (512, 512, 4224), (256, 128, 5632)]:
        Pp = (P + 127) // 128 * 128
        V = rng.standard_normal((36, C, Pp)).astype(np.float32); U = (rng.standard_normal((36, C, Kp)) * 0.05).astype(np.float32)
        ref, _ = h3_gemm(V, U, P)
        bad = 0
        for it in range(8):
            M, _ = h3_gemm(V, U, P)
            bad += int(not np.array_equal(M, ref))
        print(f"coresident gemm C={C} Kp={Kp} P={P}: {bad} of 8 runs differ from the first", flush=True)
    stop[0] = True; th.join()
