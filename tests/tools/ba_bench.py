#!/usr/bin/env python3
"""Times SURVEY.md 8d config 5 (local BA: 20 keyframes x 3000 map points) and PoseOptimization on one MI355X
against the CPU oracle on the same box.  Prints one JSON object.  Not part of bench.py's contract."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def best(f, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return min(ts), r


def main():
    import torch
    from conftest import make_ba_scene, perturb_pose
    from oracle import oracle
    from sivo_amd import optimizer
    poses, pts, edges, intr = make_ba_scene()
    rng = np.random.default_rng(3)
    fixed = np.zeros(len(poses), np.uint8); fixed[:2] = 1
    P0 = poses.copy()
    for i in range(2, len(poses)):
        P0[i] = perturb_pose(poses[i], rng, 0.002, 0.02)
    X0 = pts + rng.normal(0, 0.05, pts.shape)
    nE = len(edges)
    out = {"edges": int(nE), "keyframes": len(poses), "points": len(pts)}

    # per-edge linearisation, device-resident (the hot loop the north star names)
    dP = torch.from_numpy(P0).cuda(); dX = torch.from_numpy(X0).cuda()
    dE = torch.from_numpy(edges.view(np.uint8).reshape(-1)).cuda()
    bufs = {"err": torch.empty(nE, 3, dtype=torch.float64, device="cuda"), "Jx": torch.empty(nE, 9, dtype=torch.float64, device="cuda"),
            "Jp": torch.empty(nE, 18, dtype=torch.float64, device="cuda"), "chi2": torch.empty(nE, dtype=torch.float64, device="cuda"),
            "rho": torch.empty(nE, dtype=torch.float64, device="cuda"), "w": torch.empty(nE, dtype=torch.float64, device="cuda"),
            "depth_ok": torch.empty(nE, dtype=torch.uint8, device="cuda")}
    for _ in range(20):
        optimizer.linearize_dev(dP, dX, dE, nE, intr, bufs)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200):
        optimizer.linearize_dev(dP, dX, dE, nE, intr, bufs)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 200
    alg_bytes = nE * (48 + 96 + 24 + (3 + 9 + 18 + 3) * 8 + 1)       # edge + pose + point gathers in, err/Jx/Jp/chi2/rho/w/flag out
    out["linearize"] = {"ms": round(ms, 5), "edges_per_s": round(nE / ms * 1e3), "algorithmic_bytes": alg_bytes,
                        "achieved_GBps": round(alg_bytes / ms / 1e6, 1), "hbm_peak_GBps": 8000,
                        "note": "36 k edges = 15.6 MB per launch: launch-latency bound, not bandwidth bound, at this size"}
    t_cpu, _ = best(lambda: oracle.ba_linearize(P0, X0, edges, intr), 5)
    out["linearize"]["cpu_oracle_ms"] = round(t_cpu * 1e3, 3)

    optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)                       # warm-up (module load)
    t_gpu, g = best(lambda: optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19), 5)
    t_cpu, o = best(lambda: oracle.local_ba(P0, fixed, X0, edges, intr, cov_pose=19), 3)
    out["local_ba"] = {"gpu_ms_whole_call": round(t_gpu * 1e3, 2), "cpu_oracle_ms": round(t_cpu * 1e3, 2),
                       "lm_iterations": g["iterations"], "trials": g["trials"],
                       "gpu_ms_per_lm_iteration": round(t_gpu * 1e3 / max(g["iterations"], 1), 3),
                       "max_pose_diff_vs_oracle": float(np.abs(g["poses"] - o["poses"]).max()),
                       "note": "whole call = host CSR build + H2D + 15 LM iterations (all kernels on the GPU) + D2H; oracle is single-threaded C"}

    ek = edges[edges["pose"] == 5].copy()
    p0 = perturb_pose(poses[5], np.random.default_rng(0))
    optimizer.pose_optimize(p0, pts, ek, intr)
    t_gpu, g = best(lambda: optimizer.pose_optimize(p0, pts, ek, intr), 10)
    t_cpu, o = best(lambda: oracle.pose_optimize(p0, pts, ek, intr), 10)
    out["pose_optimization"] = {"edges": len(ek), "gpu_ms_whole_call": round(t_gpu * 1e3, 3), "cpu_oracle_ms": round(t_cpu * 1e3, 3),
                                "lm_iterations": g["iterations"], "trials": g["trials"],
                                "note": "one launch of one persistent workgroup; whole call includes H2D/D2H and hipMalloc"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
