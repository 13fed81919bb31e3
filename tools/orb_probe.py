#!/usr/bin/env python3
"""ORB stereo extraction + stereo match ALONE (no network beside it): wall time per stereo pair, sequential and with the two images
extracted on two host threads (as the frame pipeline of bench.py does).  Run under `rocprofv3 --kernel-trace --stats` for the number of
kernel launches and copies per pair (tools/gpu_session.sh step `orb`).  Usage: python tools/orb_probe.py [pairs [launch mode]]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_stereo  # noqa: E402
from sivo_amd import orb  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else None          # sivo_orb_set_launch_mode (None: the library's default)
L, R = synthetic_stereo(21, disparity=8)
ex_l, ex_r = orb.ORBextractor(launch_mode=MODE), orb.ORBextractor(launch_mode=MODE)
print("launch mode", MODE)
dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
BF, B = 386.1448, 386.1448 / 718.856


def pair_sequential():
    kl, dl = ex_l(dL); kr, dr = ex_r(dR)
    orb.stereo_match(ex_l, ex_r, kl, dl, kr, dr, BF, B)
    return len(kl), len(kr)


def pair_threaded():
    out = [None, None]

    def run(i, ex, img):
        out[i] = ex(img)
    t = threading.Thread(target=run, args=(1, ex_r, dR))
    t.start(); run(0, ex_l, dL); t.join()
    (kl, dl), (kr, dr) = out
    orb.stereo_match(ex_l, ex_r, kl, dl, kr, dr, BF, B)
    return len(kl), len(kr)


def pair_library():
    (kl, dl), (kr, dr) = orb.extract_pair(ex_l, ex_r, dL, dR)
    orb.stereo_match(ex_l, ex_r, kl, dl, kr, dr, BF, B)
    return len(kl), len(kr)


for name, fn in (("sequential", pair_sequential), ("two host threads", pair_threaded), ("sivo_orb_extract_pair_dev", pair_library)):
    for _ in range(5):
        n = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(N):
        t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    print(f"stereo pair, {name}: mean {ts.mean():.3f} ms, median {np.median(ts):.3f}, min {ts.min():.3f}, max {ts.max():.3f}  (keypoints {n}, {N} pairs)")
ex_l.profile(True)
for _ in range(20):
    ex_l(dL)
ms, calls, keys = ex_l.profile_read()
print("kernel groups of one extraction (HIP events, us):", {k: round(1e3 * v, 1) for k, v in ms.items()}, "keypoints", keys)
