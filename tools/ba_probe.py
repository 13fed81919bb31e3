#!/usr/bin/env python3
"""sivo_local_ba on SURVEY 8d config 5 (20 keyframes x 3000 points): the distribution of the whole-call time over 30 calls
(bench.py reports their mean).  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sivo_amd import optimizer  # noqa: E402

poses, pts, edges, intr = bench.ba_scene()
rng = np.random.default_rng(3)
fixed = np.zeros(len(poses), np.uint8); fixed[:2] = 1
P0 = poses.copy(); P0[2:, 9:] += rng.normal(0, 0.02, (len(poses) - 2, 3))
X0 = pts + rng.normal(0, 0.05, pts.shape)
optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    t0 = time.perf_counter(); g = optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19); ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts)
print(f"edges {len(edges)} iterations {g['iterations']} trials {g['trials']}: mean {ts.mean():.3f} ms, median {np.median(ts):.3f}, min {ts.min():.3f}, max {ts.max():.3f}")
print("calls:", " ".join(f"{t:.2f}" for t in ts))
