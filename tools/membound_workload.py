#!/usr/bin/env python3
"""Workload for the memory-bound / integer kernels in isolation (no network running beside them): run under
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o mb -- python tools/membound_workload.py
then tools/membound_report.py <dir>/*mb_kernel_stats.csv turns the per-kernel averages into achieved GB/s."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_stereo
from sivo_amd import matcher, orb
from sivo_amd.segnet import mc_finalize, mc_reduce

L, R = synthetic_stereo(21, disparity=8)
ex_l, ex_r = orb.ORBextractor(), orb.ORBextractor()
dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
for _ in range(30):
    kl, dl = ex_l(dL); kr, dr = ex_r(dR)
    orb.stereo_match(ex_l, ex_r, kl, dl, kr, dr, 386.1448, 386.1448 / 718.856)
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.integers(0, 256, (2000, 32), dtype=np.uint8)).cuda()
B = torch.from_numpy(rng.integers(0, 256, (2000, 32), dtype=np.uint8)).cuda()
for _ in range(30):
    matcher.descriptor_distance_matrix(A, B)
    matcher.bruteforce(A, B)
lg = torch.randn(12, 15, 352, 1024, device="cuda")
for _ in range(30):
    ps, _ = mc_reduce(lg)
    mc_finalize(ps, 12)
torch.cuda.synchronize()
print("keypoints", len(kl), len(kr))
