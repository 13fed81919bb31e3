#!/usr/bin/env python3
"""Localises an error of the direct f16x3 convolution (conv3_h3.hip): one input channel, one tap and one output channel at a
time (the output must be a shifted copy of the input plane), then random data per item position.  Prints only what differs."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from sivo_amd import segnet


def shifted(x, ky, kx):
    H, W = x.shape
    xp = np.zeros((H + 2, W + 2), x.dtype)
    xp[1:-1, 1:-1] = x
    return xp[ky:ky + H, kx:kx + W]


def main():
    torch.manual_seed(0)
    N, Cin, Cout, H, W = 2, 32, 64, 16, 128
    rng = np.random.default_rng(0)
    bad = 0
    for (c0, co, ky, kx) in [(0, 0, 1, 1), (0, 0, 0, 0), (0, 0, 2, 2), (1, 0, 1, 1), (8, 0, 1, 1), (16, 0, 1, 1), (31, 63, 0, 2), (5, 33, 2, 0), (17, 4, 1, 0)]:
        x = np.zeros((N, Cin, H, W), np.float32)
        plane = rng.integers(1, 100, (N, H, W)).astype(np.float32)
        x[:, c0] = plane
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, c0, ky, kx] = 1.0
        out, _, ov = segnet.conv3_h3(torch.from_numpy(x).cuda(), w, np.ones(Cout, np.float32), np.zeros(Cout, np.float32), relu=False)
        out = out.cpu().numpy()
        ref = np.zeros_like(out)
        for n in range(N):
            ref[n, co] = shifted(plane[n], ky, kx)
        d = np.argwhere(out != ref)
        if len(d):
            bad += 1
            print(f"c0={c0} co={co} tap=({ky},{kx}): {len(d)} differ; first {d[:6].tolist()}; out there {[float(out[tuple(i)]) for i in d[:6]]} ref {[float(ref[tuple(i)]) for i in d[:6]]}")
            print("   couts hit:", sorted(set(d[:, 1].tolist()))[:20], " rows:", sorted(set(d[:, 2].tolist()))[:20], " cols:", sorted(set(d[:, 3].tolist()))[:12], "...")
        else:
            print(f"c0={c0} co={co} tap=({ky},{kx}): exact")
    print("one-hot cases failing:", bad)


if __name__ == "__main__":
    main()
