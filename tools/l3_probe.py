#!/usr/bin/env python3
"""GPU box: what the 256 MiB Infinity Cache gives a streaming kernel on this part — read (sum), write (fill) and copy rates of
torch's own streaming kernels over working sets from 8 MB to 2 GB, each repeated back to back so that a working set that fits a
cache level is served from it from the second pass on.  The question behind it (task 1a): would the F(4x4) bridge read M
faster if M (26-208 MB per sample group) were still in the Infinity Cache when it runs?"""
import sys

import torch


def rate(fn, nbytes, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def main():
    print("working set MB | read TB/s (sum) | write TB/s (fill) | copy TB/s (read + write, src + dst = 2 x working set / 2)")
    for mb in (8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048):
        n = mb * (1 << 20) // 4
        a = torch.ones(n, dtype=torch.float32, device="cuda")
        h = n // 2
        reps = max(8, min(400, (8 << 30) // (mb << 20)))
        rd = rate(lambda: a.sum(), n * 4, reps)
        wr = rate(lambda: a.fill_(1.0), n * 4, reps)
        cp = rate(lambda: a[:h].copy_(a[h:2 * h]), h * 4 * 2, reps)       # src + dst together = the working set
        print(f"{mb:6d} | {rd:6.2f} | {wr:6.2f} | {cp:6.2f}", flush=True)
        del a
    # producer -> consumer through memory, the GEMM -> bridge case: kernel A writes a buffer, kernel B reads it right behind
    print("producer -> consumer (fill then sum of the same buffer, back to back): MB | TB/s of the pair (2 x bytes / time)")
    for mb in (26, 52, 104, 208, 312, 624):
        n = mb * (1 << 20) // 4
        a = torch.empty(n, dtype=torch.float32, device="cuda")

        def pair():
            a.fill_(2.0)
            a.sum()
        print(f"{mb:6d} | {rate(pair, 2 * n * 4, max(8, min(200, (8 << 30) // (mb << 20)))):6.2f}", flush=True)
        del a


if __name__ == "__main__":
    sys.exit(main())
