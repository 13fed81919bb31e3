#!/usr/bin/env python3
"""How many host threads the CPU baseline should use on this box: cores the process may run on, the cgroup's CPU quota, and the
reference-equivalent convolution (oracle/caffe_cpu.c) of one 128 -> 128 layer at 176 x 512 under OMP_NUM_THREADS = 8 ... all."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 128, 176, 512)).astype(np.float32); w = (rng.standard_normal((128, 128, 3, 3)) * 0.03).astype(np.float32)
    for f in (O.caffe_conv2d, O.conv2d):
        f(x, w, None, 1)
        t0 = time.perf_counter()
        for _ in range(3):
            f(x, w, None, 1)
        t = (time.perf_counter() - t0) / 3
        print(f"  OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}: {f.__name__} {t * 1e3:.1f} ms {2 * 128 * 128 * 9 * 176 * 512 / t / 1e9:.0f} GFLOP/s", flush=True)
else:
    print("os.cpu_count()", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(p):
            print(p, open(p).read().strip())
    print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
    for n in (8, 16, 32, 64, 128, 256):
        if n <= os.cpu_count():
            subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=dict(os.environ, OMP_NUM_THREADS=str(n)))
