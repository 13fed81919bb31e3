#!/usr/bin/env python3
"""GPU box: what compiling the WHOLE product library without packed-FP32 VALU instructions would cost (the blanket form
of the mitigation).  sivo_amd/libsivo_hip_nopk.so: `make -C sivo_amd/csrc nopk` (every source with -Xclang -target-feature -Xclang
-packed-fp32-ops; not part of `make all`); bench.py's main line runs once with the product library and once with that one.
    python tools/nopk_cost_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = ("import sys, runpy; sys.path.insert(0, {root!r}); import sivo_amd._lib as L; L.LIB_PATH = {lib!r}; "
       "sys.argv = ['bench.py', '--configs', 'none', '--no-cpu-baseline', '--steps', '40', '--warmup', '5']; runpy.run_path({bench!r}, run_name='__main__')")

if __name__ == "__main__":
    for name, lib in (("product library", "libsivo_hip.so"), ("every kernel without packed-FP32 instructions", "libsivo_hip_nopk.so"), ("product library, again", "libsivo_hip.so")):
        path = os.path.join(ROOT, "sivo_amd", lib)
        if not os.path.exists(path):
            print(f"[{name}] {lib} is not built"); continue
        out = subprocess.run([sys.executable, "-c", RUN.format(root=ROOT, lib=path, bench=os.path.join(ROOT, "bench.py"))], capture_output=True, text=True, timeout=200, cwd=ROOT)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"[{name}] {d['value']} frames/s, {d['ms_per_step']} ms per frame, serial {d['config'].get('serial_fps')}, roofline.frac {d['roofline']['frac']}", flush=True)
        except Exception as e:
            print(f"[{name}] failed: {e}\n{out.stdout[-500:]}\n{out.stderr[-1500:]}", flush=True)
