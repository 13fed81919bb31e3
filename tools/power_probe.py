#!/usr/bin/env python3
"""Board power, shader clock and power cap while a command runs (is the f16x3 GEMM power-limited?).

    python tools/power_probe.py [--interval 0.02] -- <command ...>

Samples the amdgpu hwmon files of the first GPU (power1_average / power1_input in microwatts, freq1_input in Hz, power1_cap)
from a thread while the command runs, and prints per-sample statistics plus one `rocm-smi` snapshot taken half-way.  Needs no
privileges: the files are world-readable.  NOTEBOOK 11 quotes its output (profiles/r06_power_*.log)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def hwmon_dir():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if any(os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input")):
            return d
    return None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def main():
    args = sys.argv[1:]
    interval = 0.02
    if args and args[0] == "--interval":
        interval = float(args[1]); args = args[2:]
    if args and args[0] == "--":
        args = args[1:]
    if not args:
        print(__doc__); return 2
    d = hwmon_dir()
    print("hwmon:", d)
    if d:
        for f in ("power1_cap", "power1_cap_max", "power1_cap_default"):
            v = read_int(os.path.join(d, f))
            if v is not None:
                print(f"  {f}: {v / 1e6:.0f} W")
    pfile = None
    if d:
        for f in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(d, f)):
                pfile = os.path.join(d, f); break
    ffile = os.path.join(d, "freq1_input") if d and os.path.exists(os.path.join(d, "freq1_input")) else None
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            t = time.time()
            p = read_int(pfile) if pfile else None
            f = read_int(ffile) if ffile else None
            samples.append((t, p, f))
            time.sleep(interval)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    proc = subprocess.Popen(args)
    smi = None
    while proc.poll() is None:
        time.sleep(0.25)
        if smi is None and time.time() - t0 > float(os.environ.get("POWER_PROBE_SMI_AT", "6")):
            try:
                smi = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showperflevel", "--json"],
                                     capture_output=True, text=True, timeout=20).stdout
            except Exception as e:          # noqa: BLE001
                smi = f"rocm-smi failed: {e}"
    stop.set(); th.join()
    ps = [p / 1e6 for _, p, _ in samples if p]
    fs = [f / 1e6 for _, _, f in samples if f]
    out = {"command": " ".join(args), "rc": proc.returncode, "seconds": round(time.time() - t0, 2), "samples": len(samples)}
    if ps:
        ps_sorted = sorted(ps)
        out["power_W"] = {"mean": round(sum(ps) / len(ps), 1), "p50": round(ps_sorted[len(ps) // 2], 1), "p90": round(ps_sorted[int(len(ps) * 0.9)], 1),
                          "max": round(ps_sorted[-1], 1), "min": round(ps_sorted[0], 1)}
    if fs:
        fs_sorted = sorted(fs)
        out["sclk_MHz"] = {"mean": round(sum(fs) / len(fs), 1), "p10": round(fs_sorted[int(len(fs) * 0.1)], 1), "p50": round(fs_sorted[len(fs) // 2], 1),
                           "max": round(fs_sorted[-1], 1), "min": round(fs_sorted[0], 1)}
    print(json.dumps(out))
    if smi:
        print("rocm-smi snapshot:", smi.strip()[:2000])
    return 0


if __name__ == "__main__":
    sys.exit(main())
