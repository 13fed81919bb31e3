#!/usr/bin/env python3
"""Clock and matrix-core occupancy of wino4_gemm_x6p_kernel under its compile-time ablations (is the GEMM held by
issue contention or by the power budget?).  GPU box only; prepared at the end of round 2, to be run first thing in round 3.

  cd /tmp && export TMPDIR=/tmp
  for a in 0 1 2 3 4 7 8 23; do
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
        -d /tmp/x6clk_$a -- python $GRAFT_REPO_ROOT/tools/x6_clock_probe.py run $a
  done
  python $GRAFT_REPO_ROOT/tools/x6_clock_probe.py summary /tmp/x6clk_*

`run a`: conv4_2 (512 -> 512, 44 x 128, T = 12) through sivo_debug_conv with ablation a of the bf16x6 GEMM (tools/x6_probe.py
lists the bits).  `summary`: per directory the x6p kernel's mean duration (kernel trace), GRBM_GUI_ACTIVE per XCD / duration =
effective clock, SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = matrix cores busy."""
import collections
import csv
import ctypes as C
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(abl):
    from sivo_amd._lib import lib, check
    ms = C.c_double()
    check(lib().sivo_debug_conv(12, 512, 512, 44, 128, 3, 10, 512 | 2048 | (abl << 12), C.byref(ms)))
    print(f"abl {abl}: whole layer {ms.value:.4f} ms")


def summary(dirs):
    for d in sorted(dirs):
        cnt = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "x6p_kernel" in r["Kernel_Name"]:
                    c = cnt[r["Counter_Name"]]
                    c[0] += float(r["Counter_Value"]); c[1] += 1
        dur = []
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "x6p_kernel" in r["Kernel_Name"]:
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        mean = {k: v[0] / v[1] for k, v in cnt.items() if v[1]}
        ns = sum(dur) / len(dur) if dur else float("nan")
        cyc = mean.get("GRBM_GUI_ACTIVE", float("nan")) / 8          # summed over the 8 XCDs
        print(f"{d}: launches {len(dur)}  {ns / 1e3:8.1f} us  clock {cyc / ns:5.3f} GHz  "
              f"mfma busy {mean.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / (cyc * 1024):5.3f}  counters {sorted(mean)}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(int(sys.argv[2]))
    elif len(sys.argv) >= 3 and sys.argv[1] == "summary":
        summary(sys.argv[2:])
    else:
        sys.exit(__doc__)
