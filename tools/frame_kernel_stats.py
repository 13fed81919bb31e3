#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel statistics of the FRAMES only, in the format of rocprofv3's own --stats file.

A handle launches kernels while it is constructed (calibration passes, the load-time accuracy guard: partly the SAME kernels as a
frame, at two samples per launch); rocprofv3 --stats averages over everything.  The guard's comparison kernel (absdiff_max_kernel)
is the last thing a construction launches: every dispatch up to the last one of it is dropped.
    python tools/frame_kernel_stats.py <kernel_trace.csv> <out.csv>"""
import collections
import csv
import math
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    start = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Begin_Timestamp"
    rows.sort(key=lambda r: int(r[start]))
    last = max((i for i, r in enumerate(rows) if "absdiff_max_kernel" in r["Kernel_Name"]), default=-1)
    kept = rows[last + 1:]
    acc = collections.defaultdict(list)
    for r in kept:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r[start]))
    total = sum(sum(v) for v in acc.values()) or 1
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (len(v) - 1)) if len(v) > 1 else 0.0
            w.writerow([name, len(v), sum(v), round(mean, 6), round(100.0 * sum(v) / total, 4), min(v), max(v), round(sd, 6)])
    print(f"frames only: {len(kept)} of {len(rows)} dispatches kept ({last + 1} construction dispatches dropped)")


if __name__ == "__main__":
    main()
