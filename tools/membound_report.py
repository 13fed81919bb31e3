#!/usr/bin/env python3
"""rocprofv3 kernel stats of tools/membound_workload.py -> achieved GB/s per kernel against the 8 TB/s HBM3E roofline.
Algorithmic bytes per call: pyramid level areas for (2000, 1.2, 8) at 1024x352 (+19 px border)."""
import csv, json, sys
LV = [(1024, 352), (853, 293), (711, 244), (593, 204), (494, 170), (412, 141), (343, 118), (286, 98)]
area = [w * h for w, h in LV]
padded = sum((w + 38) * (h + 38) for w, h in LV)
pyr = sum(area)
NKP = 2000
BYTES = {
    "pyramid_kernel": 2 * area[0] + sum(area[:-1]) + sum(area[1:]),     # the source once, every level written once and read once by the next
    "blur_border_kernel": 2 * pyr + padded - pyr + pyr * 0.1,           # blur: every level in and out; borders: the 19-px frames
    "fast_cells_kernel": pyr + 2 * 4 * 20000,                           # every level once + the candidates (+ the cell counts)
    "orient_describe_kernel": NKP * (749 + 512 + 36),                   # disc r = 15 + 512 test pixels per keypoint, 36-byte record
    "stereo_sad_kernel": 600 * (11 * 11 + 11 * 21),                     # ~600 matched keypoints, 11x11 window + 21-wide strip
    "hamming_matrix_kernel": 32 * 4000 + 4 * 2000 * 2000,
    "hamming_argmin2_kernel": 32 * 4000 + 3 * 4 * 2000,                 # brute force: every B row per query from L2
    "mc_reduce_kernel": 12 * 15 * 352 * 1024 * 4 + 15 * 352 * 1024 * 4,
    "mc_finalize_kernel": 352 * 1024 * (60 + 17),
}
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("void ", "").replace("sivo::", "").split("(")[0].split("<")[0]
    if name in BYTES:
        us = float(r["AverageNs"]) / 1e3
        rows[name] = {"calls": int(r["Calls"]), "avg_us": round(us, 2), "algorithmic_bytes": int(BYTES[name]),
                      "achieved_GBps": round(BYTES[name] / us / 1e3, 1), "frac_of_8TBps": round(BYTES[name] / us / 1e3 / 8000, 4),
                      "time_at_roofline_us": round(BYTES[name] / 8e6, 3)}
out = {"_how": "rocprofv3 --kernel-trace --stats over tools/membound_workload.py (ORB stereo pair + stereo match, 2000x2000 Hamming, "
               "MC reduction; nothing else on the GPU); achieved = algorithmic bytes / mean kernel duration; a kernel whose whole "
               "working set is 1-2 MB needs 0.1-0.3 us at 8 TB/s, far below the ~2-5 us a launch costs: those rows are "
               "launch/latency bound by construction", "kernels": rows}
json.dump(out, open(sys.argv[2], "w"), indent=1) if len(sys.argv) > 2 else None
for k, v in rows.items():
    print(f"{k:26s} {v['avg_us']:9.2f} us  {v['algorithmic_bytes']/1e6:9.3f} MB  {v['achieved_GBps']:8.1f} GB/s  ({100*v['frac_of_8TBps']:.2f}% of 8 TB/s)")
