#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports, for the busiest queue, kernel time vs idle gaps between
consecutive kernels (launch / dependency bubbles).  Usage: gap_probe.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
q, ks = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
ks.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(ks) // 4
ks = ks[skip:]
busy = sum(e - s for s, e, _ in ks)
gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
small = [g for g in gaps if g < 200_000]       # ignore the frame-to-frame host gaps
print(f"queue {q}: {len(ks)} kernels, busy {busy/1e6:.3f} ms, span {(ks[-1][1]-ks[0][0])/1e6:.3f} ms")
print(f"gaps < 0.2 ms: n={len(small)} sum={sum(small)/1e6:.3f} ms mean={sum(small)/max(len(small),1)/1e3:.2f} us; larger gaps: n={len(gaps)-len(small)} sum={(sum(gaps)-sum(small))/1e6:.3f} ms")
by = collections.Counter()
for i, g in enumerate(gaps):
    if g < 200_000: by[ks[i + 1][2][:60]] += g
for k, v in by.most_common(8): print(f"  {v/1e3:9.1f} us before {k}")
