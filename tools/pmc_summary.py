#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-kernel, per-dispatch figures.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-orb
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python bench.py ... (same)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -- ...
  python tools/pmc_summary.py out.json /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m

HBM bytes per dispatch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024: FETCH_SIZE under-counts by 2x on gfx950
(MI355X_MICROARCH.md, HBM / rocprofv3 section); the correction is re-checked here on mc_reduce_kernel, whose
algorithmic read is known (T * 15 * H * W * 4 bytes)."""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0].replace("sivo::", "")
    return re.sub(r",\s*", ",", name).replace(",0>", ">") if name.startswith("conv_wino_kernel") else re.sub(r",\s*", ",", name)


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dropped = 0
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            # frames only: a handle's construction (calibration passes, the accuracy guard) launches kernels too, partly the same ones at
            # two samples per launch; the guard's comparison kernel is the last thing construction launches
            key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
            last = max((int(r[key]) for r in rows if "absdiff_max_kernel" in r["Kernel_Name"]), default=-1) if key else -1
            for r in rows:
                if key and int(r[key]) <= last:
                    dropped += 1
                    continue
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    res = {"_construction_rows_dropped": dropped, "_how": "FRAMES ONLY (every dispatch up to the accuracy guard's last absdiff_max_kernel dropped: construction); rocprofv3 --pmc passes (one counter group per pass, no trace domains) over `python bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-orb`; per-dispatch means; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 with the gfx950 "
                   "FETCH_SIZE correction of MI355X_MICROARCH.md; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs / 8 XCD-summed)"}
    for k, cs in sorted(acc.items()):
        e = {"dispatches": max(v[1] for v in cs.values())}
        for c, (tot, n) in cs.items():
            e[c] = tot / n
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["bytes"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all SIMDs (4 per CU, 256 CUs)
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
        res[k] = e
    json.dump(res, open(out_path, "w"), indent=1)
    for k, e in res.items():
        if not k.startswith("_"):
            print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()})


if __name__ == "__main__":
    main()
