#!/bin/bash
# Matrix-core busy cycles, LDS instruction / wait counters and the sustained shader clock of the f16x3 GEMM's forms and ablations
# (diagnostic build, tools/h3_probe.py on one shape).  One counter group per pass, no trace domains beside --pmc; durations from a
# separate --kernel-trace pass.  Usage (GPU box): bash tools/h3_pmc.sh <tag> [shape substring]   -> gpurun_out/<tag>/h3_pmc.txt
set -u
export TMPDIR=/tmp
TAG=${1:-r06_h3pmc}
SHAPE=${2:-conv4_2}
# PMC_PROBE / PMC_KERNEL: another probe script and kernel-name filter (e.g. "tools/d3pk_probe.py 3" and conv3_h3_kernel, with SIVO_PROBE_SHAPE / SIVO_PROBE_ZEROS)
PROBE=${PMC_PROBE:-tools/h3_probe.py 5}
KERNEL=${PMC_KERNEL:-wino4_gemm_h3}
export SIVO_PROBE_SHAPE=$SHAPE
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
rm -rf /tmp/h3pmc_*
(H3_PROBE_SHAPE=$SHAPE timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/h3pmc_t -o t -- python $R/$PROBE > $O/h3_pmc_probe_under_trace.log 2>&1)
(H3_PROBE_SHAPE=$SHAPE timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/h3pmc_m -- python $R/$PROBE > /dev/null 2>&1)
(H3_PROBE_SHAPE=$SHAPE timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/h3pmc_l -- python $R/$PROBE > /dev/null 2>&1)
(H3_PROBE_SHAPE=$SHAPE timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/h3pmc_w -- python $R/$PROBE > /dev/null 2>&1)
cd $R
python - <<PY > $O/h3_pmc.txt
import collections, csv, glob, re
def short(n):
    n = re.sub(r"^void ", "", n).split("(")[0].replace("sivo::", "")
    return re.sub(r",\s*", ",", n)
dur = {}
for f in glob.glob("/tmp/h3pmc_t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("m", "l", "w"):
    for f in glob.glob(f"/tmp/h3pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
print("shape: $SHAPE (tools/h3_probe.py, 5 launches per variant + 1 warm-up); template arguments <BM, BN, ABL, FORM>: ABL bits 1 no V' loads, 2 no U' DMA, 4 no M stores, 8 no MFMA,")
print("16 V' by LDS-DMA, 32 start skew, 64 / 128 nt hints; FORM 2 product, 1 4-byte V' loads, 0 phased (round 5).  Counters are per-dispatch means; GRBM_GUI_ACTIVE is summed over the")
print("8 XCDs; clock = GRBM_GUI_ACTIVE / 8 / duration (duration from the separate --kernel-trace pass); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 * 1024 SIMDs)")
print(f"{'kernel':46s} {'us':>8s} {'clock MHz':>9s} {'mfma_busy':>9s} {'INSTS_LDS':>12s} {'WAIT_INST_LDS':>14s} {'WAIT_ANY/WAVE_CYC':>18s}")
for k in sorted(acc):
    if "$KERNEL" not in k: continue
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    ns = dur.get(k, (0, 0))[0]
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
    clk = gui / ns * 1e3 if ns else 0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024) if gui else 0
    wa = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
    print(f"{k[:46]:46s} {ns / 1e3:8.1f} {clk:9.0f} {busy:9.3f} {c.get('SQ_INSTS_LDS', 0):12.0f} {c.get('SQ_WAIT_INST_LDS', 0):14.0f} {wa:18.3f}")
PY
cat $O/h3_pmc.txt
