#!/bin/bash
# (Round 5: a handle launches kernels while it is constructed — calibration passes, the accuracy guard's two frames, partly the SAME
# kernels at two samples per launch.  The per-kernel figures below count the FRAMES only: tools/frame_kernel_stats.py and
# tools/pmc_summary.py drop every dispatch up to the guard's last absdiff_max_kernel; rocprofv3's own --stats file, which counts
# everything, is kept beside it as *_incl_construction.csv.  `rocprofv3 --selected-regions` produced no output here.)
# Round profile on the GPU box: bench line, rocprofv3 kernel statistics of every configuration of the line (main run plain and
# SIVO_LANES=1 — the one whose average launch duration must agree with the HIP events of `roofline` —, SegNet-Basic T = 6,
# Standard T = 48, local BA), PMC passes (HBM traffic / matrix-core busy; one counter group per pass, no trace domains) for the
# main run, Basic and T = 48, optionally the GPU test suite.  Every profiler pass runs under `timeout` and with --serial: round 4's
# last session lost 36 GPU-minutes to the one-lane `--kernel-trace` pass hanging with two frames in flight (the three-lane pass of
# the same build completed; without rocprofv3 the one-lane bench runs, tools/gpu_session.sh).   Usage: bash tools/profile_round.sh <tag> [tests|onlytests]
#   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
set -u
TAG=${1:-r03_x}
WITH_TESTS=${2:-}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$WITH_TESTS" = onlytests ]; then timeout 1700 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_gpu_tests.log; exit 0; fi
pmc() {   # name, bench args
  local name=$1 args=$2
  for pass in "f FETCH_SIZE" "w WRITE_SIZE" "m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass; p=$1; shift
    rm -rf /tmp/pmc_${name}_$p
    (cd /tmp && SIVO_LANES=1 timeout 240 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_${name}_$p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-orb --configs none $args > /dev/null 2>&1)
  done
  python tools/pmc_summary.py $O/${TAG}_pmc_traffic_$name.json /tmp/pmc_${name}_f /tmp/pmc_${name}_w /tmp/pmc_${name}_m > /dev/null && python - <<PY
import json
d=json.load(open("$O/${TAG}_pmc_traffic_$name.json"))
for k,v in d.items():
    if isinstance(v,dict) and v.get("bytes",0)>5e7: print("$name", k[:60], v.get("dispatches"), "MB", round(v["bytes"]/1e6,1), "mfma_busy", round(v.get("mfma_busy_frac",0),3))
PY
}
# PROFILE_LIGHT=1: the bench line and the kernel statistics of the main configuration only (the PMC summaries in profiles/ stay)
LIGHT=${PROFILE_LIGHT:-}
if [ -z "$LIGHT" ]; then
pmc main ""
pmc basic "--net basic --T 6"
pmc t48 "--T 48"
fi
# the line reads its `traffic` from profiles/: the PMC passes of THIS build first
[ -z "$LIGHT" ] && cp $O/${TAG}_pmc_traffic_*.json $R/profiles/ 2>/dev/null
timeout 600 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench_line.json")); r=d["roofline"]
print(d["value"], "fps", d["ms_per_step"], "ms frac", r["frac"], "executed", r.get("executed_frac"), [(c["name"][:28], c["value"]) for c in d.get("configs", [])])
for m in d.get("membound", []): print("  ", m["kernel"][:70], m["avg_us"], "us", m["achieved_GBps"], "GB/s")
PY
stats() {   # name, bench args (quoted string), env...
  local name=$1 args=$2; shift 2
  rm -rf /tmp/prof_$name
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- python $R/bench.py --serial --no-cpu-baseline --configs none $args > $O/${TAG}_bench_line_under_rocprof_$name.json 2>/dev/null)
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_${name}_incl_construction.csv
  local t=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/frame_kernel_stats.py $t $O/${TAG}_kernel_stats_$name.csv && head -5 $O/${TAG}_kernel_stats_$name.csv | cut -c1-150
}
stats main "--steps 20" SIVO_DUMMY=1
stats onelane "--steps 20" SIVO_LANES=1
[ -n "$LIGHT" ] && exit 0
stats basic "--net basic --T 6 --steps 20 --no-orb" SIVO_LANES=1
stats t48 "--T 48 --steps 4 --warmup 1 --no-orb" SIVO_LANES=1
timeout 200 python tools/layer_times.py 10 > $O/${TAG}_layer_times.txt 2>&1; tail -3 $O/${TAG}_layer_times.txt
rm -rf /tmp/prof_ba
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -o ba -- python $R/tests/tools/ba_bench.py > $O/${TAG}_ba_bench_under_rocprof.json 2>/dev/null)
f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_ba.csv && head -6 $O/${TAG}_kernel_stats_ba.csv | cut -c1-150
timeout 300 python tests/tools/ba_bench.py > $O/${TAG}_ba_bench.json 2>/dev/null; cut -c1-400 $O/${TAG}_ba_bench.json
if [ "$WITH_TESTS" = tests ]; then
  timeout 1700 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_gpu_tests.log
fi
