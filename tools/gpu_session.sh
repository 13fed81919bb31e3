#!/bin/bash
# One GPU-box session: the steps named on the command line, in order, every log under gpurun_out/<tag>/.
#   bash tools/gpu_session.sh <tag> step...      steps: gemm segnet bench benchvar e2e fullsize ba orb power alltests
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bench() {   # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --per-layer --configs none --no-cpu-baseline --steps 20 > $O/bench_$name.json 2> $O/bench_$name.err
  echo "bench $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); r=d["roofline"]
    print("  ", d["value"], "fps", d["ms_per_step"], "ms; dom", r["kernel"], "avg", round(r["avg_launch_ms"],4), "ms frac", r["frac"], "exec", r.get("executed_frac"))
    print("  ", {k: v for k, v in r["kernels_ms_per_frame"].items()})
except Exception as e: print("  no line:", e)
PY
}
for step in "$@"; do
  case $step in
    pk) timeout 900 python -m pytest tests/test_gpu_conv3_pk.py -q -s > $O/pk.log 2>&1; echo "pk rc=$?"; grep -E "packed|passed|failed|FAILED|Error|assert" $O/pk.log | tail -70;;
    d3) timeout 600 python -m pytest tests/test_gpu_conv3_h3.py -q -s > $O/d3.log 2>&1; echo "d3 rc=$?"; grep -E "TFLOP|passed|failed|FAILED|Error" $O/d3.log | tail -20;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log;;
    gemm) timeout 600 python -m pytest tests/test_gpu_h3_gemm.py -x -q -s > $O/gemm.log 2>&1; echo "gemm rc=$?"; grep -E "h3 gemm|passed|failed|Error|assert" $O/gemm.log | tail -15;;
    segnet) timeout 900 python -m pytest tests/test_gpu_segnet.py -q -s > $O/segnet.log 2>&1; echo "segnet rc=$?"; grep -E "passed|failed|FAILED|dlogit" $O/segnet.log | tail -25;;
    bench) bench default SIVO_DUMMY=1;;
    benchvar) bench nostagger SIVO_H3_STAGGER=0; bench lanes1 SIVO_LANES=1; bench x6 SIVO_GEMM=x6;;
    probe) timeout 600 python tools/h3_probe.py 20 > $O/probe_256.log 2>&1; cat $O/probe_256.log;;
    lanes) timeout 300 python tools/h3_debug2.py default > $O/lanes.log 2>&1; grep "lanes ==" $O/lanes.log;;
    e2e) timeout 600 python -m pytest tests/test_gpu_frame_e2e.py -q -s > $O/e2e.log 2>&1; echo "e2e rc=$?"; grep -E "e2e|passed|failed|Error" $O/e2e.log | tail -8;;
    fullsize) timeout 1500 python -m pytest tests/test_gpu_segnet_fullsize.py -q -s > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "dlogit|passed|failed|FAILED|Error" $O/fullsize.log | tail -40;;
    track) timeout 900 python -m pytest tests/test_gpu_ba_solve.py tests/test_gpu_search.py tests/test_gpu_match_ba.py tests/test_pin_matcher.py tests/test_pin_optimizer.py tests/test_cpp_api.py tests/test_pin_helpers.py -m gpu -q -s > $O/track.log 2>&1; echo "track rc=$?"; grep -E "pose_optimize|passed|failed|FAILED|Error|FAIL " $O/track.log | tail -30;;
    benchtrack) timeout 600 python bench.py --configs track,ba,shards --no-cpu-baseline --steps 20 > $O/bench_track.json 2> $O/bench_track.err; echo "benchtrack rc=$?"; tail -5 $O/bench_track.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_track.json"))
    print("  ", d["value"], "fps", d["ms_per_step"], "ms; gate", d["config"].get("entropy_gate"))
    for c in d.get("configs", []):
        print("  ", {k: v for k, v in c.items() if k not in ("name", "note", "parity", "roofline")})
except Exception as e: print("  no line:", e)
PY
;;
    ba) timeout 900 python -m pytest tests/test_gpu_ba_solve.py tests/test_pin_optimizer.py tests/test_gpu_match_ba.py tests/test_cpp_api.py -m gpu -q > $O/ba_tests.log 2>&1; echo "ba tests rc=$?"; tail -4 $O/ba_tests.log
        timeout 300 python tests/tools/ba_bench.py > $O/ba_bench.json 2> $O/ba_bench.err; cut -c1-900 $O/ba_bench.json
        timeout 120 python tools/ba_probe.py 30 2>/dev/null | tee $O/ba_probe.log
        rm -rf /tmp/prof_ba; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -o ba -- python $R/tools/ba_probe.py 10 > /dev/null 2>&1)
        f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_ba.csv && head -14 $O/kernel_stats_ba.csv | cut -c1-170
        timeout 300 python bench.py --configs ba --no-cpu-baseline --no-orb --steps 3 --warmup 1 > $O/bench_ba.json 2> $O/bench_ba.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_ba.json"))
    for c in d.get("configs", []): print("  ", {k: v for k, v in c.items() if k not in ("name", "note", "parity", "roofline")})
except Exception as e: print("  no line:", e)
PY
;;
    orb) timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_pin_orb.py tests/test_pin_frame.py tests/test_gpu_frame_e2e.py tests/test_codeobj.py tests/test_cpp_api.py -m gpu -q > $O/orb_tests.log 2>&1; echo "orb tests rc=$?"; tail -4 $O/orb_tests.log
        for m in 0 15; do timeout 200 python tools/orb_probe.py 50 $m 2>/dev/null; done | tee $O/orb_probe.log
        for m in 0 15; do
          rm -rf /tmp/prof_orb; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_orb -o orb -- python $R/tools/orb_probe.py 20 $m > /dev/null 2>&1)
          f=$(find /tmp/prof_orb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_orb_mode$m.csv && cat $O/kernel_stats_orb_mode$m.csv | cut -c1-150
        done
        bash tools/ab_orb_modes.sh 2>&1 | tee $O/orb_launch_modes.log
;;
    power)   # board power / shader clock under the f16x3 GEMM and its ablations (is the kernel power-limited?)
        for v in "as built" "MFMA + LDS only" "no MFMA" "no M stores"; do
          H3_PROBE_SHAPE=conv4_2 H3_PROBE_ONLY="$v" POWER_PROBE_SMI_AT=9 timeout 200 python tools/power_probe.py -- python tools/h3_probe.py 20000 2>&1 | grep -v "^$" | cut -c1-1500
        done > $O/power.log 2>&1
        H3_PROBE_ZEROS=1 H3_PROBE_SHAPE=conv4_2 H3_PROBE_ONLY="as built" POWER_PROBE_SMI_AT=99 timeout 200 python tools/power_probe.py -- python tools/h3_probe.py 20000 2>&1 | cut -c1-600 >> $O/power.log
        H3_PROBE_ZEROS=1 H3_PROBE_SHAPE=conv4_2 H3_PROBE_ONLY="MFMA + LDS only" POWER_PROBE_SMI_AT=99 timeout 200 python tools/power_probe.py -- python tools/h3_probe.py 20000 2>&1 | cut -c1-600 >> $O/power.log
        POWER_PROBE_SMI_AT=99 timeout 200 python tools/power_probe.py -- python bench.py --steps 600 --warmup 3 --no-cpu-baseline --configs none 2>/dev/null | cut -c1-600 >> $O/power.log
        cat $O/power.log;;
    alltests) timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "alltests rc=$?"; tail -5 $O/gpu_tests.log;;
  esac
done
