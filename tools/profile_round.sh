#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel statistics (plain run and SIVO_LANES=1 — the one whose average
# launch duration must agree with the HIP events of `roofline`), PMC passes (traffic / MFMA busy; one counter group per pass,
# no trace domains), optionally the GPU test suite.  Usage: bash tools/profile_round.sh <tag> [tests]
#   (tests only: bash tools/profile_round.sh <tag> onlytests)
#   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
set -u
TAG=${1:-r02_x}
WITH_TESTS=${2:-}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$WITH_TESTS" = onlytests ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_gpu_tests.log; exit 0; fi
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/${TAG}_bench_line.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],[ (c['name'][:28],c['value']) for c in d.get('configs',[])])"
stats() {   # name, env..., -- bench args
  local name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- python $R/bench.py --steps 20 --no-cpu-baseline --configs none > $O/${TAG}_bench_line_under_rocprof_$name.json 2>/dev/null)
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_$name.csv && head -6 $O/${TAG}_kernel_stats_$name.csv | cut -c1-150
}
stats main SIVO_DUMMY=1
stats onelane SIVO_LANES=1
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass; p=$1; shift
  rm -rf /tmp/pmc_$p
  (cd /tmp && SIVO_LANES=1 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-orb --configs none > /dev/null 2>&1)
done
python tools/pmc_summary.py $O/${TAG}_pmc_traffic.json /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m && python - <<PY
import json
d=json.load(open("$O/${TAG}_pmc_traffic.json"))
for k,v in d.items():
    if isinstance(v,dict) and v.get("bytes",0)>5e7: print(k[:60], v.get("dispatches"), "MB", round(v["bytes"]/1e6,1), "mfma_busy", round(v.get("mfma_busy_frac",0),3))
PY
if [ "$WITH_TESTS" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_gpu_tests.log
fi
