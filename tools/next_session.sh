#!/bin/bash
# First GPU session of the next round, every step under `timeout` (round 4 lost 36 GPU-minutes to one profiler pass without one):
#   1. the full -m gpu suite on the tree as it stands (round 4's last build was covered file by file)
#   2. the co-residency experiment that was built and not run: the bridge's compared second run beside a synthetic occupant
#      (idle / ds traffic / LDS-DMA traffic) that sits on every CU for 3 ms (NOTEBOOK 3.1e; run in round 5: NOTEBOOK 9.2(a))
#   3. the default bench line, and the same under rocprofv3 with ONE lane and two frames in flight — the combination that hung
#      (120 s limit): does it reproduce, and where does it stand (py-spy is not in the image: the last lines of stderr)
# Usage: gpurun --timeout 1500 -- 'bash tools/next_session.sh r05_a'      -> gpurun_out/<tag>/
set -u
TAG=${1:-r05_a}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 800 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
PROBE_ONLY="second run beside" timeout 300 python tools/coresident_probe.py > $O/coresident_occupant.log 2>&1; echo "occupant rc=$?"
grep "^\[" $O/coresident_occupant.log | cut -c1-400
timeout 300 python bench.py --configs none --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
(cd /tmp && SIVO_LANES=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hang -o hang -- python $R/bench.py --no-cpu-baseline --configs none --steps 10 > $O/bench_onelane_pipelined_under_rocprof.json 2> $O/bench_onelane_pipelined_under_rocprof.err); echo "one lane + two frames in flight under rocprofv3: rc=$? (124 = hung)"
tail -5 $O/bench_onelane_pipelined_under_rocprof.err
