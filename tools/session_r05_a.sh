#!/bin/bash
# Round 5, GPU session A: the co-residency experiments (self-checking victim beside occupants / the real GEMM; the prepared occupant
# experiment of round 4), the Infinity-Cache probe + per-layer rates at 2 / 4 / 12 samples per launch, the new overflow test, one bench line.
set -u
TAG=${1:-r05_a}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x -k "overflow" > $O/overflow_tests.log 2>&1; echo "overflow tests rc=$?"; tail -3 $O/overflow_tests.log
timeout 600 python tools/coresident_repro.py > $O/coresident_repro.log 2>&1; echo "repro rc=$?"
grep "^\[\|GEMM beside" $O/coresident_repro.log | cut -c1-600
PROBE_ONLY="second run beside" timeout 400 python tools/coresident_probe.py > $O/coresident_occupant.log 2>&1; echo "occupant rc=$?"
grep "^\[" $O/coresident_occupant.log | cut -c1-500
timeout 200 python tools/l3_probe.py > $O/l3_probe.log 2>&1; echo "l3 rc=$?"; cat $O/l3_probe.log
for t in 2 4 12; do
  SIVO_LANES=1 timeout 200 python bench.py --T $t --no-orb --no-cpu-baseline --configs none --per-layer --steps 8 --warmup 2 > $O/per_layer_T$t.json 2> $O/per_layer_T$t.txt; echo "per-layer T=$t rc=$?"
  grep "wino4_bridge\|wino4_input\|wino4_output" $O/per_layer_T$t.txt | cut -c1-150
done
timeout 300 python bench.py --configs none --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench_line.json; tail -3 $O/bench.err
