#!/bin/bash
# Round 5, GPU session C: the banded prefix (bit identity at full size, emulated multi-device handle), the accuracy guard with its
# new budget rule (test + probe), the segnet test file, the shard ceilings of the bench line.
set -u
TAG=${1:-r05_c}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_prefix_bands.py -m gpu -q -x -s > $O/prefix_bands_tests.log 2>&1; echo "prefix band tests rc=$?"; grep "^\[bands\|passed\|failed\|Error\|error" $O/prefix_bands_tests.log | cut -c1-300 | tail -15
timeout 900 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x -s > $O/segnet_tests.log 2>&1; echo "segnet tests rc=$?"; grep "^\[guard\|passed\|failed" $O/segnet_tests.log | cut -c1-400 | tail -8
timeout 600 python tools/guard_probe.py 0 3 100 > $O/guard_probe.log 2>&1; echo "guard probe rc=$?"; grep "^\[bn" $O/guard_probe.log | cut -c1-500
timeout 600 python bench.py --configs shards --no-cpu-baseline --steps 10 > $O/bench_shards.json 2> $O/bench_shards.err; echo "bench shards rc=$?"
python - <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r05_c/bench_shards.json").read())
    print("value", d["value"], d["ms_per_step"], "chain_frac", d["roofline"].get("chain_frac"), "conv_stack_frac", d["roofline"].get("conv_stack_frac"))
    for c in d.get("configs", []):
        print(c["name"][:60]); [print("  ", r) for r in c.get("shards", [])]
except Exception as e:
    print("parse failed", e)
P
tail -3 $O/bench_shards.err
