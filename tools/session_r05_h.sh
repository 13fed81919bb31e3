#!/bin/bash
# Round 5, GPU session H: the fork dropout in the input transform (A/B test, bands + full-size + e2e tests on it), bench line
set -u
TAG=${1:-r05_h}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x > $O/segnet_tests.log 2>&1; echo "segnet tests rc=$?"; tail -3 $O/segnet_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_prefix_bands.py tests/test_gpu_frame_e2e.py tests/test_gpu_coresidency.py -m gpu -q -x > $O/bands_e2e_tests.log 2>&1; echo "bands / e2e / coresidency rc=$?"; tail -3 $O/bands_e2e_tests.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_segnet_fullsize.py -m gpu -q -x -s > $O/fullsize_tests.log 2>&1; echo "fullsize rc=$?"; grep -E "passed|failed|class map vs" $O/fullsize_tests.log | tail -12 | cut -c1-300
timeout 300 python bench.py --configs shards --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r05_h/bench_line.json").read())
    r=d["roofline"]
    print("value", d["value"], d["ms_per_step"], "serial", d["config"].get("serial_fps"), "frac", r["frac"], "chain", r.get("chain_frac"), r.get("chain_ms_per_frame"), "stack", r.get("conv_stack_frac"))
    print({k: v for k, v in r["kernels_ms_per_frame"].items()})
    for c in d.get("configs", []):
        for x in c.get("shards", []): print("  ", x)
except Exception as e:
    print("parse failed", e)
P
tail -2 $O/bench.err
