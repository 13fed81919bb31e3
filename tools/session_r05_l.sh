#!/bin/bash
# Round 5, GPU session L: the bridge without packed-FP32 instructions in the product — the co-residency tests (exact-LDS GEMM beside the
# bridge as shipped: reproducible; the packed form: the reproducer), the main bench line, the SegNet kernel tests, smoke
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_gpu_coresidency.py -q -s > $O/coresidency_tests.log 2>&1; echo "coresidency rc=$?"; grep -E "^\[coresident\]|passed|failed|^FAILED|^ERROR" $O/coresidency_tests.log | cut -c1-400
for rep in a b; do
  timeout 150 python bench.py --configs none --no-cpu-baseline --steps 40 --warmup 5 > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench rc=$?"
  python - $O/bench_$rep.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(d["value"], "fps", d["ms_per_step"], "ms; frac", d["roofline"]["frac"], "serial", d["config"].get("serial_fps"))
except Exception as e: print("parse failed", e)
P
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout 500 python -m pytest tests/test_gpu_segnet.py tests/test_gpu_prefix_bands.py -q -x > $O/segnet_tests.log 2>&1; echo "segnet rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/segnet_tests.log | tail -5 | cut -c1-300
