#!/bin/bash
# Round 5, GPU session J: four more hazard variants (plane offset, GEMM tile forms, one lane), then the whole -m gpu suite on the final tree
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_j
mkdir -p $O
cd $R
PROBE_ONLY="HZ2 exact" PROBE_SEEDS8=1 timeout 600 python tools/coresident_probe.py > $O/hazard_variants2.log 2>&1; echo "hazard rc=$?"
grep "^\[" $O/hazard_variants2.log | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/gpu_tests.log | tail -20 | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 300 python bench.py --configs ba --no-cpu-baseline > $O/bench_ba.json 2> $O/bench_ba.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r05_j/bench_ba.json").read())
    print("value", d["value"], d["ms_per_step"])
    for c in d.get("configs", []): print({k: v for k, v in c.items() if k not in ("name", "roofline", "parity")})
except Exception as e:
    print("parse failed", e)
P
