#!/bin/bash
# Round 5, GPU session B: co-residency repro with the victim capped at 72 VGPRs (fits beside the real GEMM) + GEMM-like occupants;
# the load-time accuracy guard: per-layer table for growing BN offsets, guarded vs unguarded logit error; segnet tests + smoke with the guard on.
set -u
TAG=${1:-r05_b}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python tools/coresident_repro.py > $O/coresident_repro.log 2>&1; echo "repro rc=$?"
grep "^\[\|GEMM beside" $O/coresident_repro.log | cut -c1-700
timeout 900 python tools/guard_probe.py 0 3 30 100 > $O/guard_probe.log 2>&1; echo "guard probe rc=$?"
grep -v "amdgpu.ids" $O/guard_probe.log | cut -c1-400
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x > $O/segnet_tests.log 2>&1; echo "segnet tests rc=$?"; tail -3 $O/segnet_tests.log
