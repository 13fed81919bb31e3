#!/bin/bash
# Round 5, GPU session G: the whole -m gpu suite on the tree as it stands (no -x: every failure listed)
set -u
TAG=${1:-r05_g}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/gpu_tests.log | tail -30 | cut -c1-300
