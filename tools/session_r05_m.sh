#!/bin/bash
# Round 5, GPU session M (last): the full-size SegNet tests with their printed figures (pytest -s) and the light round profile of the
# final tree (bench line with cpu_baseline, rocprofv3 kernel statistics of the main configuration plain and on one lane)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_segnet_fullsize.py -q -s > $O/r05_c_fullsize_tests.log 2>&1; echo "fullsize rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/r05_c_fullsize_tests.log | tail -3 | cut -c1-300
PROFILE_LIGHT=1 bash tools/profile_round.sh r05_c
