#!/bin/bash
# Round 5, GPU session D: co-residency tests (LDS claims, frame kernels beside each other), guard test + probe at forced scales,
# the banded rank under rocprofv3 (which kernels its time is made of).
set -u
TAG=${1:-r05_d}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_coresidency.py -m gpu -q -x -s > $O/coresidency_tests.log 2>&1; echo "coresidency tests rc=$?"; grep "^\[coresident\|passed\|failed\|Error\|assert" $O/coresidency_tests.log | cut -c1-400 | tail -8
timeout 600 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x -s -k "guard or overflow" > $O/guard_tests.log 2>&1; echo "guard tests rc=$?"; grep "^\[guard\|passed\|failed" $O/guard_tests.log | cut -c1-500 | tail -6
timeout 600 python tools/guard_probe.py boost:-16 > $O/guard_probe_boost.log 2>&1; echo "guard probe rc=$?"; grep -v amdgpu.ids $O/guard_probe_boost.log | cut -c1-300
timeout 300 python tools/band_probe.py 8 30 > $O/band_probe.log 2>&1; echo "band probe rc=$?"; grep "^\[band" $O/band_probe.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_band -o band -- python $R/tools/band_probe.py 8 30 banded > $O/band_probe_rocprof.log 2>&1); echo "rocprof rc=$?"
f=$(find /tmp/prof_band -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/band_kernel_stats.csv && head -40 $O/band_kernel_stats.csv | cut -c1-170
