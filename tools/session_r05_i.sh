#!/bin/bash
# Round 5, GPU session I: the round profile with rocprofv3 restricted to the frames (--selected-regions), then the co-residency hazard
# variants of the bridge (diag build: second barrier / sleep / split reads around the barrier that separates the plane's writes from its reads)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r05_a
mkdir -p $R/gpurun_out/r05_i
PROBE_ONLY="HZ exact" PROBE_SEEDS8=1 timeout 900 python tools/coresident_probe.py > $R/gpurun_out/r05_i/hazard_variants.log 2>&1; echo "hazard rc=$?"
grep "^\[\|^==" $R/gpurun_out/r05_i/hazard_variants.log | cut -c1-330
