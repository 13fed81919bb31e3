#!/bin/bash
# Round 5, GPU session K: SIVO_LANES sweep of the main line with the current kernels
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_k
mkdir -p $O
cd $R
for l in 1 2 3 4; do
  SIVO_LANES=$l timeout 200 python bench.py --configs none --no-cpu-baseline --steps 40 --warmup 5 > $O/lanes_$l.json 2> $O/lanes_$l.err
  python - $O/lanes_$l.json $l <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print("SIVO_LANES", sys.argv[2], d["value"], "fps", d["ms_per_step"], "ms; serial", d["config"].get("serial_fps"))
except Exception as e: print("parse failed", e)
P
done
