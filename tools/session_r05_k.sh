#!/bin/bash
# Round 5, GPU session K: SIVO_LANES sweep with the current kernels — main line (twice, for the spread), SegNet-Basic T = 6, Standard T = 48
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_k
mkdir -p $O
cd $R
run() {  # tag lanes args...
  local tag=$1 l=$2; shift 2
  SIVO_LANES=$l timeout 200 python bench.py --configs none --no-cpu-baseline "$@" > $O/${tag}_$l.json 2> $O/${tag}_$l.err
  python - $O/${tag}_$l.json $tag $l <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "SIVO_LANES", sys.argv[3], d["value"], "fps", d["ms_per_step"], "ms; serial", d["config"].get("serial_fps"))
except Exception as e: print("parse failed", e)
P
}
for rep in a b; do for l in 2 3; do run main_$rep $l --steps 40 --warmup 5; done; done
for l in 1 2 3; do run basic $l --net basic --T 6 --no-orb --steps 40 --warmup 5; done
for l in 2 3 4; do run t48 $l --T 48 --no-orb --steps 8 --warmup 2; done
for l in 1 2; do run t6 $l --T 6 --no-orb --steps 40 --warmup 5; done
for l in 1 2; do run t3 $l --T 3 --no-orb --steps 40 --warmup 5; done
