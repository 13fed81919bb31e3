#!/bin/bash
# Round 5, GPU session F: shard ceilings with the next frame's band beside the samples; rehearsal of bench's N = 2 / 4 loop on one GPU
# (gloo, every rank on cuda:0) with bands in sequence, overlapped, and off; ORB with the newer OpenCV's Gaussian taps.
set -u
TAG=${1:-r05_f}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_orb.py -m gpu -q -x > $O/orb_tests.log 2>&1; echo "orb tests rc=$?"; tail -2 $O/orb_tests.log
timeout 600 python bench.py --configs shards --no-cpu-baseline --steps 10 > $O/bench_shards.json 2> $O/bench_shards.err; echo "bench shards rc=$?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r05_f/bench_shards.json").read())
    print("value", d["value"], d["ms_per_step"])
    for c in d.get("configs", []):
        for r in c.get("shards", []): print("  ", r)
except Exception as e:
    print("parse failed", e)
P
for mode in "overlap:1:1" "sequence:1:0" "recompute:0:0"; do
  name=${mode%%:*}; rest=${mode#*:}; b=${rest%%:*}; o=${rest#*:}
  for n in 2 4; do
    SIVO_BENCH_SHARE_GPU=1 SIVO_BENCH_BACKEND=gloo SIVO_BENCH_BANDS=$b SIVO_BENCH_BAND_OVERLAP=$o timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 8 --warmup 3 > $O/rehearsal_${name}_$n.json 2> $O/rehearsal_${name}_$n.err; echo "rehearsal $name N=$n rc=$?"
    python - $O/rehearsal_${name}_$n.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   fps", d["value"], "ms", d["ms_per_step"], {k: v for k, v in d.get("multi_gpu", {}).items() if k != "note"})
except Exception as e:
    print("   parse failed", e)
P
    tail -2 $O/rehearsal_${name}_$n.err | cut -c1-300
  done
done
