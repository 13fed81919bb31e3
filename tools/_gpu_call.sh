export TMPDIR=/tmp
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
B="--steps 40 --configs none --no-cpu-baseline"
for v in hi 0 1; do
  if [ $v = hi ]; then unset SIVO_ORB_PRIO; else export SIVO_ORB_PRIO=$v; fi
  SIVO_BENCH_NO_EVENTS=1 timeout 300 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; python -c "import json;d=json.load(open('$O/bench_$v.json'));print('orb prio $v',d['value'],d['ms_per_step'])"
done
unset SIVO_ORB_PRIO
SIVO_BENCH_NO_EVENTS=1 timeout 300 python bench.py $B --no-orb > $O/bench_noorb.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_noorb.json'));print('no orb',d['value'],d['ms_per_step'])"
