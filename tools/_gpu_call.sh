export TMPDIR=/tmp
SIVO_BENCH_NO_EVENTS=1 SIVO_BENCH_TAIL_PROBE=1 timeout 300 python bench.py --steps 40 --configs none --no-cpu-baseline 2>&1 | grep "host tail"
