export TMPDIR=/tmp
mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
B="--steps 30 --configs none --no-cpu-baseline"
for l in 3 1; do SIVO_LANES=$l timeout 300 python bench.py $B > $O/bench_l$l.json 2> $O/bench_l$l.err; python -c "import json;d=json.load(open('$O/bench_l$l.json'));print('lanes $l',d['value'],d['ms_per_step'],d['config']['semantic_keys'],d['config']['stereo_matches'])"; done
