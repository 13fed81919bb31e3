export TMPDIR=/tmp
mkdir -p gpurun_out/r02y
O=gpurun_out/r02y
timeout 1200 python -m pytest tests/test_gpu_segnet_fullsize.py -x -q -s -k "basic" > $O/t1.log 2>&1; echo "t1 rc=$?"; grep "^\[" $O/t1.log | head -12; tail -2 $O/t1.log
timeout 300 python bench.py --net basic --T 6 --steps 20 --configs none --no-cpu-baseline --no-orb --per-layer > $O/bench_basic.json 2> $O/bench_basic.err; python -c "import json;d=json.load(open('$O/bench_basic.json'));print('basic',d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['kernels_ms_per_frame'])"; grep "N=" $O/bench_basic.err | awk '{print $1,$2,$3,$4,$5}' | head -20
