export TMPDIR=/tmp
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
B="--steps 30 --configs none --no-cpu-baseline --per-layer"
timeout 300 python bench.py $B > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print('pipe',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"
SIVO_WINO_PIPE=0 timeout 300 python bench.py $B > $O/bench0.json 2> $O/bench0.err; python -c "import json;d=json.load(open('$O/bench0.json'));print('old ',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"
grep "conv_wino_kernel" $O/bench.err | head; grep "conv_wino_kernel" $O/bench0.err | head
