export TMPDIR=/tmp
mkdir -p gpurun_out/r02v
O=gpurun_out/r02v
timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q -k "persistent or winograd_and_direct or full_size" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
B="--steps 30 --configs none --no-cpu-baseline"
timeout 300 python bench.py $B > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print('new',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"
