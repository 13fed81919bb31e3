export TMPDIR=/tmp
mkdir -p gpurun_out/r02ab
O=gpurun_out/r02ab
SIVO_X6_PRODUCERS=8 timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q -k "bridged or pooling_fused or winograd_and_direct or reference_nets or fused_upsample" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -2 $O/t1.log
timeout 200 python tools/x6_probe.py 2>&1 | grep -v amdgpu > $O/probe4.log; SIVO_X6_PRODUCERS=8 timeout 200 python tools/x6_probe.py 2>&1 | grep -v amdgpu > $O/probe8.log; paste -d'\n' $O/probe4.log $O/probe8.log | grep -v "^$" | head -12
B="--steps 30 --configs none --no-cpu-baseline --per-layer"
SIVO_X6_PRODUCERS=8 timeout 300 python bench.py $B > $O/bench8.json 2> $O/bench8.err; python -c "import json;d=json.load(open('$O/bench8.json'));print('prod8',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame']['wino4_gemm_x6p_kernel'])"
timeout 300 python bench.py $B > $O/bench4.json 2> $O/bench4.err; python -c "import json;d=json.load(open('$O/bench4.json'));print('prod4',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame']['wino4_gemm_x6p_kernel'])"
