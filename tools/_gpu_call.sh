export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
timeout 300 python tools/conv_probe.py conv1_2_D conv2_1_D 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q -k "winograd_and_direct or reference_nets or fused_upsample or classifier_fused or full_size" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
B="--steps 20 --configs none --no-cpu-baseline --per-layer"
timeout 300 python bench.py $B > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print('new',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"
