export TMPDIR=/tmp
mkdir -p gpurun_out/r02x
O=gpurun_out/r02x
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/probe7.log
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from sivo_amd._lib import lib, check
for name, shp in (("conv_decode1 64->64 352x1024 N=6", (6, 64, 64, 352, 1024)), ("conv_decode2 176x512", (6, 64, 64, 176, 512)), ("conv4 44x128", (6, 64, 64, 44, 128)), ("conv2 176x512 N=1", (1, 64, 64, 176, 512))):
    row = []
    for v, label in ((0, "fp32 MFMA direct"), (65536, "bf16x6")):
        ms = C.c_double()
        check(lib().sivo_debug_conv(*shp, 7, 5, v, C.byref(ms)))
        N, ci, co, H, W = shp
        row.append(f"{label} {ms.value:.3f} ms ({2.0 * 49 * ci * co * H * W * N / ms.value / 1e9:.0f} TF alg)")
    print(name, " | ".join(row), flush=True)
PY
timeout 900 python -m pytest tests/test_gpu_segnet.py -x -q -k "reference_nets or full_size or tiny_net or sharding" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
