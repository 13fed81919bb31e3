export TMPDIR=/tmp
mkdir -p gpurun_out/r02ad
O=gpurun_out/r02ad
timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -2 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_segnet_fullsize.py -x -q -k "three_lanes or standard-12-kitti-7" > $O/t2.log 2>&1; echo "t2 rc=$?"; tail -2 $O/t2.log
