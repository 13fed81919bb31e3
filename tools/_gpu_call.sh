export TMPDIR=/tmp
mkdir -p gpurun_out/r02w
O=gpurun_out/r02w
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_segnet.py -x -q -k "persistent or winograd_and_direct or full_size or reference_nets" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
timeout 300 python tools/conv_probe.py conv1_2_D conv2_1_D 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
for pass in "f FETCH_SIZE" "w WRITE_SIZE"; do
  set -- $pass; p=$1; shift
  rm -rf /tmp/pmc_$p
  (cd /tmp && SIVO_LANES=1 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-orb --configs none > /dev/null 2>&1)
done
python tools/pmc_summary.py $O/pmc_traffic.json /tmp/pmc_f /tmp/pmc_w > /dev/null; python - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in d.items():
    if isinstance(v,dict) and v.get("bytes",0)>5e7: print(k[:60], v.get("dispatches"), "MB", round(v["bytes"]/1e6,1), "fetch MB", round(2*v.get("FETCH_SIZE",0)*1024/1e6,1))
PY
