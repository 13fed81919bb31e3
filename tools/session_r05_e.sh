#!/bin/bash
# Round 5, GPU session E: guard (dispatch fix, classifier + 7x7 guarded), bands with one-launch pack / unpack + graph (tests, probe A/B, shards)
set -u
TAG=${1:-r05_e}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_segnet.py -m gpu -q -x -s -k "guard or overflow or reference_nets" > $O/guard_tests.log 2>&1; echo "guard tests rc=$?"; grep "^\[guard\|passed\|failed" $O/guard_tests.log | cut -c1-500 | tail -6
timeout 900 python -m pytest tests/test_gpu_prefix_bands.py -m gpu -q -x > $O/prefix_bands_tests.log 2>&1; echo "prefix band tests rc=$?"; tail -3 $O/prefix_bands_tests.log | cut -c1-300
timeout 300 python tools/band_probe.py 8 40 > $O/band_probe.log 2>&1; echo "band probe rc=$?"; grep "^\[band" $O/band_probe.log
PROBE_DIAG=1 SIVO_BAND_GRAPH=0 timeout 300 python tools/band_probe.py 8 40 > $O/band_probe_eager.log 2>&1; echo "band probe eager rc=$?"; grep "^\[band" $O/band_probe_eager.log
timeout 300 python tools/band_probe.py 4 40 > $O/band_probe4.log 2>&1; grep "^\[band" $O/band_probe4.log
timeout 600 python tools/guard_probe.py 0 boost:-16 > $O/guard_probe.log 2>&1; echo "guard probe rc=$?"; grep "^\[" $O/guard_probe.log | cut -c1-400
SIVO_NET=basic timeout 300 python - > $O/guard_basic.log 2>&1 <<'P'
import sys; sys.path.insert(0, ".")
from sivo_amd import netspec, weights as wts
from sivo_amd.segnet import BayesianSegNet
text = netspec.basic_prototxt(2, 352, 1024)
layers = netspec.parse_layers(text)
sn = BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=2)
g = sn.guard_report()
print("basic guard", g["builds"], g["predicted"], g["budget"], g["ms"])
for r in g["layers"]: print("   ", r)
P
grep -v amdgpu.ids $O/guard_basic.log | cut -c1-250
