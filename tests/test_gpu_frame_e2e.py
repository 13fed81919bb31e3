"""BASELINE configs[2] end to end, as bench.py runs it: the object bench.py times (sivo_amd.frame.StereoFramePipeline.frame:
SegNet-Standard T = 12 -> class map; ORB 2000 x 8 on the left and right image; SelectSemanticKeys; ComputeStereoMatches) on
bench.py's own inputs (make_inputs), against the reference order of Frame.cc:125-174 evaluated by the CPU oracle on the same
images and fed the DEVICE's class map (the class map itself is checked against the oracle network in
tests/test_gpu_segnet_fullsize.py; feeding it here keeps the comparison of the integer stages bit-exact, as
tests/test_pin_frame.py does with a prepared class map).  Bit-exact: kept keys (all seven cv::KeyPoint fields), descriptors,
mvuRight, mvDepth; and the two counters bench.py prints in its line."""
import numpy as np
import pytest
import torch

from sivo_amd import netspec, weights as wts
from sivo_amd.frame import TERRAIN, StereoFramePipeline
from sivo_amd.segnet import BayesianSegNet

pytestmark = pytest.mark.gpu
H, W, T = 352, 1024, 12
BF, FX = 386.1448, 718.856


def test_bench_frame_equals_the_oracle_pipeline(oracle):
    from bench import make_inputs
    bgr, left, right = make_inputs(H, W)
    text = netspec.standard_prototxt(T, H, W)
    layers = netspec.parse_layers(text)
    sn = BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=T)
    fp = StereoFramePipeline()
    maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
            torch.empty((H, W), dtype=torch.float64, device="cuda"))
    d_bgr, d_left, d_right = (torch.from_numpy(a).cuda() for a in (bgr, left, right))
    for seed in (2000, 2001):              # seeds of bench.py's timed frames
        got = fp.frame(sn, d_bgr, d_left, d_right, seed, maps)
        torch.cuda.synchronize()
        classes = maps[0].cpu().numpy()
        assert sn.gemm_status()[1] == 0

        # Frame.cc:125-174 on the oracle
        exL, exR = oracle.OrbExtractor(), oracle.OrbExtractor()
        kl, dl = exL(left); kr, dr = exR(right)
        keep = classes[kl["y"].astype(np.int32), kl["x"].astype(np.int32)] <= TERRAIN                 # Frame.cc:177-203
        ks, ds = kl[keep], dl[keep]
        uR, depth, _, _ = oracle.stereo_matches(dict(x=ks["x"], y=ks["y"], octave=ks["octave"]), ds,
                                                dict(x=kr["x"], y=kr["y"], octave=kr["octave"]), dr, exL.scale, exL.inv_scale,
                                                [exL.level(i) for i in range(8)], [exR.level(i) for i in range(8)],
                                                BF, BF / np.float32(FX))
        assert got["n_left"] == len(kl) and got["n_right"] == len(kr)
        assert got["keys"].tobytes() == ks.tobytes()
        assert np.array_equal(got["desc"], ds)
        assert np.array_equal(got["right"].view(np.uint32), uR.view(np.uint32))
        assert np.array_equal(got["depth"].view(np.uint32), depth.view(np.uint32))
        assert got["semantic_keys"] == int(keep.sum()) and got["stereo_matches"] == int((uR >= 0).sum())
        assert got["semantic_keys"] > 300 and got["stereo_matches"] > 100
        print(f"[e2e seed {seed}] left {len(kl)} right {len(kr)} semantic {got['semantic_keys']} stereo {got['stereo_matches']}")
