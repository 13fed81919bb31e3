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


def test_bench_started_plainly_with_two_gpus_launches_two_ranks():
    """The driver starts the scaling runs as `python bench.py --gpus N`: started plainly, bench.py launches its own N ranks and rank 0 prints
    ONE JSON line.  One GPU here: both ranks share it and the reduction goes through gloo (SIVO_BENCH_SHARE_GPU / SIVO_BENCH_BACKEND, the
    rehearsal switches of bench.py); a small image keeps it short.  The N > 1 loop — banded prefix, all-gather, sample shards, all-reduce,
    finalize, ORB on rank 0 — is the one the 8-GPU runs execute."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SIVO_BENCH_SHARE_GPU="1", SIVO_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--T", "4", "--height", "160", "--width", "512"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["config"]["samples_per_rank"] == [2, 2]
    assert line["multi_gpu"]["allreduce_bytes"] == 15 * 160 * 512 * 4
