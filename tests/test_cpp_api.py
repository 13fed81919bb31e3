"""The C++ classes with the reference's signatures (sivo_amd/api: SIVO::BayesianSegNet, ORBextractor,
ORBmatcher, Optimizer) through their own test program tests/cpp/test_api.cpp, which follows the
reference's tests/test_bayesian_segnet.cpp (InitializationTest :138-150, SegmentationTest :152-168)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "cpp", "test_api")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "sivo_amd", "api")], check=True)


def test_cpp_api_host_checks():
    """Exception convention (std::invalid_argument for empty paths), DescriptorDistance, ComputeThreeMaxima."""
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN, "cpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu checks ok" in r.stdout


@pytest.mark.gpu
def test_cpp_api_matches_python_binding(tmp_path, kitti_like_bgr):
    """segmentImage / operator() through the C++ classes give byte-identical results to the Python binding
    (both are thin callers of the same C ABI), whose parity with the oracle the other GPU tests establish."""
    from oracle import prototxt as oproto
    from sivo_amd import netspec, orb, weights as wts
    from sivo_amd.segnet import BayesianSegNet, BayesianSegNetParams
    if not os.path.exists(BIN):
        _build()
    T, H, W = 3, 64, 128
    text = netspec.tiny_prototxt(T, H, W)
    layers = oproto.parse(text)["layers"]
    flat = wts.pack(layers, wts.synth_weights(layers, 42))
    (tmp_path / "m.prototxt").write_text(text)
    wts.save(str(tmp_path / "w.sivow"), flat)
    frame = np.ascontiguousarray(kitti_like_bgr[:200, :400])
    with open(tmp_path / "frame.bin", "wb") as f:
        f.write(np.array(frame.shape[:2], np.int32).tobytes()); f.write(frame.tobytes())
    r = subprocess.run([BIN, "gpu", str(tmp_path / "m.prototxt"), str(tmp_path / "w.sivow"), str(tmp_path / "frame.bin"), str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "gpu checks ok" in r.stdout, r.stdout + r.stderr
    sn = BayesianSegNet(BayesianSegNetParams(str(tmp_path / "m.prototxt"), str(tmp_path / "w.sivow")))
    cls, conf, ent = sn.segment_image(frame, seed=7)
    assert np.array_equal(np.fromfile(tmp_path / "classes.bin", np.uint8).reshape(H, W), cls)
    assert np.array_equal(np.fromfile(tmp_path / "confidence.bin", np.float64).reshape(H, W), conf)
    assert np.array_equal(np.fromfile(tmp_path / "entropy.bin", np.float64).reshape(H, W), ent)
    kps, desc = orb.ORBextractor()(np.ascontiguousarray(frame[..., 0]))
    assert np.fromfile(tmp_path / "kps.bin", orb.KP_DTYPE).tobytes() == kps.tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "desc.bin", np.uint8).reshape(-1, 32), desc)
