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
    # BayesianSegNetParams::devices: the file constructor of the multi-device handle (one rank here)
    snm = BayesianSegNet(BayesianSegNetParams(str(tmp_path / "m.prototxt"), str(tmp_path / "w.sivow")), devices=[0])
    cls_m, conf_m, _ = snm.segment_image(frame, seed=7)
    np.testing.assert_allclose(conf_m, conf, atol=2e-7, rtol=0)
    assert (cls_m == cls).mean() > 0.999
    kps, desc = orb.ORBextractor()(np.ascontiguousarray(frame[..., 0]))
    assert np.fromfile(tmp_path / "kps.bin", orb.KP_DTYPE).tobytes() == kps.tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "desc.bin", np.uint8).reshape(-1, 32), desc)
    # SIVO::Frame: the same stereo frame rebuilt through the Python binding (centre crop, right = left shifted by 8 px,
    # extractors (500, 1.2, 1, 20, 7), class <= TERRAIN filter, stereo matching on the kept keys)
    g = frame[..., 0]
    gR = np.concatenate([g[:, 8:], np.repeat(g[:, -1:], 8, axis=1)], axis=1)
    y0, x0 = (frame.shape[0] - H) // 2, (frame.shape[1] - W) // 2
    cl, cr = np.ascontiguousarray(g[y0:y0 + H, x0:x0 + W]), np.ascontiguousarray(gR[y0:y0 + H, x0:x0 + W])
    cls2, _, _ = sn.segment_image(np.ascontiguousarray(frame[y0:y0 + H, x0:x0 + W]), seed=7)
    assert np.array_equal(np.fromfile(tmp_path / "frame_classes.bin", np.uint8).reshape(H, W), cls2)
    ex_l, ex_r = orb.ORBextractor(500, 1.2, 1, 20, 7), orb.ORBextractor(500, 1.2, 1, 20, 7)
    kl, dl = ex_l(cl); kr, dr = ex_r(cr)
    keep = cls2[kl["y"].astype(np.int32), kl["x"].astype(np.int32)] <= 8
    uR, depth, _ = orb.stereo_match(ex_l, ex_r, kl[keep], dl[keep], kr, dr, 386.1448, 386.1448 / 718.856)
    assert np.fromfile(tmp_path / "frame_keys.bin", orb.KP_DTYPE).tobytes() == kl[keep].tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "frame_right.bin", np.float32).view(np.uint32), uR.view(np.uint32))
    assert np.array_equal(np.fromfile(tmp_path / "frame_depth.bin", np.float32).view(np.uint32), depth.view(np.uint32))
    assert keep.sum() > 20
