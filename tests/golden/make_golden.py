#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the build container,
where /root/reference exists; the GPU box only reads the committed files).

frame_bgr_352x1024.npy   the reference's only real fixture, tests/data/test_image.png
                         (1242x375 KITTI frame), centre-cropped to the network geometry
                         by the rule of src/orbslam/System.cc:161-163 (x0 = 109, y0 = 11),
                         stored BGR as cv::imread would deliver it.
netgraph_{standard,basic}.json  layer graphs parsed from the reference prototxts; tests
                         check that sivo_amd.netspec generates the same graphs.
orb_kitti_golden.npz     oracle ORB output on that frame (keypoints + descriptors), frozen so
                         that later edits of the oracle are noticed.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    from PIL import Image
    from oracle import oracle as O, prototxt
    im = np.array(Image.open(f"{REF}/tests/data/test_image.png").convert("RGB"))
    assert im.shape == (375, 1242, 3)
    x0, y0 = 1242 // 2 - 1024 // 2, 375 // 2 - 352 // 2
    bgr = np.ascontiguousarray(im[y0:y0 + 352, x0:x0 + 1024, ::-1])
    np.save(f"{HERE}/frame_bgr_352x1024.npy", bgr)

    keys = ("name", "type", "bottom", "top", "num_output", "pad", "kernel_size", "pool", "stride", "scale",
            "dropout_ratio", "sample_weights_test", "local_size", "alpha", "beta", "bn_mode")
    for kind, path in (("standard", "standard/kitti/bayesian_segnet_kitti.prototxt"),
                       ("basic", "basic/kitti/bayesian_segnet_basic_kitti.prototxt")):
        net = prototxt.parse(open(f"{REF}/config/bayesian_segnet/{path}").read())
        graph = {"name": net["name"], "input": net["input"], "shape": net["shape"],
                 "layers": [{k: L[k] for k in keys if k in L} for L in net["layers"]]}
        json.dump(graph, open(f"{HERE}/netgraph_{kind}.json", "w"), indent=0)

    kps, desc = O.OrbExtractor()(O.bgr2gray(bgr))
    np.savez_compressed(f"{HERE}/orb_kitti_golden.npz", keypoints=kps, descriptors=desc)
    print("wrote fixtures:", os.listdir(HERE))


if __name__ == "__main__":
    main()
