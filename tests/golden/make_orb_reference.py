"""Writes tests/golden/orb_reference.json from the reference's own ORBextractor.cc (oracle/_ref/libref_orb.so, built by
`make -C oracle ref` in a container that has /root/reference): per case the number of keys and SHA-256 of the
cv::KeyPoint array + descriptors and of the pyramid levels.  Run from the repository root:
    python tests/golden/make_orb_reference.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pin_orb_common as P   # noqa: E402

out = {"_how": "reference ORBextractor.cc compiled untouched against oracle/ref_shims (OpenCV primitives restated by oracle/orb_oracle.c), "
               "mode 0 (heap addresses grow with creation order); key = image|nfeatures/scaleFactor/nlevels/iniThFAST/minThFAST"}
for name, img, cfg in P.cases():
    k, d, lv, _ = P.reference_extract(img, cfg, 0)
    out[name] = P.digest(k, d, lv)
    print(name, out[name]["n"])
with open(P.GOLDEN, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
