"""Writes tests/golden/frame_reference.json from the reference's own Frame.cc + ORBextractor.cc (oracle/_ref/libref_frame.so,
built by `make -C oracle ref` where /root/reference exists): per stereo scene the digests of mvKeysSemantic,
mDescriptorsSemantic, mvRight, mvDepth and of 400 GetFeaturesInArea queries.  Run from the repository root:
    python tests/golden/make_frame_reference.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pin_frame_common as P   # noqa: E402

out = {"_how": "reference Frame.cc + ORBextractor.cc compiled untouched (oracle/Makefile ref); key = scene|nfeatures/nlevels; "
               "first 32 hex digits of SHA-256 per field"}
for name, (left, right, classes, cfg) in P.scenes().items():
    ref = P.reference_frame(left, right, classes, cfg, P.probes(*left.shape, 5))
    out[name] = P.digest(ref, P.FRAME_FIELDS)
    out[name]["matched"] = int((ref["right"] >= 0).sum())
    print(name, out[name]["n_semantic"], out[name]["matched"])
with open(P.GOLDEN, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
