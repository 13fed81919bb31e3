"""Writes tests/golden/helpers_reference.json from the reference's own sivo_helpers.cpp (oracle/_ref/libref_helpers.so): the
stereo mutual information of 512 seeded cases as IEEE-754 hex.  Run from the repository root:
    python tests/golden/make_helpers_reference.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pin_helpers_common as P   # noqa: E402

out = {"_how": "reference sivo_helpers.cpp compiled untouched against oracle/ref_shims_eigen; cases = pin_helpers_common.cases(512, 1)",
       "stereo_mi_hex": [float(P.stereo_mutual_information(*c)[0]).hex() for c in P.cases()]}
with open(P.GOLDEN, "w") as f:
    json.dump(out, f, indent=0)
print(len(out["stereo_mi_hex"]), "cases")
