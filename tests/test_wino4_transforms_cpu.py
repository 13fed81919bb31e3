"""CPU: sivo_amd/csrc/wino4_transforms.hpp (the Winograd F(4x4,3x3) matrices on the points 0, 1, -1, 1/2, -2, inf) compiled for the host
with hipcc and checked against the correlation it must reproduce (tests/cpp/test_wino4_transforms.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_transforms_reproduce_the_correlation(tmp_path):
    exe = tmp_path / "test_wino4_transforms"
    subprocess.run([HIPCC, "-x", "hip", "--cuda-host-only", "-O1", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "test_wino4_transforms.cpp"),
                    "-o", str(exe)], check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
