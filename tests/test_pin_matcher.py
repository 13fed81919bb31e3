"""The guided-matching path pinned against the reference's OWN code.

oracle/Makefile compiles /root/reference/src/orbslam/ORBmatcher.cc as it is (stand-in SLAM types + a cv::Mat subset under
oracle/ref_shims); tests/cpp/pin_matcher.cpp runs every Search* / Fuse member of that code and of this repository's
SIVO::ORBmatcher templates on identical scenes (144 cases) and requires identical return values, output vectors and
mutation logs.  Where the reference is not available (the GPU box, a fresh checkout elsewhere) the same program, built
without it, checks against tests/golden/matcher_reference.txt, which the reference build wrote."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "matcher_reference.txt")


def _make():
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "ref"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)


def _run(which):
    """which: 'cpu' (C ABI = oracle) or 'gpu' (C ABI = libsivo_hip.so)."""
    pin = os.path.join(ROOT, "oracle", "_ref", "pin_matcher_" + which)
    gold = os.path.join(ROOT, "tests", "cpp", "golden_matcher_" + which)
    if not (os.path.exists(pin) or os.path.exists(gold)):
        _make()
    exe = pin if os.path.exists(pin) else gold
    r = subprocess.run([exe, "--golden", GOLDEN], capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0 and "pin ok" in r.stdout, tail + r.stderr
    assert "144 cases, 0 failures, 144 with matches" in r.stdout, tail
    return exe


def test_oracle_and_templates_equal_the_reference_matcher():
    """CPU: reference ORBmatcher.cc == templates over the CPU oracle (pins oracle/search_oracle.c and the gather / scatter
    code of sivo_amd/api/orbslam/ORBmatcher.h); also checks that the committed golden file is what the reference computes."""
    exe = _run("cpu")
    if os.path.isdir("/root/reference"):
        assert exe.endswith(os.path.join("_ref", "pin_matcher_cpu")), "the reference is here: the live comparison must run"


def test_golden_file_covers_every_case():
    names = [l.split("|")[0] for l in open(GOLDEN) if l.strip() and not l.startswith("#")]
    assert len(names) == 144 and len(set(names)) == 144
    for routine in ("local map", "frame", "reloc", "loop", "bow kf-frame", "bow kf-kf", "initialization", "triangulation", "sim3",
                    "fuse th", "fuse sim3"):
        assert sum(n.startswith(routine) for n in names) >= 6, routine


@pytest.mark.gpu
def test_device_path_equals_the_reference_matcher():
    """GPU: the same comparison with libsivo_hip.so behind the C ABI (prebuilt oracle/_ref/pin_matcher_gpu carries the
    reference's object code to the box; without it the golden file stands in)."""
    _run("gpu")


def test_frame_view_cache_builds_one_view_per_frame():
    """CPU: ORBmatcher.h keeps a frame's matcher view per Frame / KeyFrame (FrameCache): tests/cpp/test_frame_cache.cpp counts the
    views built through the oracle-backed C ABI — one per frame however often it is searched, a new one for another id / cloned
    descriptors / changed keys, LRU eviction beyond the capacity, views in use survive a release."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_frame_cache")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "test_frame_cache"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "frame cache: ok" in r.stdout, r.stdout + r.stderr
