import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def kitti_like_bgr():
    """1024x352 BGR test frame: the committed crop of the reference's tests/data/test_image.png
    (tests/golden/make_golden.py), or a seeded synthetic frame when the fixture is absent."""
    p = os.path.join(ROOT, "tests", "golden", "frame_bgr_352x1024.npy")
    if os.path.exists(p):
        return np.load(p)
    return synthetic_frame(1234)[..., None].repeat(3, axis=2)


def synthetic_frame(seed, rows=352, cols=1024):
    """SURVEY.md 8d input B: 40 random rectangles + sigma-4 Gaussian noise, u8."""
    rng = np.random.default_rng(seed)
    img = np.full((rows, cols), 90.0)
    for _ in range(40):
        x0, x1 = sorted(rng.integers(0, cols, 2)); y0, y1 = sorted(rng.integers(0, rows, 2))
        img[y0:y1 + 1, x0:x1 + 1] = rng.integers(0, 256)
    img += rng.normal(0, 4, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synthetic_stereo(seed, rows=352, cols=1024, disparity=8):
    left = synthetic_frame(seed, rows, cols + 64)
    rng = np.random.default_rng(seed + 1)
    right = left[:, disparity:disparity + cols].astype(np.float64) + rng.normal(0, 1.0, (rows, cols))
    return left[:, :cols].copy(), np.clip(np.rint(right), 0, 255).astype(np.uint8)
