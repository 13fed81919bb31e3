"""Scene for the guided-matching tests: two ORB frames of the KITTI test image (the second one shifted, as a camera
motion would), stereo coordinates for part of the keys, map points hanging on the keys of the first frame.

Built to exercise what the reference's sequential loops are sensitive to: several map points projecting into one
window (collisions on one keypoint -> the repair rounds), equal Hamming distances (descriptors drawn from a small
pool for part of the keys -> first / last-minimum tie rules), temporal points with Observations() == 0 that do not
block a keypoint, occupied slots, empty windows, points outside the image, negative depth."""
import numpy as np


def orb_frames(oracle, kitti_like_bgr, shift=(6, 2), nfeatures=1000):
    gray = oracle.bgr2gray(kitti_like_bgr)
    ex = oracle.OrbExtractor(nfeatures=nfeatures)
    k1, d1 = ex(gray)
    moved = np.roll(np.roll(gray, shift[0], axis=1), shift[1], axis=0)
    k2, d2 = ex(np.ascontiguousarray(moved))
    return ex, (k1, d1), (k2, d2), gray.shape


def tie_descriptors(desc, rng, frac=0.3, pool=6):
    """Replace a fraction of the descriptors by members of a small pool: produces many exactly equal distances."""
    d = desc.copy()
    base = desc[rng.integers(0, len(desc), pool)]
    sel = rng.random(len(desc)) < frac
    d[sel] = base[rng.integers(0, pool, int(sel.sum()))]
    return d


def stereo_right(keys, rng, frac=0.7, disparity=(3.0, 60.0)):
    ur = np.full(len(keys), -1.0, np.float32)
    sel = rng.random(len(keys)) < frac
    ur[sel] = (keys["x"][sel] - rng.uniform(*disparity, int(sel.sum()))).astype(np.float32)
    return ur


def node_lists(n1, n2, rng, n_nodes=40, cover=0.9):
    """Two CSR lists over a common set of vocabulary nodes: every key falls into at most one node."""
    a1 = np.where(rng.random(n1) < cover, rng.integers(0, n_nodes, n1), -1)
    a2 = np.where(rng.random(n2) < cover, rng.integers(0, n_nodes, n2), -1)
    off1, idx1, off2, idx2 = [0], [], [0], []
    for k in range(n_nodes):
        i1 = np.nonzero(a1 == k)[0]; i2 = np.nonzero(a2 == k)[0]
        if len(i1) == 0 or len(i2) == 0:
            continue                                  # the reference walks only nodes both feature vectors hold
        idx1 += list(rng.permutation(i1)); idx2 += list(rng.permutation(i2))
        off1.append(len(idx1)); off2.append(len(idx2))
    return (np.array(off1, np.int32), np.array(idx1, np.int32), np.array(off2, np.int32), np.array(idx2, np.int32))
