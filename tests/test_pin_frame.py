"""The stereo Frame constructor pinned against the reference's OWN code.

`make -C oracle ref` compiles /root/reference/src/orbslam/Frame.cc and ORBextractor.cc as they are into
oracle/_ref/libref_frame.so (stand-ins: the network — it hands over the class map prepared here —, the vocabulary,
MapPoint; OpenCV primitives: oracle/orb_oracle.c).  Compared on 6 stereo scenes: the semantic keys and descriptors that
SelectSemanticKeys keeps, mvRight / mvDepth of ComputeStereoMatches bit for bit, and 400 GetFeaturesInArea queries (empty
windows, windows over the border, level filters) on the grid AssignFeaturesToGrid built.  Where the library is absent the
committed digests of tests/golden/frame_reference.json (make_frame_reference.py) stand in."""
import os

import numpy as np
import pytest

import pin_frame_common as P

HAVE_REF = os.path.exists(P.REF_LIB)


def _check(pipeline):
    golden = P.load_golden()
    n = 0
    for name, (left, right, classes, cfg) in P.scenes().items():
        pr = P.probes(*left.shape, 5)
        got = pipeline(left, right, classes, cfg, pr)
        want = golden[name]
        d = P.digest(got, P.FRAME_FIELDS)
        for f in ("n_semantic",) + P.FRAME_FIELDS:
            assert d[f] == want[f], (name, f)
        assert int((got["right"] >= 0).sum()) == want["matched"] > 20, name
        if HAVE_REF:
            ref = P.reference_frame(left, right, classes, cfg, pr)
            if P.digest(ref, P.FRAME_FIELDS) != {k: want[k] for k in ("n_semantic",) + P.FRAME_FIELDS}:
                # The reference constructor runs its two extractors on two std::threads (Frame.cc:126-129).  Once in three runs
                # of the GPU suite on the 256-core GPU box (never on the 8-core build box) the reference library returned an
                # mvuRight that differs from its own committed digest for the first scene while `got` equalled the digest; say so
                # and run the reference once more — a real regression fails both times and fails the digest comparison above
                bad = [f for f in P.FRAME_FIELDS if P.digest(ref, (f,))[f] != want[f]]
                print(f"[pin_frame] reference build deviates from its committed digest on {name}: {bad}; running it again")
                ref = P.reference_frame(left, right, classes, cfg, pr)
            assert ref["keys"].tobytes() == got["keys"].tobytes() and np.array_equal(ref["desc"], got["desc"]), name
            assert np.array_equal(ref["right"].view(np.uint32), got["right"].view(np.uint32)), name
            assert np.array_equal(ref["depth"].view(np.uint32), got["depth"].view(np.uint32)), name
            assert np.array_equal(ref["query_off"], got["query_off"]) and np.array_equal(ref["query_idx"], got["query_idx"]), name
            rd = P.digest(ref, P.FRAME_FIELDS)
            assert all(rd[f] == want[f] for f in P.FRAME_FIELDS), "stale golden: " + name
            assert tuple(ref["bounds"]) == (0.0, float(left.shape[1]), 0.0, float(left.shape[0]))
        n += 1
    assert n == len(golden) - 1 == 6


def _queries(features_in_area, pr):
    off, idx = [0], []
    for q in range(len(pr["qx"])):
        r = features_in_area(float(pr["qx"][q]), float(pr["qy"][q]), float(pr["qr"][q]), int(pr["qmin"][q]), int(pr["qmax"][q]))
        idx.extend(int(v) for v in r); off.append(len(idx))
    return np.array(off, np.int32), np.array(idx, np.int32)


def test_oracle_equals_the_reference_frame():
    from oracle import oracle as O, search as S

    def pipeline(left, right, classes, cfg, pr):
        exL, exR = O.OrbExtractor(*cfg), O.OrbExtractor(*cfg)
        kl, dl = exL(left); kr, dr = exR(right)
        keep = classes[kl["y"].astype(np.int32), kl["x"].astype(np.int32)] <= P.TERRAIN            # Frame.cc:177-203
        ks, ds = kl[keep], dl[keep]
        lv = cfg[2]
        uR, depth, _, _ = O.stereo_matches(dict(x=ks["x"], y=ks["y"], octave=ks["octave"]), ds, dict(x=kr["x"], y=kr["y"], octave=kr["octave"]), dr,
                                           exL.scale, exL.inv_scale, [exL.level(i) for i in range(lv)], [exR.level(i) for i in range(lv)],
                                           P.BF, P.BF / np.float32(P.FX))
        F = S.Frame(ks, uR, ds, (0, left.shape[1], 0, left.shape[0]), exL.scale, exL.sigma2, exL.inv_sigma2)
        off, idx = _queries(F.features_in_area, pr)
        return dict(keys=ks, desc=ds, right=uR, depth=depth, query_off=off, query_idx=idx)
    _check(pipeline)


@pytest.mark.gpu
def test_device_equals_the_reference_frame():
    from sivo_amd import matcher, orb

    def pipeline(left, right, classes, cfg, pr):
        exL, exR = orb.ORBextractor(*cfg), orb.ORBextractor(*cfg)
        kl, dl = exL(left); kr, dr = exR(right)
        keep = classes[kl["y"].astype(np.int32), kl["x"].astype(np.int32)] <= P.TERRAIN
        ks, ds = kl[keep], dl[keep]
        uR, depth, _ = orb.stereo_match(exL, exR, ks, ds, kr, dr, P.BF, P.BF / np.float32(P.FX))
        F = matcher.MatchFrame(ks, uR, ds, (0, left.shape[1], 0, left.shape[0]), exL.GetScaleFactors(), exL.GetScaleSigmaSquares(), exL.GetInverseScaleSigmaSquares())
        off, idx = _queries(F.features_in_area, pr)
        return dict(keys=ks, desc=ds, right=uR, depth=depth, query_off=off, query_idx=idx)
    _check(pipeline)
