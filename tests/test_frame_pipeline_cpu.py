"""Host logic of sivo_amd.frame.StereoFramePipeline without a GPU: the ordering rule an extractor needs — ONE image at a time.

Round 6: with two frames in flight, frame k + 1's left extraction was handed to a free pool worker while frame k's was still running
inside the same extractor whenever the GPU was busy enough (one bench run in ten died; before that the two frames' candidates would
have mixed silently).  The pipeline now waits for the previous frame's ORB futures before it starts the next ones (and the C handle has
a mutex).  Here the extractors and the matcher are stand-ins that record overlap."""
import threading
import time

import numpy as np

from sivo_amd import frame as frame_mod


class _FakeExtractor:
    def __init__(self, log, name, delay):
        self.log, self.name, self.delay, self.busy = log, name, delay, threading.Lock()

    def __call__(self, image):
        assert self.busy.acquire(blocking=False), f"two extractions inside extractor {self.name} at once"
        try:
            self.log.append(("start", self.name, int(image)))
            time.sleep(self.delay)
            self.log.append(("end", self.name, int(image)))
            kp = np.zeros(3, dtype=[("x", np.float32), ("y", np.float32)])
            return kp, np.full((3, 32), int(image), np.uint8)
        finally:
            self.busy.release()


def test_next_frame_waits_for_the_previous_frames_orb_work(monkeypatch):
    log = []
    monkeypatch.setattr(frame_mod.orb, "ORBextractor", lambda *a, **k: None)
    fp = frame_mod.StereoFramePipeline()
    fp.ex_l, fp.ex_r = _FakeExtractor(log, "l", 0.05), _FakeExtractor(log, "r", 0.01)      # left much slower than right: the case that broke
    in_match = threading.Lock()

    def fake_match(ex_l, ex_r, kl, dl, kr, dr, bf, b):
        assert in_match.acquire(blocking=False)
        try:
            assert not ex_l.busy.locked() and not ex_r.busy.locked(), "matching reads both pyramids: no extraction may run beside it"
            assert int(dl[0, 0]) == int(dr[0, 0]), "left and right results of different frames met in one match"
            log.append(("match", int(dl[0, 0])))
            time.sleep(0.02)
            return np.zeros(3, np.float32), np.zeros(3, np.float32), None, np.zeros(3, np.int32)
        finally:
            in_match.release()
    monkeypatch.setattr(frame_mod.orb, "stereo_match_begin", fake_match)
    pend = [fp.start_orb(k, k) for k in range(4)]                 # four frames issued back to back: far more than the pool's three workers
    for p in pend:
        for f in p[1]:
            f.result()
    assert [e for e in log if e[0] == "match"] == [("match", k) for k in range(4)]
    # per extractor: start / end strictly alternate and the frames come in order
    for name in "lr":
        ev = [e for e in log if e[0] in ("start", "end") and e[1] == name]
        assert ev == [(kind, name, k) for k in range(4) for kind in ("start", "end")]
    # frame k + 1 starts nothing before frame k's match is over
    order = [e for e in log if e[0] in ("start", "match")]
    for k in range(3):
        assert order.index(("match", k)) < min(order.index(("start", "l", k + 1)), order.index(("start", "r", k + 1)))
    fp.pool.shutdown()
