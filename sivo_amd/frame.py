"""The perception part of the stereo Frame constructor (reference src/orbslam/Frame.cc:125-174) on device-resident images,
in the order the reference runs it — SegmentImage, two ExtractORB threads, SelectSemanticKeys, ComputeStereoMatches — with
the overlap an asynchronous device allows: the network is enqueued first, the two extractors and the matching of EVERY left
key run beside it on their own streams, and only the semantic filter and the median cull of ComputeStereoMatches wait for the
class map (sivo_stereo_match_begin / _cull).  bench.py times this object; tests/test_gpu_frame_e2e.py checks it against the
oracle pipeline."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import orb

TERRAIN = 8                      # bayesian_segnet.hpp:67-83: classes <= TERRAIN are static (Frame.cc:190)


class StereoFramePipeline:
    def __init__(self, device=0, bf=386.1448, b=386.1448 / 718.856, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7,
                 start_delay_s=0.0, orb_launch_mode=None):
        self.ex_l = orb.ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th, device=device, launch_mode=orb_launch_mode)
        self.ex_r = orb.ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th, device=device, launch_mode=orb_launch_mode)
        self.bf, self.b = bf, b
        # Frame.cc:126-129 starts two threads per frame; here three long-lived workers (starting three Python threads per frame
        # cost 0.5 ms before the network was even enqueued: tools/frame_timeline.py)
        self.pool = ThreadPoolExecutor(max_workers=3)
        self.start_delay_s = start_delay_s
        self._last = []              # the ORB work of the previous frame (futures)

    def start_orb(self, d_left, d_right):
        """ExtractORB left / right + the candidate search, Hamming matching and SAD refinement of every left key; returns the
        pending state for finish()."""
        # One extractor holds ONE image: the previous frame's extractions and its matching (which reads both pyramids) must be over before
        # the next image goes into the same two extractors.  With two frames in flight they normally ended a frame ago; when the GPU is
        # so busy that they have not (round 6: one bench run in ten died of two extractions inside one extractor), this waits.
        for f in self._last:
            f.result()
        res = {}

        def run(k, ex, im):
            if self.start_delay_s > 0:
                import time
                time.sleep(self.start_delay_s)
            res[k] = ex(im)
        fl = self.pool.submit(run, "l", self.ex_l, d_left)
        fr = self.pool.submit(run, "r", self.ex_r, d_right)

        def match():
            fl.result(); fr.result()
            (kl, dl), (kr, dr) = res["l"], res["r"]
            res["m"] = orb.stereo_match_begin(self.ex_l, self.ex_r, kl, dl, kr, dr, self.bf, self.b)
        fm = self.pool.submit(match)
        self._last = [fl, fr, fm]
        return res, [fm]

    def finish(self, pending, classes_host):
        """SelectSemanticKeys (Frame.cc:177-203: class <= TERRAIN at the truncated key position) and the median cull of
        ComputeStereoMatches over the kept keys.  Returns the frame's mvKeys / mDescriptors / mvuRight / mvDepth (kept keys
        only) and the right image's keys."""
        res, futures = pending
        for f in futures:
            f.result()
        kl, dl = res["l"]
        uR, depth, _, sad = res["m"]
        keep = classes_host[kl["y"].astype(np.int32), kl["x"].astype(np.int32)] <= TERRAIN
        orb.stereo_match_cull(keep, sad, uR, depth)
        return {"keys": kl[keep], "desc": dl[keep], "right": uR[keep], "depth": depth[keep], "n_left": len(kl), "n_right": len(res["r"][0]),
                "semantic_keys": int(keep.sum()), "stereo_matches": int((uR[keep] >= 0).sum())}

    def frame(self, segnet, d_bgr, d_left, d_right, seed, maps):
        """One frame on one device: maps = (classes u8, confidence f64, entropy f64) cuda tensors filled by the network."""
        segnet.segment_into(d_bgr, seed, maps)              # asynchronous: ~65 launches enqueued in ~0.5 ms
        pending = self.start_orb(d_left, d_right)
        classes_host = maps[0].cpu().numpy()                # 360 KB D2H; waits for this frame's class map
        if segnet.take_overflow():                          # an activation left the fp16 range of an f16x3 layer: the maps are wrong —
            segnet.segment_into(d_bgr, seed, maps)          # the same frame once more (this call runs without f16x3; the scales back off)
            classes_host = maps[0].cpu().numpy()
        return self.finish(pending, classes_host)
