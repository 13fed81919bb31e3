"""Python mirror of SIVO::ORBextractor (reference include/orbslam/ORBextractor.h:46-123,
src/orbslam/ORBextractor.cc) over the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, lib

KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])
assert KP_DTYPE.itemsize == 28
EDGE_THRESHOLD = 19


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, device=0, gaussian="rounded", launch_mode=None):
        """gaussian: the OpenCV GaussianBlur 8U taps the reference build was linked with (sivo_orb_set_gaussian): "rounded" =
        OpenCV 3.2 - 3.4.12 / 4.0 - 4.5.0 (default), "ed" = OpenCV >= 3.4.13 / >= 4.5.1.  launch_mode (sivo_orb_set_launch_mode): bits 0 - 3 = the pyramid /
        FAST + scan + emission / blur + borders / angle + descriptor in one launch each; None = the library's default (15: four launches per image)."""
        h = C.c_void_p()
        self._L = lib()          # the library this object lives in (product, or the diagnostic build inside `with _lib.use("diag")`)
        check(self._L.sivo_orb_create(nfeatures, C.c_float(scale_factor), nlevels, ini_th_fast, min_th_fast, device, C.byref(h)))
        self._h = h
        if gaussian != "rounded":
            check(self._L.sivo_orb_set_gaussian(h, {"rounded": 0, "ed": 1}[gaussian]))
        if launch_mode is not None:
            check(self._L.sivo_orb_set_launch_mode(h, int(launch_mode)))
        self.nfeatures, self.nlevels, self._scale_factor = nfeatures, nlevels, scale_factor
        arrs = [np.empty(nlevels, np.float32) for _ in range(4)] + [np.empty(nlevels, np.int32)]
        check(self._L.sivo_orb_tables(h, *[_p(a) for a in arrs]))
        self._scale, self._inv_scale, self._sigma2, self._inv_sigma2, self.features_per_level = arrs

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._L.sivo_orb_destroy(h)
            except Exception:      # interpreter shutdown: the module globals may already be gone
                pass
            self._h = None

    # accessors of ORBextractor.h:62-84
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self._scale_factor
    def GetScaleFactors(self): return self._scale
    def GetInverseScaleFactors(self): return self._inv_scale
    def GetScaleSigmaSquares(self): return self._sigma2
    def GetInverseScaleSigmaSquares(self): return self._inv_sigma2

    def __call__(self, image, mask=None):
        """operator()(image, mask(ignored), keypoints, descriptors): image is a host uint8 (rows, cols) array
        or a cuda uint8 tensor.  Returns (keypoints[KP_DTYPE], descriptors[n,32] u8)."""
        cap = self.nfeatures * 2 + 64
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        if isinstance(image, np.ndarray):
            if image.size == 0:
                return kps[:0], desc[:0]
            assert image.dtype == np.uint8 and image.ndim == 2, "image must be CV_8UC1"
            img = image if image.strides[1] == 1 else np.ascontiguousarray(image)
            check(self._L.sivo_orb_extract(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        else:
            import torch
            assert image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2 and image.stride(1) == 1
            check(self._L.sivo_orb_extract_dev(self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0),
                                             _p(kps), _p(desc), cap, C.byref(n),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def profile(self, enable=True):
        """Bracket the extractor's kernel groups with HIP events from now on (and clear the accumulators)."""
        check(self._L.sivo_orb_profile(self._h, int(bool(enable))))

    def profile_read(self):
        """{group: mean ms per extraction}, extractions, mean keypoints — groups: pyramid, blur (+ borders), fast (+ scan + emission), angle (IC angle + rBRIEF in
        one kernel since round 6), descriptor (0: kept for the layout of the C entry point)."""
        ms = (C.c_double * 5)(); n = C.c_int32(0); k = C.c_double(0)
        check(self._L.sivo_orb_profile_read(self._h, ms, C.byref(n), C.byref(k)))
        return dict(zip(("pyramid", "blur", "fast", "angle", "descriptor"), list(ms))), n.value, k.value

    def image_pyramid(self, level, with_border=False):
        """mvImagePyramid[level] of the last extraction (interior view unless with_border)."""
        r, c = C.c_int32(), C.c_int32()
        check(self._L.sivo_orb_level(self._h, level, None, 0, C.byref(r), C.byref(c)))
        b = EDGE_THRESHOLD
        buf = np.empty((r.value + 2 * b, c.value + 2 * b), np.uint8)
        check(self._L.sivo_orb_level(self._h, level, _p(buf), buf.size, C.byref(r), C.byref(c)))
        return buf if with_border else buf[b:b + r.value, b:b + c.value]

    def candidates(self, level):
        n = C.c_int32(0)
        check(self._L.sivo_orb_candidates(self._h, level, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(self._L.sivo_orb_candidates(self._h, level, _p(out), n.value, C.byref(n)))
        return out[:n.value]


def distribute_octtree(keys, min_x, max_x, min_y, max_y, n_features):
    """DistributeOctTree (host)."""
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    out = np.zeros(len(keys) + 8, KP_DTYPE)
    n = C.c_int32(0)
    check(lib().sivo_orb_distribute(_p(keys), len(keys), min_x, max_x, min_y, max_y, n_features, _p(out), len(out), C.byref(n)))
    return out[:n.value].copy()


def extract_pair(left, right, image_left, image_right):
    """Both images of a stereo frame (cuda uint8 tensors of one shape) through two extractors at once, as Frame::Frame's two ExtractORB
    threads (Frame.cc:126-131; sivo_orb_extract_pair_dev).  Returns ((keys, descriptors) left, (keys, descriptors) right)."""
    import torch
    for im in (image_left, image_right):
        assert im.is_cuda and im.dtype == torch.uint8 and im.dim() == 2 and im.stride(1) == 1
    assert image_left.shape == image_right.shape
    capl, capr = left.nfeatures * 2 + 64, right.nfeatures * 2 + 64
    kl = np.zeros(capl, KP_DTYPE); dl = np.zeros((capl, 32), np.uint8); nl = C.c_int32(0)
    kr = np.zeros(capr, KP_DTYPE); dr = np.zeros((capr, 32), np.uint8); nr = C.c_int32(0)
    check(left._L.sivo_orb_extract_pair_dev(left._h, right._h, image_left.data_ptr(), image_right.data_ptr(), image_left.shape[0], image_left.shape[1],
                                          image_left.stride(0), image_right.stride(0), _p(kl), _p(dl), capl, C.byref(nl), _p(kr), _p(dr), capr, C.byref(nr),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return (kl[:nl.value].copy(), dl[:nl.value].copy()), (kr[:nr.value].copy(), dr[:nr.value].copy())


def stereo_match(left, right, kpL, descL, kpR, descR, bf, b):
    """Frame::ComputeStereoMatches over two extractors' resident pyramids."""
    kpL = np.ascontiguousarray(kpL, KP_DTYPE); kpR = np.ascontiguousarray(kpR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    nL = len(kpL)
    uR = np.empty(nL, np.float32); depth = np.empty(nL, np.float32); best = np.empty(nL, np.int32)
    check(left._L.sivo_stereo_match(left._h, right._h, _p(kpL), _p(descL), nL, _p(kpR), _p(descR), len(kpR),
                                  C.c_float(bf), C.c_float(b), _p(uR), _p(depth), _p(best)))
    return uR, depth, best


def stereo_match_begin(left, right, kpL, descL, kpR, descR, bf, b):
    """Everything of ComputeStereoMatches except the median cull, for ALL left keypoints (see sivo_hip.h)."""
    kpL = np.ascontiguousarray(kpL, KP_DTYPE); kpR = np.ascontiguousarray(kpR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    nL = len(kpL)
    uR = np.empty(nL, np.float32); depth = np.empty(nL, np.float32); best = np.empty(nL, np.int32); sad = np.empty(nL, np.int32)
    check(left._L.sivo_stereo_match_begin(left._h, right._h, _p(kpL), _p(descL), nL, _p(kpR), _p(descR), len(kpR),
                                        C.c_float(bf), C.c_float(b), _p(uR), _p(depth), _p(best), _p(sad)))
    return uR, depth, best, sad


def stereo_match_cull(keep, sad, uR, depth):
    """Median cull over the kept keypoints (keep: bool/uint8 mask or None); uR / depth are updated in place."""
    k = None if keep is None else np.ascontiguousarray(keep, np.uint8)
    check(lib().sivo_stereo_match_cull(len(uR), _p(k) if k is not None else None, _p(sad), _p(uR), _p(depth)))
    return uR, depth
