"""ctypes front-end of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Follows the reference per-frame path:
  BayesianSegNet::segmentImage  src/bayesian_segnet/bayesian_segnet.cpp:299-318
  ORBextractor::operator()      src/orbslam/ORBextractor.cc:1019-1083
  ORBmatcher::DescriptorDistance src/orbslam/ORBmatcher.cc:1582-1596
  Frame::ComputeStereoMatches   src/orbslam/Frame.cc:444-629
  g2o stereo/mono edges         call sites src/orbslam/Optimizer.cc:651-755

PARITY UNPINNED (see the C files' headers and DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_orb_create.restype = C.c_void_p
        _LIB.orc_orb_level.restype = C.c_void_p
        _LIB.orc_cvround.argtypes = [C.c_double]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- Philox / dropout
def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, np.uint32); key = np.asarray(key, np.uint32); out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(ctr, c_u32p), _p(key, c_u32p), _p(out, c_u32p))
    return out


def dropout(x, site, sample0, seed, ratio=0.5):
    x = np.ascontiguousarray(x, np.float32).copy()
    N = x.shape[0]
    lib().orc_dropout(_p(x, c_f32p), C.c_int(N), C.c_int64(x[0].size), C.c_uint32(site), C.c_uint32(sample0),
                      C.c_uint64(seed), C.c_float(ratio))
    return x


# --------------------------------------------------------------------------- SegNet layers
def conv2d(x, w, b, pad, acc64=False):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    out = np.empty((N, Cout, H + 2 * pad - k + 1, W + 2 * pad - k + 1), np.float32)
    assert out.shape[2] == H and out.shape[3] == W, "oracle conv supports 'same' convolutions only"
    bp = _p(np.ascontiguousarray(b, np.float32), c_f32p) if b is not None else None
    lib().orc_conv2d(_p(x, c_f32p), N, Cin, H, W, _p(w, c_f32p), bp, Cout, k, pad, _p(out, c_f32p), int(acc64))
    return out


_caffe_ws = [None]


def caffe_conv2d(x, w, b, pad):
    """Caffe's CPU convolution (im2col + SGEMM per image, caffe_cpu.c): the reference-equivalent TIMING baseline of bench.py — sums in
    another order than conv2d's chain, with FMAs; held to conv2d within float round-off by tests/test_oracle_segnet.py."""
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    assert 2 * pad == k - 1, "'same' convolutions only"
    out = np.empty((N, Cout, H, W), np.float32)
    L = lib()
    L.cfc_conv_workspace.restype = C.c_int64
    need = L.cfc_conv_workspace(Cin, H, W, Cout, k)
    if _caffe_ws[0] is None or _caffe_ws[0].size < need:          # (as Caffe keeps its col_buffer_: allocated once, reused by every call)
        _caffe_ws[0] = np.empty(need, np.float32)
    ws = _caffe_ws[0]
    bp = _p(np.ascontiguousarray(b, np.float32), c_f32p) if b is not None else None
    L.cfc_conv2d(_p(x, c_f32p), N, Cin, H, W, _p(w, c_f32p), bp, Cout, k, pad, _p(out, c_f32p), _p(ws, c_f32p))
    return out


def bn_inference(x, scale, shift):
    x = np.ascontiguousarray(x, np.float32).copy()
    N, Cc, H, W = x.shape
    lib().orc_bn_inference(_p(x, c_f32p), N, Cc, C.c_int64(H * W), _p(np.ascontiguousarray(scale, np.float32), c_f32p),
                           _p(np.ascontiguousarray(shift, np.float32), c_f32p))
    return x


def relu(x):
    x = np.ascontiguousarray(x, np.float32).copy()
    lib().orc_relu(_p(x, c_f32p), C.c_int64(x.size))
    return x


def maxpool(x, k=2, s=2):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    Ho = -(-(H - k) // s) + 1; Wo = -(-(W - k) // s) + 1
    out = np.empty((N, Cc, Ho, Wo), np.float32); mask = np.empty((N, Cc, Ho, Wo), np.int32)
    lib().orc_maxpool(_p(x, c_f32p), N, Cc, H, W, k, s, _p(out, c_f32p), _p(mask, c_i32p), Ho, Wo)
    return out, mask


def unpool(x, mask, Ho, Wo):
    x = np.ascontiguousarray(x, np.float32); mask = np.ascontiguousarray(mask, np.int32)
    N, Cc, H, W = x.shape
    out = np.empty((N, Cc, Ho, Wo), np.float32)
    lib().orc_unpool(_p(x, c_f32p), _p(mask, c_i32p), N, Cc, H, W, _p(out, c_f32p), Ho, Wo)
    return out


def lrn(x, local_size, alpha, beta):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_lrn(_p(x, c_f32p), N, Cc, C.c_int64(H * W), local_size, C.c_float(alpha), C.c_float(beta), _p(out, c_f32p))
    return out


def softmax(x):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_softmax(_p(x, c_f32p), N, Cc, C.c_int64(H * W), _p(out, c_f32p))
    return out


def mc_mean(prob):
    prob = np.ascontiguousarray(prob, np.float32)
    T, Cc, H, W = prob.shape
    mean = np.empty((Cc, H, W), np.float64)
    lib().orc_mc_mean(_p(prob, c_f32p), T, Cc, C.c_int64(H * W), _p(mean, c_f64p))
    return mean


def mc_finalize(mean):
    mean = np.ascontiguousarray(mean, np.float64)
    Cc, H, W = mean.shape
    classes = np.empty((H, W), np.uint8); conf = np.empty((H, W), np.float64); ent = np.empty((H, W), np.float64)
    lib().orc_mc_finalize(_p(mean, c_f64p), Cc, C.c_int64(H * W), _p(classes, c_u8p), _p(conf, c_f64p), _p(ent, c_f64p))
    return classes, conf, ent


def mc_variance(prob, classes):
    prob = np.ascontiguousarray(prob, np.float32); classes = np.ascontiguousarray(classes, np.uint8)
    T, Cc, H, W = prob.shape
    var = np.empty((H, W), np.float64)
    lib().orc_mc_variance(_p(prob, c_f32p), T, Cc, C.c_int64(H * W), _p(classes, c_u8p), _p(var, c_f64p))
    return var


def preprocess(bgr, T, H, W):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    blob = np.empty((T, 3, H, W), np.float32)
    rc = lib().orc_preprocess(_p(bgr, c_u8p), bgr.shape[0], bgr.shape[1], T, H, W, _p(blob, c_f32p))
    return None if rc else blob


def run_net(net, weights, blob, seed, sample0=0, keep=None, acc64=False, dropout_on=True, force_masks=None, flips=None,
            expand_to=None, conv=None):
    """Execute a parsed prototxt (oracle.prototxt.parse) layer by layer, as
    caffe::Net::Forward does (bayesian_segnet.cpp:310).  `weights[name]` is the
    list of parameter blobs of layer `name` (conv: [W, b]; BN: [scale, shift]).
    Dropout sites are numbered in layer order.  Returns the blob dict (only
    names in `keep` plus the last top when keep is given).

    force_masks: {mask blob name: argmax indices (N or 1, C, Ho, Wo)} — the pooling SWITCHES of another
    implementation.  Max pooling is discontinuous: where two window elements agree to the last few ulps, two
    correct fp32 implementations may pick different ones and every logit in the receptive field of that
    switch then differs by O(1).  With the switches forced, the rest of the arithmetic is compared at the
    stated tolerance, and `flips[name] = (count, max gap)` records, per pooling layer, how many forced
    switches differ from this oracle's own choice and the largest (oracle max - forced element) among them:
    a genuine near-tie has a gap of a few ulps.

    expand_to: T — test-time shortcut.  The reference fills the T batch slots with T copies of one image
    (bayesian_segnet.cpp:174-177), so every layer upstream of the first test-time Dropout computes T identical
    results.  With expand_to the caller passes ONE slot (blob of shape (1,3,H,W)); the layers run on it and the
    bottom of the first active Dropout is repeated T times there (blobs upstream keep N = 1, pooling masks are
    broadcast where the decoder consumes them).  Same values as the T-slot run, T-1 redundant prefixes less.

    conv: another convolution routine with conv2d's (x, w, b, pad) signature — bench.py's reference-equivalent CPU baseline passes
    caffe_conv2d (im2col + SGEMM, what Caffe's CPU layer does); None = the oracle's own chain."""
    blobs = {net["input"]: blob}
    site = 0
    last = net["input"]
    for L in net["layers"]:
        t = L["type"]; bot = [blobs[b] for b in L["bottom"]]
        if t == "Convolution":
            w, b = weights[L["name"]]
            out = conv(bot[0], w, b, L["pad"]) if conv is not None else conv2d(bot[0], w, b, L["pad"], acc64)
        elif t == "BN":
            s, sh = weights[L["name"]]
            out = bn_inference(bot[0], s, sh)
        elif t == "ReLU":
            out = relu(bot[0])
        elif t == "Pooling":
            out, mask = maxpool(bot[0], L["kernel_size"], L["stride"])
            if force_masks is not None and L["top"][1] in force_masks:
                x = bot[0]
                fm = np.asarray(force_masks[L["top"][1]]).astype(np.int64)
                fm = np.broadcast_to(fm, mask.shape)
                forced = np.take_along_axis(x.reshape(x.shape[0], x.shape[1], -1), fm.reshape(fm.shape[0], fm.shape[1], -1),
                                            axis=2).reshape(out.shape)
                diff = fm != mask.astype(np.int64)
                if flips is not None:
                    flips[L["top"][1]] = (int(diff.sum()), float((out - forced)[diff].max()) if diff.any() else 0.0,
                                          float(np.abs(out[diff]).max()) if diff.any() else 0.0)
                out, mask = forced.astype(out.dtype), fm.astype(mask.dtype)
            blobs[L["top"][1]] = mask
        elif t == "Upsample":
            s = L["scale"]
            m = bot[1]
            if m.shape[0] != bot[0].shape[0]:
                m = np.ascontiguousarray(np.broadcast_to(m, bot[0].shape))
            out = unpool(bot[0], m, bot[0].shape[2] * s, bot[0].shape[3] * s)
        elif t == "Dropout":
            active = dropout_on and L["sample_weights_test"]
            x = bot[0]
            if active and expand_to is not None and x.shape[0] == 1 and expand_to > 1:
                x = np.ascontiguousarray(np.broadcast_to(x, (expand_to,) + x.shape[1:]))
            out = dropout(x, site, sample0, seed, L["dropout_ratio"]) if active else x
            site += 1
        elif t == "LRN":
            out = lrn(bot[0], L["local_size"], L["alpha"], L["beta"])
        elif t == "Softmax":
            out = softmax(bot[0])
        else:
            raise ValueError("unsupported layer type " + t)
        blobs[L["top"][0]] = out
        last = L["top"][0]
        if keep is not None:
            live = set(keep) | {last}
            # drop blobs no later layer needs
            idx = net["layers"].index(L)
            for later in net["layers"][idx + 1:]:
                live.update(later["bottom"])
            for k in list(blobs):
                if k not in live:
                    del blobs[k]
    blobs["__last__"] = blobs[last]
    return blobs


def segment(net, weights, bgr, seed, sample0=0, logits_name=None, force_masks=None, flips=None, keep=(),
            shared_prefix=False):
    """BayesianSegNet::segmentImage (bayesian_segnet.cpp:299-318) on the oracle.  shared_prefix: run the
    sample-invariant prefix once (run_net's expand_to); `keep`: further blob names to return in "blobs"."""
    T, _, H, W = net["shape"]
    blob = preprocess(bgr, 1 if shared_prefix else T, H, W)
    keep = list(keep) + ([logits_name] if logits_name else [])
    blobs = run_net(net, weights, blob, seed, sample0, keep=keep, force_masks=force_masks, flips=flips,
                    expand_to=T if shared_prefix else None)
    prob = blobs["__last__"]
    mean = mc_mean(prob)
    classes, conf, ent = mc_finalize(mean)
    return {"prob": prob, "mean": mean, "classes": classes, "confidence": conf, "entropy": ent,
            "logits": blobs.get(logits_name) if logits_name else None, "blobs": blobs}


# --------------------------------------------------------------------------- matching
def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a, c_u8p), _p(b, c_u8p))


def hamming_matrix(A, B):
    A = np.ascontiguousarray(A, np.uint8); B = np.ascontiguousarray(B, np.uint8)
    out = np.empty((A.shape[0], B.shape[0]), np.int32)
    lib().orc_hamming_matrix(_p(A, c_u8p), A.shape[0], _p(B, c_u8p), B.shape[0], _p(out, c_i32p))
    return out


def hamming_argmin2(A, B, cand_off, cand_idx):
    A = np.ascontiguousarray(A, np.uint8); B = np.ascontiguousarray(B, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32); cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    n = A.shape[0]
    bi = np.empty(n, np.int32); bd = np.empty(n, np.int32); sd = np.empty(n, np.int32)
    lib().orc_hamming_argmin2(_p(A, c_u8p), n, _p(B, c_u8p), _p(cand_off, c_i32p), _p(cand_idx, c_i32p),
                              _p(bi, c_i32p), _p(bd, c_i32p), _p(sd, c_i32p))
    return bi, bd, sd


class _Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("step", C.c_int32)]


def stereo_matches(kpL, descL, kpR, descR, scale, inv_scale, pyrL, pyrR, bf, b):
    """kp*: dict with float32 x, y and int32 octave arrays; pyr*: list of 2-D uint8 arrays (no border)."""
    nL, nR = len(kpL["x"]), len(kpR["x"])
    f = lambda a, t: np.ascontiguousarray(a, t)
    lx, ly, lo = f(kpL["x"], np.float32), f(kpL["y"], np.float32), f(kpL["octave"], np.int32)
    rx, ry, ro = f(kpR["x"], np.float32), f(kpR["y"], np.float32), f(kpR["octave"], np.int32)
    dL, dR = f(descL, np.uint8), f(descR, np.uint8)
    sc, isc = f(scale, np.float32), f(inv_scale, np.float32)
    pyrL = [f(p, np.uint8) for p in pyrL]; pyrR = [f(p, np.uint8) for p in pyrR]
    IL = (_Image * len(pyrL))(*[_Image(p.ctypes.data, p.shape[0], p.shape[1], p.strides[0]) for p in pyrL])
    IR = (_Image * len(pyrR))(*[_Image(p.ctypes.data, p.shape[0], p.shape[1], p.strides[0]) for p in pyrR])
    uR = np.empty(nL, np.float32); depth = np.empty(nL, np.float32); best = np.empty(nL, np.int32)
    kept = lib().orc_stereo_matches(nL, _p(lx, c_f32p), _p(ly, c_f32p), _p(lo, c_i32p), _p(dL, c_u8p),
                                    nR, _p(rx, c_f32p), _p(ry, c_f32p), _p(ro, c_i32p), _p(dR, c_u8p),
                                    len(pyrL), _p(sc, c_f32p), _p(isc, c_f32p), IL, IR, C.c_float(bf), C.c_float(b),
                                    _p(uR, c_f32p), _p(depth, c_f32p), _p(best, c_i32p))
    return uR, depth, best, kept


# --------------------------------------------------------------------------- BA edges
EDGE_DTYPE = np.dtype([("pose", np.int32), ("point", np.int32), ("stereo", np.int32), ("pad_", np.int32),
                       ("obs", np.float64, 3), ("inv_sigma2", np.float64)])


def ba_linearize(poses, points, edges, intr, delta_mono=np.sqrt(5.991), delta_stereo=np.sqrt(7.815)):
    poses = np.ascontiguousarray(poses, np.float64); points = np.ascontiguousarray(points, np.float64)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE); intr = np.ascontiguousarray(intr, np.float64)
    nE = edges.shape[0]
    err = np.empty((nE, 3)); Jx = np.empty((nE, 3, 3)); Jp = np.empty((nE, 3, 6))
    chi2 = np.empty(nE); rho = np.empty(nE); w = np.empty(nE); ok = np.empty(nE, np.uint8)
    lib().orc_ba_linearize(_p(poses, c_f64p), _p(points, c_f64p), edges.ctypes.data_as(C.c_void_p), C.c_int64(nE),
                           _p(intr, c_f64p), C.c_double(delta_mono), C.c_double(delta_stereo),
                           _p(err, c_f64p), _p(Jx, c_f64p), _p(Jp, c_f64p), _p(chi2, c_f64p), _p(rho, c_f64p),
                           _p(w, c_f64p), _p(ok, c_u8p))
    return {"err": err, "Jx": Jx, "Jp": Jp, "chi2": chi2, "rho": rho, "w": w, "depth_ok": ok}


def pose_optimize(pose0, points, edges, intr):
    """Optimizer::PoseOptimization (Optimizer.cc:273-491) on arrays; see ba_solve_oracle.c."""
    pose0 = np.ascontiguousarray(pose0, np.float64).reshape(12); points = np.ascontiguousarray(points, np.float64)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE); intr = np.ascontiguousarray(intr, np.float64)
    nE = edges.shape[0]
    outlier = np.zeros(nE, np.uint8); pose = np.empty(12); cov = np.zeros((6, 6)); chi2 = np.zeros(nE)
    cov_ok = C.c_int(0); iters = C.c_int(0); trials = C.c_int(0)
    f = lib().orc_pose_optimize
    f.restype = C.c_int
    n_in = f(_p(pose0, c_f64p), _p(points, c_f64p), edges.ctypes.data_as(C.c_void_p), C.c_int64(nE), _p(intr, c_f64p),
             _p(outlier, c_u8p), _p(pose, c_f64p), _p(cov, c_f64p), C.byref(cov_ok), _p(chi2, c_f64p), C.byref(iters),
             C.byref(trials))
    return {"pose": pose, "outlier": outlier, "cov": cov, "cov_ok": bool(cov_ok.value), "chi2": chi2, "inliers": n_in,
            "iterations": iters.value, "trials": trials.value}


def local_ba(poses, fixed, points, edges, intr, cov_pose=-1, stop=False):
    """Optimizer::LocalBundleAdjustment (Optimizer.cc:757-926) on arrays; returns updated copies."""
    poses = np.array(poses, np.float64).reshape(-1, 12).copy(); points = np.array(points, np.float64).reshape(-1, 3).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8); edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    intr = np.ascontiguousarray(intr, np.float64)
    nE = edges.shape[0]
    outlier = np.zeros(nE, np.uint8); cov = np.zeros((6, 6))
    cov_ok = C.c_int(0); iters = C.c_int(0); trials = C.c_int(0); stop_flag = C.c_int(1 if stop else 0)
    lib().orc_local_ba(_p(poses, c_f64p), _p(fixed, c_u8p), poses.shape[0], _p(points, c_f64p), points.shape[0],
                       edges.ctypes.data_as(C.c_void_p), C.c_int64(nE), _p(intr, c_f64p), C.byref(stop_flag),
                       _p(outlier, c_u8p), cov_pose, _p(cov, c_f64p), C.byref(cov_ok), C.byref(iters), C.byref(trials))
    return {"poses": poses, "points": points, "outlier": outlier, "cov": cov, "cov_ok": bool(cov_ok.value),
            "iterations": iters.value, "trials": trials.value}


def ba_optimize(poses, fixed, points, edges, intr, iterations, level=None, robust=None,
                delta_mono=float(np.sqrt(np.float32(5.991))), delta_stereo=float(np.sqrt(np.float32(7.815)))):
    """One g2o optimize(iterations) call (Levenberg + Schur) on arrays; returns updated copies."""
    poses = np.array(poses, np.float64).reshape(-1, 12).copy(); points = np.array(points, np.float64).reshape(-1, 3).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8); edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    intr = np.ascontiguousarray(intr, np.float64)
    nE = edges.shape[0]
    level = np.zeros(nE, np.uint8) if level is None else np.ascontiguousarray(level, np.uint8)
    robust = np.ones(nE, np.uint8) if robust is None else np.ascontiguousarray(robust, np.uint8)
    err = np.zeros((nE, 3)); hpp = np.zeros((int((fixed == 0).sum()), 6, 6)); trials = C.c_int(0)
    f = lib().orc_ba_optimize
    f.restype = C.c_int
    n = f(_p(poses, c_f64p), _p(fixed, c_u8p), poses.shape[0], _p(points, c_f64p), points.shape[0],
          edges.ctypes.data_as(C.c_void_p), C.c_int64(nE), _p(intr, c_f64p), C.c_double(delta_mono), C.c_double(delta_stereo),
          _p(level, c_u8p), _p(robust, c_u8p), iterations, _p(err, c_f64p), _p(hpp, c_f64p), C.byref(trials))
    return {"poses": poses, "points": points, "err": err, "hpp": hpp, "iterations": n, "trials": trials.value}


# --------------------------------------------------------------------------- ORB
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])


class OrbExtractor:
    """ORBextractor (ORBextractor.cc:412-475, 1019-1083) on the oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, gaussian="rounded"):
        """gaussian: which OpenCV GaussianBlur 8U kernel the descriptor image is blurred with (orb_oracle.c orc_gaussian7_taps):
        "rounded" 18 34 49 55 ... (OpenCV 3.2 - 3.4.12, 4.0 - 4.5.0) or "ed" 18 34 48 56 ... (OpenCV >= 3.4.13 / >= 4.5.1)."""
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self._h = C.c_void_p(lib().orc_orb_create(nfeatures, C.c_float(scale_factor), nlevels, ini_th, min_th))
        lib().orc_orb_set_gaussian(self._h, {"rounded": 0, "ed": 1}[gaussian])
        s = np.empty(nlevels, np.float32); i = np.empty(nlevels, np.float32)
        s2 = np.empty(nlevels, np.float32); i2 = np.empty(nlevels, np.float32)
        fpl = np.empty(nlevels, np.int32); um = np.empty(16, np.int32)
        lib().orc_orb_tables(self._h, _p(s, c_f32p), _p(i, c_f32p), _p(s2, c_f32p), _p(i2, c_f32p), _p(fpl, c_i32p), _p(um, c_i32p))
        self.scale, self.inv_scale, self.sigma2, self.inv_sigma2 = s, i, s2, i2
        self.features_per_level, self.umax = fpl, um

    def __del__(self):
        try:
            lib().orc_orb_destroy(self._h)
        except Exception:
            pass

    def __call__(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        cap = self.nfeatures * 2 + 64
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_orb_extract(self._h, _p(gray, c_u8p), gray.shape[0], gray.shape[1], gray.strides[0],
                                  kps.ctypes.data_as(C.c_void_p), _p(desc, c_u8p), cap)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l, with_border=False):
        r, c, s = C.c_int32(), C.c_int32(), C.c_int32()
        ptr = lib().orc_orb_level(self._h, l, C.byref(r), C.byref(c), C.byref(s))
        b = 19
        full = np.ctypeslib.as_array(C.cast(ptr - b * s.value - b, c_u8p), shape=(r.value + 2 * b, s.value)).copy()
        return full if with_border else full[b:b + r.value, b:b + c.value].copy()

    def candidates(self, l):
        n = lib().orc_orb_candidates(self._h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().orc_orb_candidates(self._h, l, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]


def resize_linear_u8(src, dh, dw):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src, c_u8p), src.shape[0], src.shape[1], src.strides[0], _p(dst, c_u8p), dh, dw, dw)
    return dst


def gaussian7_u8(src, variant="rounded"):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty_like(src)
    lib().orc_gaussian7_u8_v(_p(src, c_u8p), src.shape[0], src.shape[1], src.strides[0], _p(dst, c_u8p), dst.strides[0], {"rounded": 0, "ed": 1}[variant])
    return dst


def gaussian7_taps(variant="rounded"):
    k = (C.c_int * 7)()
    lib().orc_gaussian7_taps({"rounded": 0, "ed": 1}[variant], k)
    return list(k)


def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    xy = np.empty((cap, 2), np.int32); sc = np.empty(cap, np.uint8)
    n = lib().orc_fast9_16(_p(img, c_u8p), img.shape[0], img.shape[1], img.strides[0], threshold, int(nonmax),
                           _p(xy, c_i32p), _p(sc, c_u8p), cap)
    return xy[:n].copy(), sc[:n].copy()


def distribute_octtree(keys, minX, maxX, minY, maxY, N):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    out = np.zeros(len(keys) + 8, KP_DTYPE)
    n = lib().orc_distribute_octtree(keys.ctypes.data_as(C.c_void_p), len(keys), minX, maxX, minY, maxY, N,
                                     out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def bgr2gray(bgr):
    """cv::cvtColor BGR2GRAY on 8U (Tracking.cc:187-194): (B*1868 + G*9617 + R*4899 + 8192) >> 14."""
    b = bgr[..., 0].astype(np.int32); g = bgr[..., 1].astype(np.int32); r = bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


# --------------------------------------------------------------------------- entropy feature-selection gate
def stereo_mutual_information(Sx, fx, fy, bl, X, Y, Z, sigma2):
    Sx = np.ascontiguousarray(Sx, np.float64)
    f = lib().orc_stereo_mutual_information
    f.restype = C.c_double
    return f(_p(Sx, c_f64p), C.c_double(fx), C.c_double(fy), C.c_double(bl), C.c_double(X), C.c_double(Y), C.c_double(Z), C.c_double(sigma2))


def entropy_gate(kps, depth, xyz, entropy, Sx, fx, fy, bl, level_sigma2, th):
    """Tracking.cc:934-1023 over all keypoints: returns (mutual_information, entropy_reduction, accept)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float64); entropy = np.ascontiguousarray(entropy, np.float64)
    Sx = np.ascontiguousarray(Sx, np.float64); ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    n = len(kps)
    mi = np.empty(n); red = np.empty(n); acc = np.empty(n, np.uint8)
    lib().orc_entropy_gate(n, kps.ctypes.data_as(C.c_void_p), _p(depth, c_f32p), _p(xyz, c_f64p), _p(entropy, c_f64p),
                           entropy.shape[0], entropy.shape[1], _p(Sx, c_f64p), C.c_double(fx), C.c_double(fy), C.c_double(bl),
                           _p(ls2, c_f32p), C.c_double(th), _p(mi, c_f64p), _p(red, c_f64p), _p(acc, c_u8p))
    return mi, red, acc


def check_semantics(kps, depth, xyz, entropy, confidence, classes, Sx, fx, fy, bl, level_sigma2, th, th_conf):
    """LocalMapping.cc:474-538 (compute_information = true) over all keypoints: (mutual_information, entropy_reduction, detected_class)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float64); entropy = np.ascontiguousarray(entropy, np.float64)
    confidence = np.ascontiguousarray(confidence, np.float64); classes = np.ascontiguousarray(classes, np.uint8)
    Sx = np.ascontiguousarray(Sx, np.float64); ls2 = np.ascontiguousarray(level_sigma2, np.float32)
    n = len(kps)
    mi = np.empty(n); red = np.empty(n); det = np.empty(n, np.uint8)
    lib().orc_check_semantics(n, kps.ctypes.data_as(C.c_void_p), _p(depth, c_f32p), _p(xyz, c_f64p), _p(entropy, c_f64p),
                              _p(confidence, c_f64p), _p(classes, c_u8p), entropy.shape[0], entropy.shape[1], _p(Sx, c_f64p), C.c_double(fx),
                              C.c_double(fy), C.c_double(bl), _p(ls2, c_f32p), C.c_double(th), C.c_double(th_conf), _p(mi, c_f64p),
                              _p(red, c_f64p), _p(det, c_u8p))
    return mi, red, det
