"""Python mirror of SIVO::BayesianSegNet (reference include/bayesian_segnet/bayesian_segnet.hpp:85-170,
src/bayesian_segnet/bayesian_segnet.cpp) over the C ABI.  torch is used for device
memory, streams and torch.distributed only."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, dbg, lib

# bayesian_segnet.hpp:67-83
CLASSES = ["ROAD", "SIDEWALK", "BUILDING", "WALL", "POLE", "TRAFFIC_LIGHT", "TRAFFIC_SIGN", "VEGETATION",
           "TERRAIN", "SKY", "PERSON", "CAR", "COMMERCIAL_VEHICLE", "BIKE"]
VOID = 255


class BayesianSegNetParams:
    """bayesian_segnet.hpp:85-105, plus how the handle runs (SivoSegnetOptions of include/sivo_hip.h; None = the library's default)."""

    def __init__(self, model_file="", weights_file="", use_gpu=True, **options):
        self.model_file, self.weights_file, self.use_gpu = model_file, weights_file, use_gpu
        self.options = dict(options)


_GEMM = {"f16x3": 0, "": 0, "x6": 1, "bf16x6": 1, "f32": 2, "fp32": 2}


def segnet_options(lanes=None, gemm=None, direct_f16x3=None, packed_activations=None, conv7=None, wino4_workspace_mb=None, debug_sync=None):
    """SivoSegnetOptions from keyword arguments.  The LIBRARY reads no environment variable; for the tests, tools and bench.py this
    Python wrapper takes an argument left at None from the environment of the process, under the names the library used to read until
    round 5: SIVO_LANES=1..4, SIVO_GEMM=x6|f32, SIVO_D3=0, SIVO_D3_PK=0, SIVO_CONV7=f32, SIVO_WINO4_MB, SIVO_DEBUG_SYNC=1."""
    import os
    env = os.environ.get
    o = _lib.SegnetOptions()
    o.struct_size = C.sizeof(_lib.SegnetOptions)
    o.lanes = int(lanes if lanes is not None else env("SIVO_LANES", "0"))
    g = gemm if gemm is not None else env("SIVO_GEMM", "")
    o.gemm = _GEMM[g] if isinstance(g, str) else int(g)
    o.no_direct_f16x3 = int(not direct_f16x3) if direct_f16x3 is not None else int(env("SIVO_D3", "1") == "0")
    o.no_packed_activations = int(not packed_activations) if packed_activations is not None else int(env("SIVO_D3_PK", "1") == "0")
    o.conv7_fp32 = int(conv7 in ("f32", "fp32")) if conv7 is not None else int(env("SIVO_CONV7", "") == "f32")
    o.wino4_workspace_mb = int(wino4_workspace_mb if wino4_workspace_mb is not None else env("SIVO_WINO4_MB", "0"))
    o.debug_sync = int(bool(debug_sync)) if debug_sync is not None else int(env("SIVO_DEBUG_SYNC") is not None)
    return o


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BayesianSegNet:
    """Constructor from a params object (files) like the reference, or from prototxt text + flat weights."""

    def __init__(self, params=None, prototxt=None, weights=None, T=0, device=0, devices=None, **options):
        """devices: list of HIP device ids -> the T samples of a frame are spread over them inside the handle
        (sivo_segnet_create_multi: RCCL reduce-scatter / all-gather); only segment_image works on such a handle.
        options: segnet_options() keywords (lanes, gemm, direct_f16x3, packed_activations, conv7, wino4_workspace_mb, debug_sync)."""
        h = C.c_void_p()
        opt = segnet_options(**dict(getattr(params, "options", {}) or {}, **options))
        self.options = opt
        po = C.byref(opt)
        self._L = lib()          # the library this object lives in (product, or the diagnostic build inside `with _lib.use("diag")`)
        self.devices = list(devices) if devices is not None else None
        if params is not None:
            if not params.model_file:
                raise ValueError("model_file (.prototxt file) is empty!")       # bayesian_segnet.cpp:80-89
            if not params.weights_file:
                raise ValueError("weights_file (.caffemodel file) is empty!")
            if not params.use_gpu:
                raise _lib.SivoError(_lib.ERR_UNSUPPORTED, "use_gpu=false: this library has no CPU path")
            if self.devices is not None:
                ids = (C.c_int32 * len(self.devices))(*self.devices)
                rc = self._L.sivo_segnet_create_multi_from_files_opts(params.model_file.encode(), params.weights_file.encode(), T,
                                                                    ids, len(self.devices), po, C.byref(h))
                device = self.devices[0] if self.devices else 0
            else:
                rc = self._L.sivo_segnet_create_from_files_opts(params.model_file.encode(), params.weights_file.encode(), T,
                                                              device, po, C.byref(h))
        elif self.devices is not None:
            text = prototxt.encode() if isinstance(prototxt, str) else (prototxt or b"")
            w = np.ascontiguousarray(weights if weights is not None else np.zeros(0), np.float32)
            ids = (C.c_int32 * len(self.devices))(*self.devices)
            rc = self._L.sivo_segnet_create_multi_opts(text, len(text), T, w.ctypes.data_as(C.c_void_p), w.size, ids, len(self.devices), po, C.byref(h))
            device = self.devices[0] if self.devices else 0
        else:
            text = prototxt.encode() if isinstance(prototxt, str) else (prototxt or b"")
            w = np.ascontiguousarray(weights if weights is not None else np.zeros(0), np.float32)
            rc = self._L.sivo_segnet_create_opts(text, len(text), T, w.ctypes.data_as(C.c_void_p), w.size, device, po, C.byref(h))
        if rc == _lib.ERR_INVALID_ARGUMENT:
            raise ValueError(self._L.sivo_last_error().decode())                     # std::invalid_argument
        check(rc)
        self._h = h
        self.device = device
        T_, C_, H_, W_, K_ = (C.c_int32() for _ in range(5))
        check(self._L.sivo_segnet_shape(h, T_, C_, H_, W_, K_))
        self.T, self.C, self.H, self.W, self.classes = T_.value, C_.value, H_.value, W_.value, K_.value
        a, b = C.c_double(), C.c_double()
        check(self._L.sivo_segnet_flops(h, C.byref(a), C.byref(b)))
        self.flops_shared, self.flops_per_sample = a.value, b.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._L.sivo_segnet_destroy(h)
            except Exception:      # interpreter shutdown: the module globals may already be gone
                pass
            self._h = None

    def get_input_geometry(self):
        """getInputGeometry(): (width, height) like cv::Size."""
        return (self.W, self.H)

    # -- device-resident path -------------------------------------------------------------
    def forward(self, d_bgr, seed, n_samples=None, sample0=0, want_logits=False, want_prob=False):
        """d_bgr: cuda uint8 tensor (H, W, 3) BGR.  Returns (prob_sum[classes,H,W] f32, logits|None, prob|None)."""
        n = self.T if n_samples is None else n_samples
        dev = d_bgr.device
        assert d_bgr.is_cuda and d_bgr.dtype == torch.uint8 and d_bgr.is_contiguous() and tuple(d_bgr.shape) == (self.H, self.W, 3)
        prob_sum = torch.empty((self.classes, self.H, self.W), dtype=torch.float32, device=dev)
        logits = torch.empty((n, self.classes, self.H, self.W), dtype=torch.float32, device=dev) if want_logits else None
        prob = torch.empty((n, self.classes, self.H, self.W), dtype=torch.float32, device=dev) if want_prob else None
        check(self._L.sivo_segnet_forward_dev(self._h, d_bgr.data_ptr(), n, sample0, C.c_uint64(seed), prob_sum.data_ptr(),
                                            logits.data_ptr() if want_logits else None,
                                            prob.data_ptr() if want_prob else None, _stream()))
        return prob_sum, logits, prob

    def forward_into(self, d_bgr, seed, prob_sum, n_samples=None, sample0=0):
        n = self.T if n_samples is None else n_samples
        check(self._L.sivo_segnet_forward_dev(self._h, d_bgr.data_ptr(), n, sample0, C.c_uint64(seed), prob_sum.data_ptr(),
                                            None, None, _stream()))

    def finalize(self, prob_sum, t_total=None, out=None):
        """classes (u8), confidence (f64), entropy (f64) maps from the probability sum."""
        t_total = self.T if t_total is None else t_total
        dev = prob_sum.device
        if out is None:
            out = (torch.empty((self.H, self.W), dtype=torch.uint8, device=dev),
                   torch.empty((self.H, self.W), dtype=torch.float64, device=dev),
                   torch.empty((self.H, self.W), dtype=torch.float64, device=dev))
        check(self._L.sivo_mc_finalize_dev(prob_sum.data_ptr(), self.classes, self.H * self.W, t_total,
                                         out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream()))
        return out

    # -- the sample-invariant prefix as row bands over ranks (include/sivo_hip.h) ---------------
    def prefix_bands(self, world):
        """Plan the split for `world` ranks: dict(slot_bytes, rows=[world + 1] rows of the prefix output per rank,
        input_rows=[(first, end)] image rows each rank's band reads)."""
        sb = C.c_size_t()
        rows = (C.c_int32 * (world + 1))()
        inp = (C.c_int32 * (2 * world))()
        check(self._L.sivo_segnet_prefix_bands(self._h, world, C.byref(sb), rows, inp))
        return dict(slot_bytes=sb.value, rows=list(rows), input_rows=[(inp[2 * r], inp[2 * r + 1]) for r in range(world)])

    def prefix_band_into(self, d_bgr, rank, world, slot):
        """This rank's band of the prefix -> slot (cuda uint8 tensor of slot_bytes), on the current stream."""
        assert slot.is_cuda and slot.dtype == torch.uint8 and slot.is_contiguous()
        check(self._L.sivo_segnet_prefix_band_dev(self._h, d_bgr.data_ptr(), rank, world, slot.data_ptr(), _stream()))

    def forward_banded_into(self, slots, world, seed, prob_sum, n_samples=None, sample0=0, logits=None):
        """The per-sample part of the forward on the gathered prefix (slots: cuda uint8 (world, slot_bytes), rank order)."""
        n = self.T if n_samples is None else n_samples
        assert slots.is_cuda and slots.dtype == torch.uint8 and slots.is_contiguous()
        check(self._L.sivo_segnet_forward_banded_dev(self._h, slots.data_ptr(), world, n, sample0, C.c_uint64(seed), prob_sum.data_ptr(),
                                                     logits.data_ptr() if logits is not None else None, _stream()))

    def segment_into(self, d_bgr, seed, out, logits=None):
        """segmentImage on device-resident data: out = (classes u8, confidence f64, entropy f64) cuda tensors (H, W).
        logits: optional cuda f32 tensor (T, classes, H, W) that receives the logits the maps were computed from."""
        if logits is not None:
            assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
            assert tuple(logits.shape) == (self.T, self.classes, self.H, self.W)
            check(self._L.sivo_segnet_segment_logits_dev(self._h, d_bgr.data_ptr(), C.c_uint64(seed), out[0].data_ptr(),
                                                       out[1].data_ptr(), out[2].data_ptr(), logits.data_ptr(), _stream()))
            return out
        check(self._L.sivo_segnet_segment_dev(self._h, d_bgr.data_ptr(), C.c_uint64(seed), out[0].data_ptr(), out[1].data_ptr(),
                                            out[2].data_ptr(), _stream()))
        return out

    # -- host path == segmentImage --------------------------------------------------------
    def segment_image(self, image_bgr, seed=0):
        """segmentImage(const cv::Mat&, MatXu&, MatXd&, MatXd&) (bayesian_segnet.cpp:299-318)."""
        img = np.ascontiguousarray(image_bgr, np.uint8)
        classes = np.empty((self.H, self.W), np.uint8)
        conf = np.empty((self.H, self.W), np.float64)
        ent = np.empty((self.H, self.W), np.float64)
        check(self._L.sivo_segnet_segment(self._h, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1],
                                        C.c_uint64(seed), classes.ctypes.data_as(C.c_void_p),
                                        conf.ctypes.data_as(C.c_void_p), ent.ctypes.data_as(C.c_void_p)))
        return classes, conf, ent

    def profile(self, enable=True, reset=False, mfma_only=False, keep_lanes=False):
        """Bracket every kernel of the forward with HIP events on its launch stream (mfma_only: just the convolution
        kernels / the F(4x4,3x3) GEMM — a handful of events per forward, for use inside a timed run; keep_lanes (with mfma_only): the
        profiled forward keeps its sample groups on their streams, times are summed over the lanes)."""
        mode = ((5 if reset else 6) if keep_lanes else (3 if reset else 4)) if (enable and mfma_only) else 2 if (enable and reset) else int(bool(enable))
        check(self._L.sivo_segnet_profile(self._h, mode))

    def profile_read(self):
        """List of dicts: layer, kernel, samples, launches (forward passes), kernel_launches, flops_per_sample,
        bytes_per_sample, ms_total."""
        n = C.c_int32(0)
        check(self._L.sivo_segnet_profile_read(self._h, None, 0, C.byref(n)))
        arr = (_lib.OpProfile * n.value)()
        check(self._L.sivo_segnet_profile_read(self._h, arr, n.value, C.byref(n)))
        return [dict(layer=a.layer.decode(), kernel=a.kernel.decode(), samples=a.samples, launches=a.launches,
                     flops_per_sample=a.flops_per_sample, bytes_per_sample=a.bytes_per_sample, ms_total=a.ms_total,
                     kernel_launches=a.kernel_launches)
                for a in arr]

    def take_overflow(self):
        """True when a frame issued through forward_into / segment_into since the last call left the fp16 range of the f16x3
        layers (sivo_segnet_take_overflow): that frame's maps are wrong — issue the same call again, it runs without f16x3.
        Call it once the frame's results were synchronised with (it reads one pinned host word)."""
        ov = C.c_int32()
        check(self._L.sivo_segnet_take_overflow(self._h, C.byref(ov)))
        return bool(ov.value)

    def gemm_status(self):
        """(mode, overflow_frames, layers): mode 2 = f16x3 in the matrix-core layers (default), 1 = bf16x6, 0 = fp32 MFMA /
        none; overflow_frames = frames in which a value left the fp16 range (each lowered the scales by 2^2, the fourth switched
        the handle to bf16x6); layers = [(name, largest |V| of the calibration, V scale, U scale)]."""
        class _Row(C.Structure):
            _fields_ = [("layer", C.c_char * 48), ("vmax", C.c_float), ("vscale", C.c_float), ("uscale", C.c_float)]
        mode, ov, n = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._L.sivo_segnet_gemm_status(self._h, C.byref(mode), C.byref(ov), None, 0, C.byref(n)))
        rows = (_Row * max(n.value, 1))()
        check(self._L.sivo_segnet_gemm_status(self._h, C.byref(mode), C.byref(ov), rows, n.value, C.byref(n)))
        return mode.value, ov.value, [(r.layer.decode(), r.vmax, r.vscale, r.uscale) for r in rows[:n.value]]

    def guard_report(self):
        """The load-time accuracy guard (sivo_segnet_guard_report): dict(budget, predicted, logit_max, ms, builds, layers=[dict(layer, kernel,
        rel_err, rel_rms, ref_max, first_rel_err, level)]) — see include/sivo_hip.h."""
        class _Row(C.Structure):
            _fields_ = [("layer", C.c_char * 48), ("kernel", C.c_char * 24), ("rel_err", C.c_float), ("rel_rms", C.c_float),
                        ("ref_max", C.c_float), ("first_rel_err", C.c_float), ("level", C.c_int32)]
        n, builds = C.c_int32(), C.c_int32()
        budget, pred, lmax, ms = C.c_float(), C.c_float(), C.c_float(), C.c_double()
        check(self._L.sivo_segnet_guard_report(self._h, None, 0, C.byref(n), C.byref(budget), C.byref(pred), C.byref(lmax), C.byref(ms), C.byref(builds)))
        rows = (_Row * max(n.value, 1))()
        check(self._L.sivo_segnet_guard_report(self._h, rows, n.value, C.byref(n), None, None, None, None, None))
        return dict(budget=budget.value, predicted=pred.value, logit_max=lmax.value, ms=ms.value, builds=builds.value,
                    layers=[dict(layer=r.layer.decode(), kernel=r.kernel.decode(), rel_err=r.rel_err, rel_rms=r.rel_rms, ref_max=r.ref_max,
                                 first_rel_err=r.first_rel_err, level=r.level) for r in rows[:n.value]])

    def blob(self, name):
        shape = (C.c_int32 * 4)()
        check(self._L.sivo_segnet_blob(self._h, name.encode(), None, 0, shape))
        out = np.empty(tuple(shape), np.float32)
        check(self._L.sivo_segnet_blob(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.size, shape))
        return out


def h3_gemm(V, U, P, vscale=None, iters=0):
    """The f16x3 GEMM of the F(4x4,3x3) path alone (sivo_debug_h3_gemm): V (36, C, Pp) and U (36, C, Kp) fp32 numpy arrays,
    Pp = P rounded up to 128.  Returns (M (36, Kp, Pp) fp32, mean launch ms or None)."""
    V = np.ascontiguousarray(V, np.float32); U = np.ascontiguousarray(U, np.float32)
    Cc, Pp, Kp = V.shape[1], V.shape[2], U.shape[2]
    assert V.shape[0] == 36 and U.shape[:2] == (36, Cc) and Pp == (P + 127) // 128 * 128
    if vscale is None:
        vscale = float(2.0 ** (8 - np.frexp(float(np.abs(V).max()))[1]))
    M = np.empty((36, Kp, Pp), np.float32)
    ms = C.c_double(0)
    check(dbg().sivo_debug_h3_gemm(Cc, Kp, P, V.ctypes.data_as(C.c_void_p), U.ctypes.data_as(C.c_void_p), C.c_float(vscale),
                                   M.ctypes.data_as(C.c_void_p), iters, C.byref(ms)))
    return M, (ms.value if iters else None)


def conv3_h3(x, weight, scale, shift, relu=True, mask=None, vscale=None, iters=0):
    """The direct f16x3 3x3 convolution alone (sivo_debug_conv3_h3_dev).  x: cuda fp32 (N, Cin, H, W), or with `mask` (cuda
    u8 window codes of the same shape) the POOLED tensor (N, Cin, H/2, W/2) the layer reads through; weight (Cout, Cin, 3, 3),
    scale / shift (Cout) numpy.  Returns (out cuda (N, Cout, H, W), mean launch ms or None, overflowed)."""
    x = x.contiguous()
    N, Cin, h, w = x.shape
    H, W = (2 * h, 2 * w) if mask is not None else (h, w)
    weight = np.ascontiguousarray(weight, np.float32)
    scale = np.ascontiguousarray(scale, np.float32); shift = np.ascontiguousarray(shift, np.float32)
    Cout = weight.shape[0]
    assert weight.shape == (Cout, Cin, 3, 3) and scale.shape == (Cout,) and shift.shape == (Cout,)
    if mask is not None:
        mask = mask.contiguous()
        assert mask.shape == x.shape and mask.dtype == torch.uint8
    if vscale is None:
        vscale = float(2.0 ** (8 - np.frexp(float(x.abs().max()))[1]))
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    ms = C.c_double(0)
    ov = C.c_int(0)
    check(dbg().sivo_debug_conv3_h3_dev(N, Cin, Cout, H, W, x.data_ptr(), mask.data_ptr() if mask is not None else None,
                                        weight.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p),
                                        shift.ctypes.data_as(C.c_void_p), int(relu), C.c_float(vscale), out.data_ptr(), iters,
                                        C.byref(ms), C.byref(ov)))
    return out, (ms.value if iters else None), bool(ov.value)


def conv3_h3_pk(x, weight, scale, shift, relu=True, mask=None, vscale=None, out_vscale=None, pk_in=True, pk_out=False, extra_pad=False,
                iters=0):
    """The direct f16x3 convolution with packed activations (sivo_debug_conv3_h3_pk_dev): as conv3_h3, but the input is packed
    on the device first (pk_in; through an Upsample: the pooled tensor + the window codes re-laid per octet) and / or the output
    is written packed with out_vscale and unpacked afterwards (pk_out).  Returns (out fp32, ms or None, overflowed,
    border_dirty): border_dirty = the zero border of the packed output was written to."""
    x = x.contiguous()
    N, Cin, h, w = x.shape
    H, W = (2 * h, 2 * w) if mask is not None else (h, w)
    weight = np.ascontiguousarray(weight, np.float32)
    scale = np.ascontiguousarray(scale, np.float32); shift = np.ascontiguousarray(shift, np.float32)
    Cout = weight.shape[0]
    assert weight.shape == (Cout, Cin, 3, 3) and scale.shape == (Cout,) and shift.shape == (Cout,)
    if mask is not None:
        mask = mask.contiguous()
        assert mask.shape == x.shape and mask.dtype == torch.uint8
    if vscale is None:
        vscale = float(2.0 ** (8 - np.frexp(float(x.abs().max()))[1]))
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    ms = C.c_double(0)
    ov = C.c_int(0)
    mode = (1 if pk_in else 0) | (2 if pk_out else 0) | (4 if extra_pad else 0)
    check(dbg().sivo_debug_conv3_h3_pk_dev(N, Cin, Cout, H, W, x.data_ptr(), mask.data_ptr() if mask is not None else None,
                                           weight.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p),
                                           shift.ctypes.data_as(C.c_void_p), int(relu), C.c_float(vscale),
                                           C.c_float(out_vscale if out_vscale else 0.0), mode, out.data_ptr(), iters, C.byref(ms), C.byref(ov)))
    return out, (ms.value if iters else None), bool(ov.value & 1), bool(ov.value & 2)


def conv_cls_h3(x, weight, scale, shift, relu=False, vscale=None, iters=0):
    """The classifier + MC kernel on the fp16 matrix cores alone (sivo_debug_conv_cls_h3_dev).  x: cuda fp32 (T, Cin, H, W);
    weight (C, Cin, 3, 3), scale / shift (C) numpy.  Returns (logits (T, C, H, W), (classes u8, confidence f64, entropy f64), ms)."""
    x = x.contiguous()
    T, Cin, H, W = x.shape
    weight = np.ascontiguousarray(weight, np.float32)
    scale = np.ascontiguousarray(scale, np.float32); shift = np.ascontiguousarray(shift, np.float32)
    Cc = weight.shape[0]
    assert weight.shape == (Cc, Cin, 3, 3) and scale.shape == (Cc,) and shift.shape == (Cc,)
    if vscale is None:
        vscale = float(2.0 ** (8 - np.frexp(float(x.abs().max()))[1]))
    logits = torch.empty((T, Cc, H, W), dtype=torch.float32, device=x.device)
    cls = torch.empty((H, W), dtype=torch.uint8, device=x.device)
    conf = torch.empty((H, W), dtype=torch.float64, device=x.device)
    ent = torch.empty((H, W), dtype=torch.float64, device=x.device)
    ms = C.c_double(0)
    check(dbg().sivo_debug_conv_cls_h3_dev(T, Cin, Cc, H, W, x.data_ptr(), weight.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p),
                                           shift.ctypes.data_as(C.c_void_p), int(relu), C.c_float(vscale), logits.data_ptr(), cls.data_ptr(),
                                           conf.data_ptr(), ent.data_ptr(), iters, C.byref(ms)))
    return logits, (cls, conf, ent), (ms.value if iters else None)


def mc_reduce(logits, prob_sum=None, want_prob=False, accumulate=False):
    n, K, H, W = logits.shape
    if prob_sum is None:
        prob_sum = torch.zeros((K, H, W), dtype=torch.float32, device=logits.device)
    prob = torch.empty_like(logits) if want_prob else None
    check(lib().sivo_mc_reduce_dev(logits.data_ptr(), n, K, H * W, prob_sum.data_ptr(),
                                   prob.data_ptr() if want_prob else None, int(accumulate), _stream()))
    return prob_sum, prob


def mc_segment(logits):
    """The post-processing of segmentImage on given logits (T, classes, H, W): f64 mean of the per-sample softmax, then
    classes (u8), confidence (f64), entropy (f64)."""
    T, K, H, W = logits.shape
    dev = logits.device
    cls = torch.empty((H, W), dtype=torch.uint8, device=dev)
    conf = torch.empty((H, W), dtype=torch.float64, device=dev)
    ent = torch.empty((H, W), dtype=torch.float64, device=dev)
    check(lib().sivo_mc_segment_dev(logits.data_ptr(), T, K, H * W, cls.data_ptr(), conf.data_ptr(), ent.data_ptr(), _stream()))
    return cls, conf, ent


def mc_finalize(prob_sum, t_total):
    K, H, W = prob_sum.shape
    dev = prob_sum.device
    cls = torch.empty((H, W), dtype=torch.uint8, device=dev)
    conf = torch.empty((H, W), dtype=torch.float64, device=dev)
    ent = torch.empty((H, W), dtype=torch.float64, device=dev)
    check(lib().sivo_mc_finalize_dev(prob_sum.data_ptr(), K, H * W, t_total, cls.data_ptr(), conf.data_ptr(),
                                     ent.data_ptr(), _stream()))
    return cls, conf, ent


def mc_variance(prob, classes):
    T, K, H, W = prob.shape
    var = torch.empty((H, W), dtype=torch.float64, device=prob.device)
    check(lib().sivo_mc_variance_dev(prob.data_ptr(), T, K, H * W, classes.data_ptr(), var.data_ptr(), _stream()))
    return var
