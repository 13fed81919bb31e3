"""ctypes binding of libsivo_hip.so (the C ABI declared in include/sivo_hip.h).

There is no CPU fallback: if the HIP library is missing, or no HIP device is
visible when a compute entry point is called, the call fails loudly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsivo_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_RUNTIME = 2
ERR_UNSUPPORTED = 3
ERR_IMAGE_TOO_SMALL = 4
ERR_CAPACITY = 5


class SivoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsivo_hip error {code}: {msg}")
        self.code = code


class SivoInvalidArgument(SivoError, ValueError):
    """SIVO_ERR_INVALID_ARGUMENT: where the reference throws std::invalid_argument."""


class KeyPoint(C.Structure):  # == cv::KeyPoint / SivoKeyPoint (28 bytes)
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


class OpProfile(C.Structure):  # == SivoOpProfile
    _fields_ = [("layer", C.c_char * 64), ("kernel", C.c_char * 96), ("samples", C.c_int32), ("launches", C.c_int32),
                ("flops_per_sample", C.c_double), ("bytes_per_sample", C.c_double), ("ms_total", C.c_double),
                ("kernel_launches", C.c_int32), ("pad_", C.c_int32)]


class SegnetOptions(C.Structure):   # == SivoSegnetOptions
    _fields_ = [("struct_size", C.c_uint32), ("lanes", C.c_int32), ("gemm", C.c_int32), ("no_direct_f16x3", C.c_int32),
                ("no_packed_activations", C.c_int32), ("conv7_fp32", C.c_int32), ("wino4_workspace_mb", C.c_int32), ("debug_sync", C.c_int32)]


class SearchQuery(C.Structure):   # == SivoSearchQuery (36 bytes)
    _fields_ = [("u", C.c_float), ("v", C.c_float), ("radius", C.c_float), ("lvl_lo", C.c_int32), ("lvl_hi", C.c_int32),
                ("ur", C.c_float), ("gate", C.c_float), ("angle", C.c_float), ("flags", C.c_int32)]


class SearchRule(C.Structure):    # == SivoSearchRule
    _fields_ = [("th_dist", C.c_int32), ("accept_lt", C.c_int32), ("ratio_mode", C.c_int32), ("nn_ratio", C.c_float),
                ("gate_mode", C.c_int32), ("check_orientation", C.c_int32), ("dynamic", C.c_int32), ("tie_last", C.c_int32),
                ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float)]


class Edge(C.Structure):      # == SivoEdge (48 bytes)
    _fields_ = [("pose", C.c_int32), ("point", C.c_int32), ("stereo", C.c_int32), ("pad_", C.c_int32),
                ("obs", C.c_double * 3), ("inv_sigma2", C.c_double)]


_vp, _i, _i64, _u64, _sz, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t, C.c_float, C.c_double
_pi32 = C.POINTER(C.c_int32)

# name -> argtypes; every function returns int status unless listed in _RESTYPE
SIGNATURES = {
    "sivo_version": [],
    "sivo_device_count": [],
    "sivo_segnet_create": [C.c_char_p, _sz, _i, _vp, _sz, _i, C.POINTER(_vp)],
    "sivo_segnet_create_opts": [C.c_char_p, _sz, _i, _vp, _sz, _i, _vp, C.POINTER(_vp)],
    "sivo_segnet_create_from_files_opts": [C.c_char_p, C.c_char_p, _i, _i, _vp, C.POINTER(_vp)],
    "sivo_segnet_create_multi_opts": [C.c_char_p, _sz, _i, _vp, _sz, _pi32, _i, _vp, C.POINTER(_vp)],
    "sivo_segnet_create_multi_from_files_opts": [C.c_char_p, C.c_char_p, _i, _pi32, _i, _vp, C.POINTER(_vp)],
    "sivo_caffemodel_weights": [C.c_char_p, _sz, _vp, _sz, _vp, _sz, C.POINTER(_sz)],
    "sivo_segnet_create_from_files": [C.c_char_p, C.c_char_p, _i, _i, C.POINTER(_vp)],
    "sivo_segnet_create_multi": [C.c_char_p, _sz, _i, _vp, _sz, _pi32, _i, C.POINTER(_vp)],
    "sivo_segnet_create_multi_from_files": [C.c_char_p, C.c_char_p, _i, _pi32, _i, C.POINTER(_vp)],
    "sivo_segnet_num_devices": [_vp, _pi32],
    "sivo_segnet_destroy": [_vp],
    "sivo_segnet_shape": [_vp, _pi32, _pi32, _pi32, _pi32, _pi32],
    "sivo_segnet_num_params": [C.c_char_p, _sz, C.POINTER(_sz)],
    "sivo_segnet_forward_dev": [_vp, _vp, _i, _i, _u64, _vp, _vp, _vp, _vp],
    "sivo_mc_finalize_dev": [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp],
    "sivo_mc_reduce_dev": [_vp, _i, _i, _i64, _vp, _vp, _i, _vp],
    "sivo_mc_variance_dev": [_vp, _i, _i, _i64, _vp, _vp, _vp],
    "sivo_segnet_segment": [_vp, _vp, _i, _i, _u64, _vp, _vp, _vp],
    "sivo_segnet_segment_dev": [_vp, _vp, _u64, _vp, _vp, _vp, _vp],
    "sivo_segnet_segment_logits_dev": [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp],
    "sivo_mc_segment_dev": [_vp, _i, _i, _i64, _vp, _vp, _vp, _vp],
    "sivo_segnet_blob": [_vp, C.c_char_p, _vp, _sz, _pi32],
    "sivo_segnet_flops": [_vp, C.POINTER(_d), C.POINTER(_d)],
    "sivo_segnet_profile": [_vp, _i],
    "sivo_segnet_profile_read": [_vp, _vp, _i, _pi32],
    "sivo_segnet_gemm_status": [_vp, _pi32, _pi32, _vp, _i, _pi32],
    "sivo_segnet_prefix_bands": [_vp, _i, _vp, _vp, _vp],
    "sivo_segnet_prefix_band_dev": [_vp, _vp, _i, _i, _vp, _vp],
    "sivo_segnet_forward_banded_dev": [_vp, _vp, _i, _i, _i, C.c_uint64, _vp, _vp, _vp],
    "sivo_segnet_guard_report": [_vp, _vp, _i, _pi32, _vp, _vp, _vp, _vp, _pi32],
    "sivo_segnet_take_overflow": [_vp, _pi32],
    "sivo_orb_create": [_i, _f, _i, _i, _i, _i, C.POINTER(_vp)],
    "sivo_orb_destroy": [_vp],
    "sivo_orb_set_gaussian": [_vp, _i],
    "sivo_orb_set_launch_mode": [_vp, _i],
    "sivo_orb_tables": [_vp, _vp, _vp, _vp, _vp, _vp],
    "sivo_orb_extract": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _pi32],
    "sivo_orb_extract_dev": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _pi32, _vp],
    "sivo_orb_extract_pair_dev": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _pi32, _vp, _vp, _i, _pi32, _vp],
    "sivo_orb_profile": [_vp, _i],
    "sivo_orb_profile_read": [_vp, C.POINTER(_d), _pi32, C.POINTER(_d)],
    "sivo_orb_level": [_vp, _i, _vp, _sz, _pi32, _pi32],
    "sivo_orb_candidates": [_vp, _i, _vp, _i, _pi32],
    "sivo_orb_distribute": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _pi32],
    "sivo_hamming_matrix_dev": [_vp, _i, _vp, _i, _vp, _vp],
    "sivo_hamming_matrix": [_vp, _i, _vp, _i, _vp],
    "sivo_hamming_argmin2_dev": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sivo_hamming_argmin2": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sivo_hamming_bruteforce_dev": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "sivo_mframe_create": [_vp, _i, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _i, _i, C.POINTER(_vp)],
    "sivo_mframe_destroy": [_vp],
    "sivo_mframe_features_in_area": [_vp, _f, _f, _f, _i, _i, _vp, _i, _pi32],
    "sivo_search": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _pi32, _pi32],
    "sivo_search_by_projection_mappoints": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _pi32],
    "sivo_search_by_projection_frame": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _f, _i, _vp, _vp, _pi32],
    "sivo_search_by_projection_reloc": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _pi32],
    "sivo_search_by_projection_kf": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _pi32],
    "sivo_fuse": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _pi32],
    "sivo_search_by_sim3_dir": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "sivo_search_by_bow_kf_frame": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _vp, _pi32],
    "sivo_search_by_bow_kf_kf": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _vp, _pi32],
    "sivo_search_for_triangulation": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _pi32],
    "sivo_stereo_match": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp],
    "sivo_stereo_match_begin": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp],
    "sivo_stereo_match_cull": [_i, _vp, _vp, _vp, _vp],
    "sivo_entropy_gate_dev": [_i, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _d, _vp, _i, _d, _vp, _vp, _vp, _vp],
    "sivo_entropy_gate_map_dev": [_i, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _d, _vp, _i, _d, _vp, _vp, _vp],
    "sivo_entropy_gate": [_i, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _d, _vp, _i, _d, _vp, _vp, _vp],
    "sivo_check_semantics_dev": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _d, _vp, _i, _d, _d, _vp, _vp, _vp, _vp],
    "sivo_check_semantics": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _d, _vp, _i, _d, _d, _vp, _vp, _vp],
    "sivo_ba_optimize": [_vp, _vp, _i, _vp, _i, _vp, _i64, C.POINTER(_d), _d, _d, _vp, _vp, _i, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(_i)],
    "sivo_local_ba": [_vp, _vp, _i, _vp, _i, _vp, _i64, C.POINTER(_d), _vp, _vp, _i, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "sivo_pose_optimize": [_vp, _vp, _i, _vp, _i64, C.POINTER(_d), _vp, _vp, _vp, C.POINTER(_i), _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "sivo_ba_linearize_dev": [_vp, _vp, _vp, _i64, C.POINTER(_d), _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sivo_ba_linearize": [_vp, _i, _vp, _i, _vp, _i64, C.POINTER(_d), _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
}

# test / diagnostic entry points (include/sivo_hip_debug.h): libsivo_hip_dbg.so, a thin library over the product's kernels
DEBUG_SIGNATURES = {
    "sivo_debug_conv": [_i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_d)],
    "sivo_debug_h3_gemm": [_i, _i, _i, _vp, _vp, _f, _vp, _i, C.POINTER(_d)],
    "sivo_debug_conv3_h3_dev": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, C.POINTER(_d), C.POINTER(_i)],
    "sivo_debug_conv3_h3_pk_dev": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _i, _vp, _i, C.POINTER(_d), C.POINTER(_i)],
    "sivo_debug_lds_claims": [_vp, _i],
    "sivo_debug_conv_cls_h3_dev": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, C.POINTER(_d)],
}
DBG_PATH = os.path.join(_HERE, "libsivo_hip_dbg.so")

DIAG_PATH = os.path.join(_HERE, "libsivo_hip_diag.so")

_libs = {}            # "product" / "diag" -> CDLL
_tls = threading.local()
_current = "product"
_dbg = None


def dbg():
    """The sivo_debug_* entry points the kernel tests and tools call: libsivo_hip_dbg.so (`make -C sivo_amd/csrc dbg`), a thin library
    that links libsivo_hip.so (loaded first, so that both share one copy) and runs the PRODUCT's kernels — or, inside
    `with use("diag")`, the diagnostic build itself, which carries the same entry points."""
    global _dbg
    if _current == "diag":
        L = lib()
        if not getattr(L, "_sivo_debug_bound", False):
            for name, args in DEBUG_SIGNATURES.items():
                fn = getattr(L, name)
                fn.argtypes = args
                fn.restype = C.c_int
            L._sivo_debug_bound = True
        return L
    if _dbg is None:
        lib()
        if not os.path.exists(DBG_PATH):
            raise ImportError(f"{DBG_PATH} is missing: make -C sivo_amd/csrc dbg")
        L = C.CDLL(DBG_PATH)
        for name, args in DEBUG_SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _dbg = L
    return _dbg


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build the HIP extension first (make -C sivo_amd/csrc all); "
                          "sivo_amd has no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Loaded first, libsivo_hip.so
    # (linked against the same sonames) binds to that copy; loaded after /opt/rocm's, torch finds no device
    # ("No HIP GPUs are available", seen when a test touched this library before anything had imported torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.sivo_last_error.restype = C.c_char_p

    def note(result, func, args, L=L):      # every call remembers the library it went into: check() reads THAT library's error text
        _tls.last = L
        return result
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
        fn.errcheck = note
    return L


def lib():
    """The library the calling code should use: libsivo_hip.so (build it with `python -c 'import __graft_entry__ as g; g.build()'`),
    or — inside a `with use("diag")` block — libsivo_hip_diag.so, the same sources compiled with -DSIVO_DIAG: the only build that
    reads the A/B / fault-injection switches (SIVO_NO_FUSE_*, SIVO_D3_FORM, SIVO_H3_BOOST, SIVO_MULTI_EMULATE, ...; the product
    reads eight documented ones).  Every handle-owning object (BayesianSegNet, ORBextractor, MatchFrame) keeps the library it was created
    in (`._L`): its methods, the functions that take it as an argument, its error text and its destructor go there, whatever the current
    context is — a handle never crosses into a library with other statics or a -DSIVO_DIAG struct layout."""
    if _current not in _libs:
        _libs[_current] = _load(LIB_PATH if _current == "product" else DIAG_PATH)
    return _libs[_current]


class use:
    """Context manager: `with _lib.use("diag"):` makes lib() return the diagnostic build inside the block (tests that compare kernel
    forms bit for bit, force the fp16 overflow path or emulate several devices on one)."""

    def __init__(self, which):
        assert which in ("product", "diag")
        self.which = which

    def __enter__(self):
        global _current
        self.prev, _current = _current, self.which
        return lib()

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False


def check(rc):
    """The error text comes from the library the failing call went into (this thread's last call: product or diagnostic build)."""
    if rc == OK:
        return
    L = getattr(_tls, "last", None) or lib()
    if rc == ERR_INVALID_ARGUMENT:      # the reference throws std::invalid_argument there
        raise SivoInvalidArgument(rc, L.sivo_last_error().decode(errors="replace"))
    raise SivoError(rc, L.sivo_last_error().decode(errors="replace"))


def require_gpu():
    if lib().sivo_device_count() < 1:
        raise SivoError(ERR_RUNTIME, "no HIP device visible: sivo_amd has no CPU fallback")
