"""CPU: host-side logic of the product and the C ABI surface (no compute calls — there is no GPU here).
Covers: the C++ prototxt reader vs the oracle's Python reader, netspec vs the reference layer graphs,
the parameter container, the reference's constructor error conventions, the host quadtree vs the oracle,
and that libsivo_hip.so exports every symbol include/sivo_hip.h declares and fails loudly without a GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import prototxt as oproto
from sivo_amd import _lib, netspec, orb, weights as wts


def _graph(net):
    keys = ("name", "type", "bottom", "top", "num_output", "pad", "kernel_size", "pool", "stride", "scale",
            "dropout_ratio", "sample_weights_test", "local_size", "alpha", "beta", "bn_mode")
    return [{k: L[k] for k in keys if k in L} for L in net["layers"]]


@pytest.mark.parametrize("kind", ["standard", "basic"])
def test_netspec_reproduces_reference_graph(kind):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", f"netgraph_{kind}.json")))
    gen = oproto.parse((netspec.standard_prototxt if kind == "standard" else netspec.basic_prototxt)(12))
    assert _graph(gen) == gold["layers"]
    assert gen["shape"][1:] == gold["shape"][1:] and gen["name"] == gold["name"]
    ref_path = f"/root/reference/config/bayesian_segnet/{'standard/kitti/bayesian_segnet_kitti' if kind == 'standard' else 'basic/kitti/bayesian_segnet_basic_kitti'}.prototxt"
    if os.path.exists(ref_path):          # build container only
        assert _graph(oproto.parse(open(ref_path).read())) == gold["layers"]


def test_netspec_layer_parser_agrees_with_the_oracle_parser():
    """sivo_amd.netspec.parse_layers (what bench.py sizes the synthetic weights with, so that the product path never
    imports oracle/) yields the same parameter shapes as the oracle's prototxt parser, on generated and reference text."""
    texts = [netspec.standard_prototxt(12), netspec.basic_prototxt(6), netspec.tiny_prototxt(2)]
    for kind in ("standard/kitti/bayesian_segnet_kitti", "basic/kitti/bayesian_segnet_basic_kitti"):
        ref = f"/root/reference/config/bayesian_segnet/{kind}.prototxt"
        if os.path.exists(ref):
            texts.append(open(ref).read())
    for text in texts:
        assert wts.param_shapes(netspec.parse_layers(text)) == wts.param_shapes(oproto.parse(text)["layers"])


def test_bench_imports_the_oracle_only_in_its_cpu_baseline_leg():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main"):]
    code = re.sub(r'"[^"\n]*"', '""', re.sub(r"#.*", "", body))          # string literals (parity notes name the oracle) and comments out
    assert "oracle" not in code.replace("cpu_baseline", "")
    helpers = src[src.index("KERNEL_CLASS"):src.index("def main")]          # roofline / timing helpers between cpu_baseline() and main()
    assert "oracle" not in re.sub(r'"[^"\n]*"', '""', re.sub(r"#.*", "", helpers))


def test_parameter_counts_match_reference_weight_files():
    """1,415,823 fp32 parameters for Basic (the 5,670,476-byte LFS object minus protobuf framing), SURVEY.md A.2."""
    n = C.c_size_t()
    text = netspec.basic_prototxt(6).encode()
    _lib.check(_lib.lib().sivo_segnet_num_params(text, len(text), C.byref(n)))
    assert n.value == 1415823
    text = netspec.standard_prototxt(12).encode()
    _lib.check(_lib.lib().sivo_segnet_num_params(text, len(text), C.byref(n)))
    layers = oproto.parse(text.decode())["layers"]
    assert n.value == sum(int(np.prod(s)) for _, shp in wts.param_shapes(layers) for s in shp) == 29451663


def test_cpp_prototxt_reader_handles_reference_quirks():
    """Blank batch dimension ("dim: # SET SAMPLE SIZE HERE", standard prototxt :3-8), comments, nested blocks."""
    ref = "/root/reference/config/bayesian_segnet/standard/kitti/bayesian_segnet_kitti.prototxt"
    n = C.c_size_t()
    if os.path.exists(ref):
        text = open(ref, "rb").read()
        _lib.check(_lib.lib().sivo_segnet_num_params(text, len(text), C.byref(n)))
        assert n.value == 29451663
    text = b'name: "x"\ninput: "data"\ninput_shape {\n  dim: # blank\n  dim: 3\n  dim: 8\n  dim: 8\n}\n' \
           b'layer { bottom: "data" top: "c" name: "c" type: "Convolution" param { lr_mult: 1 } ' \
           b'convolution_param { weight_filler { type: "xavier" } num_output: 4 pad: 1 kernel_size: 3 } }\n'
    _lib.check(_lib.lib().sivo_segnet_num_params(text, len(text), C.byref(n)))
    assert n.value == 4 * 3 * 9 + 4
    rc = _lib.lib().sivo_segnet_num_params(b"layer {", 7, C.byref(n))
    assert rc == _lib.ERR_INVALID_ARGUMENT


def test_constructor_error_conventions():
    """Empty model / weights -> std::invalid_argument in the reference (bayesian_segnet.cpp:80-89, pinned by
    tests/test_bayesian_segnet.cpp:138-150); C != 3 and T <= 1 likewise (:64-70)."""
    from sivo_amd.segnet import BayesianSegNet, BayesianSegNetParams
    with pytest.raises(ValueError, match="model_file"):
        BayesianSegNet(BayesianSegNetParams("", "w.sivow"))
    with pytest.raises(ValueError, match="weights_file"):
        BayesianSegNet(BayesianSegNetParams("m.prototxt", ""))
    with pytest.raises(ValueError, match="model_file"):
        BayesianSegNet(prototxt="", weights=np.zeros(4, np.float32))
    h = C.c_void_p()
    lib = _lib.lib()
    assert lib.sivo_segnet_create(b"", 0, 0, None, 0, 0, C.byref(h)) == _lib.ERR_INVALID_ARGUMENT
    assert lib.sivo_segnet_create_from_files(b"", b"x", 0, 0, C.byref(h)) == _lib.ERR_INVALID_ARGUMENT


def test_weight_container_roundtrip(tmp_path):
    layers = oproto.parse(netspec.tiny_prototxt(2))["layers"]
    w = wts.synth_weights(layers, 1)
    flat = wts.pack(layers, w)
    p = tmp_path / "w.sivow"
    wts.save(str(p), flat)
    assert np.array_equal(wts.load(str(p)), flat)
    w2 = wts.synth_weights(layers, 1)
    assert all(np.array_equal(a, b) for k in w for a, b in zip(w[k], w2[k]))      # seeded => reproducible


def test_caffemodel_reader_matches_layers_by_name(tmp_path):
    """Net::CopyTrainedLayersFrom semantics (reference bayesian_segnet.cpp:61): a binary NetParameter in the
    `layer` (100) or legacy `layers` (2) encoding, blobs with BlobShape or legacy num/channels/height/width,
    layers in any order, parameter-free layers present, is turned into the flat array bit-exactly."""
    for text in (netspec.tiny_prototxt(2), netspec.basic_prototxt(6)):
        layers = oproto.parse(text)["layers"]
        w = wts.synth_weights(layers, 5)
        flat = wts.pack(layers, w)
        for v1 in (False, True):
            for legacy in (False, True):
                blob = wts.to_caffemodel(layers, w, v1=v1, legacy_dims=legacy)
                assert np.array_equal(wts.load_caffemodel(text, blob), flat)
        blob = wts.to_caffemodel(list(reversed(layers)), w)                 # file order is irrelevant, names decide
        assert np.array_equal(wts.load_caffemodel(text, blob), flat)
    # the Basic file size lands where the reference's LFS pointer says the trained model does (5,670,476 B):
    # 4 B per parameter + protobuf framing
    assert 4 * 1415823 < len(wts.to_caffemodel(layers, w)) < 5670476 + 4096
    # errors Caffe CHECK-fails on: a parametrised layer missing from the file, or a blob of the wrong size
    some = next(n for n, shp in wts.param_shapes(layers) if len(shp[0]) == 4)
    w_missing = {k: v for k, v in w.items() if k != some}
    with pytest.raises(ValueError, match=some):
        wts.load_caffemodel(text, wts.to_caffemodel(layers, w_missing))
    w_bad = dict(w); w_bad[some] = [w[some][0][:, :, :, :2], w[some][1]]
    with pytest.raises(ValueError, match="prototxt implies"):
        wts.load_caffemodel(text, wts.to_caffemodel(layers, w_bad))
    with pytest.raises(ValueError):
        wts.load_caffemodel(text, wts.to_caffemodel(layers, w)[:-7])        # truncated file
    # an un-pulled Git-LFS pointer (what /root/reference ships) is named as such
    ptr = tmp_path / "w.caffemodel"
    ptr.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 5670476\n")
    proto = tmp_path / "m.prototxt"
    proto.write_text(text)
    h = C.c_void_p()
    assert _lib.lib().sivo_segnet_create_from_files(str(proto).encode(), str(ptr).encode(), 0, 0, C.byref(h)) == _lib.ERR_INVALID_ARGUMENT
    assert b"Git-LFS" in _lib.lib().sivo_last_error()


def _caffe_messages():
    """NetParameter / LayerParameter / V1LayerParameter / BlobProto / BlobShape of BVLC Caffe's caffe.proto — the fields a
    .caffemodel of a trained net carries, with the field numbers of the published file (caffe-segnet keeps them) — as
    message classes of Google's protobuf runtime.  caffe.proto itself is not in the reference tree (empty submodule)."""
    from google.protobuf import descriptor_pb2 as D, descriptor_pool, message_factory
    F = D.FieldDescriptorProto
    fp = D.FileDescriptorProto(name="caffe_subset.proto", package="caffe", syntax="proto2")

    def msg(name, fields):
        m = fp.message_type.add(name=name)
        for fname, num, typ, label, extra in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if extra.get("type_name"):
                f.type_name = extra["type_name"]
            if extra.get("packed"):
                f.options.packed = True
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, REP, {"packed": True})])
    msg("BlobProto", [("num", 1, F.TYPE_INT32, OPT, {}), ("channels", 2, F.TYPE_INT32, OPT, {}), ("height", 3, F.TYPE_INT32, OPT, {}),
                      ("width", 4, F.TYPE_INT32, OPT, {}), ("data", 5, F.TYPE_FLOAT, REP, {"packed": True}),
                      ("diff", 6, F.TYPE_FLOAT, REP, {"packed": True}), ("shape", 7, F.TYPE_MESSAGE, OPT, {"type_name": ".caffe.BlobShape"}),
                      ("double_data", 8, F.TYPE_DOUBLE, REP, {"packed": True})])
    msg("LayerParameter", [("name", 1, F.TYPE_STRING, OPT, {}), ("type", 2, F.TYPE_STRING, OPT, {}), ("bottom", 3, F.TYPE_STRING, REP, {}),
                           ("top", 4, F.TYPE_STRING, REP, {}), ("blobs", 7, F.TYPE_MESSAGE, REP, {"type_name": ".caffe.BlobProto"}),
                           ("phase", 10, F.TYPE_INT32, OPT, {})])
    msg("V1LayerParameter", [("bottom", 2, F.TYPE_STRING, REP, {}), ("top", 3, F.TYPE_STRING, REP, {}), ("name", 4, F.TYPE_STRING, OPT, {}),
                             ("type", 5, F.TYPE_INT32, OPT, {}), ("blobs", 6, F.TYPE_MESSAGE, REP, {"type_name": ".caffe.BlobProto"})])
    msg("NetParameter", [("name", 1, F.TYPE_STRING, OPT, {}), ("layers", 2, F.TYPE_MESSAGE, REP, {"type_name": ".caffe.V1LayerParameter"}),
                         ("input", 3, F.TYPE_STRING, REP, {}), ("input_dim", 4, F.TYPE_INT32, REP, {}),
                         ("layer", 100, F.TYPE_MESSAGE, REP, {"type_name": ".caffe.LayerParameter"})])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fp)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe." + n))
            for n in ("NetParameter", "LayerParameter", "V1LayerParameter", "BlobProto", "BlobShape")}


@pytest.mark.parametrize("v1,legacy_dims,unpacked", [(False, False, False), (True, True, False), (False, True, True)])
def test_caffemodel_reader_against_googles_protobuf_writer(v1, legacy_dims, unpacked):
    """The library's .caffemodel reader (csrc/caffemodel.cpp, Net::CopyTrainedLayersFrom of bayesian_segnet.cpp:61) on
    files serialised by Google's protobuf runtime from a restatement of caffe.proto's messages — a writer that shares no
    code with this repository (test_caffemodel_reader_matches_layers_by_name uses the repository's own encoder).
    Both encodings of the layer list, both encodings of the blob shape, packed float data, an extra `diff` field and a
    double_data blob the reader has to skip / convert, layers in shuffled order with parameter-free ones in between."""
    M = _caffe_messages()
    text = netspec.basic_prototxt(6)
    layers = oproto.parse(text)["layers"]
    w = wts.synth_weights(layers, 9)
    flat = wts.pack(layers, w)
    net = M["NetParameter"](name="segnet_basic")
    net.input.append("data"); net.input_dim.extend([6, 3, 352, 1024])
    order = list(layers)
    np.random.default_rng(4).shuffle(order)
    first_conv = next(L["name"] for L in layers if L["type"] == "Convolution")
    for L in order:
        lp = net.layers.add() if v1 else net.layer.add()
        lp.name = L["name"]
        if v1:
            lp.type = 4 if L["type"] == "Convolution" else 39
        else:
            lp.type = L["type"]; lp.phase = 1
        lp.bottom.extend(L["bottom"]); lp.top.extend(L["top"])
        for i, b in enumerate(w.get(L["name"], [])):
            bp = lp.blobs.add()
            b = np.ascontiguousarray(b, np.float32)
            if legacy_dims:
                dims = (1,) * (4 - b.ndim) + b.shape
                bp.num, bp.channels, bp.height, bp.width = (int(d) for d in dims)
            else:
                bp.shape.dim.extend(int(d) for d in b.shape)
            if L["name"] == first_conv and i == 1 and not v1:
                bp.double_data.extend(float(x) for x in b.ravel())       # old Caffe snapshots of double nets
            else:
                bp.data.extend(float(x) for x in b.ravel())
            if i == 0 and L["name"] == first_conv:
                bp.diff.extend([0.0] * 5)                                 # a snapshot written with diffs: ignored on load
    blob = net.SerializeToString()
    if unpacked:
        # protobuf parsers must accept both packed and unpacked encodings of a repeated scalar: re-encode one small blob
        # (a bias) unpacked by hand — field 5, wire type 5 (32-bit) per element — and splice it in through the runtime
        bias_layer = next(l for l in net.layer if l.name == first_conv)
        raw = b"".join(bytes([5 << 3 | 5]) + np.float32(x).tobytes() for x in bias_layer.blobs[1].double_data)
        shp = bias_layer.blobs[1].shape.SerializeToString() if not legacy_dims else b""
        dims = b"".join(bytes([(n << 3) | 0, int(v)]) for n, v in ((1, 1), (2, 1), (3, 1))) if legacy_dims else b""
        bp_bytes = dims + raw + (bytes([7 << 3 | 2, len(shp)]) + shp if shp else b"")
        if legacy_dims:
            n = len(bias_layer.blobs[1].double_data)
            assert n < 128
            bp_bytes += bytes([(4 << 3) | 0, n])
        del bias_layer.blobs[1]
        body = bias_layer.SerializeToString() + bytes([7 << 3 | 2]) + wts._varint(len(bp_bytes)) + bp_bytes
        others = M["NetParameter"](); others.CopyFrom(net)
        keep = [l for l in others.layer if l.name != first_conv]
        del others.layer[:]
        others.layer.extend(keep)
        blob = others.SerializeToString() + b"\xa2\x06" + wts._varint(len(body)) + body        # field 100, wire type 2
        assert M["NetParameter"].FromString(blob).layer[-1].blobs[1].data[:2] == list(np.float32(w[first_conv][1][:2]))
    assert np.array_equal(wts.load_caffemodel(text, blob), flat)


def test_host_quadtree_matches_oracle(oracle, kitti_like_bgr):
    ex = oracle.OrbExtractor()
    ex(oracle.bgr2gray(kitti_like_bgr))
    for l in range(8):
        c = ex.candidates(l)
        rows, cols = ex.level(l).shape
        a = oracle.distribute_octtree(c, 16, cols - 16, 16, rows - 16, int(ex.features_per_level[l]))
        b = orb.distribute_octtree(c, 16, cols - 16, 16, rows - 16, int(ex.features_per_level[l]))
        assert a.tobytes() == b.tobytes(), f"level {l}"
    rng = np.random.default_rng(0)
    for n, N in ((1, 10), (5, 3), (400, 400), (2500, 100)):
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = rng.integers(0, 300, n); k["y"] = rng.integers(0, 100, n); k["response"] = rng.integers(7, 60, n)
        assert oracle.distribute_octtree(k, 16, 316, 16, 116, N).tobytes() == orb.distribute_octtree(k, 16, 316, 16, 116, N).tobytes()
    assert len(orb.distribute_octtree(np.zeros(0, oracle.KP_DTYPE), 16, 316, 16, 116, 10)) == 0


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sivo_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sivo_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared >= set(_lib.SIGNATURES)                 # the Python binding binds only declared symbols
    # the test / diagnostic entry points are NOT product ABI: declared in sivo_hip_debug.h, exported by libsivo_hip_dbg.so only
    assert not [s for s in declared if s.startswith("sivo_debug_")]
    dheader = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sivo_hip_debug.h")).read(), flags=re.S)
    ddeclared = set(re.findall(r"\b(sivo_debug_[a-z0-9_]+)\s*\(", dheader))
    assert ddeclared == set(_lib.DEBUG_SIGNATURES) and len(ddeclared) >= 4
    dbg = C.CDLL(_lib.DBG_PATH)
    assert not [s for s in sorted(ddeclared) if not hasattr(dbg, s)]
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "sivo_debug_" not in exported


def test_product_library_reads_no_environment_switch():
    """The library is configured through SivoSegnetOptions at construction (include/sivo_hip.h); every A/B, ablation and fault-injection
    switch lives behind SIVO_DIAG_ENV (sivo_amd/csrc/common.hpp) and exists in libsivo_hip_diag.so only.  The product binary does not
    contain the name of a single SIVO_ variable; the Python wrapper (sivo_amd/segnet.py segnet_options) maps the old names onto options."""
    import subprocess
    names = sorted(set(l.strip() for l in subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True).stdout.splitlines() if re.fullmatch(r"SIVO_[A-Z0-9_]+", l.strip())))
    assert names == [], names
    diag = subprocess.run(["strings", _lib.DIAG_PATH], capture_output=True, text=True).stdout
    for name in ("SIVO_H3_BOOST", "SIVO_MULTI_EMULATE", "SIVO_NO_FUSE_BRIDGE", "SIVO_D3_FORM", "SIVO_D3_ABL", "SIVO_ORB_PRIO"):
        assert name in diag, name
    from sivo_amd.segnet import segnet_options
    old = {k: os.environ.pop(k, None) for k in ("SIVO_LANES", "SIVO_GEMM", "SIVO_D3", "SIVO_D3_PK", "SIVO_CONV7", "SIVO_WINO4_MB", "SIVO_DEBUG_SYNC")}
    try:
        d = segnet_options()
        assert (d.struct_size, d.lanes, d.gemm, d.no_direct_f16x3, d.no_packed_activations, d.conv7_fp32, d.wino4_workspace_mb, d.debug_sync) == (32, 0, 0, 0, 0, 0, 0, 0)
        os.environ.update(SIVO_LANES="3", SIVO_GEMM="x6", SIVO_D3="0", SIVO_D3_PK="0", SIVO_CONV7="f32", SIVO_WINO4_MB="512", SIVO_DEBUG_SYNC="1")
        e = segnet_options()
        assert (e.lanes, e.gemm, e.no_direct_f16x3, e.no_packed_activations, e.conv7_fp32, e.wino4_workspace_mb, e.debug_sync) == (3, 1, 1, 1, 1, 512, 1)
        k = segnet_options(lanes=1, gemm="f32", direct_f16x3=True, packed_activations=True, conv7="h3", wino4_workspace_mb=64, debug_sync=False)
        assert (k.lanes, k.gemm, k.no_direct_f16x3, k.no_packed_activations, k.conv7_fp32, k.wino4_workspace_mb, k.debug_sync) == (1, 2, 0, 0, 0, 64, 0)
    finally:
        for name, v in old.items():
            os.environ.pop(name, None)
            if v is not None:
                os.environ[name] = v


def test_struct_layouts_match_reference_types():
    assert C.sizeof(_lib.KeyPoint) == 28 and orb.KP_DTYPE.itemsize == 28      # cv::KeyPoint
    assert C.sizeof(_lib.Edge) == 48


def test_no_gpu_fails_loudly():
    """The product has no CPU fallback: compute entry points report SIVO_ERR_RUNTIME without a device."""
    lib = _lib.lib()
    if lib.sivo_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert lib.sivo_orb_create(2000, C.c_float(1.2), 8, 20, 7, 0, C.byref(h)) == _lib.ERR_RUNTIME
    a = np.zeros((2, 32), np.uint8); out = np.zeros((2, 2), np.int32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert lib.sivo_hamming_matrix(p(a), 2, p(a), 2, p(out)) == _lib.ERR_RUNTIME
    text = netspec.tiny_prototxt(2).encode()
    w = np.zeros(10, np.float32)
    assert lib.sivo_segnet_create(text, len(text), 0, p(w), 10, 0, C.byref(h)) == _lib.ERR_RUNTIME
    assert b"no CPU fallback" in lib.sivo_last_error()
    with pytest.raises(_lib.SivoError):
        _lib.require_gpu()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sivo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src, f


def test_kitti_sequence_listing_and_trajectory_format(tmp_path):
    """times.txt parsing + file naming (reference sivo.cc:145-177) and the CameraTrajectory.txt line format
    (System.cc:322-329): 12 numbers, fixed, 9 decimals, [Rwc | twc] row-major; the header-only C++ twin writes
    the same bytes (exercised by tests/cpp/test_api cpu)."""
    from sivo_amd import kitti
    seq = tmp_path / "00"
    seq.mkdir()
    (seq / "times.txt").write_text("0.000000e+00\n1.037875e-01\n\n2.076350e-01\n")
    left, right, times = kitti.load_images(str(seq))
    assert times == [0.0, 0.1037875, 0.207635]
    assert left[2] == f"{seq}/image_2/000002.png" and right[0] == f"{seq}/image_3/000000.png" and len(left) == 3
    a = 0.1
    T = np.array([[np.cos(a), 0, -np.sin(a), 0, 1, 0, np.sin(a), 0, np.cos(a), 0.5, -0.25, 2.0],
                  [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]], np.float32)
    out = tmp_path / "CameraTrajectory.txt"
    kitti.save_trajectory_kitti(str(out), T)
    lines = out.read_text().splitlines()
    # identity pose: twc = -Rwc * 0 is a NEGATIVE zero, which `f << fixed` prints with its sign (as the reference does)
    assert lines[1] == " ".join(["1.000000000", "0.000000000", "0.000000000", "-0.000000000",
                                 "0.000000000", "1.000000000", "0.000000000", "-0.000000000",
                                 "0.000000000", "0.000000000", "1.000000000", "-0.000000000"])
    v = np.array(lines[0].split(), np.float64).reshape(3, 4)
    Rcw = T[0, :9].reshape(3, 3).astype(np.float64)
    np.testing.assert_allclose(v[:, :3], Rcw.T, atol=1e-7)
    np.testing.assert_allclose(v[:, 3], -Rcw.T @ T[0, 9:].astype(np.float64), atol=1e-6)
    assert all(len(x.split(".")[1]) == 9 for x in lines[0].split())
    # the C++ twin produces the same file
    import subprocess
    from test_cpp_api import BIN, _build
    if not os.path.exists(BIN):
        _build()
    T.tofile(tmp_path / "T.bin")
    r = subprocess.run([BIN, "kitti", str(seq), str(tmp_path / "T.bin"), str(tmp_path / "cpp.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "cpp.txt").read_text() == out.read_text()
    assert r.stdout.split() == ["3", f"{seq}/image_2/000002.png", f"{seq}/image_3/000002.png", "0.207635"]


def test_tools_the_gpu_tests_run_are_loadable():
    """tests/test_gpu_coresidency.py runs tools/coresident_probe.py and tools/bridge_pair_repro.py in processes of their own and looks
    their variants up by name: the tools must parse, import without a GPU, and still carry those variants and the libraries they name."""
    import importlib
    import py_compile
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(os.listdir(os.path.join(root, "tools"))):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(root, "tools", f), doraise=True)
    sys.path.insert(0, os.path.join(root, "tools"))
    cp = importlib.import_module("coresident_probe")
    bp = importlib.import_module("bridge_pair_repro")
    probe = dict(cp.VARIANTS)
    for name in ("HZ8 exact LDS, the bridge as shipped (no packed-FP32 instructions)", "HZ8 exact LDS, the bridge with packed-FP32 instructions (the reproducer)"):
        assert probe[name].get("SIVO_H3_LDS_ALL") == "0", name
    assert probe["HZ8 exact LDS, the bridge as shipped (no packed-FP32 instructions)"]["PROBE_DIAG_LIB"] == "libsivo_hip_diag.so"
    pair = {v[0]: v for v in bp.VARIANTS}
    assert pair["bridge as shipped, GEMM with its exact LDS, 2 lanes"][1:] == ("libsivo_hip_diag.so", {"SIVO_H3_LDS_ALL": "0"}, 2)
    assert pair["packed bridge, GEMM with its exact LDS, 2 lanes"][1:] == ("libsivo_hip_diag_pkbridge.so", {"SIVO_H3_LDS_ALL": "0"}, 2)
    assert pair["packed bridge, GEMM claiming 160 KB, 2 lanes"][1:] == ("libsivo_hip_diag_pkbridge.so", {}, 2)
    # both diagnostic libraries are part of the build (the packed form of the bridge is the reproducer)
    mk = open(os.path.join(root, "sivo_amd", "csrc", "Makefile")).read()
    assert "all: $(OUT) dbg diag diag_pkbridge" in mk
