"""Deterministic synthetic parameters and the flat parameter container.

The reference's trained weights are Git-LFS pointers, so every
parity test and bench uses seeded synthetic parameters of the exact shapes the
prototxt implies: He-normal conv weights, 0.01*N(0,1) biases, BN scale in
[0.5,1.5] and shift 0.1*N(0,1) (BN INFERENCE is `scale*x + shift`).  The first
convolution is scaled by 1/64 so raw 0..255 pixels give O(1) activations and
O(1..10) logits, which keeps the 1e-3 logit tolerance meaningful.

Flat layout handed to the C ABI (sivo_segnet_create): parameters of each
parametrised layer in prototxt order; Convolution = W (Cout,Cin,k,k) then bias
(Cout); BN = scale (C) then shift (C); all fp32.
"""
import numpy as np


def param_shapes(layers, cin=3):
    """layers: list of dicts with name/type/bottom/top/num_output/kernel_size (either parser's output).
    Returns [(layer_name, [shape, ...])] in prototxt order."""
    ch = {}
    out = []
    first = True
    for L in layers:
        t = L["type"]
        c_in = ch.get(L["bottom"][0], cin) if L["bottom"] else cin
        if t == "Convolution":
            k = L["kernel_size"]
            out.append((L["name"], [(L["num_output"], c_in, k, k), (L["num_output"],)]))
            ch[L["top"][0]] = L["num_output"]
            first = False
        elif t == "BN":
            out.append((L["name"], [(c_in,), (c_in,)]))
            ch[L["top"][0]] = c_in
        else:
            for tname in L["top"]:
                ch[tname] = c_in
    return out


def synth_weights(layers, seed=42, first_scale=1.0 / 64):
    rng = np.random.default_rng(seed)
    w = {}
    first = True
    for name, shapes in param_shapes(layers):
        if len(shapes[0]) == 4:
            co, ci, k, _ = shapes[0]
            std = np.sqrt(2.0 / (ci * k * k))
            W = (rng.standard_normal(shapes[0]) * std).astype(np.float32)
            if first:
                W *= np.float32(first_scale); first = False
            b = (0.01 * rng.standard_normal(shapes[1])).astype(np.float32)
            w[name] = [W, b]
        else:
            s = rng.uniform(0.5, 1.5, shapes[0]).astype(np.float32)
            sh = (0.1 * rng.standard_normal(shapes[1])).astype(np.float32)
            w[name] = [s, sh]
    return w


def pack(layers, weights):
    parts = []
    for name, shapes in param_shapes(layers):
        for arr, shp in zip(weights[name], shapes):
            assert tuple(arr.shape) == tuple(shp), (name, arr.shape, shp)
            parts.append(np.ascontiguousarray(arr, np.float32).ravel())
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


MAGIC = b"SIVOW001"


def save(path, flat):
    """.sivow container: 8-byte magic, u64 count, fp32 payload (little endian)."""
    with open(path, "wb") as f:
        f.write(MAGIC); f.write(np.uint64(flat.size).tobytes()); f.write(np.ascontiguousarray(flat, "<f4").tobytes())


def load(path):
    with open(path, "rb") as f:
        assert f.read(8) == MAGIC, "not a .sivow file"
        n = int(np.frombuffer(f.read(8), np.uint64)[0])
        return np.frombuffer(f.read(4 * n), "<f4").copy()


def load_caffemodel(prototxt_text, path_or_bytes):
    """Flat parameter array from a trained `.caffemodel` (what Net::CopyTrainedLayersFrom reads at
    reference bayesian_segnet.cpp:61), layers matched to the prototxt by name.  Host-only."""
    import ctypes as C
    from ._lib import lib, check
    blob = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    text = prototxt_text.encode() if isinstance(prototxt_text, str) else prototxt_text
    n = C.c_size_t(0)
    check(lib().sivo_caffemodel_weights(text, len(text), blob, len(blob), None, 0, C.byref(n)))
    out = np.empty(n.value, np.float32)
    check(lib().sivo_caffemodel_weights(text, len(text), blob, len(blob), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
    return out


# ---- minimal protobuf writer (tests and export only): NetParameter{name=1, layer=100{name=1,type=2,blobs=7}} ----
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _blob_proto(arr, legacy_dims=False):
    arr = np.ascontiguousarray(arr, "<f4")
    if legacy_dims:
        dims = ([1] * (4 - arr.ndim) + list(arr.shape)) if arr.ndim > 1 else [1, arr.size, 1, 1]
        head = b"".join(_varint((f << 3) | 0) + _varint(d) for f, d in zip((1, 2, 3, 4), dims))
        return head + _ld(5, arr.tobytes())
    shape = _ld(1, b"".join(_varint(d) for d in arr.shape))
    return _ld(7, shape) + _ld(5, arr.tobytes())


def to_caffemodel(layers, weights, net_name="sivo", v1=False, legacy_dims=False, extra_layers=True):
    """Encode `weights` ({layer name: [blob, blob]}) as a binary Caffe NetParameter.  v1=True writes the
    legacy `layers` field (V1LayerParameter: name=4, blobs=6).  Parameter-free layers are written too
    (as a real file has them) when extra_layers is set."""
    body = _ld(1, net_name.encode())
    for L in layers:
        blobs = weights.get(L["name"])
        if blobs is None and not extra_layers:
            continue
        if v1:
            msg = _ld(4, L["name"].encode()) + b"".join(_ld(2, b.encode()) for b in L["bottom"])
            msg += b"".join(_ld(6, _blob_proto(b, legacy_dims)) for b in (blobs or []))
            body += _ld(2, msg)
        else:
            msg = _ld(1, L["name"].encode()) + _ld(2, L["type"].encode())
            msg += b"".join(_ld(3, b.encode()) for b in L["bottom"]) + b"".join(_ld(4, t.encode()) for t in L["top"])
            msg += _varint((10 << 3) | 0) + _varint(1)             # phase: TEST
            msg += b"".join(_ld(7, _blob_proto(b, legacy_dims)) for b in (blobs or []))
            body += _ld(100, msg)
    return body
