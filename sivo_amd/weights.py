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
