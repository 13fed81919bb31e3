"""Minimal Caffe-prototxt reader for the oracle — TEST INFRASTRUCTURE ONLY.

Reads just the keys the two reference nets use
(config/bayesian_segnet/{basic,standard}/kitti/*.prototxt): input shape
(`input_dim` x4 or `input_shape { dim }`), and per `layer { }` block: name,
type, bottom*, top*, convolution_param {num_output, pad, kernel_size},
pooling_param {pool, kernel_size, stride}, upsample_param {scale},
dropout_param {dropout_ratio, sample_weights_test}, lrn_param {local_size,
alpha, beta}, bn_param {bn_mode}.  Independent of the product's C++ parser
(sivo_amd/csrc/prototxt.cpp); tests cross-check the two.
"""
import re


def _strip_comments(text):
    return "\n".join(line.split("#", 1)[0] for line in text.splitlines())


def _blocks(text, key):
    """Yield the brace-balanced bodies of every top-level `key { ... }`."""
    i = 0
    pat = re.compile(r"\b" + key + r"\s*\{")
    while True:
        m = pat.search(text, i)
        if not m:
            return
        depth, j = 1, m.end()
        while depth:
            c = text[j]
            depth += (c == "{") - (c == "}")
            j += 1
        yield text[m.end():j - 1]
        i = j


def _scalar(body, key, cast, default=None):
    m = re.search(r"\b" + key + r"\s*:\s*\"?([^\s\"}]+)\"?", body)
    return cast(m.group(1)) if m else default


def parse(text, batch=None):
    """Return {'name', 'input', 'shape': [T,C,H,W], 'layers': [dict,...]}."""
    text = _strip_comments(text)
    head = text.split("layer", 1)[0]
    dims = [int(v) for v in re.findall(r"\binput_dim\s*:\s*(\d+)", head)]
    if not dims:
        for body in _blocks(head, "input_shape"):
            dims = [int(v) for v in re.findall(r"\bdim\s*:\s*(\d+)", body)]
    if len(dims) == 3:          # the standard prototxt ships with the sample size left blank
        dims = [0] + dims
    if batch is not None:
        dims[0] = batch
    net = {"name": _scalar(head, "name", str, ""), "input": _scalar(head, "input", str, "data"),
           "shape": dims, "layers": []}
    for body in _blocks(text, "layer"):
        flat = re.sub(r"\b\w+\s*\{[^{}]*\}", "", body)      # drop nested blocks for name/type
        flat = re.sub(r"\b\w+\s*\{[^{}]*\}", "", flat)
        L = {"name": _scalar(flat, "name", str), "type": _scalar(flat, "type", str),
             "bottom": re.findall(r"\bbottom\s*:\s*\"([^\"]+)\"", flat),
             "top": re.findall(r"\btop\s*:\s*\"([^\"]+)\"", flat)}
        t = L["type"]
        if t == "Convolution":
            L.update(num_output=_scalar(body, "num_output", int), pad=_scalar(body, "pad", int, 0),
                     kernel_size=_scalar(body, "kernel_size", int), stride=_scalar(body, "stride", int, 1))
        elif t == "Pooling":
            L.update(pool=_scalar(body, "pool", str, "MAX"), kernel_size=_scalar(body, "kernel_size", int),
                     stride=_scalar(body, "stride", int, 1))
        elif t == "Upsample":
            L.update(scale=_scalar(body, "scale", int, 2))
        elif t == "Dropout":
            L.update(dropout_ratio=_scalar(body, "dropout_ratio", float, 0.5),
                     sample_weights_test=_scalar(body, "sample_weights_test", str, "false") == "true")
        elif t == "LRN":
            L.update(local_size=_scalar(body, "local_size", int, 5), alpha=_scalar(body, "alpha", float, 1.0),
                     beta=_scalar(body, "beta", float, 0.75))
        elif t == "BN":
            L.update(bn_mode=_scalar(body, "bn_mode", str, "LEARN"))
        net["layers"].append(L)
    return net
