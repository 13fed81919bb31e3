"""Programmatic Caffe-prototxt text for the two Bayesian SegNet variants SIVO ships
(reference config/bayesian_segnet/standard/kitti/bayesian_segnet_kitti.prototxt and
config/bayesian_segnet/basic/kitti/bayesian_segnet_basic_kitti.prototxt).

/root/reference does not exist on the GPU box, so benches and GPU tests build
the model description from here; tests/test_netspec.py checks (in the build
container) that the generated text parses to the same layer graph as the
reference files.  T (the MC sample count) is the batch dimension of the input
(reference bayesian_segnet.cpp:67-70).
"""


def _conv(name, bottom, top, cout, k, pad):
    return (f'layer {{\n  bottom: "{bottom}"\n  top: "{top}"\n  name: "{name}"\n  type: "Convolution"\n'
            f'  convolution_param {{\n    num_output: {cout}\n    pad: {pad}\n    kernel_size: {k}\n  }}\n}}\n')


def _bn(name, blob):
    return (f'layer {{\n  bottom: "{blob}"\n  top: "{blob}"\n  name: "{name}"\n  type: "BN"\n'
            f'  bn_param {{\n    bn_mode: INFERENCE\n  }}\n}}\n')


def _relu(name, blob):
    return f'layer {{\n  bottom: "{blob}"\n  top: "{blob}"\n  name: "{name}"\n  type: "ReLU"\n}}\n'


def _pool(name, bottom):
    return (f'layer {{\n  bottom: "{bottom}"\n  top: "{name}"\n  top: "{name}_mask"\n  name: "{name}"\n  type: "Pooling"\n'
            f'  pooling_param {{\n    pool: MAX\n    kernel_size: 2\n    stride: 2\n  }}\n}}\n')


def _drop(name, blob):
    return (f'layer {{\n  name: "{name}"\n  type: "Dropout"\n  bottom: "{blob}"\n  top: "{blob}"\n'
            f'  dropout_param {{\n    sample_weights_test: true\n    dropout_ratio: 0.5\n  }}\n}}\n')


def _up(name, bottom, mask, top):
    return (f'layer {{\n  name: "{name}"\n  type: "Upsample"\n  bottom: "{bottom}"\n  top: "{top}"\n  bottom: "{mask}"\n'
            f'  upsample_param {{\n    scale: 2\n  }}\n}}\n')


def _softmax(bottom):
    return (f'layer {{\n  name: "prob"\n  type: "Softmax"\n  bottom: "{bottom}"\n  top: "prob"\n'
            f'  softmax_param {{engine: CAFFE}}\n}}\n')


def standard_prototxt(T, H=352, W=1024, classes=15, width=(64, 128, 256, 512, 512)):
    """SegNet-Standard (VGG16 encoder/decoder, 26 conv3x3 + BN + ReLU, 6 test-time dropouts)."""
    c1, c2, c3, c4, c5 = width
    out = [f'name: "bayesian_segnet"\ninput: "data"\ninput_shape {{\n  dim: {T}\n  dim: 3\n  dim: {H}\n  dim: {W}\n}}\n']

    def cbr(name, bottom, cout, relu_name=None):
        out.append(_conv(name, bottom, name, cout, 3, 1))
        out.append(_bn(name + "_bn", name))
        out.append(_relu(relu_name or ("relu" + name[4:]), name))
        return name

    x = cbr("conv1_1", "data", c1); x = cbr("conv1_2", x, c1); out.append(_pool("pool1", x))
    x = cbr("conv2_1", "pool1", c2); x = cbr("conv2_2", x, c2); out.append(_pool("pool2", x))
    x = cbr("conv3_1", "pool2", c3); x = cbr("conv3_2", x, c3); x = cbr("conv3_3", x, c3)
    out.append(_pool("pool3", x)); out.append(_drop("pool3_drop", "pool3"))
    x = cbr("conv4_1", "pool3", c4); x = cbr("conv4_2", x, c4); x = cbr("conv4_3", x, c4)
    out.append(_pool("pool4", x)); out.append(_drop("pool4_drop", "pool4"))
    x = cbr("conv5_1", "pool4", c5); x = cbr("conv5_2", x, c5); x = cbr("conv5_3", x, c5)
    out.append(_pool("pool5", x)); out.append(_drop("pool5_drop", "pool5"))
    out.append(_up("upsample5", "pool5", "pool5_mask", "pool5_D"))
    x = cbr("conv5_3_D", "pool5_D", c5); x = cbr("conv5_2_D", x, c5); x = cbr("conv5_1_D", x, c4)
    out.append(_drop("upsample4_drop", x)); out.append(_up("upsample4", x, "pool4_mask", "pool4_D"))
    x = cbr("conv4_3_D", "pool4_D", c4); x = cbr("conv4_2_D", x, c4); x = cbr("conv4_1_D", x, c3)
    out.append(_drop("upsample3_drop", x)); out.append(_up("upsample3", x, "pool3_mask", "pool3_D"))
    x = cbr("conv3_3_D", "pool3_D", c3); x = cbr("conv3_2_D", x, c3); x = cbr("conv3_1_D", x, c2)
    out.append(_drop("upsample2_drop", x)); out.append(_up("upsample2", x, "pool2_mask", "pool2_D"))
    x = cbr("conv2_2_D", "pool2_D", c2); x = cbr("conv2_1_D", x, c1)
    out.append(_up("upsample1", x, "pool1_mask", "pool1_D"))
    x = cbr("conv1_2_D", "pool1_D", c1)
    out.append(_conv("conv1_1_D", x, "conv1_1_D", classes, 3, 1))
    out.append(_softmax("conv1_1_D"))
    return "".join(out)


def basic_prototxt(T, H=352, W=1024, classes=15, width=64):
    """SegNet-Basic (LRN + 4 enc conv7x7+ReLU+pool, 4 dec unpool+conv7x7, 1x1 classifier, 4 dropouts)."""
    out = [f'name: "bayesian_segnet_basic"\ninput: "data"\ninput_dim: {T}\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n']
    out.append('layer {\n  name: "norm"\n  type: "LRN"\n  bottom: "data"\n  top: "norm"\n'
               '  lrn_param {\n    local_size: 5\n    alpha: 9.99999974738e-05\n    beta: 0.75\n  }\n}\n')
    x = "norm"
    for i in (1, 2, 3, 4):
        out.append(_conv(f"conv{i}", x, f"conv{i}", width, 7, 3))
        out.append(_relu(f"relu{i}", f"conv{i}"))
        out.append(_pool(f"pool{i}", f"conv{i}"))
        x = f"pool{i}"
        if i >= 3:
            out.append(_drop(f"encdrop{i}", x))
    for i in (4, 3, 2, 1):
        out.append(_up(f"upsample{i}", x, f"pool{i}_mask", f"upsample{i}"))
        out.append(_conv(f"conv_decode{i}", f"upsample{i}", f"conv_decode{i}", width, 7, 3))
        x = f"conv_decode{i}"
        if i >= 3:
            out.append(_drop(f"decdrop{i}", x))
    out.append(_conv("dense_softmax_inner_prod", x, "dense_softmax_inner_prod", classes, 1, 0))
    out.append(_softmax("dense_softmax_inner_prod"))
    return "".join(out)


def tiny_prototxt(T, H=32, W=64, classes=15):
    """A 2-level SegNet-shaped net (conv/BN/ReLU/pool/dropout/unpool/LRN/softmax all present)
    for fast parity tests; same layer semantics, not a reference model."""
    out = [f'name: "tiny_segnet"\ninput: "data"\ninput_dim: {T}\ninput_dim: 3\ninput_dim: {H}\ninput_dim: {W}\n']
    out.append('layer {\n  name: "norm"\n  type: "LRN"\n  bottom: "data"\n  top: "norm"\n'
               '  lrn_param {\n    local_size: 5\n    alpha: 0.0001\n    beta: 0.75\n  }\n}\n')
    out.append(_conv("c1", "norm", "c1", 16, 3, 1)); out.append(_bn("c1_bn", "c1")); out.append(_relu("r1", "c1"))
    out.append(_pool("p1", "c1"))
    out.append(_conv("c2", "p1", "c2", 24, 3, 1)); out.append(_bn("c2_bn", "c2")); out.append(_relu("r2", "c2"))
    out.append(_pool("p2", "c2")); out.append(_drop("p2_drop", "p2"))
    out.append(_up("u2", "p2", "p2_mask", "p2_D"))
    out.append(_conv("d2", "p2_D", "d2", 16, 7, 3)); out.append(_relu("rd2", "d2"))
    out.append(_drop("d2_drop", "d2")); out.append(_up("u1", "d2", "p1_mask", "p1_D"))
    out.append(_conv("d1", "p1_D", "d1", 16, 3, 1)); out.append(_bn("d1_bn", "d1")); out.append(_relu("rd1", "d1"))
    out.append(_conv("cls", "d1", "cls", classes, 1, 0))
    out.append(_softmax("cls"))
    return "".join(out)


def parse_layers(text):
    """Layer list of a (generated or reference) prototxt: dicts with name / type / bottom / top / num_output /
    kernel_size — what weights.param_shapes needs to size the parameter array.  The C++ reader in csrc/prototxt.cpp is
    the one the library itself uses; this is its small Python twin for callers that prepare weights."""
    import re
    text = re.sub(r"#[^\n]*", "", text)
    layers = []
    i = 0
    while True:
        m = re.compile(r"\blayer\s*\{").search(text, i)
        if not m:
            break
        depth, j = 1, m.end()
        while depth and j < len(text):
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            j += 1
        body = text[m.end():j - 1]
        get = lambda k: re.findall(rf'\b{k}\s*:\s*"?([^"\s]+)"?', body)
        L = {"name": (get("name") or [""])[0], "type": (get("type") or [""])[0], "bottom": get("bottom"), "top": get("top")}
        if get("num_output"): L["num_output"] = int(get("num_output")[0])
        if get("kernel_size"): L["kernel_size"] = int(get("kernel_size")[0])
        layers.append(L)
        i = j
    return layers
