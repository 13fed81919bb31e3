"""A weight family that mimics a TRAINED VGG16-SegNet (test data; uses the CPU oracle's primitives for its calibration pass).

The reference's `.caffemodel` files are Git-LFS pointers here (config/bayesian_segnet/standard/kitti/*.caffemodel): real trained weights
never passed through the path.  What the f16x3 scales, the F(4x4) margin and the accuracy guard have to cope with in a trained net,
and what He-normal synthetic weights do not have:
  * per-layer weight magnitudes that fall with depth (VGG16's published per-layer standard deviations, approximate: the ImageNet
    model's conv1_1 filters are ~25x larger than its conv4 / conv5 filters), heavy tails (Student-t, 5 degrees of freedom), filters of
    very different norm (log-normal, sigma 0.5) and a few dead filters (3 % at 1e-2 of the norm);
  * BN layers that normalise each channel's ACTUAL statistics and then apply a learned gain spread over two decades (log-uniform
    [0.1, 10], normalised to unit mean square) and a negative offset (-0.5 gain): ~70 % of the activations behind a ReLU are zero,
    channels of very different magnitude meet in every contraction.
The BN statistics come from a calibration pass on a small image with the oracle's layer primitives (what training's running
averages are), so activations stay O(1) through all 26 convolutions whatever the weight magnitudes.
"""
import numpy as np

from oracle import oracle as O
from sivo_amd import weights as wts

# approximate standard deviation of the 3x3 filters of the ImageNet VGG16 (the encoder SegNet is initialised from); decoder layer
# conv*_D mirrors its encoder twin, the classifier conv1_1_D takes conv1_2's
VGG16_STD = {"conv1_1": 0.206, "conv1_2": 0.042, "conv2_1": 0.032, "conv2_2": 0.024, "conv3_1": 0.017, "conv3_2": 0.012, "conv3_3": 0.013,
             "conv4_1": 0.010, "conv4_2": 0.0076, "conv4_3": 0.0077, "conv5_1": 0.0086, "conv5_2": 0.0087, "conv5_3": 0.0093}


def _std_of(name):
    base = name[:-2] if name.endswith("_D") else name
    return VGG16_STD.get(base, 0.02)


def trained_like_weights(net, calib_bgr, seed=7, calib_seed=5):
    """{layer: [blobs]} for the parsed prototxt `net` (any geometry).  calib_bgr: (h, w, 3) uint8 image of the family the net will see —
    the BN statistics are those of this image (h, w multiples of 32), as training's running averages are those of the training set."""
    rng = np.random.default_rng(seed)
    layers = net["layers"]
    w = {}
    for name, shapes in wts.param_shapes(layers):
        if len(shapes[0]) == 4:
            co, ci, k, _ = shapes[0]
            t = rng.standard_t(5, shapes[0]) / np.sqrt(5.0 / 3.0)                      # unit variance, heavy tails
            norm = np.exp(rng.normal(0.0, 0.5, (co, 1, 1, 1)))
            dead = rng.random((co, 1, 1, 1)) < 0.03
            norm = np.where(dead, 1e-2 * norm, norm)
            W = (t * norm * _std_of(name)).astype(np.float32)
            b = (0.1 * _std_of(name) * rng.standard_normal(shapes[1])).astype(np.float32)
            w[name] = [W, b]
        else:
            w[name] = [np.ones(shapes[0], np.float32), np.zeros(shapes[1], np.float32)]
    # calibration pass: every BN normalises the statistics its input really has, then gain and offset
    h, wd = calib_bgr.shape[:2]
    blob = O.preprocess(np.ascontiguousarray(calib_bgr, np.uint8), 1, h, wd)
    blobs = {net["input"]: blob}
    site = 0
    for L in layers:
        t = L["type"]
        bot = [blobs[b] for b in L["bottom"]]
        if t == "Convolution":
            out = O.conv2d(bot[0], w[L["name"]][0], w[L["name"]][1], L["pad"])
        elif t == "BN":
            x = bot[0].astype(np.float64)
            mu = x.mean(axis=(0, 2, 3)); sd = x.std(axis=(0, 2, 3)) + 1e-12
            g = np.exp(rng.uniform(np.log(0.1), np.log(10.0), mu.shape)) / np.sqrt(10.86)      # E[g^2] = 1
            s = g / sd
            w[L["name"]] = [s.astype(np.float32), (-mu * s - 0.5 * g).astype(np.float32)]
            out = O.bn_inference(bot[0], *w[L["name"]])
        elif t == "ReLU":
            out = O.relu(bot[0])
        elif t == "Pooling":
            out, mask = O.maxpool(bot[0], L["kernel_size"], L["stride"])
            blobs[L["top"][1]] = mask
        elif t == "Upsample":
            out = O.unpool(bot[0], bot[1], bot[0].shape[2] * L["scale"], bot[0].shape[3] * L["scale"])
        elif t == "Dropout":
            out = O.dropout(bot[0], site, 0, calib_seed, L["dropout_ratio"]) if L["sample_weights_test"] else bot[0]
            site += 1
        elif t == "Softmax":
            out = bot[0]
        else:
            raise ValueError("unsupported layer type " + t)
        blobs[L["top"][0]] = out
    # the classifier has no BN behind it: scale it so that the logits are O(10) like a trained net's
    last_conv = [L for L in layers if L["type"] == "Convolution"][-1]
    lg = blobs[last_conv["top"][0]]
    k = np.float32(8.0 / max(float(np.abs(lg).max()), 1e-6))
    w[last_conv["name"]][0] *= k; w[last_conv["name"]][1] *= k
    return w
