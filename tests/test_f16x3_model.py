"""CPU: the arithmetic of the two f16x3 kernels (conv_wino4_h3.hip, conv3_h3.hip), stated in numpy — what the GPU tests measure
on the device (tests/test_gpu_h3_gemm.py, tests/test_gpu_conv3_h3.py) must already hold for the model:

  x -> x * 2^k (k per layer: max |x| 2^k in [2^7, 2^8)), hi = fp16(xs), lo = fp16(xs - hi);   xs = hi + lo to 2^-22 |xs|
  x y = x_lo y_hi + x_hi y_lo + x_hi y_hi    (each product of two fp16 values is exact in fp32; fp32 accumulation)

and the error of a sum of such products stays below 2^-20 of sum |x||y| — the bound the device tests use."""
import numpy as np


def split(x, scale):
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi, lo, xs


def pow2_scale(x):
    m = float(np.abs(x).max())
    return 2.0 ** (8 - np.frexp(m)[1]) if m > 0 else 1.0


def test_hi_plus_lo_is_the_value_to_2_pow_minus_22():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-6, 6, 200000))).astype(np.float32)
    s = pow2_scale(x)
    hi, lo, xs = split(x, s)
    assert np.abs(xs).max() < 256 and np.isfinite(hi).all()
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    big = np.abs(xs) >= 2.0 ** -10 * np.abs(xs).max()         # down to 2^-10 of the maximum: full precision of the pair
    assert (np.abs(rec - xs)[big] <= 2.0 ** -22 * np.abs(xs)[big]).all()
    # below that the low half runs into fp16's subnormal spacing: an absolute error of at most 2^-25 (scaled units)
    assert np.abs(rec - xs).max() <= max(2.0 ** -25, 2.0 ** -22 * 256)
    assert (np.abs(rec - xs)[~big] <= 2.0 ** -25 + 2.0 ** -22 * np.abs(xs)[~big]).all()


def test_three_products_reproduce_an_fp32_product():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(100000).astype(np.float32)
    y = (rng.standard_normal(100000) * 0.05).astype(np.float32)
    sx, sy = pow2_scale(x), pow2_scale(y)
    xh, xl, _ = split(x, sx)
    yh, yl, _ = split(y, sy)
    f = np.float32
    # every partial product is exact in fp32 (11 x 11 significand bits), so only their sum rounds
    for a, b in ((xl, yh), (xh, yl), (xh, yh)):
        p32 = a.astype(f) * b.astype(f)
        assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))
    got = ((xl.astype(f) * yh.astype(f) + xh.astype(f) * yl.astype(f)) + xh.astype(f) * yh.astype(f)).astype(np.float64) / (sx * sy)
    ref = x.astype(np.float64) * y.astype(np.float64)
    assert np.abs(got - ref).max() <= 2.0 ** -21 * np.abs(ref).max()
    nz = np.abs(ref) > 1e-3 * np.abs(ref).max()
    assert (np.abs(got - ref)[nz] <= 2.0 ** -20 * np.abs(ref)[nz]).all()


def test_direct_3x3_convolution_in_the_model_meets_the_device_bound():
    """conv3_h3.hip in numpy at a small size: taps in order, 16-channel stages, the three terms (lo, hi) (hi, lo) (hi, hi) of a
    stage added to an fp32 accumulator one after the other (the MFMA's own k-reduction is exact to fp32 rounding of the
    16-term sum, modelled here by a float64 sum rounded once)."""
    rng = np.random.default_rng(3)
    C, K, H, W = 32, 8, 10, 12
    x = np.maximum(rng.standard_normal((C, H, W)), 0).astype(np.float32) * 3
    w = (rng.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)
    sx, sw = pow2_scale(x), pow2_scale(w)
    xh, xl, _ = split(np.pad(x, ((0, 0), (1, 1), (1, 1))), sx)
    wh, wl, _ = split(w, sw)
    acc = np.zeros((K, H, W), np.float32)
    for c0 in range(0, C, 16):
        for ky in range(3):
            for kx in range(3):
                for a, b in ((wl, xh), (wh, xl), (wh, xh)):
                    part = np.einsum("kc,chw->khw", a[:, c0:c0 + 16, ky, kx].astype(np.float64),
                                     b[c0:c0 + 16, ky:ky + H, kx:kx + W].astype(np.float64))
                    acc = (acc.astype(np.float64) + part).astype(np.float32)
    got = acc.astype(np.float64) / (sx * sw)
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    ref = np.zeros((K, H, W))
    mag = np.zeros((K, H, W))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("kc,chw->khw", w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + H, kx:kx + W])
            mag += np.einsum("kc,chw->khw", np.abs(w[:, :, ky, kx]).astype(np.float64), np.abs(xp[:, ky:ky + H, kx:kx + W]))
    worst = float((np.abs(got - ref) / mag).max())
    assert worst <= 2.0 ** -20, worst / 2.0 ** -24
    # and it is no worse than the sequential fp32 chain the direct fp32 kernels compute
    chain = np.zeros((K, H, W), np.float32)
    xp32 = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for c in range(C):
        for ky in range(3):
            for kx in range(3):
                chain = chain + w[:, c, ky, kx][:, None, None] * xp32[c, ky:ky + H, kx:kx + W][None]
    assert np.abs(got - ref).max() <= 2.0 * np.abs(chain.astype(np.float64) - ref).max() + 1e-9
