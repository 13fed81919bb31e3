"""GPU: the direct 3x3 convolution on the fp16 matrix cores (f16x3; sivo_amd/csrc/conv3_h3.hip) against an fp64 evaluation
of the same convolution (nine shifted fp64 matrix products on the device), through the C ABI (sivo_debug_conv3_h3_dev).

Bound: every fp32 operand is hi + lo to 2^-22, the three products are exact in fp32, the accumulation is fp32: the error of
an output is a small multiple of 2^-24 * sum |w| |x| over its 9 Cin terms; the tests require 2^-20 of that sum (the bound
of tests/test_gpu_h3_gemm.py) and print what was reached.  Shapes cover whole and partial items (8 x 64 output pixels) in
both directions, one and two 64-cout groups, 2 .. 8 channel chunks, the layer reading through an Upsample (pooled input +
window codes), large activations with their power-of-two scale, the overflow flag, and the three network shapes the kernel
runs at (with their launch times)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref64(x, w):
    """fp64 'same' 3x3 cross-correlation of x (N, C, H, W) with w (K, C, 3, 3), and the same of the absolute values."""
    N, Cc, H, W = x.shape
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    ap = xp.abs()
    wd = w.double()
    out = torch.zeros((N, w.shape[0], H, W), dtype=torch.float64, device=x.device)
    mag = torch.zeros_like(out)
    for ky in range(3):
        for kx in range(3):
            out += torch.einsum("kc,nchw->nkhw", wd[:, :, ky, kx], xp[:, :, ky:ky + H, kx:kx + W])
            mag += torch.einsum("kc,nchw->nkhw", wd[:, :, ky, kx].abs(), ap[:, :, ky:ky + H, kx:kx + W])
    return out, mag


def _unpool(pooled, mask):
    N, Cc, h, w = pooled.shape
    up = torch.zeros((N, Cc, 2 * h, 2 * w), dtype=pooled.dtype, device=pooled.device)
    for dy in range(2):
        for dx in range(2):
            up[:, :, dy::2, dx::2] = torch.where(mask == dy * 2 + dx, pooled, torch.zeros_like(pooled))
    return up


def _case(N, Cin, Cout, H, W, relu, unpool, seed, amp=1.0, nonneg=False):
    from sivo_amd import segnet
    g = torch.Generator(device="cuda").manual_seed(seed)
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = torch.randn((N, Cin, h, w), generator=g, device="cuda", dtype=torch.float32) * amp
    if nonneg:
        x = x.clamp_min(0)
    mask = torch.randint(0, 4, (N, Cin, h, w), generator=g, device="cuda", dtype=torch.uint8) if unpool else None
    rng = np.random.default_rng(seed)
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, Cout).astype(np.float32)
    out, _, ov = segnet.conv3_h3(x, wt, scale, shift, relu=relu, mask=mask)
    assert not ov
    full = _unpool(x, mask) if unpool else x
    ref, mag = _ref64(full, torch.from_numpy(wt).cuda())
    sc = torch.from_numpy(scale).cuda().double()[None, :, None, None]
    sh = torch.from_numpy(shift).cuda().double()[None, :, None, None]
    ref = ref * sc + sh
    if relu:
        ref = ref.clamp_min(0)
    err = (out.double() - ref).abs()
    bound = mag * sc.abs() + sh.abs()
    worst = float((err / bound.clamp_min(1e-30)).max())
    return worst, float(err.max()), float(ref.abs().max())


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,unpool", [
    (2, 32, 64, 8, 64, False, False),        # exactly one item per sample, two chunks
    (1, 64, 64, 20, 100, True, False),       # partial items in both directions
    (3, 128, 64, 16, 128, True, False),      # conv2_1_D's channels
    (2, 64, 128, 24, 72, False, False),      # two cout groups
    (2, 64, 64, 16, 128, True, True),        # through an Upsample (conv1_2_D's form)
    (1, 128, 128, 12, 40, True, True),       # through an Upsample, partial items, two cout groups (conv2_2_D's form)
    (5, 48, 64, 10, 66, False, False),       # three chunks, one column and two rows beyond an item
])
def test_direct_f16x3_convolution_against_fp64(N, Cin, Cout, H, W, relu, unpool):
    worst, emax, rmax = _case(N, Cin, Cout, H, W, relu, unpool, seed=N * 1000 + Cin + H)
    print(f"[{N}x{Cin}->{Cout} {H}x{W} relu={relu} unpool={unpool}] worst |err| / sum|w||x| = {worst / 2.0 ** -24:.2f} x 2^-24, max |err| {emax:.2e}, max |out| {rmax:.2f}")
    assert worst <= 2.0 ** -20


@pytest.mark.parametrize("unpool", [False, True])
def test_the_two_stage_loops_agree_bit_for_bit(unpool, monkeypatch):
    """FORM 1 (default: loads, split and DMA dealt over the MFMA slots, patch two stages ahead) against FORM 0 (phased): the same
    products in the same order, so every output bit must agree — at a shape with several items per workgroup, partial
    items and two cout groups."""
    from sivo_amd import segnet
    g = torch.Generator(device="cuda").manual_seed(77)
    N, Cin, Cout, H, W = 6, 64, 128, 180, 520
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = torch.randn((N, Cin, h, w), generator=g, device="cuda", dtype=torch.float32)
    mask = torch.randint(0, 4, (N, Cin, h, w), generator=g, device="cuda", dtype=torch.uint8) if unpool else None
    rng = np.random.default_rng(5)
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, Cout).astype(np.float32)
    outs = []
    from sivo_amd import _lib
    for form in ("0", "1"):
        monkeypatch.setenv("SIVO_D3_FORM", form)          # (a switch of the diagnostic build: libsivo_hip_diag.so, the same sources with -DSIVO_DIAG)
        with _lib.use("diag"):
            out, _, ov = segnet.conv3_h3(x, wt, scale, shift, relu=False, mask=mask)
        assert not ov
        outs.append(out)
    ref, _, _ = segnet.conv3_h3(x, wt, scale, shift, relu=False, mask=mask)       # the product library's kernel
    assert torch.equal(outs[1], ref)
    assert torch.equal(outs[0], outs[1])


def test_large_activations_and_the_overflow_flag():
    from sivo_amd import segnet
    worst, emax, rmax = _case(2, 64, 64, 16, 64, True, False, seed=5, amp=700.0, nonneg=True)
    print(f"[amp 700] worst {worst / 2.0 ** -24:.2f} x 2^-24, max |out| {rmax:.1f}")
    assert worst <= 2.0 ** -20
    x = torch.full((1, 32, 8, 64), 3.0, device="cuda")
    x[0, 17, 3, 5] = 1000.0
    wt = np.zeros((64, 32, 3, 3), np.float32)
    one = np.ones(64, np.float32)
    _, _, ov = segnet.conv3_h3(x, wt, one, one * 0, vscale=64.0)          # 1000 * 64 < 65504
    assert not ov
    _, _, ov = segnet.conv3_h3(x, wt, one, one * 0, vscale=128.0)         # 1000 * 128 > 65504
    assert ov


@pytest.mark.parametrize("name,N,Cin,Cout,H,W,unpool", [
    ("conv1_2_D", 12, 64, 64, 352, 1024, True),
    ("conv2_1_D", 12, 128, 64, 176, 512, False),
    ("conv2_2_D", 12, 128, 128, 176, 512, True),
    ("conv1_2 (prefix)", 1, 64, 64, 352, 1024, False),
])
def test_network_shapes_at_full_size(name, N, Cin, Cout, H, W, unpool):
    """The layers of SegNet-Standard at 352 x 1024 the kernel is used for, T = 12: every output of samples 0 and N - 1 against
    fp64, and the launch time."""
    from sivo_amd import segnet
    g = torch.Generator(device="cuda").manual_seed(11)
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = torch.randn((N, Cin, h, w), generator=g, device="cuda", dtype=torch.float32).clamp_min(0) * 3.0
    mask = torch.randint(0, 4, (N, Cin, h, w), generator=g, device="cuda", dtype=torch.uint8) if unpool else None
    rng = np.random.default_rng(3)
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, Cout).astype(np.float32)
    out, ms, ov = segnet.conv3_h3(x, wt, scale, shift, relu=True, mask=mask, iters=10)
    assert not ov
    sc = torch.from_numpy(scale).cuda().double()[None, :, None, None]
    sh = torch.from_numpy(shift).cuda().double()[None, :, None, None]
    worst = 0.0
    for n in sorted({0, N - 1}):
        xs = x[n:n + 1]
        full = _unpool(xs, mask[n:n + 1]) if unpool else xs
        ref, mag = _ref64(full, torch.from_numpy(wt).cuda())
        ref = (ref * sc + sh).clamp_min(0)
        err = (out[n:n + 1].double() - ref).abs()
        worst = max(worst, float((err / (mag * sc.abs() + sh.abs()).clamp_min(1e-30)).max()))
    flops = 2.0 * 9 * Cin * Cout * H * W * N
    print(f"[{name} {N}x{Cin}->{Cout} {H}x{W}] {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s algorithmic ({3 * flops / ms / 1e9 / 2500:.2f} of the fp16 peak executed); worst {worst / 2.0 ** -24:.2f} x 2^-24")
    assert worst <= 2.0 ** -20
