"""GPU: the packed activation forms of the direct f16x3 convolution (sivo_amd/csrc/conv3_h3.hip IN_PK / IN_PK_UNPOOL / OUT_PK,
pk_format.hip) against its fp32-blob form, through the C ABI of the test library (sivo_debug_conv3_h3_pk_dev).

The packed format holds exactly what the fp32 form's patch staging computes (x * scale, hi = fp16, lo = fp16(rest)), so:
  * packed INPUT (by LDS-DMA; through an Upsample: pooled pieces & the per-octet window masks) must give every output BIT
    FOR BIT what the fp32-input kernel gives on the same tensor;
  * packed OUTPUT must be bit for bit the split of the fp32 kernel's output with the consumer's scale — compared after
    unpacking, (hi + lo) / scale — and must leave the zero border of its planes untouched;
and the fp32 form itself is checked against fp64 in tests/test_gpu_conv3_h3.py.  Shapes: whole and partial items in both
directions, one and two cout groups, 2 .. 8 channel chunks, planes padded beyond what the tiling needs; then the three
decoder layers of SegNet-Standard at 352 x 1024, T = 12, with launch times of both forms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _split_roundtrip(v, scale):
    """(hi + lo) / scale of v * scale split as fp16 hi + fp16 lo — what pk_unpack returns for a packed tensor."""
    xs = v * scale
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return (hi.float() + lo.float()) / scale


def _inputs(N, Cin, Cout, H, W, unpool, seed, amp=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    h, w = (H // 2, W // 2) if unpool else (H, W)
    x = (torch.randn((N, Cin, h, w), generator=g, device="cuda", dtype=torch.float32) * amp).clamp_min(-0.5 * amp)
    mask = torch.randint(0, 4, (N, Cin, h, w), generator=g, device="cuda", dtype=torch.uint8) if unpool else None
    rng = np.random.default_rng(seed)
    wt = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, Cout).astype(np.float32)
    return x, mask, wt, scale, shift


SHAPES = [
    (2, 32, 64, 8, 64, False, False),        # exactly one item per sample, two chunks
    (1, 64, 64, 20, 100, True, False),       # partial items in both directions
    (3, 128, 64, 16, 128, True, False),      # conv2_1_D's channels
    (2, 64, 128, 24, 72, False, False),      # two cout groups
    (2, 64, 64, 16, 128, True, True),        # through an Upsample (conv1_2_D's form)
    (1, 128, 128, 12, 40, True, True),       # through an Upsample, partial items, two cout groups (conv2_2_D's form)
    (5, 48, 64, 10, 66, False, False),       # three chunks, one column and two rows beyond an item
    (2, 32, 64, 22, 130, True, True),        # through an Upsample, three item columns, last one two pixels wide
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,unpool", SHAPES)
@pytest.mark.parametrize("extra_pad", [False, True])
def test_packed_input_is_bit_identical_to_the_fp32_form(N, Cin, Cout, H, W, relu, unpool, extra_pad):
    from sivo_amd import segnet
    x, mask, wt, scale, shift = _inputs(N, Cin, Cout, H, W, unpool, seed=N * 1000 + Cin + H)
    ref, _, ov = segnet.conv3_h3(x, wt, scale, shift, relu=relu, mask=mask)
    assert not ov
    out, _, ov, dirty = segnet.conv3_h3_pk(x, wt, scale, shift, relu=relu, mask=mask, pk_in=True, pk_out=False, extra_pad=extra_pad)
    assert not ov and not dirty
    nbad = int((out != ref).sum())
    print(f"[{N}x{Cin}->{Cout} {H}x{W} unpool={unpool} pad={extra_pad}] packed input: {nbad} of {ref.numel()} outputs differ, max |d| {float((out - ref).abs().max()):.3e}")
    assert torch.equal(out, ref)


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,unpool", SHAPES)
@pytest.mark.parametrize("pk_in", [False, True])
def test_packed_output_is_the_split_of_the_fp32_output(N, Cin, Cout, H, W, relu, unpool, pk_in):
    from sivo_amd import segnet
    x, mask, wt, scale, shift = _inputs(N, Cin, Cout, H, W, unpool, seed=N * 1000 + Cin + H + 1)
    if unpool and not pk_in:
        # Not a plan shape: the planner packs a direct layer's output only when that layer does not read fp32 through an Upsample
        # (segnet_plan.cpp, `a_direct`: !(A.unpool_in >= 0 && !A.pk_in)); the launcher refuses the combination instead of running another form
        with pytest.raises(Exception, match="not built"):
            segnet.conv3_h3_pk(x, wt, scale, shift, relu=relu, mask=mask, pk_in=False, pk_out=True, out_vscale=1.0, extra_pad=True)
        return
    ref, _, ov = segnet.conv3_h3(x, wt, scale, shift, relu=relu, mask=mask)
    assert not ov
    out_vscale = float(2.0 ** (8 - np.frexp(float(ref.abs().max()))[1]))
    out, _, ov, dirty = segnet.conv3_h3_pk(x, wt, scale, shift, relu=relu, mask=mask, pk_in=pk_in, pk_out=True, out_vscale=out_vscale,
                                           extra_pad=True)
    assert not ov and not dirty
    want = _split_roundtrip(ref, out_vscale)
    nbad = int((out != want).sum())
    print(f"[{N}x{Cin}->{Cout} {H}x{W} unpool={unpool} pk_in={pk_in}] packed output: {nbad} of {ref.numel()} differ, max |d| {float((out - want).abs().max()):.3e}; "
          f"split error {float(((want - ref).abs() / ref.abs().clamp_min(1e-20)).max()) / 2.0 ** -22:.2f} x 2^-22 relative")
    assert torch.equal(out, want)


def test_packed_output_raises_the_overflow_flag():
    """The consumer of a packed tensor no longer sees the fp32 values: the producer's output stage is where a value that
    leaves the fp16 range must raise the flag."""
    from sivo_amd import segnet
    x = torch.full((1, 32, 8, 64), 3.0, device="cuda")
    wt = np.zeros((64, 32, 3, 3), np.float32)
    wt[5, 7, 1, 1] = 1.0
    one = np.ones(64, np.float32)
    for pk_in in (False, True):
        _, _, ov, _ = segnet.conv3_h3_pk(x, wt, one, one * 0, relu=False, pk_in=pk_in, pk_out=True, vscale=8.0, out_vscale=16384.0)      # 3 * 16384 < 65504
        assert not ov
        _, _, ov, _ = segnet.conv3_h3_pk(x, wt, one, one * 0, relu=False, pk_in=pk_in, pk_out=True, vscale=8.0, out_vscale=32768.0)      # 3 * 32768 > 65504
        assert ov


@pytest.mark.parametrize("name,N,Cin,Cout,H,W,unpool,pk_out", [
    ("conv2_2_D", 12, 128, 128, 176, 512, True, True),
    ("conv2_1_D", 12, 128, 64, 176, 512, False, True),
    ("conv1_2_D", 12, 64, 64, 352, 1024, True, False),
])
def test_decoder_layers_at_full_size(name, N, Cin, Cout, H, W, unpool, pk_out):
    """The three decoder layers the packed chain covers, in the forms the network runs them (T = 12): samples 0 and N - 1
    bit for bit against the fp32 form, and the launch times of both."""
    from sivo_amd import segnet
    x, mask, wt, scale, shift = _inputs(N, Cin, Cout, H, W, unpool, seed=11, amp=3.0)
    x = x.clamp_min(0)
    ref, ms_ref, ov = segnet.conv3_h3(x, wt, scale, shift, relu=True, mask=mask, iters=10)
    assert not ov
    out_vscale = float(2.0 ** (8 - np.frexp(float(ref.abs().max()))[1]))
    out, ms, ov, dirty = segnet.conv3_h3_pk(x, wt, scale, shift, relu=True, mask=mask, pk_in=True, pk_out=pk_out, out_vscale=out_vscale, iters=10)
    assert not ov and not dirty
    flops = 2.0 * 9 * Cin * Cout * H * W * N
    print(f"[{name} {N}x{Cin}->{Cout} {H}x{W}] fp32 blobs {ms_ref:.3f} ms, packed {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s algorithmic "
          f"({3 * flops / ms / 1e9 / 2500:.2f} of the fp16 peak executed)")
    for n in sorted({0, N - 1}):
        want = _split_roundtrip(ref[n], out_vscale) if pk_out else ref[n]
        assert torch.equal(out[n], want), (name, n, int((out[n] != want).sum()))


def test_the_network_with_packed_activations_is_bit_identical_to_fp32_blobs(monkeypatch):
    """SegNet-Standard at full channel widths on a 64 x 128 image, T = 3: the handle that hands its direct f16x3 layers packed
    activations (conv3_1_D -> conv2_2_D -> conv2_1_D -> conv1_2_D through two Upsamples, and the encoder's conv -> conv pairs)
    against a handle built with SIVO_D3_PK=0 (fp32 blobs everywhere): every logit bit for bit, the maps within rounding, and the packed
    blobs — unpacked by sivo_segnet_blob — within 2^-21 of the fp32 ones."""
    from oracle import prototxt as oproto
    from sivo_amd import netspec, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    T, H, W = 3, 64, 128
    text = netspec.standard_prototxt(T, H, W)
    net = oproto.parse(text)
    w = wts.pack(net["layers"], wts.synth_weights(net["layers"], 42))
    img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    got = {}
    for pk in ("1", "0"):
        monkeypatch.setenv("SIVO_D3_PK", pk)
        sn = BayesianSegNet(prototxt=text, weights=w, T=T)
        _, logits, _ = sn.forward(img, 7, want_logits=True)
        maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
                torch.empty((H, W), dtype=torch.float64, device="cuda"))
        sn.segment_into(img, 7, maps)
        torch.cuda.synchronize()
        assert sn.gemm_status()[:2] == (2, 0)
        got[pk] = (logits.clone(), [m.clone() for m in maps], {n: sn.blob(n) for n in ("conv3_1_D", "conv2_2_D", "conv2_1_D", "conv2_1", "conv3_2")})
    assert torch.equal(got["1"][0], got["0"][0]), int((got["1"][0] != got["0"][0]).sum())
    # the maps of the segment entry point: with packed activations the classifier itself runs on the fp16 matrix cores
    # (conv_cls_h3.hip, direct f16x3) instead of Winograd F(2x2) on the fp32 pipe — different roundings of the same logits
    (c1, f1, e1), (c0, f0, e0) = got["1"][1], got["0"][1]
    print(f"[maps] classes differ at {int((c1 != c0).sum())} of {c1.numel()} pixels, max |d confidence| {float((f1 - f0).abs().max()):.2e}, max |d entropy| {float((e1 - e0).abs().max()):.2e}")
    assert int((c1 != c0).sum()) <= c1.numel() // 500 and float((f1 - f0).abs().max()) < 1e-4 and float((e1 - e0).abs().max()) < 5e-4
    for name, a in got["1"][2].items():
        b = got["0"][2][name]
        rel = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
        print(f"[{name}] packed blob vs fp32 blob: max |d| / max |x| = {rel / 2.0 ** -22:.3f} x 2^-22")
        assert rel <= 2.0 ** -21


@pytest.mark.parametrize("T,Cin,classes,H,W", [(2, 64, 15, 8, 32), (3, 64, 15, 20, 72), (2, 32, 11, 10, 24), (4, 96, 16, 16, 40), (12, 64, 15, 352, 1024)])
def test_classifier_on_the_fp16_matrix_cores(T, Cin, classes, H, W):
    """conv_cls_h3.hip alone: the 3x3 classifier on f16x3 from a packed input + Softmax + f64 mean + maps.  Its logits against
    an fp64 convolution (bound as for the direct kernel: 2^-20 of sum |w||x|), its maps bit for bit the post-processing kernel's
    on exactly these logits; whole and ragged tiles, 1 - 3 k-steps, 11 / 15 / 16 classes; the network's shape with its launch time."""
    from sivo_amd import segnet
    g = torch.Generator(device="cuda").manual_seed(T * 100 + H)
    x = (torch.randn((T, Cin, H, W), generator=g, device="cuda", dtype=torch.float32) * 2.0).clamp_min(0)
    rng = np.random.default_rng(H + W)
    wt = (rng.standard_normal((classes, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, classes).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, classes).astype(np.float32)
    big = H * W > 100000
    logits, maps, ms = segnet.conv_cls_h3(x, wt, scale, shift, iters=10 if big else 0)
    post = segnet.mc_segment(logits)
    torch.cuda.synchronize()
    for a, b in zip(maps, post):
        assert torch.equal(a, b)
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    wd = torch.from_numpy(wt).cuda().double()
    worst = 0.0
    for n in sorted({0, T - 1}):
        ref = torch.zeros((classes, H, W), dtype=torch.float64, device="cuda")
        mag = torch.zeros_like(ref)
        for ky in range(3):
            for kx in range(3):
                ref += torch.einsum("kc,chw->khw", wd[:, :, ky, kx], xp[n, :, ky:ky + H, kx:kx + W])
                mag += torch.einsum("kc,chw->khw", wd[:, :, ky, kx].abs(), xp[n, :, ky:ky + H, kx:kx + W].abs())
        sc = torch.from_numpy(scale).cuda().double()[:, None, None]
        sh = torch.from_numpy(shift).cuda().double()[:, None, None]
        err = (logits[n].double() - (ref * sc + sh)).abs()
        worst = max(worst, float((err / (mag * sc.abs() + sh.abs()).clamp_min(1e-30)).max()))
    line = f"[cls {T}x{Cin}->{classes} {H}x{W}] worst |err| / sum|w||x| = {worst / 2.0 ** -24:.2f} x 2^-24"
    if ms:
        line += f"; {ms:.3f} ms = {T * Cin * H * W * 4 / ms / 1e9:.2f} TB/s of input, {2.0 * 9 * Cin * 16 * H * W * T * 3 / ms / 1e9:.0f} TFLOP/s executed"
    print(line)
    assert worst <= 2.0 ** -20
