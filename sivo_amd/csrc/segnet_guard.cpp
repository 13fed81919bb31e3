// segnet_guard.cpp — what guards the f16x3 arithmetic of a handle: the fp16 RANGE guard (calibrated power-of-two scales, the overflow
// flag protocol: DESIGN 3.2) and the load-time ACCURACY guard (per-layer comparison with the direct fp32 kernel, rerouting: DESIGN 3.4).
// The reference computes in fp32 throughout (Caffe, src/bayesian_segnet/bayesian_segnet.cpp:310); these guards are what lets the fast
// kernels stand behind it within the 1e-3 logit tolerance for weights nobody has seen.
#include "segnet_impl.hpp"

namespace sivo {

// The kernels of an f16x3 layer store 1 into the pinned flag word when a value times the layer's scale leaves the fp16 range.
// The frame that raised it is wrong (inf / NaN in that layer).  What follows (h3_back_off): the NEXT forward of the handle runs
// without f16x3 (bf16x6 / fp32 kernels: fp32's range) — that is the recomputation of the frame, which the synchronous entry points
// do before they return and a caller of the asynchronous ones does after sivo_segnet_take_overflow told it to — and every
// f16x3 scale of the handle is lowered by 2^2: two more bits of headroom for two bits of the lo plane (2^-20 instead of 2^-22
// relative; still below the fp32 FMA chain's own error).  The fourth such frame switches f16x3 off for good: activations that
// outgrow 2^14 times the calibration's are not what the scales were made for.
bool h3_flag_take(sivo_segnet &S) {
    if (!S.h3_flag || !*S.h3_flag) return false;
    *S.h3_flag = 0;
    ++S.h3_overflow_frames;
    return true;
}
void h3_back_off(sivo_segnet &S) {
    S.h3_pause = true;
    if (++S.h3_back_offs > 3) { S.h3_on = false; return; }
    for (Op &op : S.ops) {
        if (op.h3_vscale > 0.f) op.h3_vscale *= 0.25f;
        if (op.d3_vscale > 0.f) op.d3_vscale *= 0.25f;
    }
}
// One overflow EVENT = every frame that was issued with the scales that overflowed.  With several frames in flight the flag can
// go up more than once per event (the frames still running when the first one was noticed carry the same scales): the scales are
// lowered once per event — a flag that shows up while the back-off's pause has not been consumed by a forward yet belongs to the
// event that caused the back-off.
bool h3_tripped(sivo_segnet &S) {
    if (!h3_flag_take(S)) return false;
    if (!S.h3_pause) h3_back_off(S);
    return true;
}
// A place that is not the caller's question (the start of a forward, a status query) found the flag up: react, and remember that
// sivo_segnet_take_overflow has not told anybody yet — with two frames in flight forward(k) runs before the caller asks about
// frame k-1, and consuming the flag silently would let k-1's wrong maps through.
void h3_absorb(sivo_segnet &S) {
    if (h3_tripped(S)) S.h3_unreported = true;
}

// Deterministic frame for the calibration pass: rectangles of random colour over a gradient plus per-pixel noise — edges,

std::vector<uint8_t> calibration_frame(int H, int W, int variant) {
    std::vector<uint8_t> img((size_t)H * W * 3);
    uint32_t st = 0x51f0u + 7919u * (uint32_t)variant;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return st >> 8; };
    std::vector<int> acc((size_t)H * W * 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) acc[((size_t)y * W + x) * 3 + c] = 40 + (c == 0 ? 120 * y / H : c == 1 ? 120 * x / W : 60);
    for (int r = 0; r < 40; ++r) {
        const int x0 = (int)(rnd() % (uint32_t)W), y0 = (int)(rnd() % (uint32_t)H);
        const int w = 8 + (int)(rnd() % (uint32_t)(W / 3 + 1)), h = 8 + (int)(rnd() % (uint32_t)(H / 2 + 1));
        const int col[3] = {(int)(rnd() % 256u), (int)(rnd() % 256u), (int)(rnd() % 256u)};
        for (int y = y0; y < std::min(H, y0 + h); ++y)
            for (int x = x0; x < std::min(W, x0 + w); ++x)
                for (int c = 0; c < 3; ++c) acc[((size_t)y * W + x) * 3 + c] = col[c];
    }
    // variant 1: the same kind of scene at full contrast (black / white rectangles dominate); variant 2: heavy sensor noise
    const int amp = variant == 2 ? 61 : 25;
    for (size_t i = 0; i < acc.size(); ++i) {
        int v = acc[i];
        if (variant == 1) v = v < 100 ? v / 4 : v > 156 ? 255 - (255 - v) / 4 : v;
        v += (int)(rnd() % (uint32_t)amp) - amp / 2;
        img[i] = (uint8_t)std::min(255, std::max(0, v));
    }
    return img;
}

// f16x3: per-layer power-of-two scale of the (transformed) input, from calibration passes on the fp32 kernels whose transform /
// absmax kernels record each layer's largest |V| — three synthetic frames (calibration_frame variants: a scene, the same at
// full contrast, heavy noise) x the MC samples 0 .. 11 of each.  The largest value is put at [2^7, 2^8): 2^8 of headroom below
// fp16's 65504 for frames with larger activations, full hi + lo precision (2^-22) down to 2^-10 of the maximum and an absolute
// error of 2^-25 below that.  The scales depend on the weights and the network geometry only — not on T (the 36 passes are the
// same (frame, global sample) pairs for every T), the device or the frames seen — so every handle of one model computes
// identical bits, until a frame overflows (h3_back_off).
// SIVO_H3_BOOST=k multiplies the scales by 2^k (tests: k = 9 forces the overflow path).
void calibrate_h3(sivo_segnet &S) {
    bool any = false;
    for (const Op &op : S.ops) any = any || op.d_wh3 || op.d3 || op.c3 || op.c7h3;
    if (!any) return;
    uint32_t *flag = nullptr;
    SIVO_HIP(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    *flag = 0;
    S.h3_flag = flag;
    S.d_h3_vmax = dev_alloc<uint32_t>(2 * S.ops.size());         // [op]: largest |V| of an F(4x4) layer; [ops + op]: largest |input| of a direct f16x3 layer
    S.owned.push_back(S.d_h3_vmax);
    SIVO_HIP(hipMemset(S.d_h3_vmax, 0, 2 * S.ops.size() * sizeof(uint32_t)));
    // THREE frames (calibration_frame variants 0, 1, 2), the MC samples 0 .. 11 of each — the dropout masks decide which activations
    // survive, and a layer's largest value is not in every sample — in passes of as many samples as the handle holds: the same 36
    // (frame, global sample index) pairs whatever T is, so that handles of one model that shard the samples compute identical scales.
    constexpr int CAL_FRAMES = 3, CAL_SAMPLES = 12;
    S.calibrating = true;
    try {
        for (int f = 0; f < CAL_FRAMES; ++f) {
            const std::vector<uint8_t> img = calibration_frame(S.H, S.W, f);
            SIVO_HIP(hipMemcpy(S.d_image, img.data(), img.size(), hipMemcpyHostToDevice));
            for (int s0 = 0; s0 < CAL_SAMPLES; s0 += S.T) {
                forward(S, S.d_image, std::min(S.T, CAL_SAMPLES - s0), s0, 0x5157ca11b8a7e5ull + (uint64_t)f, S.d_prob_sum, nullptr, nullptr, S.stream, nullptr);
                SIVO_HIP(hipStreamSynchronize(S.stream));
            }
        }
    } catch (...) {
        S.calibrating = false;
        throw;
    }
    S.calibrating = false;
    std::vector<uint32_t> bits(2 * S.ops.size());
    SIVO_HIP(hipMemcpy(bits.data(), S.d_h3_vmax, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const int boost = SIVO_DIAG_ENV("SIVO_H3_BOOST") ? std::atoi(SIVO_DIAG_ENV("SIVO_H3_BOOST")) : 0;
    auto scale_for = [&](uint32_t b, float *vmax) {
        float v;
        std::memcpy(&v, &b, 4);
        *vmax = v;
        int e = 0;
        if (v > 0.f && std::isfinite(v)) (void)std::frexp(v, &e);        // v = m 2^e, m in [0.5, 1)
        return std::ldexp(1.f, (v > 0.f && std::isfinite(v) ? 8 - e : 0) + boost);
    };
    for (size_t i = 0; i < S.ops.size(); ++i) {
        Op &op = S.ops[i];
        if (op.d_wh3) op.h3_vscale = scale_for(bits[i], &op.h3_vmax);
        if (op.d3 || op.c3 || op.c7h3) op.d3_vscale = scale_for(bits[S.ops.size() + i], &op.d3_vmax);
    }
    S.h3_on = true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Load-time accuracy guard.  The fp16 RANGE of the f16x3 layers is guarded by calibrate_h3 + the overflow flag; this guards their
// ACCURACY for the weights at hand: how much of the 1e-3 logit budget Winograd F(4x4,3x3) (4 d0 - 5 d2 + d4 cancels the common mode
// of a tile) and the fp16 hi + lo split use depends on the weights' and activations' dynamic range, and trained weights are not the
// synthetic ones the tests sweep.  On two calibration frames x MC samples 0, 1 the network is evaluated once more, UNFUSED, along a
// reference chain — every 3x3 layer that production runs on F(4x4) or f16x3 is computed by the direct fp32 matrix-core kernel
// (conv_v2.hip: v_mfma_f32, the fp32 FMA chain) from the reference chain's own input — and beside it the layer's production kernel
// (input transform + f16x3 / bf16x6 GEMM + output transform, or the direct f16x3 kernel) runs on the SAME input: err_l = max |fast -
// ref| / max |ref| is that layer's own error, free of propagated differences and of pooling-switch flips.
// Budget: the tolerance is 1e-3 at the logit range of the reference configuration (|logit| <= 30), i.e. 3.3e-5 of the logits' scale.
// Errors of independent layers add in quadrature and a relative error of the activations carries to the logits with a factor <= 0.5
// (measured: predicted 0.5 sqrt(sum err_l^2) = 1.7 - 1.9e-5 against 0.95 - 1.9e-5 found against the oracle for the synthetic weights,
// BN offsets 3 / 30 / 100).  While the prediction is above the budget the largest contributors move one level down —
// F(4x4) -> direct f16x3 (no transform) -> F(2x2) / direct fp32 -> direct fp32 — the handle is planned again (fusions depend on
// the kernels) and guarded again.  Two samples, two frames: ~1 s at load, nothing per
// frame.  The decisions depend on the weights and the geometry only (never on T: always samples 0 and 1), so shard handles of one

namespace {
struct GuardVerdict { bool any_over = false; std::map<std::string, int> levels; };

GuardVerdict accuracy_guard(sivo_segnet &S, const float *weights, const std::map<std::string, int> &levels_in, float tol) {
    GuardVerdict verdict;
    verdict.levels = levels_in;
    const int n = std::min(S.T, 2);
    std::vector<size_t> guarded;
    auto d3_runs = [&](const Op &op) { return op.d3 && S.h3_on && op.d3_vscale > 0.f && op.drop_site < 0; };
    auto c7_runs = [&](const Op &op) { return op.c7h3 && S.h3_on && op.d3_vscale > 0.f; };
    auto cls_runs = [&](size_t i) { const Op &op = S.ops[i]; return (int)i == S.cls_op && op.c3 && op.pk_in && S.pk_on && S.h3_on && op.d3_vscale > 0.f; };
    for (size_t i = 0; i < S.ops.size(); ++i) {
        const Op &op = S.ops[i];
        // (F(2x2) fp32 layers only when the guard itself put them there: they can still go one level down, to the direct kernel)
        if (op.kind != OP_CONV) continue;
        if (op.ks == 3 && (int)i != S.cls_op && (op.wino4 || op.wino4f || d3_runs(op) || (op.wino && op.guard_level >= 2))) guarded.push_back(i);
        else if (cls_runs(i) || c7_runs(op)) guarded.push_back(i);        // the f16x3 classifier (fused with the MC statistics) / 7x7 layer
    }
    if (guarded.empty()) return verdict;
    const auto t_begin = std::chrono::steady_clock::now();
    hipStream_t st = S.stream;
    // every blob of the net, materialised for n samples (shared ones once); freed when the guard returns
    std::vector<void *> buf(S.blobs.size(), nullptr);
    std::vector<void *> scratch;
    auto release = [&] { for (void *p : buf) if (p) (void)hipFree(p); for (void *p : scratch) if (p) (void)hipFree(p); };
    try {
        int64_t max_out = 0;
        for (size_t b = 0; b < S.blobs.size(); ++b) {
            const Blob &B = S.blobs[b];
            const size_t cnt = (size_t)(B.shared ? 1 : n) * B.chw();
            SIVO_HIP(hipMalloc(&buf[b], cnt * (B.is_mask ? 1 : sizeof(float))));
            if (!B.is_mask) max_out = std::max<int64_t>(max_out, (int64_t)cnt);
        }
        float *d_fast = nullptr;
        void *d_cls_pk = nullptr;
        uint32_t *d_bits = nullptr;
        double *d_sums = nullptr;
        std::vector<float *> d_wref(S.ops.size(), nullptr);      // per guarded layer: its Caffe weights packed for the direct fp32 kernel
        std::vector<int> wref_pad(S.ops.size(), 0);
        SIVO_HIP(hipMalloc((void **)&d_fast, (size_t)max_out * sizeof(float))); scratch.push_back(d_fast);
        SIVO_HIP(hipMalloc((void **)&d_bits, (2 * S.ops.size() + 2) * sizeof(uint32_t))); scratch.push_back(d_bits);
        SIVO_HIP(hipMalloc((void **)&d_sums, 2 * S.ops.size() * sizeof(double))); scratch.push_back(d_sums);
        SIVO_HIP(hipMemset(d_bits, 0, (2 * S.ops.size() + 2) * sizeof(uint32_t)));
        SIVO_HIP(hipMemset(d_sums, 0, 2 * S.ops.size() * sizeof(double)));
        auto fp = [&](int b) { return (float *)buf[b]; };
        const uint64_t seed = 0x6a09e667f3bcc908ull;
        for (int frame = 0; frame < 2; ++frame) {
            const std::vector<uint8_t> img = calibration_frame(S.H, S.W, frame);
            SIVO_HIP(hipMemcpyAsync(S.d_image, img.data(), img.size(), hipMemcpyHostToDevice, st));
            launch_preprocess(S.d_image, fp(S.input_blob), (int64_t)S.H * S.W, st);
            for (size_t oi = 0; oi < S.ops.size(); ++oi) {
                const Op &op = S.ops[oi];
                const Blob &bi = S.blobs[op.in], &bo = S.blobs[op.out];
                const int N = bo.shared ? 1 : n;
                switch (op.kind) {
                    case OP_CONV: {
                        ConvArgs a{};
                        a.in = fp(op.in); a.in_sample_stride = bi.shared ? 0 : bi.chw();
                        a.wt = op.d_w; a.ep_scale = op.d_scale; a.ep_shift = op.d_shift;
                        a.out = fp(op.out);
                        a.N = N; a.Cin = op.cin; a.H = bi.H; a.W = bi.W; a.Cout = op.cout; a.CoutPad = op.cout_pad;
                        a.relu = op.relu; a.drop_site = op.drop_site; a.sample0 = 0; a.seed = seed + (uint64_t)frame;
                        a.wt_x6 = op.d_wx6;
                        if (op.in_drop_site >= 0) {
                            // the fork pooling's dropout lives in this layer's input transform (drop_moved): the pooling above wrote the
                            // sample-invariant values once; here every guard sample gets its own dropped copy, so that the reference chain and
                            // the production kernel both see per-sample masks (x 2 or 0: exact) through their plain input path
                            float *dropped = nullptr;
                            SIVO_HIP(hipMalloc((void **)&dropped, (size_t)n * bi.chw() * sizeof(float))); scratch.push_back(dropped);
                            launch_dropout(fp(op.in), 0, dropped, n, bi.chw(), op.in_drop_site, 0, seed + (uint64_t)frame, st);
                            a.in = dropped; a.in_sample_stride = bi.chw();
                        }
                        const bool is_guarded = std::find(guarded.begin(), guarded.end(), oi) != guarded.end();
                        if (!is_guarded) {
                            // the layer's own fp32 kernel (no F(4x4), no f16x3 in it): part of the reference chain as it is
                            if (op.c7x6) launch_conv7_x6(a, st);
                            else if (op.wino) launch_conv_wino(a, op.wino_cfg, st);
                            else if (op.v2) launch_conv2(a, op.ks, st);
                            else launch_conv(a, op.ks, st);
                            break;
                        }
                        if (c7_runs(op)) {
                            // 7x7: the layer's own weights are the direct fp32 kernel's (conv_mfma_kernel<7>); beside it the f16x3 form
                            launch_conv(a, op.ks, st);
                            ConvArgs f = a;
                            f.out = d_fast;
                            f.wt_h3 = op.d_wd3; f.h3_vscale = op.d3_vscale; f.h3_uscale = op.d3_uscale; f.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                            launch_conv7_h3(f, st);
                            launch_absdiff_max(d_fast, fp(op.out), (int64_t)N * bo.chw(), d_bits + 2 * oi, d_sums + 2 * oi, st);
                            break;
                        }
                        if (cls_runs(oi)) {
                            // classifier: logits of its plain fp32 kernel against those of conv_cls_h3_kernel on the packed form of the same input
                            if (op.v2) launch_conv2(a, op.ks, st); else launch_conv(a, op.ks, st);
                            if (!d_cls_pk) {
                                SIVO_HIP(hipMalloc(&d_cls_pk, pk_bytes(N, bi.C, bi.pk_Hp, bi.pk_Wp))); scratch.push_back(d_cls_pk);
                                SIVO_HIP(hipMemsetAsync(d_cls_pk, 0, pk_bytes(N, bi.C, bi.pk_Hp, bi.pk_Wp), st));
                            }
                            launch_pk_pack(a.in, bi.chw(), d_cls_pk, N, bi.C, bi.H, bi.W, bi.pk_Hp, bi.pk_Wp, op.d3_vscale, const_cast<uint32_t *>(S.h3_flag), st);
                            ClsMcArgs c{};
                            c.in = a.in; c.in_sample_stride = bi.chw(); c.wt = op.d_w_mc; c.ep_scale = op.d_scale; c.ep_shift = op.d_shift;
                            c.T = N; c.Cin = op.cin; c.H = bi.H; c.W = bi.W; c.C = op.cout; c.relu = op.relu;
                            c.logits = d_fast; c.prob_sum = S.d_prob_sum; c.sum_chunk = 0;
                            c.in_pk = d_cls_pk; c.in_pk_sample_bytes = bi.pk_sample_bytes(); c.in_Hp = bi.pk_Hp; c.in_Wp = bi.pk_Wp;
                            c.wt_h3 = op.d_wd3; c.h3_vscale = op.d3_vscale; c.h3_uscale = op.d3_uscale;
                            launch_conv_cls_h3(c, st);
                            launch_absdiff_max(d_fast, fp(op.out), (int64_t)N * bo.chw(), d_bits + 2 * oi, d_sums + 2 * oi, st);
                            break;
                        }
                        // reference: the direct fp32 matrix-core kernel on weights packed for it from the Caffe array
                        if (!d_wref[oi]) {
                            std::vector<float> wt;
                            conv2_pack_weights(weights + op.w_off, op.ks, op.cin, op.cout, wt, &wref_pad[oi]);
                            SIVO_HIP(hipMalloc((void **)&d_wref[oi], wt.size() * sizeof(float))); scratch.push_back(d_wref[oi]);
                            SIVO_HIP(hipMemcpy(d_wref[oi], wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
                        }
                        ConvArgs r = a;
                        r.wt = d_wref[oi]; r.CoutPad = wref_pad[oi]; r.wt_x6 = nullptr;
                        launch_conv2(r, op.ks, st);
                        // the production kernel of this layer on the same input, standalone (no bridge, no fused pooling / Upsample)
                        ConvArgs f = a;
                        f.out = d_fast;
                        // (the order of run_ops: a direct f16x3 layer also carries the flags of the fp32 kernel it falls back to)
                        if (d3_runs(op)) {
                            f.wt_h3 = op.d_wd3; f.h3_vscale = op.d3_vscale; f.h3_uscale = op.d3_uscale; f.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                            f.CoutPad = op.cout;
                            launch_conv3_h3(f, st);
                        } else if (op.wino4) {
                            if (S.h3_on && op.d_wh3 && op.h3_vscale > 0.f) { f.wt_h3 = op.d_wh3; f.h3_vscale = op.h3_vscale; f.h3_uscale = op.h3_uscale; }
                            f.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                            launch_conv_wino4(f, S.d_wino4_ws, op.wino4_group, st, nullptr, false, nullptr);
                        } else if (op.wino4f) {
                            f.variant |= 4096;
                            launch_conv_wino4f(f, st);
                        } else {
                            launch_conv_wino(f, op.wino_cfg, st);
                        }
                        launch_absdiff_max(d_fast, fp(op.out), (int64_t)N * bo.chw(), d_bits + 2 * oi, d_sums + 2 * oi, st);
                        break;
                    }
                    case OP_POOL: {
                        PoolArgs a{};
                        a.in = fp(op.in); a.in_sample_stride = bi.shared ? 0 : bi.chw();
                        a.out = fp(op.out); a.mask = (uint8_t *)buf[op.out2];
                        a.mask_N = S.blobs[op.out2].shared ? 1 : n;
                        a.N = N; a.C = bi.C; a.H = bi.H; a.W = bi.W; a.Ho = bo.H; a.Wo = bo.W;
                        a.drop_site = op.drop_moved ? -1 : op.drop_site; a.sample0 = 0; a.seed = seed + (uint64_t)frame;
                        launch_maxpool2(a, st);
                        break;
                    }
                    case OP_UNPOOL: {
                        UnpoolArgs a{};
                        const Blob &bm = S.blobs[op.in2];
                        a.in = fp(op.in); a.mask = (const uint8_t *)buf[op.in2];
                        a.mask_sample_stride = bm.shared ? 0 : bm.chw();
                        a.out = fp(op.out); a.N = N; a.C = bi.C; a.H = bi.H; a.W = bi.W;
                        launch_unpool2(a, st);
                        break;
                    }
                    case OP_DROPOUT:
                        launch_dropout(fp(op.in), bi.shared ? 0 : bi.chw(), fp(op.out), n, bi.chw(), op.drop_site, 0, seed + (uint64_t)frame, st);
                        break;
                    case OP_LRN:
                        launch_lrn(fp(op.in), fp(op.out), N, bi.C, (int64_t)bi.H * bi.W, op.local_size, op.alpha, op.beta, st);
                        break;
                }
            }
            launch_absmax(fp(S.logits_blob), (int64_t)n * S.blobs[S.logits_blob].chw(), d_bits + 2 * S.ops.size(), st);
        }
        SIVO_HIP(hipStreamSynchronize(st));
        SIVO_HIP(hipGetLastError());
        std::vector<uint32_t> bits(2 * S.ops.size() + 2);
        std::vector<double> sums(2 * S.ops.size());
        SIVO_HIP(hipMemcpy(bits.data(), d_bits, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(sums.data(), d_sums, sums.size() * sizeof(double), hipMemcpyDeviceToHost));
        const bool overflowed = S.h3_flag && *S.h3_flag;
        if (S.h3_flag) *S.h3_flag = 0;            // (the guard's frames are the calibration's: nothing to report to a caller)
        auto as_float = [](uint32_t b) { float v; std::memcpy(&v, &b, 4); return v; };
        // a value left the fp16 range DURING the guard's own frames: they are the calibration's frames, so this is a handle whose scales
        // were forced (SIVO_H3_BOOST) — the range guard's business (overflow flag, back-off), not an accuracy verdict
        if (overflowed) { S.guard_rows.clear(); S.guard_budget = 0.f; S.guard_predicted = 0.f; release(); return verdict; }
        const float L = std::max(1.f, as_float(bits[2 * S.ops.size()]));
        // the tolerance is stated at the logit range of the reference configuration (|logit| <= 30): relative to the logits' scale
        const float budget = tol / 30.f;
        std::vector<sivo_segnet::GuardRow> rows;
        std::vector<std::pair<float, size_t>> by_err;
        double sum2 = 0.0;
        for (size_t oi : guarded) {
            const Op &op = S.ops[oi];
            sivo_segnet::GuardRow r;
            r.layer = op.name;
            r.kernel = cls_runs(oi) ? "classifier f16x3" : c7_runs(op) ? "direct 7x7 f16x3" : d3_runs(op) ? "direct f16x3" : op.wino4 ? (S.h3_on && op.d_wh3 && op.h3_vscale > 0.f ? "F(4x4) f16x3 GEMM" : op.d_wx6 ? "F(4x4) bf16x6 GEMM" : "F(4x4) fp32 GEMM") : op.wino4f ? "F(4x4) fp32 fused" : "F(2x2) fp32 fused";
            r.ref_max = as_float(bits[2 * oi + 1]);
            r.rel_err = as_float(bits[2 * oi]) / std::max(r.ref_max, 1e-30f);
            r.rel_rms = (float)std::sqrt(sums[2 * oi] / std::max(sums[2 * oi + 1], 1e-300));
            r.level = op.guard_level;
            r.first_rel_err = r.rel_err;
            for (const auto &prev : S.guard_rows) if (prev.layer == r.layer) r.first_rel_err = prev.first_rel_err;
            sum2 += (double)r.rel_err * r.rel_err;
            by_err.push_back({r.rel_err, oi});
            rows.push_back(r);
        }
        // predicted error of the logits relative to their scale: the layers' own errors in quadrature, times GUARD_CARRY (how much
        // of a layer's LARGEST error reaches the logits: measured 0.3 - 0.5 over the weight families of the full-size sweep)
        constexpr double GUARD_CARRY = 0.5, REROUTED_ERR = 2e-6;
        double predicted = GUARD_CARRY * std::sqrt(sum2);
        // The prediction is an estimate: found / predicted was 0.6 - 0.9 for the weight families of the sweep and 2.4 for a plan whose
        // scales were forced wrong (tests/test_gpu_segnet.py).  A plan that never needed correction is held to the budget itself; once a
        // plan HAS needed correction the weights (or scales) are not of the kind the estimate was fitted on, and the corrected plan is
        // held to a third of it.
        const double target = levels_in.empty() ? budget : budget / 3.0;
        if (!(predicted <= target)) {
            // take the largest contributors one level down until the prediction (a rerouted layer counted at the direct kernels' ~2e-6) fits
            std::sort(by_err.begin(), by_err.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
            double s2 = sum2;
            for (const auto &[err, oi] : by_err) {
                const Op &op = S.ops[oi];
                if (GUARD_CARRY * std::sqrt(std::max(s2, 0.0)) <= budget / 3.0 || !(err > REROUTED_ERR)) break;
                if (op.guard_level >= 3) continue;
                const int next = (cls_runs(oi) || c7_runs(op)) ? 1 : d3_runs(op) ? std::max(2, op.guard_level + 1) : (op.wino4 || op.wino4f) ? std::max(1, op.guard_level + 1) : 3;
                verdict.levels[op.name] = next;
                verdict.any_over = true;
                s2 += REROUTED_ERR * REROUTED_ERR - (double)err * err;
            }
        }
        S.guard_predicted = (float)predicted;
        S.guard_rows = rows;
        S.guard_budget = budget; S.guard_logit_max = L;
    } catch (...) {
        release();
        throw;
    }
    release();
    S.guard_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return verdict;
}

}  // namespace

// build + guard + (when a layer is over its budget) plan again with that layer one level down, until nothing moves
std::unique_ptr<sivo_segnet> build_guarded(const ProtoNet &net, int t_override, const float *weights, size_t n_weights, int device,
                                           const SivoSegnetOptions &opt) {
    const bool off = SIVO_DIAG_ENV("SIVO_GUARD") && std::atoi(SIVO_DIAG_ENV("SIVO_GUARD")) == 0;
    const float tol = SIVO_DIAG_ENV("SIVO_GUARD_TOL") ? (float)std::atof(SIVO_DIAG_ENV("SIVO_GUARD_TOL")) : 1e-3f;
    std::map<std::string, int> levels;
    std::unique_ptr<sivo_segnet> S;
    std::vector<sivo_segnet::GuardRow> carried;
    double ms = 0.0;
    for (int round = 0; round < 5; ++round) {
        S.reset();                                   // (the previous plan's 16 GB go back before the next one allocates)
        S = build(net, t_override, weights, n_weights, device, opt, levels);
        S->guard_builds = round + 1;
        if (off) break;
        DeviceGuard dg(device);
        S->guard_rows = carried; S->guard_ms = ms;
        const GuardVerdict v = accuracy_guard(*S, weights, levels, tol);
        carried = S->guard_rows; ms = S->guard_ms;
        S->guard_over_budget = v.any_over;
        if (!v.any_over) break;
        levels = v.levels;
    }
    // five plans and the last one still over its budget (never seen: three plans settle a handle whose scales are 2^16 off): the handle is
    // returned — its layers are one to three levels down already — and says so in sivo_segnet_guard_report (predicted > budget, builds = 5)
    if (S->guard_over_budget)
        std::fprintf(stderr, "sivo_segnet: the accuracy guard could not bring the predicted logit error (%.3g of the logit scale) under its budget (%.3g) in %d plans\n",
                     (double)S->guard_predicted, (double)S->guard_budget, S->guard_builds);
    return S;
}

}  // namespace sivo
