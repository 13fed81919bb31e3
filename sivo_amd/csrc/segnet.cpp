// segnet.cpp — host runtime of the Bayesian SegNet path: prototxt -> fused launch
// plan -> per-frame forward.  Stands behind SIVO::BayesianSegNet
// (reference src/bayesian_segnet/bayesian_segnet.cpp:46-78 constructor,
// :299-318 segmentImage).
// The plan is built in segnet_plan.cpp, guarded in segnet_guard.cpp, split into row bands in segnet_bands.cpp; this file runs it per
// frame and holds the C ABI.
#include "segnet_impl.hpp"

namespace sivo {

void harvest(sivo_segnet &S) {
    if (!S.pending) return;
    for (Op &op : S.ops) {
        if (!op.timed_mask) continue;
        int n = 0;
        for (int l = 0; l < Op::PROF_LANES; ++l) {
            if (!(op.timed_mask & (1u << l))) continue;
            float ms = 0.f;
            if (!op.w4_gemm_only_last) {
                SIVO_HIP(hipEventSynchronize(op.ev1[l]));
                SIVO_HIP(hipEventElapsedTime(&ms, op.ev0[l], op.ev1[l]));
                op.ms_total += ms;
            }
            for (int g = 0; g < op.w4_groups_last[l]; ++g)
                for (int k = op.w4_gemm_only_last ? 1 : 0; k < (op.w4_gemm_only_last ? 2 : 3); ++k) {
                    SIVO_HIP(hipEventSynchronize(op.w4_ev[l][4 * g + k + 1]));
                    SIVO_HIP(hipEventElapsedTime(&ms, op.w4_ev[l][4 * g + k], op.w4_ev[l][4 * g + k + 1]));
                    op.w4_ms[k] += ms;
                }
            op.w4_launches += op.w4_groups_last[l];
            op.w4_groups_last[l] = 0;
            op.kernel_launches += 1;
            n += op.lane_n[l];
        }
        op.last_n = n;               // samples of the forward pass (all lanes)
        op.launches += 1;
        op.timed_mask = 0;
    }
    S.pending = false;
}

// Runs the ops [first, last) for the samples [n0, n0 + n) of the per-sample blobs on stream st, using workspace
// region `lane` (0 / 1).  Shared (sample-invariant) blobs are addressed as they are.
void run_ops(sivo_segnet &S, size_t first, size_t last, int n0, int n, int sample0, uint64_t seed, hipStream_t st, int lane) {
    auto fptr = [&](const Blob &b) { return (float *)b.d + (b.shared ? 0 : (int64_t)n0 * b.chw()); };
    auto mptr = [&](const Blob &b) { return (uint8_t *)b.d + (b.shared ? 0 : (int64_t)n0 * b.chw()); };
    float *ws = S.d_wino4_ws ? S.d_wino4_ws + (size_t)lane * S.wino4_ws_floats : nullptr;
    sample0 += n0;
    int w4_vslot = 0;
    for (size_t oi = first; oi < last; ++oi) {
        Op &op = S.ops[oi];
        if (op.skip) continue;
        const Blob &bi = S.blobs[op.in];
        const Blob &bo = S.blobs[op.out];
        const int N = bo.shared ? 1 : n;
        const bool timed = S.profile && (!S.profile_mfma_only || op.kind == OP_CONV);
        const bool bracket = timed && !(S.profile_mfma_only && op.wino4);      // an F(4x4) layer then only times its GEMM
        if (timed) { op.timed_mask |= 1u << lane; op.w4_gemm_only_last = S.profile_mfma_only && op.wino4; op.lane_n[lane] = N; }
        if (bracket) {
            if (!op.ev0[lane]) { SIVO_HIP(hipEventCreate(&op.ev0[lane])); SIVO_HIP(hipEventCreate(&op.ev1[lane])); }
            SIVO_HIP(hipEventRecord(op.ev0[lane], st));
        }
        switch (op.kind) {
            case OP_CONV: {
                ConvArgs a{};
                a.in = fptr(bi); a.in_sample_stride = bi.shared ? 0 : bi.chw();
                a.wt = op.d_w; a.ep_scale = op.d_scale; a.ep_shift = op.d_shift;
                a.out = fptr(bo);
                a.N = N; a.Cin = op.cin; a.H = bi.H; a.W = bi.W; a.Cout = op.cout; a.CoutPad = op.cout_pad;
                a.relu = op.relu; a.drop_site = op.drop_site; a.sample0 = sample0; a.seed = seed;
                a.in_drop_site = op.in_drop_site;
                a.wt_x6 = op.d_wx6;
                const auto h3_active = [&](const Op &o) { return S.h3_on && !S.calibrating && o.d_wh3 && o.h3_vscale > 0.f; };
                if (op.wino4) {
                    if (h3_active(op)) { a.wt_h3 = op.d_wh3; a.h3_vscale = op.h3_vscale; a.h3_uscale = op.h3_uscale; }
                    a.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                    if (S.calibrating && S.d_h3_vmax && !op.w4_bridged_in) a.vmax = S.d_h3_vmax + oi;
                }
                if (op.pool_op >= 0) {
                    const Op &P = S.ops[op.pool_op];
                    a.pool_out = fptr(S.blobs[P.out]); a.pool_mask = mptr(S.blobs[P.out2]); a.pool_drop_site = P.drop_site;
                    a.out = nullptr;
                }
                if (op.unpool_in >= 0) {
                    const Blob &bp = S.blobs[op.unpool_in], &bm = S.blobs[op.unpool_mask];
                    a.in = fptr(bp); a.in_sample_stride = bp.shared ? 0 : bp.chw();
                    a.unpool_mask = mptr(bm); a.unpool_mask_stride = bm.shared ? 0 : bm.chw();
                }
                const bool d3_now = op.d3 && S.h3_on && !S.calibrating && op.d3_vscale > 0.f && a.drop_site < 0 && !a.pool_out &&
                                    conv3_h3_supported(op.ks, op.cin, op.cout, a.H, a.W, a.unpool_mask != nullptr);
                // packed links (decided at plan time from static conditions only) are live while the handle runs f16x3
                const bool pk_live = S.pk_on && S.h3_on && !S.calibrating;
                const bool pk_out_now = pk_live && op.pk_to >= 0 && (op.pk_to != S.cls_op || S.cls_pk_now);
                if (pk_live && ((op.pk_in && (int)oi != S.cls_op) || (pk_out_now && !op.wino4)) && !d3_now)
                    throw std::runtime_error("layer '" + op.name + "': planned for packed activations but not running its f16x3 kernel");
                Blob &bo_w = S.blobs[op.out];
                if ((op.d3 || op.c7h3) && S.calibrating && S.d_h3_vmax) {
                    // the layer's largest |input| (the pooled tensor holds the same values as its Upsample)
                    const int64_t plane_in = a.unpool_mask ? (int64_t)(a.H / 2) * (a.W / 2) : (int64_t)a.H * a.W;
                    launch_absmax(a.in, (int64_t)(a.in_sample_stride ? N : 1) * op.cin * plane_in, S.d_h3_vmax + S.ops.size() + oi, st);
                }
                if (d3_now) {
                    ConvArgs b = a;
                    b.wt_h3 = op.d_wd3; b.h3_vscale = op.d3_vscale; b.h3_uscale = op.d3_uscale; b.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                    b.CoutPad = op.cout;
                    if (op.pk_in && pk_live) {
                        // the producer wrote the packed form this frame (the same condition on its side): read it
                        const Blob &bx = S.blobs[op.unpool_in >= 0 ? op.unpool_in : op.in];
                        b.in_pk = static_cast<unsigned char *>(bx.d_pk) + (bx.shared ? 0 : (int64_t)n0 * bx.pk_sample_bytes());
                        b.in_pk_sample_bytes = bx.shared ? 0 : bx.pk_sample_bytes();
                        b.in_Hp = bx.pk_Hp; b.in_Wp = bx.pk_Wp;
                        if (op.unpool_in >= 0) {
                            const Blob &bm = S.blobs[op.unpool_mask];
                            b.unpool_bits = bm.d_bits + (bm.shared ? 0 : (int64_t)n0 * bm.bits_sample_dwords());
                            b.unpool_bits_stride = bm.shared ? 0 : bm.bits_sample_dwords();
                        }
                    }
                    if (pk_out_now) {
                        b.out_pk = static_cast<unsigned char *>(bo_w.d_pk) + (bo_w.shared ? 0 : (int64_t)n0 * bo_w.pk_sample_bytes());
                        b.out_Hp = bo_w.pk_Hp; b.out_Wp = bo_w.pk_Wp; b.out_vscale = S.ops[op.pk_to].d3_vscale;
                        bo_w.pk_fresh = true; bo_w.pk_scale = b.out_vscale;
                    } else {
                        bo_w.pk_fresh = false;
                    }
                    launch_conv3_h3(b, st);
                } else if (op.wino4f) {
                    static const bool epi4 = !(SIVO_DIAG_ENV("SIVO_W4F_EPI") && std::atoi(SIVO_DIAG_ENV("SIVO_W4F_EPI")) == 0);
                    if (epi4) a.variant |= 4096;      // float4 form of the output stage (conv_wino4f.hip)
                    launch_conv_wino4f(a, st);
                } else if (op.wino4) {
                    hipEvent_t *sub = nullptr;
                    if (S.profile) {
                        op.w4_groups_last[lane] = cdiv(N, op.wino4_group);
                        while ((int)op.w4_ev[lane].size() < 4 * op.w4_groups_last[lane]) {
                            hipEvent_t e;
                            SIVO_HIP(hipEventCreate(&e));
                            op.w4_ev[lane].push_back(e);
                        }
                        sub = op.w4_ev[lane].data();
                    }       // (not profiling: the counts of an earlier profiled forward stay until they are harvested)
                    Wino4Plan plan{};
                    const bool planned = op.wino4_group >= N && S.wino4_slot_floats;
                    if (planned) {
                        // three rotating slots: V of this layer, its M, V of the next layer (when bridged)
                        if (!op.w4_bridged_in) w4_vslot = 0;
                        plan.V = ws + (size_t)w4_vslot * S.wino4_slot_floats;
                        plan.M = ws + (size_t)((w4_vslot + 1) % 3) * S.wino4_slot_floats;
                        plan.Vnext = ws + (size_t)((w4_vslot + 2) % 3) * S.wino4_slot_floats;
                        plan.skip_input = op.w4_bridged_in; plan.bridge = op.w4_bridge;
                        if (op.w4_bridge) {
                            w4_vslot = (w4_vslot + 2) % 3;
                            const Op &next = S.ops[op.bridge_to];
                            plan.next_vscale = h3_active(next) ? next.h3_vscale : 0.f;
                            plan.next_vmax = S.calibrating && S.d_h3_vmax ? S.d_h3_vmax + op.bridge_to : nullptr;
                        }
                    }
                    launch_conv_wino4(a, ws, op.wino4_group, st, sub, S.profile_mfma_only, planned ? &plan : nullptr);
                    if (pk_out_now) {
                        // the F(4x4) output transform writes fp32; the next layer (direct f16x3) reads the packed form
                        launch_pk_pack(a.out, bo_w.chw(), static_cast<unsigned char *>(bo_w.d_pk) + (bo_w.shared ? 0 : (int64_t)n0 * bo_w.pk_sample_bytes()), N, bo_w.C,
                                       bo_w.H, bo_w.W, bo_w.pk_Hp, bo_w.pk_Wp, S.ops[op.pk_to].d3_vscale, const_cast<uint32_t *>(S.h3_flag), st);
                        bo_w.pk_scale = S.ops[op.pk_to].d3_vscale;
                    }
                    bo_w.pk_fresh = false;
                }
                else if (op.c7h3 && S.h3_on && !S.calibrating && op.d3_vscale > 0.f) {
                    ConvArgs b = a;
                    b.wt_h3 = op.d_wd3; b.h3_vscale = op.d3_vscale; b.h3_uscale = op.d3_uscale; b.h3_flag = const_cast<uint32_t *>(S.h3_flag);
                    launch_conv7_h3(b, st);
                }
                else if (op.c7x6) launch_conv7_x6(a, st);
                else if (op.wino) launch_conv_wino(a, op.wino_cfg, st);
                else if (op.v2) launch_conv2(a, op.ks, st);
                else launch_conv(a, op.ks, st);
                if (!d3_now) bo_w.pk_fresh = false;
                break;
            }
            case OP_POOL: {
                PoolArgs a{};
                a.in = fptr(bi); a.in_sample_stride = bi.shared ? 0 : bi.chw();
                a.out = fptr(bo); a.mask = mptr(S.blobs[op.out2]);
                a.mask_N = S.blobs[op.out2].shared ? 1 : n;
                a.N = N; a.C = bi.C; a.H = bi.H; a.W = bi.W; a.Ho = bo.H; a.Wo = bo.W;
                a.drop_site = op.drop_moved ? -1 : op.drop_site; a.sample0 = sample0; a.seed = seed;
                launch_maxpool2(a, st);
                if (op.make_bits && S.pk_on && S.h3_on && !S.calibrating) {
                    const Blob &bm = S.blobs[op.out2];
                    launch_pool_bits(a.mask, bm.d_bits + (bm.shared ? 0 : (int64_t)n0 * bm.bits_sample_dwords()), a.mask_N, bi.C, bo.H, bo.W, bm.bits_Hp, bm.bits_Wp, st);
                }
                break;
            }
            case OP_UNPOOL: {
                UnpoolArgs a{};
                const Blob &bm = S.blobs[op.in2];
                a.in = fptr(bi); a.mask = mptr(bm);
                a.mask_sample_stride = bm.shared ? 0 : bm.chw();
                a.out = fptr(bo); a.N = N; a.C = bi.C; a.H = bi.H; a.W = bi.W;
                if (bi.shared && !bo.shared) throw std::runtime_error("unpool of a shared blob with a per-sample mask is not supported");
                launch_unpool2(a, st);
                break;
            }
            case OP_DROPOUT:
                launch_dropout(fptr(bi), bi.shared ? 0 : bi.chw(), fptr(bo), n, bi.chw(), op.drop_site,
                               sample0, seed, st);
                break;
            case OP_LRN:
                launch_lrn(fptr(bi), fptr(bo), N, bi.C, (int64_t)bi.H * bi.W, op.local_size, op.alpha,
                           op.beta, st);
                break;
        }
        if (bracket) SIVO_HIP(hipEventRecord(op.ev1[lane], st));
        const bool debug_sync = S.opt.debug_sync != 0;      // debugging aid: serialise every op of every lane
        if (debug_sync) SIVO_HIP(hipDeviceSynchronize());
    }
}

// d_prob_sum: fp32 sums of the softmax probabilities over the n samples (layout S.sum_chunk), or null.  mc: the f64 mean
// and its maps (exact: no probability sum goes through memory).  When the plan ends in a classifier convolution that
// conv_cls_mc.hip supports and neither the per-sample probabilities nor (outside mc) the logits are asked for, that
// convolution, the Softmax and the reduction over the samples are ONE kernel and the logits blob is not written.
void forward(sivo_segnet &S, const uint8_t *d_bgr, int n, int sample0, uint64_t seed, float *d_prob_sum,
             float *d_logits, float *d_prob, hipStream_t st, const McTargets *mc, const BandInput *pre) {
    const int64_t hw = (int64_t)S.H * S.W;
    S.last_seed = seed; S.last_sample0 = sample0;
    if (S.profile) harvest(S);
    h3_absorb(S);               // an earlier (asynchronous) frame left the fp16 range and nobody asked yet: back off now, report later
    // the recomputation of a frame that raised the flag: this forward is enqueued without f16x3
    struct Pause {
        sivo_segnet &S; bool was;
        explicit Pause(sivo_segnet &s) : S(s), was(s.h3_on) { if (S.h3_pause) S.h3_on = false; S.h3_pause = false; }
        ~Pause() { S.h3_on = was; }
    } pause(S);
    const Blob &lg = S.blobs[S.logits_blob];
    if (lg.shared) throw std::runtime_error("the network has no test-time dropout: nothing to sample");
    if (!pre) launch_preprocess(d_bgr, (float *)S.blobs[S.input_blob].d, hw, st);
    const bool fuse = S.cls_op >= 0 && !d_prob && !d_logits && (mc || d_prob_sum || S.d_sum64);
    const size_t last = fuse ? (size_t)S.cls_op : S.ops.size();
    // the fused classifier on f16x3 takes its input packed from its producer: decided per forward (an unfused pass runs the
    // classifier as a plain convolution on the fp32 blob)
    S.cls_pk_now = fuse && S.ops[S.cls_op].c3 && S.ops[S.cls_op].pk_in && S.ops[S.cls_op].d3_vscale > 0.f && S.pk_on && S.h3_on && !S.calibrating;
    // the sample-invariant ops form a prefix of the plan
    size_t fork = 0;
    while (fork < last && (S.ops[fork].skip || S.blobs[S.ops[fork].out].shared)) ++fork;
    // SIVO_LANES = 1..4 (default 2 since round 5: 137.8 / 141.3 - 143.2 / 139.1 - 140.3 / 129.5 frames/s for 1 / 2 / 3 / 4 lanes at T = 12
    // with the f16x3 kernels, profiles/r05_lanes_sweep.log; T = 48: 40.9 / 40.5 / 40.1 for 2 / 3 / 4; round 1's fp32 kernels preferred 3): how many
    // sample groups run side by side; profiling keeps one launch per op, and lanes of fewer than 2 samples gain nothing
    int lanes = S.d_wino4_ws ? S.ws_lanes : 2;
    if (S.profile && !S.profile_keep_lanes) lanes = 1;
    while (lanes > 1 && n < 2 * lanes) --lanes;
    // The op at the fork (pool3 with its fused dropout in SegNet-Standard) produces per-sample values but also writes a
    // SHARED blob, the pooling switches every sample's decoder reads.  It runs once for all samples on the caller's
    // stream, ahead of the lane fork, so that exactly one kernel writes the switches and every lane is ordered after it.
    while (lanes > 1 && fork < last && !S.ops[fork].skip && S.ops[fork].out2 >= 0 && S.blobs[S.ops[fork].out2].shared &&
           !S.blobs[S.ops[fork].out].shared)
        ++fork;
    // pre: the sample-invariant prefix came in as row bands computed by `world` ranks (PrefixBands): its results are unpacked into
    // the shared blobs and the fork pooling's dropout is applied per sample; the ops from behind that pooling run as always
    if (pre) bands_unpack(S, *pre, n, sample0, seed, st, &fork);
    else run_ops(S, 0, fork, 0, n, sample0, seed, st, 0);
    if (lanes == 1) {
        run_ops(S, fork, last, 0, n, sample0, seed, st, 0);
    } else {
        if (!S.lane_fork) SIVO_HIP(hipEventCreateWithFlags(&S.lane_fork, hipEventDisableTiming));
        SIVO_HIP(hipEventRecord(S.lane_fork, st));
        int n0 = 0;
        for (int l = 0; l < lanes; ++l) {
            const int nl = n / lanes + (l < n % lanes ? 1 : 0);
            hipStream_t ls = st;
            if (l > 0) {
                if (!S.lane_stream[l]) {
                    SIVO_HIP(hipStreamCreateWithFlags(&S.lane_stream[l], hipStreamNonBlocking));
                    SIVO_HIP(hipEventCreateWithFlags(&S.lane_join[l], hipEventDisableTiming));
                }
                ls = S.lane_stream[l];
                SIVO_HIP(hipStreamWaitEvent(ls, S.lane_fork, 0));
            }
            run_ops(S, fork, last, n0, nl, sample0, seed, ls, l);
            if (l > 0) SIVO_HIP(hipEventRecord(S.lane_join[l], ls));
            n0 += nl;
        }
        for (int l = 1; l < lanes; ++l) SIVO_HIP(hipStreamWaitEvent(st, S.lane_join[l], 0));
    }
    if (fuse) {
        // all samples are back on the caller's stream: classifier + Softmax + sum over the samples (+ maps) in one launch
        Op &op = S.ops[S.cls_op];
        const Blob &bi = S.blobs[op.in];
        ClsMcArgs a{};
        a.in = (const float *)bi.d; a.in_sample_stride = bi.chw();
        a.wt = op.d_w_mc; a.ep_scale = op.d_scale; a.ep_shift = op.d_shift;
        a.T = n; a.Cin = op.cin; a.H = bi.H; a.W = bi.W; a.C = op.cout; a.relu = op.relu;
        a.logits = mc ? mc->logits : nullptr;
        a.prob_sum = d_prob_sum; a.prob_sum64 = S.d_sum64; a.sum_chunk = S.sum_chunk;
        if (mc) { a.classes = mc->classes; a.confidence = mc->conf; a.entropy = mc->ent; }
        op.mc_fused_last = true;
        op.cls_h3_last = S.cls_pk_now;
        if (S.cls_pk_now) {
            a.in_pk = bi.d_pk; a.in_pk_sample_bytes = bi.pk_sample_bytes(); a.in_Hp = bi.pk_Hp; a.in_Wp = bi.pk_Wp;
            a.wt_h3 = op.d_wd3; a.h3_vscale = op.d3_vscale; a.h3_uscale = op.d3_uscale;
        }
        if (S.calibrating && op.c3 && S.d_h3_vmax)       // the classifier's largest |input| (calibration runs the fp32 chain)
            launch_absmax((const float *)bi.d, (int64_t)n * bi.chw(), S.d_h3_vmax + S.ops.size() + S.cls_op, st);
        if (S.profile) {
            op.timed_mask = 1u; op.w4_gemm_only_last = false; op.w4_groups_last[0] = 0; op.lane_n[0] = n;
            if (!op.ev0[0]) { SIVO_HIP(hipEventCreate(&op.ev0[0])); SIVO_HIP(hipEventCreate(&op.ev1[0])); }
            SIVO_HIP(hipEventRecord(op.ev0[0], st));
        }
        if (S.cls_pk_now) launch_conv_cls_h3(a, st);
        else launch_conv_cls_mc(a, st);
        if (S.profile) SIVO_HIP(hipEventRecord(op.ev1[0], st));
    } else {
        if (S.cls_op >= 0) S.ops[S.cls_op].mc_fused_last = false;
        if (d_prob_sum || d_prob || S.d_sum64)
            launch_mc_reduce((const float *)lg.d, n, S.classes, hw, d_prob_sum ? d_prob_sum : S.d_sum64 ? nullptr : S.d_prob_sum, d_prob, 0, st, S.sum_chunk, S.d_sum64);
        if (mc) {
            launch_mc_reduce_finalize((const float *)lg.d, n, S.classes, hw, mc->classes, mc->conf, mc->ent, st);
            if (mc->logits) SIVO_HIP(hipMemcpyAsync(mc->logits, lg.d, (size_t)n * lg.chw() * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        if (d_logits)
            SIVO_HIP(hipMemcpyAsync(d_logits, lg.d, (size_t)n * lg.chw() * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (S.profile) S.pending = true;
    SIVO_HIP(hipGetLastError());
}

namespace {
std::string read_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::invalid_argument(std::string("cannot open '") + path + "'");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
}  // namespace
}  // namespace sivo

size_t sivo::segnet_prefix_slot_bytes(sivo_segnet_t h, int world) {
    DeviceGuard dg(h->device);
    try {
        return plan_bands(*h, world).slot_bytes;
    } catch (const std::invalid_argument &) {
        return 0;
    }
}
void sivo::segnet_prefix_band(sivo_segnet_t h, const uint8_t *d_bgr, int rank, int world, void *d_slot, hipStream_t st) {
    DeviceGuard dg(h->device);
    bands_run(*h, d_bgr, rank, world, d_slot, st);
}

void sivo::segnet_forward_chunked(sivo_segnet_t h, const uint8_t *d_bgr, int n, int sample0, uint64_t seed, double *d_sum_chunked,
                                  int64_t chunk, hipStream_t st, const void *d_slots, int world) {
    DeviceGuard dg(h->device);
    h->sum_chunk = chunk;
    h->d_sum64 = d_sum_chunked;
    try {
        const BandInput pre{d_slots, world};
        forward(*h, d_bgr, n, sample0, seed, nullptr, nullptr, nullptr, st, nullptr, d_slots ? &pre : nullptr);
    } catch (...) {
        h->sum_chunk = 0; h->d_sum64 = nullptr;
        throw;
    }
    h->sum_chunk = 0; h->d_sum64 = nullptr;
}

// The multi-device form's view of the fp16 range guard (segnet_multi.cpp).  overflowed: did a kernel of this handle raise the flag since
// the last question — seen now, or absorbed by a forward() of the same frame (the banded prefix runs BEFORE the frame's forward, whose
// h3_absorb consumes the band's flag and backs this one handle off: *backed_off then says that the scales of this handle are lowered
// already).  back_off: lower the scales once per event on every device — a handle that backed off by itself only gets the pause back
// that its forward consumed, so that all devices run the recomputation with the same arithmetic.
bool sivo::segnet_fp16_overflowed(sivo_segnet_t h, bool *backed_off) {
    const bool now = h3_flag_take(*h), earlier = h->h3_unreported;
    h->h3_unreported = false;
    *backed_off = earlier;
    return now || earlier;
}
void sivo::segnet_fp16_back_off(sivo_segnet_t h, bool already_backed_off) {
    if (already_backed_off) h->h3_pause = true;
    else h3_back_off(*h);
}

using namespace sivo;

extern "C" int sivo_segnet_create_multi_opts(const char *text, size_t len, int t_override, const float *weights, size_t n_weights,
                                             const int *device_ids, int ndev, const SivoSegnetOptions *opts, sivo_segnet_t *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("out is NULL");
        *out = nullptr;
        if (!text || !len) throw std::invalid_argument("model_file (.prototxt file) is empty!");
        if (!weights || !n_weights) throw std::invalid_argument("weights_file (.caffemodel file) is empty!");
        if (!device_ids || ndev < 1) throw std::invalid_argument("device_ids is empty");
        for (int d = 0; d < ndev; ++d)
            if (device_ids[d] < 0 || device_ids[d] >= sivo_device_count())
                return fail(SIVO_ERR_RUNTIME, "HIP device %d is not available (%d visible): libsivo_hip has no CPU fallback", device_ids[d],
                            sivo_device_count());
        std::unique_ptr<sivo_segnet> S(new sivo_segnet);
        const SivoSegnetOptions opt = segnet_options(opts);
        S->multi = segnet_multi_create(text, len, t_override, weights, n_weights, device_ids, ndev, &opt);
        int32_t T, H, W, K;
        segnet_multi_shape(S->multi, &T, &H, &W, &K, nullptr);
        S->device = device_ids[0]; S->T = T; S->H = H; S->W = W; S->classes = K;
        *out = S.release();
        return SIVO_OK;
    });
}
extern "C" int sivo_segnet_create_multi(const char *text, size_t len, int t_override, const float *weights, size_t n_weights,
                                        const int *device_ids, int ndev, sivo_segnet_t *out) {
    return sivo_segnet_create_multi_opts(text, len, t_override, weights, n_weights, device_ids, ndev, nullptr, out);
}

extern "C" int sivo_segnet_num_devices(sivo_segnet_t h, int *ndev) {
    if (!h || !ndev) return fail(SIVO_ERR_INVALID_ARGUMENT, "null argument");
    int32_t n = 1;
    if (h->multi) segnet_multi_shape(h->multi, nullptr, nullptr, nullptr, nullptr, &n);
    *ndev = n;
    return SIVO_OK;
}

extern "C" int sivo_segnet_num_params(const char *text, size_t len, size_t *n_params) {
    return guarded([&] {
        if (!text || !len) throw std::invalid_argument("model_file (.prototxt file) is empty!");
        *n_params = count_params(parse_prototxt(std::string(text, len)));
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_create_opts(const char *text, size_t len, int t_override, const float *weights, size_t n_weights, int device,
                                       const SivoSegnetOptions *opts, sivo_segnet_t *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("out is NULL");
        *out = nullptr;
        if (!text || !len) throw std::invalid_argument("model_file (.prototxt file) is empty!");
        if (!weights || !n_weights) throw std::invalid_argument("weights_file (.caffemodel file) is empty!");
        const SivoSegnetOptions opt = segnet_options(opts);
        if (sivo_device_count() <= device || device < 0)
            return fail(SIVO_ERR_RUNTIME, "HIP device %d is not available (%d visible): libsivo_hip has no CPU fallback",
                        device, sivo_device_count());
        ProtoNet net = parse_prototxt(std::string(text, len));
        *out = build_guarded(net, t_override, weights, n_weights, device, opt).release();
        return SIVO_OK;
    });
}
extern "C" int sivo_segnet_create(const char *text, size_t len, int t_override, const float *weights,
                                  size_t n_weights, int device, sivo_segnet_t *out) {
    return sivo_segnet_create_opts(text, len, t_override, weights, n_weights, device, nullptr, out);
}

extern "C" int sivo_caffemodel_weights(const char *prototxt_text, size_t prototxt_len, const void *model_bytes,
                                       size_t model_len, float *out, size_t capacity, size_t *n_weights) {
    return guarded([&] {
        if (!prototxt_text || !prototxt_len || !model_bytes || !n_weights) throw std::invalid_argument("null argument");
        const std::vector<float> w = weights_from_caffemodel(std::string((const char *)model_bytes, model_len),
                                                             parse_prototxt(std::string(prototxt_text, prototxt_len)));
        *n_weights = w.size();
        if (!out) return SIVO_OK;
        if (capacity < w.size()) return fail(SIVO_ERR_CAPACITY, "output capacity is smaller than the parameter count");
        std::memcpy(out, w.data(), w.size() * sizeof(float));
        return SIVO_OK;
    });
}

// The weights of BayesianSegNetParams::weights_file as the flat fp32 array sivo_segnet_create takes: a .caffemodel is
// read like Net::CopyTrainedLayersFrom does (bayesian_segnet.cpp:61, layers matched by name), a .sivow container as is.
static std::vector<float> weights_from_file(const char *model_file, const char *weights_file, std::string &text) {
    if (!model_file || !*model_file) throw std::invalid_argument("model_file (.prototxt file) is empty!");
    if (!weights_file || !*weights_file) throw std::invalid_argument("weights_file (.caffemodel file) is empty!");
    text = read_file(model_file);
    const std::string wb = read_file(weights_file);
    std::vector<float> w;
    if (wb.size() >= 16 && std::memcmp(wb.data(), "SIVOW001", 8) == 0) {
        uint64_t n = 0;
        std::memcpy(&n, wb.data() + 8, 8);
        if (wb.size() != 16 + 4 * n) throw std::invalid_argument("weights_file is truncated");
        w.resize(n);
        std::memcpy(w.data(), wb.data() + 16, 4 * n);
    } else if (wb.size() < 200 && wb.compare(0, 7, "version") == 0) {
        throw std::invalid_argument("weights_file is a Git-LFS pointer, not the trained model (run `git lfs pull`)");
    } else if (looks_like_caffemodel(wb)) {
        w = weights_from_caffemodel(wb, parse_prototxt(text));
    } else {
        throw std::invalid_argument("weights_file is neither a .caffemodel (protobuf NetParameter) nor a .sivow container");
    }
    return w;
}

extern "C" int sivo_segnet_create_from_files_opts(const char *model_file, const char *weights_file, int t_override, int device,
                                                  const SivoSegnetOptions *opts, sivo_segnet_t *out) {
    return guarded([&] {
        std::string text;
        const std::vector<float> w = weights_from_file(model_file, weights_file, text);
        return sivo_segnet_create_opts(text.data(), text.size(), t_override, w.data(), w.size(), device, opts, out);
    });
}
extern "C" int sivo_segnet_create_from_files(const char *model_file, const char *weights_file, int t_override,
                                             int device, sivo_segnet_t *out) {
    return sivo_segnet_create_from_files_opts(model_file, weights_file, t_override, device, nullptr, out);
}

extern "C" int sivo_segnet_create_multi_from_files_opts(const char *model_file, const char *weights_file, int t_override, const int *device_ids,
                                                        int ndev, const SivoSegnetOptions *opts, sivo_segnet_t *out) {
    return guarded([&] {
        std::string text;
        const std::vector<float> w = weights_from_file(model_file, weights_file, text);
        return sivo_segnet_create_multi_opts(text.data(), text.size(), t_override, w.data(), w.size(), device_ids, ndev, opts, out);
    });
}
extern "C" int sivo_segnet_create_multi_from_files(const char *model_file, const char *weights_file, int t_override,
                                                   const int *device_ids, int ndev, sivo_segnet_t *out) {
    return sivo_segnet_create_multi_from_files_opts(model_file, weights_file, t_override, device_ids, ndev, nullptr, out);
}

extern "C" int sivo_segnet_destroy(sivo_segnet_t h) {
    return guarded([&] {
        if (h) {
            DeviceGuard dg(h->device);
            delete h;
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_shape(sivo_segnet_t h, int32_t *T, int32_t *C, int32_t *H, int32_t *W, int32_t *classes) {
    if (!h) return fail(SIVO_ERR_INVALID_ARGUMENT, "null handle");
    if (T) *T = h->T;
    if (C) *C = h->C;
    if (H) *H = h->H;
    if (W) *W = h->W;
    if (classes) *classes = h->classes;
    return SIVO_OK;
}

extern "C" int sivo_segnet_flops(sivo_segnet_t h, double *shared, double *per_sample) {
    if (!h) return fail(SIVO_ERR_INVALID_ARGUMENT, "null handle");
    if (shared) *shared = h->flops_shared;
    if (per_sample) *per_sample = h->flops_sample;
    return SIVO_OK;
}

extern "C" int sivo_segnet_forward_dev(sivo_segnet_t h, const uint8_t *d_bgr, int n_samples, int sample0,
                                       uint64_t seed, float *d_prob_sum, float *d_logits, float *d_prob,
                                       void *stream) {
    return guarded([&] {
        if (!h || !d_bgr || !d_prob_sum) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle shards the samples itself: use sivo_segnet_segment");
        if (n_samples < 1 || n_samples > h->T) throw std::invalid_argument("n_samples must be in [1, T]");
        DeviceGuard dg(h->device);
        forward(*h, d_bgr, n_samples, sample0, seed, d_prob_sum, d_logits, d_prob, (hipStream_t)stream);
        return SIVO_OK;
    });
}

// ---- row bands of the sample-invariant prefix over ranks (include/sivo_hip.h; PrefixBands above)
extern "C" int sivo_segnet_prefix_bands(sivo_segnet_t h, int world, size_t *slot_bytes, int32_t *rows /* [world + 1], optional */,
                                        int32_t *input_rows /* [2 * world], optional */) {
    return guarded([&] {
        if (!h || !slot_bytes) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle splits its prefix itself");
        DeviceGuard dg(h->device);
        const PrefixBands &B = plan_bands(*h, world);
        *slot_bytes = B.slot_bytes;
        if (rows) for (int r = 0; r <= world; ++r) rows[r] = B.y0[(size_t)r];
        if (input_rows) for (int r = 0; r < world; ++r) { input_rows[2 * r] = B.in0[(size_t)r]; input_rows[2 * r + 1] = B.in1[(size_t)r]; }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_prefix_band_dev(sivo_segnet_t h, const uint8_t *d_bgr, int rank, int world, void *d_slot, void *stream) {
    return guarded([&] {
        if (!h || !d_bgr || !d_slot) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle splits its prefix itself");
        DeviceGuard dg(h->device);
        bands_run(*h, d_bgr, rank, world, d_slot, (hipStream_t)stream);
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_forward_banded_dev(sivo_segnet_t h, const void *d_slots, int world, int n_samples, int sample0, uint64_t seed,
                                              float *d_prob_sum, float *d_logits, void *stream) {
    return guarded([&] {
        if (!h || !d_slots) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle splits its prefix itself");
        if (n_samples < 1 || n_samples > h->T) throw std::invalid_argument("n_samples out of range");
        DeviceGuard dg(h->device);
        const BandInput pre{d_slots, world};
        forward(*h, nullptr, n_samples, sample0, seed, d_prob_sum, d_logits, nullptr, (hipStream_t)stream, nullptr, &pre);
        return SIVO_OK;
    });
}

extern "C" int sivo_mc_reduce_dev(const float *d_logits, int n, int classes, int64_t hw, float *d_prob_sum,
                                  float *d_prob, int accumulate, void *stream) {
    return guarded([&] {
        if (!d_logits || !d_prob_sum || n < 1 || classes < 1 || classes > 16 || hw < 1)
            throw std::invalid_argument("bad argument (1 <= classes <= 16)");
        launch_mc_reduce(d_logits, n, classes, hw, d_prob_sum, d_prob, accumulate, (hipStream_t)stream);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_mc_finalize_dev(const float *d_prob_sum, int classes, int64_t hw, int t_total,
                                    uint8_t *d_classes, double *d_confidence, double *d_entropy, void *stream) {
    return guarded([&] {
        if (!d_prob_sum || classes < 1 || hw < 1 || t_total < 1) throw std::invalid_argument("bad argument");
        launch_mc_finalize(d_prob_sum, classes, hw, t_total, d_classes, d_confidence, d_entropy, (hipStream_t)stream);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_mc_variance_dev(const float *d_prob, int T, int classes, int64_t hw, const uint8_t *d_classes,
                                    double *d_variance, void *stream) {
    return guarded([&] {
        if (!d_prob || !d_classes || !d_variance || T < 2) throw std::invalid_argument("bad argument (T >= 2)");
        launch_mc_variance(d_prob, T, classes, hw, d_classes, d_variance, (hipStream_t)stream);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_segment(sivo_segnet_t h, const uint8_t *bgr, int rows, int cols, uint64_t seed,
                                   uint8_t *classes, double *confidence, double *entropy) {
    return guarded([&] {
        if (!h || !bgr) throw std::invalid_argument("null argument");
        // resizeImage (bayesian_segnet.cpp:142-162): exact size -> as is; larger -> centre crop; smaller -> empty
        if (rows < h->H || cols < h->W)
            return fail(SIVO_ERR_IMAGE_TOO_SMALL, "image %dx%d is smaller than the network geometry %dx%d", cols, rows, h->W, h->H);
        if (h->multi) {
            segnet_multi_segment(h->multi, bgr, rows, cols, seed, classes, confidence, entropy);
            return SIVO_OK;
        }
        DeviceGuard dg(h->device);
        const int x_tl = (rows == h->H && cols == h->W) ? 0 : cols / 2 - h->W / 2;
        const int y_tl = (rows == h->H && cols == h->W) ? 0 : rows / 2 - h->H / 2;
        hipStream_t st = h->stream;
        SIVO_HIP(hipMemcpy2DAsync(h->d_image, (size_t)h->W * 3, bgr + ((size_t)y_tl * cols + x_tl) * 3, (size_t)cols * 3,
                                  (size_t)h->W * 3, (size_t)h->H, hipMemcpyHostToDevice, st));
        const McTargets mc{h->d_classes, h->d_conf, h->d_ent, nullptr};
        const int64_t hw = (int64_t)h->H * h->W;
        for (int attempt = 0; attempt < 2; ++attempt) {
            forward(*h, h->d_image, h->T, 0, seed, nullptr, nullptr, nullptr, st, &mc);
            if (classes) SIVO_HIP(hipMemcpyAsync(classes, h->d_classes, hw, hipMemcpyDeviceToHost, st));
            if (confidence) SIVO_HIP(hipMemcpyAsync(confidence, h->d_conf, hw * sizeof(double), hipMemcpyDeviceToHost, st));
            if (entropy) SIVO_HIP(hipMemcpyAsync(entropy, h->d_ent, hw * sizeof(double), hipMemcpyDeviceToHost, st));
            SIVO_HIP(hipStreamSynchronize(st));
            if (!h3_tripped(*h)) break;       // a value left the fp16 range in this frame: once more, without f16x3
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_segment_dev(sivo_segnet_t h, const uint8_t *d_bgr, uint64_t seed, uint8_t *d_classes,
                                       double *d_confidence, double *d_entropy, void *stream) {
    return guarded([&] {
        if (!h || !d_bgr || !d_classes || !d_confidence || !d_entropy) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle takes host buffers: use sivo_segnet_segment");
        DeviceGuard dg(h->device);
        hipStream_t st = (hipStream_t)stream;
        const McTargets mc{d_classes, d_confidence, d_entropy, nullptr};
        forward(*h, d_bgr, h->T, 0, seed, nullptr, nullptr, nullptr, st, &mc);
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_segment_logits_dev(sivo_segnet_t h, const uint8_t *d_bgr, uint64_t seed, uint8_t *d_classes,
                                              double *d_confidence, double *d_entropy, float *d_logits, void *stream) {
    return guarded([&] {
        if (!h || !d_bgr || !d_classes || !d_confidence || !d_entropy || !d_logits) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle takes host buffers: use sivo_segnet_segment");
        DeviceGuard dg(h->device);
        const McTargets mc{d_classes, d_confidence, d_entropy, d_logits};
        forward(*h, d_bgr, h->T, 0, seed, nullptr, nullptr, nullptr, (hipStream_t)stream, &mc);
        return SIVO_OK;
    });
}

extern "C" int sivo_mc_segment_dev(const float *d_logits, int T, int classes, int64_t hw, uint8_t *d_classes,
                                   double *d_confidence, double *d_entropy, void *stream) {
    return guarded([&] {
        if (!d_logits || !d_classes || !d_confidence || !d_entropy || T < 1 || classes < 1 || classes > 16 || hw < 1)
            throw std::invalid_argument("bad argument (1 <= classes <= 16)");
        launch_mc_reduce_finalize(d_logits, T, classes, hw, d_classes, d_confidence, d_entropy, (hipStream_t)stream);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_blob(sivo_segnet_t h, const char *name, float *host_out, size_t capacity,
                                int32_t shape[4]) {
    return guarded([&] {
        if (!h || !name) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("blobs live in the per-device handles of a multi-device handle");
        auto it = h->blob_id.find(name);
        if (it == h->blob_id.end()) throw std::invalid_argument(std::string("no blob named '") + name + "'");
        const Blob &b = h->blobs[it->second];
        if (b.fused_away)
            throw std::invalid_argument(std::string("blob '") + name + "' is not materialised: it only exists on chip, fused into the next convolution (the diagnostic library libsivo_hip_diag.so has switches that keep Upsample outputs / conv-to-conv activations / pooled convolutions in HBM: DESIGN.md appendix)");
        const int N = (b.shared && b.drop_pending < 0) ? 1 : h->T;
        if (shape) { shape[0] = N; shape[1] = b.C; shape[2] = b.H; shape[3] = b.W; }
        const size_t n = (size_t)N * b.chw();
        if (!host_out) return SIVO_OK;
        if (capacity < n) return fail(SIVO_ERR_CAPACITY, "blob '%s' holds %zu values, capacity %zu", name, n, capacity);
        DeviceGuard dg(h->device);
        SIVO_HIP(hipDeviceSynchronize());
        if (b.drop_pending >= 0) {
            // Caffe's blob of this name is the per-sample DROPPED tensor; here its dropout is applied inside the consumer's input
            // transform and only the values in front of it are stored: re-create the T samples of the last forward
            float *tmp = dev_alloc<float>(n);
            launch_dropout((const float *)b.d, 0, tmp, N, b.chw(), b.drop_pending, h->last_sample0, h->last_seed, nullptr);
            SIVO_HIP(hipMemcpy(host_out, tmp, n * sizeof(float), hipMemcpyDeviceToHost));
            SIVO_HIP(hipFree(tmp));
        } else if (b.is_mask) {
            float *tmp = dev_alloc<float>(n);
            launch_mask_to_index((const uint8_t *)b.d, tmp, (int64_t)n, b.H, b.W, b.src_W, nullptr);
            SIVO_HIP(hipMemcpy(host_out, tmp, n * sizeof(float), hipMemcpyDeviceToHost));
            SIVO_HIP(hipFree(tmp));
        } else if (b.pk_fresh) {
            // the last forward wrote this blob in its packed form only: (hi + lo) / scale, exact to 2^-22 of the fp32 value
            float *tmp = dev_alloc<float>(n);
            launch_pk_unpack(b.d_pk, tmp, N, b.C, b.H, b.W, b.pk_Hp, b.pk_Wp, b.pk_scale, nullptr);
            SIVO_HIP(hipMemcpy(host_out, tmp, n * sizeof(float), hipMemcpyDeviceToHost));
            SIVO_HIP(hipFree(tmp));
        } else {
            SIVO_HIP(hipMemcpy(host_out, b.d, n * sizeof(float), hipMemcpyDeviceToHost));
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_profile(sivo_segnet_t h, int enable) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        DeviceGuard dg(h->device);
        // switching profiling OFF does not wait for the events of the last profiled forward (a caller inside a throughput loop would
        // drain its pipeline): they are harvested by the next sivo_segnet_profile_read, or before profiling is switched on again
        if (enable != 0 && h->pending) harvest(*h);
        h->profile = enable != 0;
        h->profile_mfma_only = enable >= 3 && enable <= 6;
        h->profile_keep_lanes = enable == 5 || enable == 6;
        if (enable == 2 || enable == 3 || enable == 5)   // reset the accumulators
            for (Op &op : h->ops) {
                op.ms_total = 0.0; op.launches = 0; op.w4_launches = 0; op.kernel_launches = 0;
                op.w4_ms[0] = op.w4_ms[1] = op.w4_ms[2] = 0.0;
            }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_profile_read(sivo_segnet_t h, SivoOpProfile *out, int capacity, int *n_out) {
    return guarded([&] {
        if (!h || !n_out) throw std::invalid_argument("null argument");
        DeviceGuard dg(h->device);
        harvest(*h);
        int rows = 0;
        for (const Op &op : h->ops) rows += op.wino4 ? (op.w4_bridged_in ? 2 : 3) : 1;   // an F(4x4,3x3) layer reports its kernels separately
        *n_out = rows;
        if (!out) return SIVO_OK;
        if (capacity < *n_out) return fail(SIVO_ERR_CAPACITY, "%d rows, capacity %d", *n_out, capacity);
        int r = 0;
        for (size_t i = 0; i < h->ops.size(); ++i) {
            const Op &op = h->ops[i];
            if (op.wino4) {
                const bool h3 = h->h3_on && op.d_wh3 && op.h3_vscale > 0.f;
                const char *kn[3] = {"wino4_input_kernel", h3 ? "wino4_gemm_h3_kernel" : op.d_wx6 ? "wino4_gemm_x6p_kernel" : "wino4_gemm_kernel", op.w4_bridge ? "wino4_bridge_kernel" : "wino4_output_kernel"};
                const Blob &bi = h->blobs[op.in];
                const double tiles = (double)((bi.H + 3) / 4) * (bi.W / 4), kp = wino4_cout_pad(op.cout);
                // input transform: activation in, V out; GEMM: V in, M out; output transform: M in, activation out — a bridge
                // (output transform + the next layer's input transform) reads M and writes the next V (36 positions x cout
                // channels) instead of the activation
                const double bytes[3] = {4.0 * (op.cin * (double)bi.H * bi.W + 36.0 * op.cin * tiles),
                                         4.0 * 36.0 * tiles * (op.cin + kp),
                                         op.w4_bridge ? 4.0 * 36.0 * tiles * (kp + op.cout) : 4.0 * (36.0 * kp * tiles + op.cout * (double)bi.H * bi.W)};
                const int groups = op.launches ? op.w4_launches / op.launches : 1;
                for (int k = op.w4_bridged_in ? 1 : 0; k < 3; ++k) {
                    SivoOpProfile &p = out[r++];
                    std::memset(&p, 0, sizeof p);
                    std::snprintf(p.layer, sizeof p.layer, "%s", op.name.c_str());
                    std::snprintf(p.kernel, sizeof p.kernel, "%s", kn[k]);
                    p.samples = op.last_n;                           // per forward pass; `launches` below counts passes
                    p.flops_per_sample = k == 1 ? op.flops : 0.0;    // algorithmic (direct-convolution) flops, on the GEMM row
                    p.bytes_per_sample = bytes[k];
                    p.ms_total = op.w4_ms[k];
                    p.launches = op.launches;
                    p.kernel_launches = op.w4_launches;
                    (void)groups;
                }
                continue;
            }
            SivoOpProfile &p = out[r++];
            std::memset(&p, 0, sizeof p);
            std::snprintf(p.layer, sizeof p.layer, "%s", op.name.c_str());
            const bool d3_on = op.d3 && h->h3_on && op.d3_vscale > 0.f && op.drop_site < 0 && op.pool_op < 0;
            std::snprintf(p.kernel, sizeof p.kernel, "%s", op.mc_fused_last ? (op.cls_h3_last ? "conv_cls_h3_kernel" : "conv_wino_cls_mc_kernel") : d3_on ? "conv3_h3_kernel" : (op.c7h3 && !(h->h3_on && op.d3_vscale > 0.f)) ? "conv7_x6_kernel" : op.kernel.c_str());
            p.samples = op.last_n;
            p.flops_per_sample = op.flops;
            p.bytes_per_sample = op.bytes;
            p.ms_total = op.ms_total;
            p.launches = op.launches;
            p.kernel_launches = op.kernel_launches;
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_segnet_take_overflow(sivo_segnet_t h, int *overflowed) {
    return guarded([&] {
        if (!h || !overflowed) throw std::invalid_argument("null argument");
        if (h->multi) throw std::invalid_argument("a multi-device handle has synchronous entry points only: they recompute such a frame themselves");
        const bool now = h3_tripped(*h);
        *overflowed = (now || h->h3_unreported) ? 1 : 0;
        h->h3_unreported = false;
        return SIVO_OK;
    });
}

// The load-time accuracy guard's report (accuracy_guard): one row per guarded layer of the final plan.
extern "C" int sivo_segnet_guard_report(sivo_segnet_t h, SivoGuardLayer *rows, int capacity, int *n_rows, float *budget, float *predicted,
                                        float *logit_max, double *guard_ms, int *builds) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        if (h->multi) throw std::invalid_argument("per-device state: query the handles of a multi-device handle one by one");
        if (n_rows) *n_rows = (int)h->guard_rows.size();
        if (budget) *budget = h->guard_budget;
        if (logit_max) *logit_max = h->guard_logit_max;
        if (predicted) *predicted = h->guard_predicted;
        if (guard_ms) *guard_ms = h->guard_ms;
        if (builds) *builds = h->guard_builds;
        if (rows)
            for (int i = 0; i < capacity && i < (int)h->guard_rows.size(); ++i) {
                const sivo_segnet::GuardRow &g = h->guard_rows[(size_t)i];
                SivoGuardLayer &r = rows[i];
                std::memset(&r, 0, sizeof r);
                std::snprintf(r.layer, sizeof r.layer, "%s", g.layer.c_str());
                std::snprintf(r.kernel, sizeof r.kernel, "%s", g.kernel.c_str());
                r.rel_err = g.rel_err; r.rel_rms = g.rel_rms; r.ref_max = g.ref_max; r.first_rel_err = g.first_rel_err; r.level = g.level;
            }
        return SIVO_OK;
    });
}

// Which GEMM the F(4x4,3x3) layers of this handle run (2 = f16x3, 1 = bf16x6, 0 = fp32 MFMA / none) and how many frames
// raised the fp16 overflow flag since the handle was created (each lowered the scales by 2^2; the fourth switched f16x3 off).
// per_layer (optional, capacity rows): layer name, largest |V| of the calibration frame and the scale chosen.
extern "C" int sivo_segnet_gemm_status(sivo_segnet_t h, int *mode, int *overflow_frames, SivoH3Layer *per_layer, int capacity, int *n_layers) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        if (h->multi) throw std::invalid_argument("per-device state: query the handles of a multi-device handle one by one");
        DeviceGuard dg(h->device);
        h3_absorb(*h);
        bool any_h3 = false, any_x6 = false;
        int rows = 0;
        for (const Op &op : h->ops) {
            if (op.d3 || op.c3 || op.c7h3) {        // direct f16x3 layer / classifier: vmax / vscale are those of its input activation
                any_h3 = any_h3 || op.d3_vscale > 0.f;
                if (per_layer && rows < capacity) {
                    SivoH3Layer &r = per_layer[rows];
                    std::memset(&r, 0, sizeof r);
                    std::snprintf(r.layer, sizeof r.layer, "%s", op.name.c_str());
                    r.vmax = op.d3_vmax; r.vscale = op.d3_vscale; r.uscale = op.d3_uscale;
                }
                ++rows;
                continue;
            }
            if (!op.wino4) continue;
            any_h3 = any_h3 || (op.d_wh3 && op.h3_vscale > 0.f);
            any_x6 = any_x6 || op.d_wx6;
            if (op.d_wh3) {
                if (per_layer && rows < capacity) {
                    SivoH3Layer &r = per_layer[rows];
                    std::memset(&r, 0, sizeof r);
                    std::snprintf(r.layer, sizeof r.layer, "%s", op.name.c_str());
                    r.vmax = op.h3_vmax; r.vscale = op.h3_vscale; r.uscale = op.h3_uscale;
                }
                ++rows;
            }
        }
        if (mode) *mode = (h->h3_on && any_h3) ? 2 : any_x6 ? 1 : 0;
        if (overflow_frames) *overflow_frames = h->h3_overflow_frames;
        if (n_layers) *n_layers = rows;
        return SIVO_OK;
    });
}

