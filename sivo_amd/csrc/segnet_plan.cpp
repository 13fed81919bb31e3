// segnet_plan.cpp — prototxt + weights -> fused launch plan of the Bayesian SegNet path (sivo::build).  Stands behind the constructor
// of SIVO::BayesianSegNet (reference src/bayesian_segnet/bayesian_segnet.cpp:46-78: Net from the prototxt, CopyTrainedLayersFrom).
//
// Plan construction ("what Caffe runs layer by layer, regrouped for the GPU"):
//   * BN(INFERENCE), ReLU and Dropout that follow a Convolution in place are folded
//     into the convolution's epilogue; Dropout that follows a Pooling in place is
//     folded into the pooling kernel.
//   * Everything upstream of the first Dropout does not depend on the Monte-Carlo
//     sample: those blobs are "shared" (N = 1, computed once per frame instead of
//     T times — 134.1 of 446.0 GFLOP per sample for SegNet-Standard) and are
//     broadcast with a zero sample stride into the first sample-dependent op.
//   * Softmax is not a kernel of its own: the plan ends at the logits and
//     sivo_mc_reduce fuses softmax with the sum over samples.
#include "segnet_impl.hpp"

namespace sivo {

SivoSegnetOptions segnet_options(const SivoSegnetOptions *opts) {
    SivoSegnetOptions o{};
    if (opts) std::memcpy(&o, opts, std::min<size_t>(sizeof o, opts->struct_size));       // (a caller built against a shorter struct: the rest stays 0 = default)
    o.struct_size = sizeof o;
    if (o.lanes < 1 || o.lanes > sivo_segnet::MAX_LANES) {
        if (o.lanes != 0) throw std::invalid_argument("SivoSegnetOptions.lanes must be 0 (default) or 1 .. 4");
        o.lanes = 2;                     // 141 - 143 frames/s against 139 - 140 with three (DESIGN appendix)
    }
    if (o.gemm < 0 || o.gemm > 2) throw std::invalid_argument("SivoSegnetOptions.gemm must be 0 (f16x3), 1 (bf16x6) or 2 (fp32)");
    if (o.wino4_workspace_mb < 0) throw std::invalid_argument("SivoSegnetOptions.wino4_workspace_mb is negative");
    if (o.wino4_workspace_mb == 0) o.wino4_workspace_mb = 16384;
    return o;
}

size_t count_params(const ProtoNet &net) {
    std::map<std::string, int> ch;
    ch[net.input] = net.shape[1];
    size_t n = 0;
    for (const ProtoLayer &L : net.layers) {
        const int cin = L.bottom.empty() ? net.shape[1] : ch[L.bottom[0]];
        if (L.type == "Convolution") {
            n += (size_t)L.num_output * cin * L.kernel_size * L.kernel_size + (size_t)L.num_output;
            ch[L.top[0]] = L.num_output;
        } else if (L.type == "BN") {
            n += 2 * (size_t)cin;
            ch[L.top[0]] = cin;
        } else {
            for (auto &t : L.top) ch[t] = cin;
        }
    }
    return n;
}

int new_blob(sivo_segnet &S, const std::string &name, int C, int H, int W, bool shared, bool is_mask = false) {
    Blob b;
    b.name = name; b.C = C; b.H = H; b.W = W; b.shared = shared; b.is_mask = is_mask;
    S.blobs.push_back(b);
    S.blob_id[name] = (int)S.blobs.size() - 1;
    return (int)S.blobs.size() - 1;
}

// Re-layout Caffe (Cout,Cin,k,k) weights to [ceil(Cin/KC)][k*k][KC][CoutPad] and fold
// bias (+ BN scale/shift) into the epilogue's per-channel affine.
void upload_conv(sivo_segnet &S, Op &op, const float *W, const float *bias, int H, int Wd, bool keep_ties, int guard_level = 0) {
    op.guard_level = guard_level;
    const int ks = op.ks, cin = op.cin, cout = op.cout;
    std::vector<float> wt;
    static const bool force_v1 = SIVO_DIAG_ENV("SIVO_CONV_V1") != nullptr;
    static const bool no_wino = SIVO_DIAG_ENV("SIVO_NO_WINOGRAD") != nullptr;
    // F(4x4,3x3) for the wide layers (4x fewer MFMA flops; costs ~2e-4 of the 1e-3 logit budget) — SIVO_NO_WINO4 disables
    static const bool no_wino4 = SIVO_DIAG_ENV("SIVO_NO_WINO4") != nullptr;
    const size_t wino4_budget = (size_t)S.opt.wino4_workspace_mb << 20;
    // keep_ties: the layer belongs to the sample-invariant encoder prefix (conv1_1 .. conv3_3), whose outputs decide the
    // switches of pool1..pool3.  Over a flat image region (sky, saturated pixels) the four elements of a pooling window
    // are EXACTLY equal in the reference, which then takes the first; the direct and the F(2x2) kernels reproduce that (a
    // constant patch gives bit-identical outputs at every position of a tile), F(4x4) does not (4d - 5d + d is not
    // exactly 0 in fp32), its noise survives the following layers, and the switch picked instead moves the value by a
    // pixel after unpooling.  Measured on the KITTI test frame with F(4x4) in the prefix: 5672 instead of 27 differing
    // switches at pool1, 0.46 % instead of 0.04 % of the final class map differing from the oracle.  The prefix runs once
    // per frame, so keeping it on F(2x2) costs 0.13 ms.
    const bool f4_ok = !no_wino && !keep_ties && guard_level < 1;      // (a layer the accuracy guard took off F(4x4): level >= 1)
    // Narrow layers (<= SIVO_D3_MAXC = 128 channels in and out): the direct f16x3 kernel (conv3_h3.hip), whenever the handle
    // runs its F(4x4) GEMMs on f16x3 as well (SIVO_GEMM unset) — SIVO_D3=0 disables.  A direct kernel treats every output
    // position alike, so it also keeps the exact pooling ties of the prefix.
    const bool gemm_default = S.opt.gemm == 0;
    const bool no_d3 = S.opt.no_direct_f16x3 != 0;
    // (the sample-invariant prefix runs once per frame with N = 1: there the alternative is the fused F(2x2) kernel on the fp32
    // pipe, not the F(4x4) GEMM, and the direct kernel wins up to 256 channels — SIVO_D3_MAXC_SHARED)
    // (guard level 1: the layer left the F(4x4) GEMM for accuracy, not for speed — the direct f16x3 kernel takes it at any width)
    const int d3_maxc = guard_level == 1 ? (1 << 30) : keep_ties ? (SIVO_DIAG_ENV("SIVO_D3_MAXC_SHARED") ? std::atoi(SIVO_DIAG_ENV("SIVO_D3_MAXC_SHARED")) : 256)
                                  : (SIVO_DIAG_ENV("SIVO_D3_MAXC") ? std::atoi(SIVO_DIAG_ENV("SIVO_D3_MAXC")) : 128);
    const bool d3_prefix = !(SIVO_DIAG_ENV("SIVO_D3_PREFIX") && std::atoi(SIVO_DIAG_ENV("SIVO_D3_PREFIX")) == 0);
    op.d3 = !no_d3 && !no_wino && gemm_default && guard_level < 2 && (d3_prefix || !keep_ties) && cin <= d3_maxc && cout <= d3_maxc && conv3_h3_supported(ks, cin, cout, H, Wd, false);
    if (op.d3) {
        std::vector<uint16_t> planes;
        op.d3_uscale = conv3_h3_pack_weights(W, cin, cout, planes);
        op.d_wd3 = dev_alloc<uint16_t>(planes.size());
        S.owned.push_back(op.d_wd3);
        SIVO_HIP(hipMemcpy(op.d_wd3, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    op.wino4 = !op.d3 && f4_ok && !no_wino4 && wino4_supported(ks, cin, cout, H, Wd);
    // narrow layers (below the F(4x4) GEMM threshold): the fused F(4x4) kernel — SIVO_NO_WINO4F falls back to fused F(2x2)
    static const bool no_wino4f = SIVO_DIAG_ENV("SIVO_NO_WINO4F") != nullptr;
    op.wino4f = !op.wino4 && f4_ok && !no_wino4f && wino4f_supported(ks, cin, cout, H, Wd);
    op.wino = !op.wino4 && !op.wino4f && !no_wino && guard_level < 3 && wino_supported(ks, cin, cout, H, Wd);
    op.v2 = !op.wino4 && !op.wino4f && !op.wino && conv2_supported(ks) && !force_v1;
    // SegNet-Basic's 64 -> 64 7x7 layers: bf16x6 on the bf16 matrix cores (SIVO_CONV7=f32 keeps the fp32-MFMA direct kernel)
    const bool conv7_f32 = S.opt.conv7_fp32 != 0;
    op.c7x6 = !conv7_f32 && conv7_x6_supported(ks, cin, cout, H, Wd);
    if (op.c7x6) {
        std::vector<uint16_t> planes;
        conv7_x6_pack_weights(W, cin, cout, planes);
        op.d_wx6 = dev_alloc<uint16_t>(planes.size());
        S.owned.push_back(op.d_wx6);
        SIVO_HIP(hipMemcpy(op.d_wx6, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (!no_d3 && gemm_default && guard_level < 1 && conv7_h3_supported(ks, cin, cout, H, Wd)) {
            std::vector<uint16_t> hp;
            op.d3_uscale = conv7_h3_pack_weights(W, cin, cout, hp);
            op.d_wd3 = dev_alloc<uint16_t>(hp.size());
            S.owned.push_back(op.d_wd3);
            SIVO_HIP(hipMemcpy(op.d_wd3, hp.data(), hp.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            op.c7h3 = true;
        }
    }
    if (op.wino4f) {
        wino4f_pack_weights(W, cin, cout, wt, &op.cout_pad);
    } else if (op.wino4) {
        wino4_pack_weights(W, cin, cout, wt, &op.cout_pad);
        // SIVO_GEMM=f32 keeps the batched GEMM on the fp32 matrix-core instructions; default: bf16x6 (conv_wino4.hip)
        const bool gemm_f32 = S.opt.gemm == 2;
        if (!gemm_f32 && wino4_x6_supported(cin, op.cout_pad)) {
            std::vector<uint16_t> planes;
            wino4_x6_pack_weights(wt, cin, op.cout_pad, planes);
            op.d_wx6 = dev_alloc<uint16_t>(planes.size());
            S.owned.push_back(op.d_wx6);
            SIVO_HIP(hipMemcpy(op.d_wx6, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        // SIVO_GEMM=x6 / f32 keep the bf16x6 / fp32 GEMM; default: f16x3 (conv_wino4_h3.hip), with the bf16 planes resident
        // as well: they run the calibration pass and any frame whose values leave the fp16 range
        const bool gemm_x6 = S.opt.gemm == 1, gemm_f32_now = S.opt.gemm == 2;
        if (!gemm_x6 && !gemm_f32_now && wino4_h3_supported(cin, op.cout_pad)) {
            std::vector<uint16_t> planes;
            op.h3_uscale = wino4_h3_pack_weights(wt, cin, op.cout_pad, planes);
            op.d_wh3 = dev_alloc<uint16_t>(planes.size());
            S.owned.push_back(op.d_wh3);
            SIVO_HIP(hipMemcpy(op.d_wh3, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        op.wino4_group = wino4_group(S.T, cin, cout, H, Wd, wino4_budget);
        S.wino4_ws_floats = std::max(S.wino4_ws_floats, wino4_workspace_floats(op.wino4_group, cin, cout, H, Wd));
    } else if (op.wino) {
        static const int env_cfg = SIVO_DIAG_ENV("SIVO_WINO_CFG") ? std::atoi(SIVO_DIAG_ENV("SIVO_WINO_CFG")) : 0;
        op.wino_cfg = env_cfg;
        wino_pack_weights(W, cin, cout, op.wino_cfg, wt, &op.cout_pad);
    } else if (op.v2) {
        conv2_pack_weights(W, ks, cin, cout, wt, &op.cout_pad);
    } else {
        const int KC = conv_k_chunk(ks, cin), BN = conv_cout_tile(ks, cout);
        op.cout_pad = cdiv(cout, BN) * BN;
        const int nchunks = cdiv(cin, KC), taps = ks * ks;
        wt.assign((size_t)nchunks * taps * KC * op.cout_pad, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t) {
                    const size_t dst = (((size_t)(ci / KC) * taps + t) * KC + (ci % KC)) * op.cout_pad + co;
                    wt[dst] = W[((size_t)co * cin + ci) * taps + t];
                }
    }
    op.d_w = dev_alloc<float>(wt.size());
    S.owned.push_back(op.d_w);
    SIVO_HIP(hipMemcpy(op.d_w, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> sc(cout, 1.f), sh(bias, bias + cout);
    op.d_scale = dev_alloc<float>(cout);
    op.d_shift = dev_alloc<float>(cout);
    S.owned.push_back(op.d_scale);
    S.owned.push_back(op.d_shift);
    SIVO_HIP(hipMemcpy(op.d_scale, sc.data(), cout * sizeof(float), hipMemcpyHostToDevice));
    SIVO_HIP(hipMemcpy(op.d_shift, sh.data(), cout * sizeof(float), hipMemcpyHostToDevice));
}

void fold_bn(Op &op, const float *scale, const float *shift) {
    // y = scale*(acc*s0 + b0) + shift = (scale*s0)*acc + (scale*b0 + shift)
    std::vector<float> s0(op.cout), b0(op.cout);
    SIVO_HIP(hipMemcpy(s0.data(), op.d_scale, op.cout * sizeof(float), hipMemcpyDeviceToHost));
    SIVO_HIP(hipMemcpy(b0.data(), op.d_shift, op.cout * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < op.cout; ++c) {
        b0[c] = scale[c] * b0[c] + shift[c];
        s0[c] = scale[c] * s0[c];
    }
    SIVO_HIP(hipMemcpy(op.d_scale, s0.data(), op.cout * sizeof(float), hipMemcpyHostToDevice));
    SIVO_HIP(hipMemcpy(op.d_shift, b0.data(), op.cout * sizeof(float), hipMemcpyHostToDevice));
}


// prefix_rows > 0: build only the SAMPLE-INVARIANT PREFIX of the net (the layers in front of the first test-time Dropout) at a
// geometry of prefix_rows x W — the row band one rank computes when the prefix is split over ranks (PrefixBands below).  Such a
// handle has shared blobs only, no Softmax / classifier / workspace, and is not calibrated: its owner copies its own scales in.
std::unique_ptr<sivo_segnet> build(const ProtoNet &net, int t_override, const float *weights, size_t n_weights,
                                   int device, const SivoSegnetOptions &opt, const std::map<std::string, int> &guard_levels, int prefix_rows) {
    std::unique_ptr<sivo_segnet> Sp(new sivo_segnet);
    sivo_segnet &S = *Sp;
    S.opt = opt;
    S.device = device;
    S.T = t_override > 0 ? t_override : net.shape[0];
    S.C = net.shape[1]; S.H = prefix_rows > 0 ? prefix_rows : net.shape[2]; S.W = net.shape[3];
    // reference constructor checks (bayesian_segnet.cpp:64-70)
    if (S.C != 3) throw std::invalid_argument("Input layer must have 3 channels!");
    if (S.T <= 1) throw std::invalid_argument("Input layer must have a batch size greater than 1!");
    if (S.H <= 0 || S.W <= 0) throw std::invalid_argument("Input layer must have a positive geometry!");
    if (prefix_rows <= 0 && count_params(net) != n_weights) {
        std::ostringstream m;
        m << "weights hold " << n_weights << " values but the prototxt implies " << count_params(net);
        throw std::invalid_argument(m.str());
    }

    DeviceGuard dg(device);
    S.proto = net;
    S.guard_levels_used = guard_levels;
    S.input_blob = new_blob(S, net.input, S.C, S.H, S.W, true);
    size_t woff = 0;
    int site = 0;
    // producer[blob] = index of the op that can still absorb in-place BN/ReLU/Dropout
    std::map<int, int> absorber;
    for (const ProtoLayer &L : net.layers) {
        auto bottom = [&](size_t i) -> int {
            auto it = S.blob_id.find(L.bottom.at(i));
            if (it == S.blob_id.end()) throw std::invalid_argument("layer '" + L.name + "': unknown bottom '" + L.bottom[i] + "'");
            return it->second;
        };
        const bool inplace = !L.top.empty() && !L.bottom.empty() && L.top[0] == L.bottom[0];
        if (L.type == "Convolution") {
            if (L.stride != 1 || (L.kernel_size != 1 && L.kernel_size != 3 && L.kernel_size != 7) ||
                L.pad != L.kernel_size / 2)
                throw std::runtime_error("Convolution '" + L.name + "': only stride-1 'same' 1x1/3x3/7x7 kernels are supported");
            const int bi = bottom(0);
            const Blob b = S.blobs[bi];
            Op op;
            op.kind = OP_CONV; op.in = bi; op.ks = L.kernel_size; op.cin = b.C; op.cout = L.num_output;
            op.out = new_blob(S, L.top[0], L.num_output, b.H, b.W, b.shared);
            const size_t nw = (size_t)op.cout * op.cin * op.ks * op.ks;
            bool keep_ties = b.shared;
            if (const char *extra = SIVO_DIAG_ENV("SIVO_KEEP_TIES_LAYERS"))        // comma-separated layer names (experiments)
                keep_ties = keep_ties || ("," + std::string(extra) + ",").find("," + L.name + ",") != std::string::npos;
            op.w_off = woff;
            const auto gl = guard_levels.find(L.name);
            upload_conv(S, op, weights + woff, weights + woff + nw, b.H, b.W, keep_ties, gl == guard_levels.end() ? 0 : gl->second);
            woff += nw + op.cout;
            op.flops = 2.0 * op.ks * op.ks * op.cin * op.cout * (double)b.H * b.W;
            op.name = L.name;
            {
                char kn[96];
                const int bn = conv_cout_tile(op.ks, op.cout), kc = conv_k_chunk(op.ks, op.cin);
                if (op.c7h3)
                    snprintf(kn, sizeof kn, "conv7_h3_kernel");
                else if (op.c7x6)
                    snprintf(kn, sizeof kn, "conv7_x6_kernel");
                else if (op.wino4f)
                    snprintf(kn, sizeof kn, "conv_wino4f_kernel");
                else if (op.wino4)
                    snprintf(kn, sizeof kn, "conv_wino4 (input + gemm + output kernels)");
                else if (op.wino)
                    snprintf(kn, sizeof kn, op.wino_cfg == 2 ? "conv_wino_kernel<6,2,2,4>" : op.wino_cfg == 1 ? "conv_wino_kernel<4,1,2,8>" : "conv_wino_kernel<2,2,2,4>");
                else if (op.v2)
                    snprintf(kn, sizeof kn, "conv_mfma2_kernel<%d,%d,32,%d,%d,%d>", op.ks, bn == 128 ? 4 : 8, bn, bn == 128 ? 2 : 4,
                             bn == 128 ? 2 : 1);
                else
                    snprintf(kn, sizeof kn, "conv_mfma_kernel<%d,%d,32,%d,%d,%d,%d>", op.ks, bn == 128 ? 4 : 8, bn, kc,
                             bn == 128 ? 2 : 4, bn == 128 ? 2 : 1);
                op.kernel = kn;
            }
            // algorithmic HBM bytes: input + output activations once, weights once
            op.bytes = 4.0 * ((double)b.C * b.H * b.W + (double)op.cout * b.H * b.W + (double)nw);
            S.ops.push_back(op);
            absorber[op.out] = (int)S.ops.size() - 1;
        } else if (L.type == "BN") {
            if (L.bn_mode != "INFERENCE") throw std::runtime_error("BN '" + L.name + "': only bn_mode INFERENCE is supported");
            const int bi = bottom(0);
            auto it = absorber.find(bi);
            if (!inplace || it == absorber.end() || S.ops[it->second].kind != OP_CONV || S.ops[it->second].relu ||
                S.ops[it->second].drop_site >= 0)
                throw std::runtime_error("BN '" + L.name + "' must follow a Convolution in place");
            const int C = S.blobs[bi].C;
            fold_bn(S.ops[it->second], weights + woff, weights + woff + C);
            woff += 2 * (size_t)C;
        } else if (L.type == "ReLU") {
            const int bi = bottom(0);
            auto it = absorber.find(bi);
            if (!inplace || it == absorber.end() || S.ops[it->second].kind != OP_CONV || S.ops[it->second].drop_site >= 0)
                throw std::runtime_error("ReLU '" + L.name + "' must follow a Convolution in place");
            S.ops[it->second].relu = true;
        } else if (L.type == "Pooling") {
            if (L.pool != "MAX" || L.kernel_size != 2 || L.stride != 2 || L.top.size() != 2)
                throw std::runtime_error("Pooling '" + L.name + "': only MAX 2x2 stride 2 with a mask top is supported");
            const int bi = bottom(0);
            const Blob b = S.blobs[bi];
            Op op;
            op.kind = OP_POOL; op.in = bi;
            const int Ho = (b.H - 2 + 1) / 2 + 1, Wo = (b.W - 2 + 1) / 2 + 1;   // ceil((H-k)/s)+1
            op.out = new_blob(S, L.top[0], b.C, Ho, Wo, b.shared);
            op.out2 = new_blob(S, L.top[1], b.C, Ho, Wo, b.shared, true);
            S.blobs[op.out2].src_W = b.W;
            op.name = L.name; op.kernel = "maxpool2_kernel";
            op.bytes = 4.0 * b.C * b.H * b.W + 5.0 * b.C * Ho * Wo;
            S.ops.push_back(op);
            absorber.erase(bi);
            absorber[op.out] = (int)S.ops.size() - 1;
        } else if (L.type == "Dropout") {
            const int my_site = site++;
            if (!L.sample_weights_test) continue;  // plain Caffe dropout is the identity at test time
            if (prefix_rows > 0) break;            // the prefix ends in front of the first test-time dropout
            if (S.prefix_weights.empty() && weights) S.prefix_weights.assign(weights, weights + woff);
            if (std::fabs(L.dropout_ratio - 0.5f) > 1e-6f)
                throw std::runtime_error("Dropout '" + L.name + "': only dropout_ratio 0.5 is supported");
            const int bi = bottom(0);
            auto it = absorber.find(bi);
            if (inplace && it != absorber.end() && S.ops[it->second].drop_site < 0 && !S.blobs[bi].shared) {
                S.ops[it->second].drop_site = my_site;       // conv / pool epilogue
            } else if (inplace && it != absorber.end() && S.ops[it->second].kind == OP_POOL && S.blobs[bi].shared) {
                // pooled output of a shared blob becomes per-sample: pool kernel broadcasts + drops
                S.ops[it->second].drop_site = my_site;
                S.blobs[bi].shared = false;
            } else {
                // general case: separate kernel, out of place into a per-sample blob that takes over the name
                Op op;
                op.kind = OP_DROPOUT; op.in = bi; op.drop_site = my_site;
                const Blob b = S.blobs[bi];
                op.out = new_blob(S, L.top[0], b.C, b.H, b.W, false);
                op.name = L.name; op.kernel = "dropout_kernel"; op.bytes = 8.0 * b.chw();
                S.ops.push_back(op);
            }
            absorber.erase(bi);
        } else if (L.type == "Upsample") {
            if (L.scale != 2 || L.bottom.size() != 2) throw std::runtime_error("Upsample '" + L.name + "': only scale 2 with a mask bottom");
            const int bi = bottom(0), mi = bottom(1);
            const Blob b = S.blobs[bi], m = S.blobs[mi];
            if (!m.is_mask || m.C != b.C || m.H != b.H || m.W != b.W)
                throw std::runtime_error("Upsample '" + L.name + "': mask does not match the bottom");
            Op op;
            op.kind = OP_UNPOOL; op.in = bi; op.in2 = mi;
            op.out = new_blob(S, L.top[0], b.C, b.H * 2, b.W * 2, b.shared && m.shared);
            op.name = L.name; op.kernel = "unpool2_kernel"; op.bytes = 5.0 * b.chw() + 16.0 * b.chw();
            S.ops.push_back(op);
            absorber.erase(bi);
        } else if (L.type == "LRN") {
            const int bi = bottom(0);
            const Blob b = S.blobs[bi];
            Op op;
            op.kind = OP_LRN; op.in = bi; op.local_size = L.local_size; op.alpha = L.alpha; op.beta = L.beta;
            op.out = new_blob(S, L.top[0], b.C, b.H, b.W, b.shared);
            op.name = L.name; op.kernel = "lrn_kernel"; op.bytes = 8.0 * b.chw();
            S.ops.push_back(op);
        } else if (L.type == "Softmax") {
            S.has_softmax = true;
            S.logits_blob = bottom(0);
        } else {
            throw std::runtime_error("layer '" + L.name + "': unsupported type '" + L.type + "'");
        }
    }
    if (prefix_rows <= 0) {
        if (!S.has_softmax) throw std::runtime_error("the network must end in a Softmax layer");
        S.classes = S.blobs[S.logits_blob].C;
        if (S.classes > 16) throw std::runtime_error("at most 16 classes are supported");
    }

    // sharedness must propagate forward through ops built before a later flip (pool+dropout flips its output)
    for (Op &op : S.ops) {
        bool sh = S.blobs[op.in].shared && (op.in2 < 0 || S.blobs[op.in2].shared) && op.drop_site < 0;
        if (op.kind == OP_DROPOUT) sh = false;
        S.blobs[op.out].shared = sh;
        if (op.out2 >= 0) S.blobs[op.out2].shared = S.blobs[op.in].shared;   // the argmax only depends on the input
        (sh ? S.flops_shared : S.flops_sample) += op.flops;
    }
    // Upsample -> Winograd convolution: the F(4x4) input transform / the F(2x2) patch loader reads the pooled tensor and
    // the window codes directly (4x fewer input bytes, no unpool kernel, the unpooled tensor is never written).
    // SIVO_NO_FUSE_UNPOOL disables.
    if (!SIVO_DIAG_ENV("SIVO_NO_FUSE_UNPOOL"))
        for (Op &u : S.ops) {
            if (u.kind != OP_UNPOOL) continue;
            Op *consumer = nullptr;
            int uses = u.out == S.logits_blob ? 2 : 0;
            for (Op &c : S.ops)
                if (c.in == u.out || c.in2 == u.out) { ++uses; consumer = &c; }
            if (uses != 1 || consumer->kind != OP_CONV || consumer->in != u.out) continue;
            if (!consumer->wino4 && !consumer->wino4f && !consumer->c7x6 && !(consumer->wino && consumer->wino_cfg == 0)) continue;   // every Winograd path and the 7x7 bf16x6 kernel read through the pooling
            const Blob &pooled = S.blobs[u.in], &mask = S.blobs[u.in2], &up = S.blobs[u.out];
            if (pooled.shared && !up.shared) continue;            // (not produced by the reference nets)
            if (up.H != 2 * pooled.H || up.W != 2 * pooled.W || (pooled.W & 1)) continue;
            (void)mask;
            consumer->unpool_in = u.in; consumer->unpool_mask = u.in2;
            u.skip = true;
            S.blobs[u.out].fused_away = true;
        }
    // F(4x4) conv -> F(4x4) conv at the same resolution: the activation in between stays on chip (wino4_bridge_kernel).
    // SIVO_NO_FUSE_BRIDGE disables (the intermediate blob is then materialised and can be inspected).
    for (size_t i = 0; i < S.ops.size(); ++i) {
        Op &A = S.ops[i];
        if (A.kind != OP_CONV || !A.wino4) continue;
        const Blob &bo = S.blobs[A.out];
        const int N = bo.shared ? 1 : S.T;
        if (A.wino4_group < N) continue;                           // several passes over the workspace: plain path
        const int64_t P = (int64_t)N * ((bo.H + 3) / 4) * (bo.W / 4), Pp = (P + 127) / 128 * 128;
        S.wino4_slot_floats = std::max(S.wino4_slot_floats, (size_t)(36 * Pp * std::max<int64_t>(A.cin, A.cout_pad)));
        if (SIVO_DIAG_ENV("SIVO_NO_FUSE_BRIDGE") || A.out == S.logits_blob) continue;
        Op *B = nullptr;
        int uses = 0;
        for (Op &c : S.ops)
            if (c.in == A.out || c.in2 == A.out) { ++uses; B = &c; }
        if (uses != 1 || B->kind != OP_CONV || !B->wino4 || B->in != A.out || B->unpool_in >= 0) continue;
        const Blob &bn = S.blobs[B->out];
        if (bn.shared != bo.shared || bn.H != bo.H || bn.W != bo.W || B->wino4_group < N) continue;
        if (wino4_bridge_lds_bytes(bo.H, bo.W) > 150 * 1024) continue;
        A.w4_bridge = true; B->w4_bridged_in = true;
        A.bridge_to = (int)(B - S.ops.data());
        S.blobs[A.out].fused_away = true;
    }
    if (S.wino4_slot_floats) S.wino4_ws_floats = std::max(S.wino4_ws_floats, 3 * S.wino4_slot_floats);
    // F(4x4) conv -> MAX 2x2 pooling (per-sample part: conv4_3 -> pool4, conv5_3 -> pool5): the output transform holds
    // whole pooling windows, so it writes the pooled tensor + window codes (+ the pooling layer's dropout) directly.
    // SIVO_NO_FUSE_POOL disables.
    if (!SIVO_DIAG_ENV("SIVO_NO_FUSE_POOL"))
        for (size_t i = 0; i < S.ops.size(); ++i) {
            Op &A = S.ops[i];
            if (A.kind != OP_CONV || !A.wino4 || A.w4_bridge || A.out == S.logits_blob || S.blobs[A.out].shared) continue;
            int uses = 0, pi = -1;
            for (size_t k = 0; k < S.ops.size(); ++k)
                if (S.ops[k].in == A.out || S.ops[k].in2 == A.out) { ++uses; pi = (int)k; }
            if (uses != 1 || S.ops[pi].kind != OP_POOL || S.ops[pi].in != A.out || S.blobs[S.ops[pi].out2].shared) continue;
            if (S.blobs[A.out].W % 4) continue;
            A.pool_op = pi;
            S.ops[pi].skip = true;
            S.blobs[A.out].fused_away = true;
        }
    // Fork pooling (sample-invariant input, test-time Dropout in place on its output) -> F(4x4) convolution: the pooling kernel would
    // write T dropped copies of the same tensor (SegNet-Standard pool3: 12 x 5.8 MB) for the input transform to read back; instead the
    // pooling writes its values once and the input transform applies the dropout as it reads — the same counter-based word per
    // (element, site, global sample), so V is bit-identical.  SIVO_NO_FUSE_INDROP disables (diagnostic build: the A/B of the test).
    if (!SIVO_DIAG_ENV("SIVO_NO_FUSE_INDROP"))
        for (size_t i = 0; i < S.ops.size(); ++i) {
            Op &P = S.ops[i];
            if (P.kind != OP_POOL || P.skip || P.drop_site < 0 || !S.blobs[P.in].shared || S.blobs[P.out].shared || P.out == S.logits_blob) continue;
            int uses = 0, ci = -1;
            for (size_t k = 0; k < S.ops.size(); ++k)
                if (S.ops[k].in == P.out || S.ops[k].in2 == P.out || S.ops[k].unpool_in == P.out) { ++uses; ci = (int)k; }
            if (uses != 1) continue;
            Op &Cv = S.ops[(size_t)ci];
            if (Cv.kind != OP_CONV || !Cv.wino4 || Cv.in != P.out || Cv.unpool_in >= 0 || Cv.w4_bridged_in || S.blobs[P.out].W % 4) continue;
            Cv.in_drop_site = P.drop_site;
            P.drop_moved = true;
            S.blobs[P.out].shared = true;
            S.blobs[P.out].drop_pending = P.drop_site;
        }
    // classifier convolution -> Softmax -> mean over the samples -> argmax / max / entropy in one kernel (conv_cls_mc.hip):
    // the logits stay on chip whenever the caller asks for the maps or the probability sums only.  SIVO_NO_FUSE_MC disables.
    if (!SIVO_DIAG_ENV("SIVO_NO_FUSE_MC") && !S.ops.empty()) {
        Op &L = S.ops.back();
        const Blob &bi = S.blobs[L.in], &bo = S.blobs[L.out];
        if (L.kind == OP_CONV && L.out == S.logits_blob && !bo.shared && !bi.shared && !bi.fused_away && L.pool_op < 0 &&
            L.unpool_in < 0 && !L.w4_bridged_in && L.drop_site < 0 && cls_mc_supported(L.ks, L.cin, L.cout, bi.H, bi.W)) {
            std::vector<float> wt;
            cls_mc_pack_weights(weights + L.w_off, L.cin, L.cout, wt);
            L.d_w_mc = dev_alloc<float>(wt.size());
            S.owned.push_back(L.d_w_mc);
            SIVO_HIP(hipMemcpy(L.d_w_mc, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
            S.cls_op = (int)S.ops.size() - 1;
            // the f16x3 form (conv_cls_h3.hip), when the handle runs f16x3 at all (SIVO_GEMM unset, SIVO_D3 not 0)
            const bool f16x3_handle = S.opt.gemm == 0 && !S.opt.no_direct_f16x3;
            const auto cgl = guard_levels.find(L.name);
            if (cgl != guard_levels.end()) L.guard_level = cgl->second;
            if (f16x3_handle && L.guard_level < 1 && cls_h3_supported(L.ks, L.cin, L.cout, bi.H, bi.W)) {       // (level >= 1: the accuracy guard took it off f16x3)
                std::vector<uint16_t> planes;
                L.d3_uscale = cls_h3_pack_weights(weights + L.w_off, L.cin, L.cout, planes);
                L.d_wd3 = dev_alloc<uint16_t>(planes.size());
                S.owned.push_back(L.d_wd3);
                SIVO_HIP(hipMemcpy(L.d_wd3, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                L.c3 = true;
            }
        }
    }
    // Direct f16x3 layer <- direct f16x3 layer (or an F(4x4) layer's output transform): the activation in between goes in the
    // consumer's packed form (conv3_h3.hip header).  The fp32 blob stays allocated: the calibration pass and every frame after
    // an fp16 overflow run the fp32 kernels.  SIVO_D3_PK=0 disables (read per handle: the tests build both).
    S.pk_on = !S.opt.no_packed_activations;
    if (S.pk_on)
        for (size_t bi_ = 0; bi_ < S.ops.size(); ++bi_) {
            Op &B = S.ops[bi_];
            const bool b_cls = B.c3 && (int)bi_ == S.cls_op;          // the fused classifier + MC kernel on f16x3 (conv_cls_h3.hip)
            if (B.kind != OP_CONV || !(B.d3 || b_cls) || B.skip || B.drop_site >= 0 || B.pool_op >= 0) continue;
            const bool unpool = B.unpool_in >= 0;
            if (b_cls && unpool) continue;
            const int X = unpool ? B.unpool_in : B.in;
            if (X == S.input_blob || X == S.logits_blob || S.blobs[X].fused_away || S.blobs[X].C % 16) continue;
            int uses = 0, ai = -1, pi = -1;
            for (size_t k = 0; k < S.ops.size(); ++k) {
                const Op &c = S.ops[k];
                if (c.out == X && !c.skip) ai = (int)k;
                if (c.skip) continue;
                if ((c.in == X && c.unpool_in < 0) || c.in2 == X || c.unpool_in == X) ++uses;
                if (unpool && c.kind == OP_POOL && c.out2 == B.unpool_mask) pi = (int)k;
            }
            if (uses != 1 || ai < 0 || (unpool && pi < 0)) continue;
            Op &A = S.ops[ai];
            if (A.kind != OP_CONV || A.pool_op >= 0 || A.w4_bridge) continue;
            // (a direct producer must itself run whenever the handle runs f16x3 — the conditions of d3_now in run_ops —, and fp32
            // through an Upsample + packed output is not built)
            const bool a_direct = A.d3 && A.drop_site < 0 && !(A.unpool_in >= 0 && !A.pk_in) &&
                                  conv3_h3_supported(A.ks, A.cin, A.cout, S.blobs[A.in].H, S.blobs[A.in].W, A.unpool_in >= 0);
            if (!a_direct && !A.wino4) continue;
            const Blob &bin = S.blobs[B.in];                       // the layer's input geometry (the Upsample's output when it reads through one)
            if (!b_cls && !conv3_h3_supported(B.ks, B.cin, B.cout, bin.H, bin.W, unpool)) continue;
            int tile_h = 8, tile_w = 64;
            if (b_cls) cls_h3_tile(&tile_h, &tile_w);
            const int tx = (bin.W + tile_w - 1) / tile_w, ty = (bin.H + tile_h - 1) / tile_h;
            Blob &bx = S.blobs[X];
            bx.pk_Hp = (unpool ? ty * 4 : ty * tile_h) + 2; bx.pk_Wp = (unpool ? tx * 32 : tx * tile_w) + 2;
            if (bx.pk_Hp < bx.H + 2 || bx.pk_Wp < bx.W + 2 || (int64_t)bx.C * bx.pk_Hp * bx.pk_Wp * 4 >= (1ll << 31)) { bx.pk_Hp = bx.pk_Wp = 0; continue; }
            if (unpool) {
                Blob &bm = S.blobs[B.unpool_mask];
                bm.bits_Hp = bx.pk_Hp; bm.bits_Wp = bx.pk_Wp;
                S.ops[pi].make_bits = true;
            }
            B.pk_in = true;
            A.pk_to = (int)bi_;
        }
    // allocate
    for (Blob &b : S.blobs) {
        if (b.fused_away) continue;
        const size_t n = (size_t)(b.shared ? 1 : S.T) * b.chw();
        b.d = b.is_mask ? (void *)dev_alloc<uint8_t>(n) : (void *)dev_alloc<float>(n);
        S.owned.push_back(b.d);
        if (b.pk_Hp) {          // zeroed once: producers write the interior only, the border stays zero for good
            const size_t nb = pk_bytes(b.shared ? 1 : S.T, b.C, b.pk_Hp, b.pk_Wp);
            SIVO_HIP(hipMalloc(&b.d_pk, nb));
            S.owned.push_back(b.d_pk);
            SIVO_HIP(hipMemset(b.d_pk, 0, nb));
        }
        if (b.bits_Hp) {
            const size_t nd = (size_t)(b.shared ? 1 : S.T) * b.bits_sample_dwords();
            b.d_bits = dev_alloc<uint32_t>(nd);
            S.owned.push_back(b.d_bits);
            SIVO_HIP(hipMemset(b.d_bits, 0, nd * sizeof(uint32_t)));
        }
    }
    if (S.wino4_ws_floats) {
        const int env_lanes = S.opt.lanes;
        S.ws_lanes = std::max(1, std::min(env_lanes, (int)sivo_segnet::MAX_LANES));
        S.d_wino4_ws = dev_alloc<float>((size_t)S.ws_lanes * S.wino4_ws_floats);      // one region per lane
        S.owned.push_back(S.d_wino4_ws);
    }
    const int64_t hw = (int64_t)S.H * S.W;
    S.d_image = dev_alloc<uint8_t>(hw * 3);
    S.d_prob_sum = dev_alloc<float>(std::max(S.classes, 1) * hw);
    S.d_classes = dev_alloc<uint8_t>(hw);
    S.d_conf = dev_alloc<double>(hw);
    S.d_ent = dev_alloc<double>(hw);
    for (void *p : {(void *)S.d_image, (void *)S.d_prob_sum, (void *)S.d_classes, (void *)S.d_conf, (void *)S.d_ent})
        S.owned.push_back(p);
    SIVO_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (prefix_rows <= 0) calibrate_h3(S);
    return Sp;
}

}  // namespace sivo
