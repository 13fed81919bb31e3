// conv_cls_mc.hip — the classifier convolution fused with the Monte-Carlo post-processing.
//
// BayesianSegNet::segmentImage (reference src/bayesian_segnet/bayesian_segnet.cpp:299-318) ends in
//   Forward() -> "prob" blob (T, classes, H, W)              the last Convolution (conv1_1_D, 64 -> 15, 3x3) + Softmax
//   extractMeanConfidence  (:278-297)  f32 -> f64, mean over T
//   computeClasses / computeMaxConfidence / computeClassificationEntropy (:180-203, :262-276)
// Layer by layer that is T x 21.6 MB of logits written by the convolution and read back by the reduction, and a 3x3
// convolution with 15 output channels is a poor MFMA shape on top (80 GFLOP of direct products at T = 12 = 0.51 ms at the
// fp32 matrix-core peak for what is 1.1 GB of input, 0.18 ms of HBM time).  This kernel keeps the logits on chip:
//
//   workgroup = 8 x 32 output pixels, ALL T samples, 4 waves; per sample the 3x3 convolution runs as Winograd
//   F(2x2,3x3) on the fp32 matrix cores exactly like conv_wino.hip (lane-local B^T d B and A^T M A, K-chunks of 4
//   channels double-buffered in LDS, the pre-transformed weight slab copied by LDS-DMA) with ONE 16-cout block
//   (15 classes + 1 zero column): 16 accumulators x 4 VGPRs, 2.25x fewer matrix-core products than the direct form;
//   after the output transform a lane holds ONE class at 16 pixels; a 16 x 16 transpose inside each 16-lane group
//   (four xor-shuffle stages, no LDS, no barrier) gives it the 15 logits of ONE pixel, and it then does the Softmax
//   layer's arithmetic (fp32, sequential over the classes) and adds the probabilities to 15 f64 registers — the same
//   operations in the same order as mc_reduce_finalize_kernel (segnet_kernels.hip), so the maps equal that kernel's
//   on the same logits bit for bit;
//   (sample, K-chunk) form ONE software pipeline: the first chunk of sample s + 1 is staged under the last chunk of
//   sample s; after the last sample the thread writes its pixel's class / confidence / entropy (17 bytes).
// Optional outputs: the fp32 probability sums in the pixel-chunk-major layout of the multi-device reduce-scatter
// (segnet_multi.cpp), and the logits themselves (diagnostics / parity tests: the values the maps were computed from).
// HBM: input read once (+ halo, L2), 17 B per pixel written: 1.1 GB per frame at T = 12 instead of 1.1 + 0.26 + 0.26.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <vector>

#include "segnet_kernels.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K_NTHR = 256;                    // 4 waves = 4 m-tiles stacked vertically
constexpr int K_TH = 8, K_TW = 32;             // output pixels per workgroup
constexpr int K_PH = K_TH + 2, K_PWp = 36, K_EOFF = 19;     // de-interleaved patch rows as in conv_wino.hip
constexpr int K_CS = 368;                      // channel stride: 360 padded to 16 (mod 32)
static_assert(K_PH * K_PWp <= K_CS && K_CS % 32 == 16, "patch channel stride");

constexpr int cls_slab(int kc) { return 16 * kc * 16; }     // floats: [position][channel of the chunk][16 couts]

template <int KC>
__global__ __launch_bounds__(K_NTHR, 3) void conv_wino_cls_mc_kernel(ClsMcArgs a) {
    constexpr int WSLAB = cls_slab(KC), PATCH = KC * K_CS, BUF = PATCH + WSLAB;
    static_assert(WSLAB % 256 == 0 && (PATCH % 4) == 0, "whole KiB slabs, 16-byte aligned");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;

    // XCD-aware order (workgroup L runs on XCD L % 8): every XCD walks its own contiguous band of pixel tiles in row-major
    // order, so the halo rows / columns shared by neighbouring tiles are served by that XCD's L2
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) >> 3;
    const int bid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (bid >= P) return;
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x;
    const int x0 = tx * K_TW, y0 = ty * K_TH;
    const int64_t plane = (int64_t)a.H * a.W;

    // 4x4 input patch of tile (row wm, column li): LDS rows 2 wm .. 2 wm + 3, columns q = 2 li .. 2 li + 3
    const int a_base = lk * K_CS + (2 * wm) * K_PWp + li;
    const int b_base = PATCH + lk * 16 + li;                 // + (p * KC + c4 * 4) * 16: 64 consecutive dwords per read

    // ---- staging plan: interior float4s + 2 halo scalars per patch row
    constexpr int NV4 = KC * K_PH * (K_TW / 4), V4IT = (NV4 + K_NTHR - 1) / K_NTHR;
    constexpr int NSC = KC * K_PH * 2, SCIT = (NSC + K_NTHR - 1) / K_NTHR;
    int v_goff[V4IT], v_dst[V4IT];
    bool v_ok[V4IT];
#pragma unroll
    for (int it = 0; it < V4IT; ++it) {
        const int idx = tid + it * K_NTHR;
        const int seg = idx % (K_TW / 4), r = idx / (K_TW / 4);
        const int py = r % K_PH, c = r / K_PH;
        const int gy = y0 + py - 1, gx = x0 + seg * 4;
        v_ok[it] = idx < NV4 && gy >= 0 && gy < a.H && gx + 3 < a.W;
        v_goff[it] = v_ok[it] ? (int)(c * plane + (int64_t)gy * a.W + gx) : 0;
        // pixels x0+4s..+3 are q = 4s+1..4s+4: (v0, v2) -> O[2s], O[2s+1]; (v1, v3) -> E[2s+1], E[2s+2]
        v_dst[it] = idx < NV4 ? ((c * K_CS + py * K_PWp + 2 * seg) | (c << 24)) : -1;
    }
    int s_goff[SCIT], s_dst[SCIT];
#pragma unroll
    for (int it = 0; it < SCIT; ++it) {
        const int idx = tid + it * K_NTHR;
        const int h = idx % 2, r = idx / 2;
        const int py = r % K_PH, c = r / K_PH;
        const int px = h == 0 ? K_EOFF : 16;          // x = x0-1 is q = 0 -> E[0]; x = x0+32 is q = 33 -> O[16]
        const int gy = y0 + py - 1, gx = h == 0 ? x0 - 1 : x0 + K_TW;
        const bool ok = idx < NSC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        s_goff[it] = ok ? (int)(c * plane + (int64_t)gy * a.W + gx) : -1;
        s_dst[it] = idx < NSC ? ((c * K_CS + py * K_PWp + px) | (c << 24)) : -1;
    }
    f32x4 pv4[V4IT];
    float psc[SCIT];
    const int nchunks = (a.Cin + KC - 1) / KC;

    // the loads of a chunk are issued a K-chunk ahead and only touched again by commit_patch: out-of-image / out-of-range
    // items read a safe address (the select is on the ADDRESS) and are zeroed when they are written to LDS
    auto issue_patch = [&](int s, int chunk) {
        const float *psrc = a.in + (int64_t)s * a.in_sample_stride + (int64_t)chunk * KC * plane;
        const int cleft = a.Cin - chunk * KC;
#pragma unroll
        for (int it = 0; it < V4IT; ++it) {
            const bool ok = v_ok[it] && (v_dst[it] >> 24) < cleft;
            pv4[it] = *reinterpret_cast<const f32x4 *>(psrc + (ok ? v_goff[it] : 0));
        }
#pragma unroll
        for (int it = 0; it < SCIT; ++it) {
            const bool ok = s_goff[it] >= 0 && (s_dst[it] >> 24) < cleft;
            psc[it] = psrc[ok ? s_goff[it] : 0];
        }
    };
    auto commit_patch = [&](int buf, int chunk) {
        float *sp = lds + buf * BUF;
        const int cleft = a.Cin - chunk * KC;
#pragma unroll
        for (int it = 0; it < V4IT; ++it)
            if (v_dst[it] >= 0) {
                const bool ok = v_ok[it] && (v_dst[it] >> 24) < cleft;
                const f32x4 v = ok ? pv4[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
                float *q = sp + (v_dst[it] & 0xffffff);
                *reinterpret_cast<float2 *>(q) = make_float2(v[0], v[2]);                   // O[2s], O[2s+1]
                *reinterpret_cast<float2 *>(q + K_EOFF + 1) = make_float2(v[1], v[3]);      // E[2s+1], E[2s+2]
            }
#pragma unroll
        for (int it = 0; it < SCIT; ++it)
            if (s_dst[it] >= 0) {
                const bool ok = s_goff[it] >= 0 && (s_dst[it] >> 24) < cleft;
                sp[s_dst[it] & 0xffffff] = ok ? psc[it] : 0.f;
            }
    };
    constexpr int NDMA = WSLAB / 256 / 4;            // 1 KiB copies per wave
    static_assert(NDMA * 4 * 256 == WSLAB, "slab = whole KiB per wave");
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + (int64_t)chunk * WSLAB;
        float *dst = lds + buf * BUF + PATCH;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int kib = i * 4 + wm;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + kib * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + kib * 256), 16, 0, 0);
        }
    };

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double sum[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.0;

    // epilogue constants of this lane's class (column li of the 16-cout block; column 15 is padding)
    const bool cok = li < a.C;
    const float ep_sc = cok ? a.ep_scale[li] : 0.f, ep_sh = cok ? a.ep_shift[li] : 0.f;
    // the pixel this lane post-processes: number li of the lane's 2 x 8 output pixels
    const int prow = 2 * wm + (li >> 3), pcol = 8 * lk + (li & 7);
    const int gy = y0 + prow, gx = x0 + pcol;
    const bool pix_ok = gy < a.H && gx < a.W;
    const int64_t pix = (int64_t)gy * a.W + gx;

    issue_patch(0, 0);
    dma_weights(0, 0);
    commit_patch(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the LDS-DMA has landed
    __syncthreads();

    const int total = a.T * nchunks;
    int s = 0, chunk = 0;
    for (int g = 0; g < total; ++g) {
        const int cur = g & 1;
        const bool more = g + 1 < total;
        int s2 = s, c2 = chunk + 1;
        if (c2 == nchunks) { c2 = 0; ++s2; }
        if (more) { issue_patch(s2, c2); dma_weights(c2, cur ^ 1); }
        const float *sp = lds + cur * BUF;
#pragma unroll
        for (int c4 = 0; c4 < KC / 4; ++c4) {
            // ---- input transform V = B^T d B of this lane's (tile, channel c4*4 + lk)
            float d[4][4], t[4][4], V[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = sp[a_base + c4 * 4 * K_CS + r * K_PWp + ((c & 1) ? 0 : K_EOFF) + (c >> 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[r * 4 + 0] = t[r][0] - t[r][2];
                V[r * 4 + 1] = t[r][1] + t[r][2];
                V[r * 4 + 2] = t[r][2] - t[r][1];
                V[r * 4 + 3] = t[r][1] - t[r][3];
            }
            // ---- 16 positions, one MFMA each (A: 16 tiles x 4 channels, B: 4 channels x 16 couts)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float bf = sp[b_base + (p * KC + c4 * 4) * 16];
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf, acc[p], 0, 0, 0);
                if (c4 == KC / 4 - 1 && p == 10 && more) commit_patch(cur ^ 1, c2);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();

        if (chunk == nchunks - 1) {
            // ---- output transform Y = A^T M A (lane-local): this lane's class li, rows 2 wm + {0,1}, columns 8 lk .. 8 lk + 7
            float y[2][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sx[2][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float m0 = acc[0 + c][r], m1 = acc[4 + c][r], m2 = acc[8 + c][r], m3 = acc[12 + c][r];
                    sx[0][c] = m0 + m1 + m2;
                    sx[1][c] = m1 - m2 - m3;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    y[i][2 * r + 0] = sx[i][0] + sx[i][1] + sx[i][2];
                    y[i][2 * r + 1] = sx[i][1] - sx[i][2] - sx[i][3];
                }
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // x[q] = logit of class li at pixel q = 8 i + j of this lane
            float x[16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = y[i][j] * ep_sc + ep_sh;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    x[8 * i + j] = v;
                }
            // 16 x 16 transpose over the 16 lanes of a group: afterwards x[c] = logit of class c at pixel li
#pragma unroll
            for (int k = 1; k < 16; k <<= 1) {
                const bool up = (li & k) != 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & k) == 0) {
                        const float send = up ? x[r] : x[r | k];
                        const float recv = __shfl_xor(send, k, 64);
                        if (up) x[r] = recv; else x[r | k] = recv;
                    }
            }
            // ---- Softmax over the classes of this lane's pixel + the f64 sum over the samples
            if (a.logits && pix_ok) {
                float *lp = a.logits + (int64_t)s * a.C * plane + pix;
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < a.C) lp[(int64_t)c * plane] = x[c];
            }
            float m = x[0];
#pragma unroll
            for (int c = 1; c < 16; ++c)
                if (c < a.C) m = x[c] > m ? x[c] : m;
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < a.C) { x[c] = expf(x[c] - m); den = __fadd_rn(den, x[c]); }
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < a.C) sum[c] += (double)__fdiv_rn(x[c], den);
        }
        s = s2; chunk = c2;
    }
    if (!pix_ok) return;

    if (a.prob_sum) {
        // chunk == hw: [class][pixel]; otherwise pixel-chunk-major [pixel / chunk][class][pixel % chunk] (launch_mc_reduce)
        float *dst = a.prob_sum + (pix / a.sum_chunk) * a.C * a.sum_chunk + (pix % a.sum_chunk);
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < a.C) dst[(int64_t)c * a.sum_chunk] = (float)sum[c];
    }
    if (a.classes) {
        // mean (f64) / argmax with first-wins ties / max / entropy in bits with the exact-zero guard (bayesian_segnet.cpp:38-44)
        const double dT = (double)a.T;
        int best = 0;
        double bv = sum[0] / dT;
        double ent = bv == 0 ? 0 : -1.0 * bv * log2(bv);
#pragma unroll
        for (int c = 1; c < 16; ++c)
            if (c < a.C) {
                const double v = sum[c] / dT;
                if (v > bv) { bv = v; best = c; }
                ent += v == 0 ? 0 : -1.0 * v * log2(v);
            }
        a.classes[pix] = (uint8_t)best;
        a.confidence[pix] = bv;
        a.entropy[pix] = ent;
    }
}

bool cls_mc_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 3 && cin >= 4 && cout >= 1 && cout <= 16 && (W % 8) == 0 && (H % 2) == 0;
}
int cls_mc_k_chunk() {
    static const int kc = [] {
        const char *e = std::getenv("SIVO_CLS_KC");
        return (e && std::atoi(e) == 8) ? 8 : 4;
    }();
    return kc;
}

// Caffe (Cout,Cin,3,3) -> U = G g G^T (f64, rounded once), [ceil(Cin/KC)][position * KC + ci % KC][16 couts], zero padded
void cls_mc_pack_weights(const float *W, int cin, int cout, std::vector<float> &out) {
    const int kc = cls_mc_k_chunk(), slab = cls_slab(kc), nchunks = (cin + kc - 1) / kc;
    out.assign((size_t)nchunks * slab, 0.f);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = W + ((size_t)co * cin + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) tmp[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const size_t base = (size_t)(ci / kc) * slab;
            for (int p = 0; p < 16; ++p) out[base + (size_t)(p * kc + ci % kc) * 16 + co] = (float)U[p / 4][p % 4];
        }
}

void launch_conv_cls_mc(const ClsMcArgs &a0, hipStream_t s) {
    ClsMcArgs a = a0;
    a.tiles_x = (a.W + K_TW - 1) / K_TW;
    a.tiles_y = (a.H + K_TH - 1) / K_TH;
    if (a.sum_chunk <= 0 || a.sum_chunk > (int64_t)a.H * a.W) a.sum_chunk = (int64_t)a.H * a.W;
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) / 8;
    if (cls_mc_k_chunk() == 8) hipLaunchKernelGGL((conv_wino_cls_mc_kernel<8>), dim3((unsigned)(8 * per)), dim3(K_NTHR), 0, s, a);
    else hipLaunchKernelGGL((conv_wino_cls_mc_kernel<4>), dim3((unsigned)(8 * per)), dim3(K_NTHR), 0, s, a);
}

}  // namespace sivo
