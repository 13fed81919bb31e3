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
#include <string>
#include <vector>

#include "common.hpp"
#include "segnet_kernels.hpp"
#include "softmax.hpp"

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
#include "conv_cls_mc_sample.inc"
        }
        s = s2; chunk = c2;
    }
    if (!pix_ok) return;
#include "conv_cls_mc_maps.inc"
}


// ---------------------------------------------------------------------------------------------------------------------
// Second form of the same kernel: EVERYTHING is staged by LDS-DMA into a ring of NB stage buffers, D = NB - 1 K-chunks ahead.
// The register-staged form above has its loads in flight for less than one K-chunk (they are issued at the top of an
// iteration and must be in LDS at its end): measured on MI355X that is what bounds it — 0.70 ms per frame at T = 12 with the
// matrix cores 33 % busy and 1.7 TB/s of input, i.e. one 5.9 KB patch per workgroup in flight against ~1.7 us of loaded
// memory latency.  Here a stage is 6 dword-gather DMAs per wave for the patch (each instruction fills 64 consecutive LDS
// dwords; the lanes' source offsets are chosen so that the LDS image IS the de-interleaved patch layout; halo positions
// outside the image and padding slots pass an out-of-range offset and receive the buffer load's 0) + one 1 KiB weight DMA per wave: 7 vector-memory operations per
// wave and stage, counted with s_waitcnt vmcnt(7 (D - 1)) — the stage needed now has landed, the D - 1 younger ones stay
// in flight.  One fence-free barrier per K-chunk (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would drain vmcnt).
// No staging registers, no LDS writes by the waves, D x 10 KB per workgroup in flight.  Requires Cin % 4 == 0.
constexpr int D_PATCH = 1536;                  // dword slots of a stage's patch image: 4 channels x 368, padded to 24 x 64
constexpr int D_WSLAB = cls_slab(4);
constexpr int D_BUF = D_PATCH + D_WSLAB;       // 10 KiB
constexpr int D_NI = 7;                        // vector-memory operations per wave and stage

__device__ __forceinline__ void cls_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NB>
__global__ __launch_bounds__(K_NTHR, 3) void conv_wino_cls_mc_dma_kernel(ClsMcArgs a) {
    constexpr int D = NB - 1;
    static_assert(D >= 1 && D_NI * (D - 1) < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) float lds[NB * D_BUF];

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) >> 3;
    const int bid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (bid >= P) return;
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x;
    const int x0 = tx * K_TW, y0 = ty * K_TH;
    const int64_t plane = (int64_t)a.H * a.W;

    const int a_base = lk * K_CS + (2 * wm) * K_PWp + li;
    const int b_base = D_PATCH + lk * 16 + li;

    // ---- DMA plan: instruction i of wave wm fills the LDS slots (4 i + wm) * 64 + lane of a stage's patch image.
    // slot -> (channel c, patch row py, column col): col 0..16 = O[col] (q = 2 col + 1), col 19..35 = E[col - 19] (q = 2 (col - 19)),
    // q = x - x0 + 1; everything else (row padding, channel padding, positions outside the image) reads the zero word.
    // Byte offsets from the base of a stage's 4 input planes.  The DMAs are BUFFER loads (buffer_load_dword ... offen lds):
    // a wave-uniform descriptor of the sample's input (base, Cin planes), the stage's plane offset in the scalar offset, the
    // lane's part in a 32-bit VGPR — and the hardware bounds check returns 0 for an offset beyond the descriptor's range,
    // which is what the lanes without a source (halo outside the image, padding slots) pass.
    uint32_t d_off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int slot = (4 * i + wm) * 64 + lane;
        const int c = slot / K_CS, rem = slot % K_CS;
        const int py = rem / K_PWp, col = rem % K_PWp;
        const int q = col < 17 ? 2 * col + 1 : 2 * (col - K_EOFF);
        const int gyy = y0 + py - 1, gxx = x0 - 1 + q;
        const bool ok = c < 4 && rem < K_PH * K_PWp && (col < 17 || col >= K_EOFF) && gyy >= 0 && gyy < a.H && gxx >= 0 && gxx < a.W;
        d_off[i] = ok ? (uint32_t)((c * plane + (int64_t)gyy * a.W + gxx) * 4) : 0xfffffff0u;
    }
    const uint32_t w_off = (uint32_t)((wm * 256 + lane * 4) * 4);
    const int nchunks = a.Cin / 4;
    const int total = a.T * nchunks;
    const uint32_t in_bytes = (uint32_t)((int64_t)a.Cin * plane * 4);        // one sample's input (launcher: < 2^31)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt, 0, nchunks * D_WSLAB * 4, 0x00020000);
    // stage counter of the next stage to issue (sample, chunk)
    int is = 0, ic = 0;
    auto issue_stage = [&](int g) {
        float *buf = lds + (g % NB) * D_BUF;
        const __amdgpu_buffer_rsrc_t in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (int64_t)is * a.in_sample_stride), 0, in_bytes, 0x00020000);
        const int soff = (int)((int64_t)ic * 4 * plane * 4);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void *)(buf + (4 * i + wm) * 64), 4,
                                                     (int)d_off[i], soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void *)(buf + D_PATCH + wm * 256), 16,
                                                 (int)w_off, ic * D_WSLAB * 4, 0, 0);
        if (++ic == nchunks) { ic = 0; ++is; }
    };

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double sum[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.0;
    const bool cok = li < a.C;
    const float ep_sc = cok ? a.ep_scale[li] : 0.f, ep_sh = cok ? a.ep_shift[li] : 0.f;
    const int prow = 2 * wm + (li >> 3), pcol = 8 * lk + (li & 7);
    const int gy = y0 + prow, gx = x0 + pcol;
    const bool pix_ok = gy < a.H && gx < a.W;
    const int64_t pix = (int64_t)gy * a.W + gx;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the epilogue constants: nothing but stage DMAs is counted from here on

#pragma unroll
    for (int g = 0; g < D; ++g)
        if (g < total) issue_stage(g);

    int s = 0, chunk = 0;
    for (int g = 0; g < total; ++g) {
        // stage g has landed once at most the younger stages' DMAs are outstanding (vmcnt counts in issue order)
        const int younger = total - 1 - g < D - 1 ? total - 1 - g : D - 1;
        if (D >= 4 && younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D_NI * 3) : "memory");
        else if (D >= 3 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D_NI * 2) : "memory");
        else if (D >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D_NI * 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cls_barrier();       // every wave's part of stage g is in LDS; every wave is done reading stage g - 1
        if (g + D < total) issue_stage(g + D);      // into the buffer stage g - 1 has just left
        const float *sp = lds + (g % NB) * D_BUF;
        {
            float d[4][4], t[4][4], V[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = sp[a_base + r * K_PWp + ((c & 1) ? 0 : K_EOFF) + (c >> 1)];
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
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float bf = sp[b_base + p * 4 * 16];
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf, acc[p], 0, 0, 0);
            }
        }
        if (chunk == nchunks - 1) {
#include "conv_cls_mc_sample.inc"
            // (diagnostic logits stores share the vmcnt counter with the DMAs and may complete out of order with them)
            if (a.logits) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (++chunk == nchunks) { chunk = 0; ++s; }
    }
    if (!pix_ok) return;
#include "conv_cls_mc_maps.inc"
}

bool cls_mc_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 3 && cin >= 4 && cout >= 1 && cout <= 16 && (W % 8) == 0 && (H % 2) == 0;
}
int cls_mc_k_chunk() {
    static const int kc = [] {
        const char *e = SIVO_DIAG_ENV("SIVO_CLS_KC");
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

// Default: the register-staged form (K-chunk SIVO_CLS_KC = 4 | 8).  SIVO_CLS_MC=dma: the LDS-DMA ring of SIVO_CLS_NB = 3 | 4 | 5
// stage buffers — bit-identical, measured 0.83 ms against 0.70 ms per frame whatever the depth (see the note at the kernel)
static int cls_mc_ring() {
    static const int nb = [] {
        const char *m = SIVO_DIAG_ENV("SIVO_CLS_MC");
        if (!m || std::string(m) != "dma") return 0;
        const char *e = SIVO_DIAG_ENV("SIVO_CLS_NB");
        const int v = e ? std::atoi(e) : 4;
        return v == 3 || v == 5 ? v : 4;
    }();
    return nb;
}

void launch_conv_cls_mc(const ClsMcArgs &a0, hipStream_t s) {
    ClsMcArgs a = a0;
    a.tiles_x = (a.W + K_TW - 1) / K_TW;
    a.tiles_y = (a.H + K_TH - 1) / K_TH;
    if (a.sum_chunk <= 0 || a.sum_chunk > (int64_t)a.H * a.W) a.sum_chunk = (int64_t)a.H * a.W;
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) / 8;
    const dim3 grid((unsigned)(8 * per)), block(K_NTHR);
    const int ring = (a.Cin % 4 == 0 && cls_mc_k_chunk() == 4 && (int64_t)a.Cin * a.H * a.W * 4 < (1ll << 31)) ? cls_mc_ring() : 0;
    if (ring == 3) hipLaunchKernelGGL((conv_wino_cls_mc_dma_kernel<3>), grid, block, 0, s, a);
    else if (ring == 4) hipLaunchKernelGGL((conv_wino_cls_mc_dma_kernel<4>), grid, block, 0, s, a);
    else if (ring == 5) hipLaunchKernelGGL((conv_wino_cls_mc_dma_kernel<5>), grid, block, 0, s, a);
    else if (cls_mc_k_chunk() == 8) hipLaunchKernelGGL((conv_wino_cls_mc_kernel<8>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_wino_cls_mc_kernel<4>), grid, block, 0, s, a);
}

}  // namespace sivo
