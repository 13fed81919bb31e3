// conv_wino4f.hip — FUSED Winograd F(4x4, 3x3) for the narrow layers (64 output channels per workgroup).
//
// The three-kernel F(4x4) path (conv_wino4.hip) moves 2.25x the activation through HBM twice; with 64..128 channels
// its position-GEMMs are bandwidth-bound and the fused F(2x2) kernel (conv_wino.hip) was the faster choice — at 2.25x
// instead of 4x fewer MFMA flops, and with a full per-lane tile transform in front of every 32 MFMAs.  This kernel
// keeps everything on chip AND gets the 4x, by giving every wave ONE ROW of the 6x6 transform:
//
//   workgroup = 12 waves = 2 m-tiles (16 horizontally adjacent 4x4-output tiles each: 4 rows x 64 px) x 6 transform
//   rows; 64 couts.  Wave (mt, i), lane (tile li, channel c0 + lk):
//     * reads the 3-4 raw patch rows that row i of B^T touches (b128 + b64 per row, one row live at a time), forms
//       t = (B^T d)[i][0..5] and V[i][0..5] = t B in registers (~40 VALU ops) — the A operands of positions (i, 0..5);
//     * B operand of position (i, j): ONE ds_read_b128 gives the four 16-cout blocks (slab is stored [position]
//       [channel][cout%16][cout/16]);  acc[6][4] = 96 accumulator VGPRs, 24 MFMAs per 4-channel step;
//     * after the channel loop A^T M A is split the same way: the column half (over j) is lane-local, the row half
//       (over i) is a 6-way reduction across the waves of an m-tile through LDS, done per 16-cout block in the space
//       the staging buffers no longer need; the final stage writes 256-byte row segments.
//   Per 24 MFMAs a wave issues ~7 patch reads + 6 slab reads: far less staging per MFMA than the F(2x2) kernel.
//   K-chunks of 4 channels, double-buffered LDS (2 x (patch 10.8 KB + slab 36 KB) = 93.5 KB, one workgroup per CU at
//   3 waves/SIMD); the pre-transformed weight slab is the LDS image and is copied by LDS-DMA; the patch goes through
//   registers one chunk ahead (the Upsample in front of the layer can be folded into that load: UNPOOL).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "lds_dma.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t w4f_dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
    uint32_t c0 = e >> 7, c1 = site, c2 = sample, c3 = 0u, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t sel = (e >> 5) & 3u;
    return sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
}

constexpr int F_MT = 2;                       // m-tiles per workgroup (stacked vertically)
constexpr int F_NW = F_MT * 6, F_NTHR = F_NW * 64;
constexpr int F_TH = 4 * F_MT, F_TW = 64;     // output pixels per workgroup
constexpr int F_PR = F_TH + 2;                // patch rows
constexpr int F_PC = 68;                      // patch row stride (66 used): image column x0 - 1 + q at patch column q
constexpr int F_CS = 688;                     // channel stride: F_PR * F_PC = 680, padded to 16 (mod 32)
constexpr int F_PATCH = 4 * F_CS;             // floats per patch buffer
constexpr int F_SLAB = 36 * 4 * 64;           // floats per weight slab (36 KiB)
constexpr int F_BUF = F_PATCH + F_SLAB;
constexpr int F_RS = 17;                      // epilogue exchange: [mt][i][tile*4 + j'][cout16 + pad]

// (Lavin's interpolation points 0, +-1, +-2, inf; the three-kernel path moved to 0, +-1, 1/2, -2 in round 6, wino4_transforms.hpp)
__device__ __forceinline__ void w4f_bt(const float d0, const float d1, const float d2, const float d3, const float d4,
                                       const float d5, float *t) {
    const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    t[0] = 4.f * d0 - 5.f * d2 + d4;
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d1 - 5.f * d3 + d5;
}
__device__ __forceinline__ void w4f_at(const float m0, const float m1, const float m2, const float m3, const float m4,
                                       const float m5, float *s) {
    const float p12 = m1 + m2, q12 = m1 - m2, p34 = m3 + m4, q34 = m3 - m4;
    s[0] = m0 + p12 + p34;
    s[1] = q12 + 2.f * q34;
    s[2] = p12 + 4.f * p34;
    s[3] = q12 + 8.f * q34 + m5;
}

// ABL (tools/conv_probe.py only; 0 in production): 1 no patch staging after the prologue, 2 no weight DMA after the prologue,
// 4 no per-chunk barrier / wait, 8 no input transform (raw patch values as operands), 32 no MFMAs, 64 no output stores,
// 128 no exchange writes, 256 no barriers in the output stage.  (There is no "no output stage" switch: without it the
// compiler removes most MFMAs as dead code and the variant measures nothing.)
// Workgroup barrier for the output stage: this wave's LDS traffic done + s_barrier.  __syncthreads() also waits for vmcnt(0),
// i.e. for the output stores issued just before it to be acknowledged (1-2 us each time, eight times per workgroup: measured
// 0.6 of the 1.65 ms of conv1_2_D); the exchange through LDS only needs lgkmcnt.
__device__ __forceinline__ void w4f_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int ABL> __device__ __forceinline__ void w4f_out_barrier() { if (!(ABL & 256)) w4f_lds_barrier(); }

template <bool UNPOOL, int ABL = 0>
__global__ __launch_bounds__(F_NTHR, 1) void conv_wino4f_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 * F_BUF floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int mt = wave / 6, wi = wave % 6;

    // XCD-aware order as in conv_wino.hip: workgroup L -> XCD L % 8, the Cout tiles of one pixel tile back to back
    const int ntiles = a.CoutPad / 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = slot % ntiles;
    int bid = (slot / ntiles) * 8 + xcd;
    if (bid >= a.tiles_x * a.tiles_y * a.N) return;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * F_TW, y0 = ty * F_TH;
    const int n0 = ntile * 64;

    const float *in_n = a.in + (int64_t)n * a.in_sample_stride;
    const int64_t plane = (int64_t)a.H * a.W;
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;
    const uint8_t *mk_n = UNPOOL ? a.unpool_mask + (int64_t)n * a.unpool_mask_stride : nullptr;

    // ---- staging plan: one interior float4 per thread (4 ch x 10 rows x 16 segments = 640) + 80 halo scalars
    constexpr int NV4 = 4 * F_PR * (F_TW / 4), NSC = 4 * F_PR * 2;
    static_assert(NV4 <= F_NTHR && NSC <= F_NTHR, "one staging item per thread");
    int v_goff = 0, v_dst = -1, v_c = 0;
    bool v_ok = false;
    if (tid < NV4) {
        const int seg = tid % (F_TW / 4), r = tid / (F_TW / 4);
        const int py = r % F_PR, c = r / F_PR;
        const int gy = y0 + py - 1, gx = x0 + seg * 4;
        v_ok = gy >= 0 && gy < a.H && gx + 3 < a.W;
        if (UNPOOL) v_goff = v_ok ? ((int)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) | ((gy & 1) << 30)) : 0;
        else v_goff = v_ok ? (int)(c * plane + (int64_t)gy * a.W + gx) : 0;
        v_dst = c * F_CS + py * F_PC + seg * 4 + 1;       // image column x0 + 4 seg sits at patch column 4 seg + 1
        v_c = c;
    }
    int s_goff = -1, s_dst = -1, s_c = 0;
    if (tid < NSC) {
        const int h = tid % 2, r = tid / 2;
        const int py = r % F_PR, c = r / F_PR;
        const int gy = y0 + py - 1, gx = h == 0 ? x0 - 1 : x0 + F_TW;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if (UNPOOL) s_goff = ok ? ((int)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) | ((gy & 1) << 30) | ((gx & 1) << 29)) : -1;
        else s_goff = ok ? (int)(c * plane + (int64_t)gy * a.W + gx) : -1;
        s_dst = c * F_CS + py * F_PC + (h == 0 ? 0 : F_TW + 1);
        s_c = c;
    }
    f32x4 pv4 = {0.f, 0.f, 0.f, 0.f};
    float psc = 0.f;
    const int nchunks = (a.Cin + 3) / 4;

    auto issue_patch = [&](int chunk) {
        const float *psrc = in_n + (int64_t)chunk * 4 * plane_in;
        const int cleft = a.Cin - chunk * 4;
        if (UNPOOL) {
            const uint8_t *msrc = mk_n + (int64_t)chunk * 4 * plane_in;
            {
                const bool ok = v_ok && v_c < cleft;
                const int off = ok ? (v_goff & 0x1fffffff) : 0, code0 = (v_goff >> 30) << 1;
                const float2 v = *reinterpret_cast<const float2 *>(psrc + off);
                const uchar2 m = *reinterpret_cast<const uchar2 *>(msrc + off);
                pv4 = ok ? (f32x4){m.x == code0 ? v.x : 0.f, m.x == code0 + 1 ? v.x : 0.f, m.y == code0 ? v.y : 0.f, m.y == code0 + 1 ? v.y : 0.f}
                         : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            {
                const bool ok = s_goff >= 0 && s_c < cleft;
                const int off = ok ? (s_goff & 0x1fffffff) : 0, code = ((s_goff >> 30) & 1) * 2 + ((s_goff >> 29) & 1);
                const float v = psrc[off];
                const int m = msrc[off];
                psc = (ok && m == code) ? v : 0.f;
            }
            return;
        }
        {
            const bool ok = v_ok && v_c < cleft;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(psrc + (ok ? v_goff : 0));
            pv4 = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        {
            const bool ok = s_goff >= 0 && s_c < cleft;
            const float v = psrc[ok ? s_goff : 0];
            psc = ok ? v : 0.f;
        }
    };
    auto commit_patch = [&](int buf) {
        float *sp = lds + buf * F_BUF;
        if (v_dst >= 0) {
            float *q = sp + v_dst;
            q[0] = pv4[0]; q[1] = pv4[1]; q[2] = pv4[2]; q[3] = pv4[3];
        }
        if (s_dst >= 0) sp[s_dst] = psc;
    };
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + ((int64_t)chunk * ntiles + ntile) * F_SLAB;
        float *dst = lds + buf * F_BUF + F_PATCH;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kib = i * F_NW + wave;          // 36 KiB: three 1 KiB copies per wave
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + kib * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + kib * 256), 16, 0, 0);
        }
    };

    f32x4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_patch(0);
    dma_weights(0, 0);
    commit_patch(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    // this lane's 6 x 6 window: patch rows 4 mt .. 4 mt + 5, columns 4 li .. 4 li + 5 of channel lk
    const int a_base = lk * F_CS + (4 * mt) * F_PC + 4 * li;
    const int b_base = F_PATCH + (wi * 6 * 4 + lk) * 64 + li * 4;        // position (wi, j): + j * 256

    // row wi of B^T (wave-uniform)
    float bt_row[6];
    {
        const float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
#pragma unroll
        for (int r = 0; r < 6; ++r) bt_row[r] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(BT[wi][r])));
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        if (more && !(ABL & 1)) issue_patch(chunk + 1);
        if (more && !(ABL & 2)) dma_weights(chunk + 1, cur ^ 1);
        const float *sp = lds + cur * F_BUF;
        // ---- row wi of B^T d, then the row transform: V[j] = (B^T d B)[wi][j]
        // (one raw row live at a time: t += B^T[wi][r] * d_r with a wave-uniform coefficient, zero rows skipped)
        float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const float c = bt_row[r];
            if (c != 0.f) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(sp + a_base + r * F_PC);
                const float2 e = *reinterpret_cast<const float2 *>(sp + a_base + r * F_PC + 4);
                t[0] = __builtin_fmaf(c, q[0], t[0]); t[1] = __builtin_fmaf(c, q[1], t[1]); t[2] = __builtin_fmaf(c, q[2], t[2]);
                t[3] = __builtin_fmaf(c, q[3], t[3]); t[4] = __builtin_fmaf(c, e.x, t[4]); t[5] = __builtin_fmaf(c, e.y, t[5]);
            }
        }
        float V[6];
        if (ABL & 8) { V[0] = t[0]; V[1] = t[1]; V[2] = t[2]; V[3] = t[3]; V[4] = t[4]; V[5] = t[5]; }
        else w4f_bt(t[0], t[1], t[2], t[3], t[4], t[5], V);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(sp + b_base + j * 256);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (ABL & 32) acc[j][nb][0] += V[j] * bq[nb];
                else acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[j], bq[nb], acc[j][nb], 0, 0, 0);
            }
            if (j == 3 && more && !(ABL & 1)) commit_patch(cur ^ 1);
        }
        if (!(ABL & 4)) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
        }
    }

#include "conv_wino4f_out.inc"
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel with its memory traffic actually in flight during the matrix-core phase.  In the kernel above hipcc puts
// `s_waitcnt vmcnt(0)` in front of the first LDS read behind the weight DMA (lds_dma.hpp), i.e. at the top of every K-chunk the
// (single) workgroup of the CU waits for everything it has just requested for the NEXT chunk: measured on conv1_2_D
// (tools/conv_probe.py w4f; 1.63 ms) 1.31 ms without the patch staging, 1.47 without the weight DMA, 1.10 without both.  Here
//   * the weight DMA is issued through inline assembly (invisible to the waitcnt pass);
//   * the patch loads are BUFFER loads (descriptor of the sample's planes, per-lane byte offset; an offset beyond the
//     descriptor returns 0, which is exactly the zero padding outside the image and beyond Cin — no select, no branch, the
//     same number of vector-memory instructions in every wave);
//   * iteration c: write the patch of chunk c + 1 (requested a whole iteration ago) to LDS, start the DMA of its weights,
//     request the patch of chunk c + 2, compute chunk c, then `s_waitcnt vmcnt(NL)` — vector-memory loads complete in issue
//     order, so "at most the NL patch loads issued behind the DMA are outstanding" means the DMA has landed while those
//     loads stay in flight — and a barrier without the vmcnt(0) drain of __syncthreads().
// PABL (tools/conv_probe.py only; 0 in production): 16 no prologue traffic, 1 no staging after the prologue, 2 no patch reads / transform (operands from
// the first chunk), 4 no B reads (fragments of the first chunk), 8 no end-of-chunk wait / barrier.  The output stage stays, so
// every accumulator is live.
template <bool UNPOOL, int PABL = 0>
__global__ __launch_bounds__(F_NTHR, 1) void conv_wino4f_p_kernel(ConvArgs a) {
    constexpr int ABL = 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 * F_BUF floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int mt = wave / 6, wi = wave % 6;

    const int ntiles = a.CoutPad / 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = slot % ntiles;
    // one cout tile: every XCD owns a contiguous band of pixel tiles (row-major), so the cache lines two neighbouring tiles
    // share (a pooled row of a tile is 132 bytes) are fetched into ONE L2; several cout tiles: the tiles of a pixel tile back
    // to back on one XCD (the patch is what they share)
    const int ptiles_all = a.tiles_x * a.tiles_y * a.N, band = ((ptiles_all + 7) >> 3);
    int bid = ntiles == 1 ? xcd * band + slot : (slot / ntiles) * 8 + xcd;
    if (bid >= ptiles_all || (ntiles == 1 && slot >= band)) return;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * F_TW, y0 = ty * F_TH;
    const int n0 = ntile * 64;

    const int64_t plane = (int64_t)a.H * a.W;
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;
    // descriptors of this sample's input planes (and window codes): wave-uniform (kernel arguments and blockIdx only)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (int64_t)n * a.in_sample_stride), 0, (int)(a.Cin * plane_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mk_rsrc =
        UNPOOL ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.unpool_mask + (int64_t)n * a.unpool_mask_stride), 0, (int)(a.Cin * plane_in), 0x00020000)
               : in_rsrc;
    constexpr uint32_t INV = 0xfffffff0u;           // beyond any descriptor: the load returns 0
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // ---- staging plan: one interior float4 per thread (4 ch x 10 rows x 16 segments = 640) + 80 halo scalars.
    // v_idx / s_idx: element index inside the chunk's 4 planes (UNPOOL: of the pooled plane; bits 30 / 29 = window row / column)
    constexpr int NV4 = 4 * F_PR * (F_TW / 4), NSC = 4 * F_PR * 2;
    uint32_t v_idx = INV, s_idx = INV;
    int v_dst = -1, s_dst = -1, v_code0 = 0, s_code = 0;
    if (tid < NV4) {
        const int seg = tid % (F_TW / 4), r = tid / (F_TW / 4);
        const int py = r % F_PR, c = r / F_PR;
        const int gy = y0 + py - 1, gx = x0 + seg * 4;
        const bool ok = gy >= 0 && gy < a.H && gx + 3 < a.W;
        if (UNPOOL) { if (ok) v_idx = (uint32_t)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)); v_code0 = (gy & 1) << 1; }
        else if (ok) v_idx = (uint32_t)(c * plane + (int64_t)gy * a.W + gx);
        v_dst = c * F_CS + py * F_PC + seg * 4 + 1;
    }
    if (tid < NSC) {
        const int h = tid % 2, r = tid / 2;
        const int py = r % F_PR, c = r / F_PR;
        const int gy = y0 + py - 1, gx = h == 0 ? x0 - 1 : x0 + F_TW;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if (UNPOOL) { if (ok) s_idx = (uint32_t)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)); s_code = (gy & 1) * 2 + (gx & 1); }
        else if (ok) s_idx = (uint32_t)(c * plane + (int64_t)gy * a.W + gx);
        s_dst = c * F_CS + py * F_PC + (h == 0 ? 0 : F_TW + 1);
    }
    constexpr int NL = UNPOOL ? 4 : 2;              // vector-memory loads per issue_patch, in every wave
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct PatchRegs { u32x4 v4; unsigned sc, vm, sm; };     // non-UNPOOL: v4 + sc; UNPOOL: v4.xy = two pooled values, vm / sm = codes
    const int nchunks = (a.Cin + 3) / 4;

    auto issue_patch = [&](int chunk, PatchRegs &R) {
        const uint32_t cb = (uint32_t)(chunk * 4 * plane_in);
        if (UNPOOL) {
            const uint32_t vi = v_idx == INV ? INV : v_idx + cb, si = s_idx == INV ? INV : s_idx + cb;
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)(vi == INV ? INV : vi * 4), 0, 0);
            R.v4[0] = v[0]; R.v4[1] = v[1];
            R.vm = __builtin_amdgcn_raw_buffer_load_b16(mk_rsrc, (int)vi, 0, 0);
            R.sc = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(si == INV ? INV : si * 4), 0, 0);
            R.sm = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc, (int)si, 0, 0);
        } else {
            R.v4 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(v_idx == INV ? INV : (v_idx + cb) * 4), 0, 0);
            R.sc = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(s_idx == INV ? INV : (s_idx + cb) * 4), 0, 0);
        }
    };
    auto commit_patch = [&](int buf, const PatchRegs &R) {
        float *sp = lds + buf * F_BUF;
        if (v_dst >= 0) {
            float *q = sp + v_dst;
            if (UNPOOL) {
                const float vx = __uint_as_float(R.v4[0]), vy = __uint_as_float(R.v4[1]);
                const int mx = (int)(R.vm & 0xffu), my = (int)((R.vm >> 8) & 0xffu);
                q[0] = mx == v_code0 ? vx : 0.f; q[1] = mx == v_code0 + 1 ? vx : 0.f;
                q[2] = my == v_code0 ? vy : 0.f; q[3] = my == v_code0 + 1 ? vy : 0.f;
            } else {
                q[0] = __uint_as_float(R.v4[0]); q[1] = __uint_as_float(R.v4[1]); q[2] = __uint_as_float(R.v4[2]); q[3] = __uint_as_float(R.v4[3]);
            }
        }
        if (s_dst >= 0) sp[s_dst] = (!UNPOOL || (int)(R.sm & 0xffu) == s_code) ? __uint_as_float(R.sc) : 0.f;
    };
    const uint32_t slab_lds = lds_addr_uniform(lds + F_PATCH) + (uint32_t)wave_u * 1024u;      // this wave's first KiB of slab buffer 0
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + ((int64_t)chunk * ntiles + ntile) * F_SLAB;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kib = i * F_NW + wave;          // 36 KiB: three 1 KiB copies per wave
            lds_dma16(wsrc + kib * 256 + lane * 4, slab_lds + (uint32_t)(buf * F_BUF * 4 + i * F_NW * 1024));
        }
    };

    f32x4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int a_base = lk * F_CS + (4 * mt) * F_PC + 4 * li;
    const int b_base = F_PATCH + (wi * 6 * 4 + lk) * 64 + li * 4;        // position (wi, j): + j * 256
    // row wi of B^T as (patch row, coefficient) pairs in increasing row order, padded with a zero coefficient (wave-uniform)
    int row_off[4];
    float row_cf[4];
    {
        const int RW[6][4] = {{0, 2, 4, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, 5}};
        const float CF[6][4] = {{4, -5, 1, 0}, {-4, -4, 1, 1}, {4, -4, -1, 1}, {-2, -1, 2, 1}, {2, -1, -2, 1}, {4, -5, 1, 0}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            row_off[k] = __builtin_amdgcn_readfirstlane(RW[wi][k]) * F_PC;
            row_cf[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(CF[wi][k])));
        }
    }

    PatchRegs R;
    R.vm = 0; R.sm = 0; R.sc = 0; R.v4 = (u32x4){0u, 0u, 0u, 0u};
    if (!(PABL & 16)) {         // (16: no prologue traffic either — probe: what the first chunk's latency costs per workgroup)
        issue_patch(0, R);
        dma_weights(0, 0);
        commit_patch(0, R);
        if (nchunks > 1) issue_patch(1, R);
        if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    w4f_lds_barrier();

    float Vk[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 bk[6];
    for (int j = 0; j < 6; ++j) bk[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks, more2 = chunk + 2 < nchunks;
        if (more && !(PABL & 1)) {
            commit_patch(cur ^ 1, R);                 // chunk + 1: requested a whole iteration ago
            // every wave (also one without a staging item) is done with R here: the compiler's own wait for these loads
            // comes now, before the DMA it cannot see is in flight, and not at a later reuse of the registers
            asm volatile("" ::"v"(R.v4), "v"(R.sc), "v"(R.vm), "v"(R.sm));
            dma_weights(chunk + 1, cur ^ 1);
            asm volatile("" ::: "memory");           // the DMA stays ahead of the loads in program order
            if (more2) issue_patch(chunk + 2, R);
        }
        const float *sp = lds + ((PABL & 1) ? 0 : cur) * F_BUF;
        // row wi of B^T d: the (at most four) patch rows with a non-zero coefficient, read in ONE batch (a branch per row
        // made each row its own LDS round trip: read, wait, multiply — 3-4 latencies in front of the first MFMA of a chunk,
        // in all 12 waves at the same time); same products in the same order as the row-by-row form, a padding term is +0 * d
        f32x4 q4[4];
        float2 e2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((PABL & 2) && chunk > 0) { q4[k] = (f32x4){1.f, 2.f, 3.f, 4.f}; e2[k] = make_float2(5.f, 6.f); continue; }
            q4[k] = *reinterpret_cast<const f32x4 *>(sp + a_base + row_off[k]);
            e2[k] = *reinterpret_cast<const float2 *>(sp + a_base + row_off[k] + 4);
        }
        float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c = row_cf[k];
            t[0] = __builtin_fmaf(c, q4[k][0], t[0]); t[1] = __builtin_fmaf(c, q4[k][1], t[1]); t[2] = __builtin_fmaf(c, q4[k][2], t[2]);
            t[3] = __builtin_fmaf(c, q4[k][3], t[3]); t[4] = __builtin_fmaf(c, e2[k].x, t[4]); t[5] = __builtin_fmaf(c, e2[k].y, t[5]);
        }
        float V[6];
        w4f_bt(t[0], t[1], t[2], t[3], t[4], t[5], V);
        if (PABL & 2) {
            if (chunk == 0) { for (int j = 0; j < 6; ++j) Vk[j] = V[j]; }
            for (int j = 0; j < 6; ++j) V[j] = Vk[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 bq;
            if (PABL & 4) {
                if (chunk == 0) bk[j] = *reinterpret_cast<const f32x4 *>(sp + b_base + j * 256);
                bq = bk[j];
            } else {
                bq = *reinterpret_cast<const f32x4 *>(sp + b_base + j * 256);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[j], bq[nb], acc[j][nb], 0, 0, 0);
        }
        if (!(PABL & 8)) {
            if (more2 && !(PABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w4f_lds_barrier();
        }
    }
    if (PABL & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); w4f_lds_barrier(); }

#include "conv_wino4f_out.inc"
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the in-flight kernel: one workgroup per CU walks a list of pixel tiles, and (tile, K-chunk) form ONE
// stream of stages.  With one workgroup per CU (93.5 KB of staging LDS) nothing overlaps a workgroup's prologue and
// epilogue: measured on conv1_2_D (tools/conv_probe.py w4fp; 33 workgroups per CU) the kernel reduced to its MFMAs and its
// output stage takes 0.88 ms against 0.55 ms of matrix-core time — 0.115 ms of that is the first chunk's memory latency
// (3.5 us per workgroup: 0.77 ms without any prologue traffic), the rest workgroup turnover and the output stage.  Here
// the stages of the NEXT tile are requested during the last chunks of the current one (stage g + 2 is written to LDS and
// its weight DMA started, stage g + 3 requested, right behind the barrier that ends stage g), the output stage has its own
// 52 KB exchange region (148 KB of LDS in all) and runs while that traffic is in flight, and the first chunk of the next
// tile starts without a memory wait.  Output stores share the vmcnt counter with the DMA: "at most NL outstanding" still
// implies the DMA has landed (loads, DMA included, complete in issue order among themselves; a pending store only makes
// the wait longer).  Requires Cout == 64 (one cout tile: every tile of a workgroup has the same epilogue affine).
template <bool UNPOOL>
__global__ __launch_bounds__(F_NTHR, 1) void conv_wino4f_pp_kernel(ConvArgs a) {
    constexpr int ABL = 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 * F_BUF floats of staging + the exchange region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int mt = wave / 6, wi = wave % 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // every XCD owns a contiguous band of pixel tiles (row-major); its gridDim.x / 8 workgroups walk the band side by side, so
    // neighbouring tiles are in flight on one XCD at the same time and share their cache lines in its L2
    const int ptiles = a.tiles_x * a.tiles_y * a.N, band = (ptiles + 7) >> 3;
    const int G = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, s0 = (int)blockIdx.x >> 3;      // G: workgroups per XCD
    const int band_n = ptiles - xcd * band < band ? ptiles - xcd * band : band;                    // tiles in this XCD's band
    if (s0 >= band_n) return;
    const int L0 = xcd * band + s0;
    const int my_tiles = (band_n - s0 + G - 1) / G;
    const int nchunks = (a.Cin + 3) / 4;
    const int total = my_tiles * nchunks;
    const int n0 = 0;

    const int64_t plane = (int64_t)a.H * a.W;
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;
    constexpr uint32_t INV = 0xfffffff0u;           // beyond any descriptor: the load returns 0
    constexpr int NV4 = 4 * F_PR * (F_TW / 4), NSC = 4 * F_PR * 2;
    constexpr int NL = UNPOOL ? 4 : 2;              // vector-memory loads per issue_patch, in every wave
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct PatchRegs { u32x4 v4; unsigned sc, vm, sm; };

    // ---- what does not depend on the tile: LDS destinations and (UNPOOL) the window codes of this thread's staging items
    // (y0 is a multiple of 8 and x0 of 64, so the parities of the item's row and column are fixed)
    int v_dst = -1, s_dst = -1, v_code0 = 0, s_code = 0;
    if (tid < NV4) {
        const int v_seg = tid % (F_TW / 4), r = tid / (F_TW / 4);
        const int v_py = r % F_PR, v_c = r / F_PR;
        v_code0 = ((v_py - 1) & 1) << 1;
        v_dst = v_c * F_CS + v_py * F_PC + v_seg * 4 + 1;
    }
    if (tid < NSC) {
        const int s_h = tid % 2, r = tid / 2;
        const int s_py = r % F_PR, s_c = r / F_PR;
        s_code = ((s_py - 1) & 1) * 2 + (s_h == 0 ? 1 : 0);       // x0 - 1 is odd, x0 + 64 even
        s_dst = s_c * F_CS + s_py * F_PC + (s_h == 0 ? 0 : F_TW + 1);
    }

    // ---- the issue side of the stream: tile (x0, y0, sample) of the stage whose patch is requested next
    __amdgpu_buffer_rsrc_t in_rsrc, mk_rsrc;
    uint32_t v_idx = INV, s_idx = INV;
    auto tile_of = [&](int k, int &x0, int &y0, int &n) {
        int bid = L0 + k * G;
        const int tx = bid % a.tiles_x; bid /= a.tiles_x;
        const int ty = bid % a.tiles_y; bid /= a.tiles_y;
        n = bid; x0 = tx * F_TW; y0 = ty * F_TH;
    };
    auto plan_tile = [&](int k) {
        int x0, y0, n;
        tile_of(k, x0, y0, n);
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (int64_t)n * a.in_sample_stride), 0, (int)(a.Cin * plane_in * 4), 0x00020000);
        mk_rsrc = UNPOOL ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.unpool_mask + (int64_t)n * a.unpool_mask_stride), 0, (int)(a.Cin * plane_in), 0x00020000)
                         : in_rsrc;
        v_idx = INV; s_idx = INV;
        if (tid < NV4) {
            const int v_seg = tid % (F_TW / 4), r = tid / (F_TW / 4);
            const int v_py = r % F_PR, v_c = r / F_PR;
            const int gy = y0 + v_py - 1, gx = x0 + v_seg * 4;
            if (gy >= 0 && gy < a.H && gx + 3 < a.W)
                v_idx = UNPOOL ? (uint32_t)(v_c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(v_c * plane + (int64_t)gy * a.W + gx);
        }
        if (tid < NSC) {
            const int s_h = tid % 2, r = tid / 2;
            const int s_py = r % F_PR, s_c = r / F_PR;
            const int gy = y0 + s_py - 1, gx = s_h == 0 ? x0 - 1 : x0 + F_TW;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                s_idx = UNPOOL ? (uint32_t)(s_c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(s_c * plane + (int64_t)gy * a.W + gx);
        }
    };
    auto issue_patch = [&](int chunk, PatchRegs &R) {
        const uint32_t cb = (uint32_t)(chunk * 4 * plane_in);
        if (UNPOOL) {
            const uint32_t vi = v_idx == INV ? INV : v_idx + cb, si = s_idx == INV ? INV : s_idx + cb;
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)(vi == INV ? INV : vi * 4), 0, 0);
            R.v4[0] = v[0]; R.v4[1] = v[1];
            R.vm = __builtin_amdgcn_raw_buffer_load_b16(mk_rsrc, (int)vi, 0, 0);
            R.sc = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(si == INV ? INV : si * 4), 0, 0);
            R.sm = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc, (int)si, 0, 0);
        } else {
            R.v4 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(v_idx == INV ? INV : (v_idx + cb) * 4), 0, 0);
            R.sc = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(s_idx == INV ? INV : (s_idx + cb) * 4), 0, 0);
        }
    };
    auto commit_patch = [&](int buf, const PatchRegs &R) {
        float *sp = lds + buf * F_BUF;
        if (v_dst >= 0) {
            float *q = sp + v_dst;
            if (UNPOOL) {
                const float vx = __uint_as_float(R.v4[0]), vy = __uint_as_float(R.v4[1]);
                const int mx = (int)(R.vm & 0xffu), my = (int)((R.vm >> 8) & 0xffu);
                q[0] = mx == v_code0 ? vx : 0.f; q[1] = mx == v_code0 + 1 ? vx : 0.f;
                q[2] = my == v_code0 ? vy : 0.f; q[3] = my == v_code0 + 1 ? vy : 0.f;
            } else {
                q[0] = __uint_as_float(R.v4[0]); q[1] = __uint_as_float(R.v4[1]); q[2] = __uint_as_float(R.v4[2]); q[3] = __uint_as_float(R.v4[3]);
            }
        }
        if (s_dst >= 0) sp[s_dst] = (!UNPOOL || (int)(R.sm & 0xffu) == s_code) ? __uint_as_float(R.sc) : 0.f;
        // every wave (also one without a staging item) is done with R here: the compiler's own wait for these loads comes
        // now, before a DMA it cannot see is in flight, and not at a later reuse of the registers
        asm volatile("" ::"v"(R.v4), "v"(R.sc), "v"(R.vm), "v"(R.sm));
    };
    const uint32_t slab_lds = lds_addr_uniform(lds + F_PATCH) + (uint32_t)wave_u * 1024u;      // this wave's first KiB of slab buffer 0
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + (int64_t)chunk * F_SLAB;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kib = i * F_NW + wave;          // 36 KiB: three 1 KiB copies per wave
            lds_dma16(wsrc + kib * 256 + lane * 4, slab_lds + (uint32_t)(buf * F_BUF * 4 + i * F_NW * 1024));
        }
    };

    f32x4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int a_base = lk * F_CS + (4 * mt) * F_PC + 4 * li;
    const int b_base = F_PATCH + (wi * 6 * 4 + lk) * 64 + li * 4;        // position (wi, j): + j * 256
    int row_off[4];
    float row_cf[4];
    {
        const int RW[6][4] = {{0, 2, 4, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, 5}};
        const float CF[6][4] = {{4, -5, 1, 0}, {-4, -4, 1, 1}, {4, -4, -1, 1}, {-2, -1, 2, 1}, {2, -1, -2, 1}, {4, -5, 1, 0}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            row_off[k] = __builtin_amdgcn_readfirstlane(RW[wi][k]) * F_PC;
            row_cf[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(CF[wi][k])));
        }
    }
    // the epilogue affine of the 64 couts goes to LDS once (behind the exchange region): the output stages read it from there
    float *ep_lds = lds + 2 * F_BUF + F_MT * 6 * 64 * F_RS;
    if (tid < 64) { ep_lds[tid] = tid < a.Cout ? a.ep_scale[tid] : 0.f; ep_lds[64 + tid] = tid < a.Cout ? a.ep_shift[tid] : 0.f; }

    // ---- prologue: stages 0 and 1 in LDS, stage 2 requested.  ik / ic: tile and chunk of the stage requested next.
    PatchRegs R;
    R.vm = 0; R.sm = 0; R.sc = 0; R.v4 = (u32x4){0u, 0u, 0u, 0u};
    int ik = 0, ic = 0;               // issue side
    int dc = 0;                       // chunk of the stage whose DMA / LDS write comes next (stage index = g + 2 in the loop)
    auto step_issue = [&]() { if (++ic == nchunks) { ic = 0; ++ik; if (ik < my_tiles) plan_tile(ik); } };
    plan_tile(0);
    issue_patch(ic, R); step_issue();
    dma_weights(0, 0);
    commit_patch(0, R);
    if (total > 1) {
        issue_patch(ic, R); step_issue();
        dma_weights(1 % nchunks, 1);
        commit_patch(1, R);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total > 2) { issue_patch(ic, R); step_issue(); }
    dc = 2 % nchunks;
    w4f_lds_barrier();

    int ck = 0, cc = 0;               // compute side: tile and chunk of stage g
    for (int g = 0; g < total; ++g) {
        const int cur = g & 1;
        const float *sp = lds + cur * F_BUF;
        // row wi of B^T d, one patch row at a time (the batched form of the non-persistent kernel costs 17 VGPRs more, which
        // this kernel does not have: its output stage lives inside the stage loop)
        float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c = row_cf[k];
            if (c != 0.f) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(sp + a_base + row_off[k]);
                const float2 e = *reinterpret_cast<const float2 *>(sp + a_base + row_off[k] + 4);
                t[0] = __builtin_fmaf(c, q[0], t[0]); t[1] = __builtin_fmaf(c, q[1], t[1]); t[2] = __builtin_fmaf(c, q[2], t[2]);
                t[3] = __builtin_fmaf(c, q[3], t[3]); t[4] = __builtin_fmaf(c, e.x, t[4]); t[5] = __builtin_fmaf(c, e.y, t[5]);
            }
        }
        float V[6];
        w4f_bt(t[0], t[1], t[2], t[3], t[4], t[5], V);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(sp + b_base + j * 256);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[j], bq[nb], acc[j][nb], 0, 0, 0);
        }
        // stage g + 1 must be complete in LDS: its DMA was issued ahead of the loads of stage g + 2
        if (g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w4f_lds_barrier();
        // ---- behind the barrier: buffer `cur` is free.  Stage g + 2 goes in (patch from R, weights by DMA), stage g + 3 is requested.
        if (g + 2 < total) {
            commit_patch(cur, R);
            dma_weights(dc, cur);
            if (++dc == nchunks) dc = 0;
            asm volatile("" ::: "memory");           // the DMA stays ahead of the loads in program order
            if (g + 3 < total) { issue_patch(ic, R); step_issue(); }
        }
        // ---- end of a tile: its output stage, in its own LDS region, while the next tile's first stages are in flight
        if (++cc == nchunks) {
            cc = 0;
            int x0, y0, n;
            tile_of(ck, x0, y0, n);
            ++ck;
            // the lane-dependent addresses of the output stage are recomputed here from an opaque copy of the thread index:
            // hoisted out of the stage loop they were loop-invariant values with nowhere to live but scratch, and every
            // reload came with a vmcnt(0) that drained the traffic just started above
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int tid = tid_o, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4, mt = wave / 6, wi = wave % 6;
            (void)li; (void)lk; (void)mt; (void)wi;
#define W4F_RB (lds + 2 * F_BUF)
#define W4F_EPILOGUE_IN_LDS ep_lds
#define W4F_FLOAT4_ONLY
#include "conv_wino4f_out.inc"
#undef W4F_FLOAT4_ONLY
#undef W4F_EPILOGUE_IN_LDS
#undef W4F_RB
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}
constexpr int F_EXCH = F_MT * 6 * 64 * F_RS + 128;    // floats of the output stage's exchange region + the epilogue affine

bool wino4f_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 3 && cin >= 4 && cout % 64 == 0 && (W % 8) == 0 && (H % 2) == 0;
}
int wino4f_slab_floats() { return F_SLAB; }

// Caffe (Cout,Cin,3,3) -> [ceil(Cin/4)][Cout/64][36][4 ch][16 = cout%16][4 = (cout%64)/16], U = G g G^T in double
void wino4f_pack_weights(const float *W, int cin, int cout, std::vector<float> &out, int *cout_pad) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int ntiles = cout / 64, nchunks = (cin + 3) / 4;
    *cout_pad = cout;
    out.assign((size_t)nchunks * ntiles * F_SLAB, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = W + ((size_t)co * cin + ci) * 9;
            double t[6][3];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            const size_t base = ((size_t)(ci / 4) * ntiles + co / 64) * F_SLAB;
            const int cl = co % 64;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    out[base + (size_t)(((i * 6 + j) * 4 + ci % 4) * 16 + cl % 16) * 4 + cl / 16] =
                        (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

void launch_conv_wino4f(const ConvArgs &a0, hipStream_t s) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
    }
    ConvArgs a = a0;
    a.tiles_x = (a.W + F_TW - 1) / F_TW;
    a.tiles_y = (a.H + F_TH - 1) / F_TH;
    const int ptiles = a.tiles_x * a.tiles_y * a.N;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8 * (a.CoutPad / 64)));
    const int abl = (a.variant >> 16) & 1023;
    // SIVO_W4F_PIPE=0 (or variant bit 8192, or a sample of 2 GiB and more): the one-chunk-ahead kernel
    static const bool pipe_env = !(SIVO_DIAG_ENV("SIVO_W4F_PIPE") && std::atoi(SIVO_DIAG_ENV("SIVO_W4F_PIPE")) == 0);
    const bool pipe = pipe_env && !(a.variant & 8192) && (int64_t)a.Cin * a.H * a.W * 4 < (1ll << 31);
    if (abl && !pipe) {      // probe only
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
            hipLaunchKernelGGL(kern, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
        };
        switch (abl) {
            case 1: return go(conv_wino4f_kernel<false, 1>);
            case 2: return go(conv_wino4f_kernel<false, 2>);
            case 3: return go(conv_wino4f_kernel<false, 3>);
            case 7: return go(conv_wino4f_kernel<false, 7>);
            case 8: return go(conv_wino4f_kernel<false, 8>);
            case 32: return go(conv_wino4f_kernel<false, 32>);
            case 64: return go(conv_wino4f_kernel<false, 64>);
            case 128: return go(conv_wino4f_kernel<false, 128>);
            case 192: return go(conv_wino4f_kernel<false, 192>);
            case 256: return go(conv_wino4f_kernel<false, 256>);
            case 448: return go(conv_wino4f_kernel<false, 448>);
            default: break;
        }
    }
    if (pipe && abl && !a.unpool_mask) {      // probe only
        auto gop = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F_BUF * 4);
            hipLaunchKernelGGL(kern, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
        };
        switch (abl) {
            case 1: return gop(conv_wino4f_p_kernel<false, 1>);
            case 2: return gop(conv_wino4f_p_kernel<false, 2>);
            case 4: return gop(conv_wino4f_p_kernel<false, 4>);
            case 6: return gop(conv_wino4f_p_kernel<false, 6>);
            case 7: return gop(conv_wino4f_p_kernel<false, 7>);
            case 8: return gop(conv_wino4f_p_kernel<false, 8>);
            case 9: return gop(conv_wino4f_p_kernel<false, 9>);
            case 15: return gop(conv_wino4f_p_kernel<false, 15>);
            case 31: return gop(conv_wino4f_p_kernel<false, 31>);
            default: break;
        }
    }
    // persistent form (one cout tile, default; SIVO_W4F_PERSIST=0 or variant bit 16384: one workgroup per pixel tile)
    // (read at every launch: tests switch it; 0 = never, 2 = always, default = where the deal is even)
    const char *pe = SIVO_DIAG_ENV("SIVO_W4F_PERSIST");
    const bool persist_env = !(pe && std::atoi(pe) == 0), persist_force = pe && std::atoi(pe) == 2;
    // Measured: conv1_2_D (8448 tiles = 33 per CU) 1.354 -> 1.307 ms; conv2_1_D (2112 tiles = 8.25 per CU: a quarter of the CUs
    // gets a ninth tile) 0.675 -> 0.683 ms.  Hence only where the static deal is even enough: at least 16 tiles per workgroup.
    static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    if (pipe && persist_env && !abl && !(a.variant & 16384) && a.CoutPad == 64 && (ptiles >= 16 * n_cu || persist_force || (a.variant & 32768))) {
        static int attr_pp[64] = {0};
        const size_t lds_pp = (size_t)(2 * F_BUF + F_EXCH) * 4;
        if (FirstUse once(attr_pp); once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_pp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4f_pp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp);
        }
        a.variant |= 4096;
        const int gmax = (n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8;
        const dim3 gp((unsigned)(ptiles < gmax ? ((ptiles + 7) / 8) * 8 : gmax));
        if (a.unpool_mask) hipLaunchKernelGGL(conv_wino4f_pp_kernel<true>, gp, dim3(F_NTHR), lds_pp, s, a);
        else hipLaunchKernelGGL(conv_wino4f_pp_kernel<false>, gp, dim3(F_NTHR), lds_pp, s, a);
        return;
    }
    if (pipe) {
        if (a.unpool_mask) hipLaunchKernelGGL(conv_wino4f_p_kernel<true>, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
        else hipLaunchKernelGGL(conv_wino4f_p_kernel<false>, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
        return;
    }
    if (a.unpool_mask) hipLaunchKernelGGL(conv_wino4f_kernel<true>, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
    else hipLaunchKernelGGL(conv_wino4f_kernel<false>, grid, dim3(F_NTHR), 2 * F_BUF * 4, s, a);
}

}  // namespace sivo
