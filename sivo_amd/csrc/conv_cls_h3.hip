// conv_cls_h3.hip — the classifier convolution (conv1_1_D: 3x3, 64 -> 15) fused with the Monte-Carlo post-processing, on the
// fp16 matrix cores (f16x3), fed by LDS-DMA from the PACKED activation its producer wrote (conv3_h3.hip OUT_PK).
//
// What it replaces.  conv_cls_mc.hip runs this layer as Winograd F(2x2) on the fp32 matrix pipe: 0.70 ms of an 8 ms frame at
// T = 12, issue-bound (per 16 MFMAs a wave also issues ~70 transform / staging instructions), for what is 1.1 GB of input —
// 0.18 ms of HBM time.  Here the input arrives as fp16 hi / lo pieces (8 channels of one pixel = the B fragment of one lane of
// v_mfma_f32_16x16x32_f16), so staging is address arithmetic only, and a direct fp32-equivalent product costs 3 MFMA flops
// on the fp16 pipe where the fp32 pipe pays 16: 80 GFLOP x 3 = 0.1 ms of matrix-core time at the peak.  The kernel is then
// what the layer is: a stream of 1.1 GB through LDS.
//
// Structure (reference call site: src/bayesian_segnet/bayesian_segnet.cpp:299-318; arithmetic after the convolution as
// conv_cls_mc.hip, whose Softmax / f64-sum / maps code is shared by textual include, so the maps equal
// mc_reduce_finalize_kernel on the same logits bit for bit):
//   workgroup = 8 x 32 output pixels, ALL T samples, 4 waves (one per SIMD); wave w owns rows 2 w, 2 w + 1 = four blocks of
//   16 pixels; implicit GEMM per tap with M = the 16 couts (15 classes + a zero row), N = 16 pixels, K = 32 channels:
//   D[cout][pixel] += W_tap[cout][ci] * X[ci][pixel + tap], three v_mfma_f32_16x16x32_f16 per fp32-equivalent product
//   (lo hi, hi lo, hi hi; fp32 accumulation).
//   * weights: the WHOLE filter bank as fp16 hi / lo planes (Cin / 32 x 18 KiB, split on the host) is copied into LDS once per
//     workgroup and stays;
//   * a stage = 32 channels of one sample: the 10 x 34 halo patch as [plane][channel octet][patch row][patch column] pieces
//     (octet stride padded to 352 pieces: every ds_read_b128 of 16 pixels x 4 octets covers the 64 banks once), 44 LDS-DMA
//     instructions per workgroup, issued one stage ahead into the other buffer; (sample, stage) form one stream: the first
//     stage of sample s + 1 is in flight under the last stage of sample s;
//   * after a sample's last stage a lane holds classes 4 g .. 4 g + 3 of pixel p of each of its wave's four blocks (C/D:
//     column = lane & 15 = pixel, row = 4 (lane >> 4) + register); a 4 x 4 transpose over the four 16-lane rows
//     (v_permlane16_swap / v_permlane32_swap, no LDS) gives lane (p, g) the 16 logits of pixel p of block g; it applies the
//     epilogue affine and runs the shared Softmax / f64-sum code.
// LDS: 2 x 45,056 (patch) + Cin / 32 x 18,432 (weights) = 126,976 bytes at 64 channels (the kernel asks for the CU's whole LDS,
// as the other DMA-staged persistent kernels do).  Cin % 32 == 0, Cin <= 96, <= 16 classes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "common.hpp"
#include "lds_dma.hpp"
#include "segnet_kernels.hpp"
#include "softmax.hpp"

namespace sivo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C_TH = 8, C_TW = 32;                         // output pixels of a workgroup
constexpr int C_PR = C_TH + 2, C_PW = C_TW + 2;            // patch rows / columns
constexpr int C_NPX = C_PR * C_PW;                         // 340 patch pixels
constexpr int C_OST = 352;                                 // pieces per (plane, octet) of a stage image: 340 + 12 pad (a multiple of 16: 256-byte strides)
constexpr int C_STAGE = 8 * C_OST * 16;                    // 45,056 bytes: [plane 2][octet 4][352] pieces
constexpr int C_NDMA = 8 * C_OST / 64;                     // 44 DMA instructions per stage, 11 per wave
constexpr int C_WKS = 9 * 2 * 4 * 16 * 16;                 // weight bytes per 32-channel k-step: [tap][plane][octet][cout 16][8 halfs] = 18,432
static_assert(C_NDMA == 44 && C_NDMA % 4 == 0, "DMA pieces per wave");
static_assert(C_STAGE + 2 * C_WKS <= 80 * 1024, "two workgroups per CU (64 input channels)");

// One wave per SIMD (NW = 4): wave w owns rows 2 w, 2 w + 1 of the tile (four blocks of 16 pixels).  (A form with two waves per
// SIMD, one row each, was measured slower — 0.54 against 0.46 ms, round 4 — and removed: the weight fragments are read once per
// two blocks instead of once per four, and half the lanes idle through every sample's Softmax.)
// Pipeline (round 4, final form): ONE patch buffer and TWO workgroups per CU (45,056 + 36,864 bytes = exactly half of the CU's LDS
// each).  A sample's last stage ends in ~4 us of Softmax + statistics per workgroup, the stages before it in ~0.2 us of MFMAs,
// and the input is a 1.1 GB stream: with two buffers in one 160 KB workgroup per CU the kernel took the SUM of its stream (0.22 ms
// alone, 4.9 TB/s) and its Softmax tail (0.37 ms alone) — 0.48–0.52 ms in three different issue orders, one wave per SIMD having
// nothing to put beside either.  Two independent workgroups per CU do what the issue orders could not: one streams while the
// other is in its Softmax, and the SIMDs have two waves to choose from: **0.36 ms**, outputs bit-identical (tools/cls_probe.py).
// Within a workgroup the next stage is requested as soon as every wave has read the current one (one barrier), i.e. under the
// Softmax when there is one.  The two workgroups of a CU both issue LDS-DMA next to each other's ds_read traffic — the
// constellation of DESIGN 3.3's co-residency finding; this pair was checked bit-identical against the one-workgroup form and
// by the full-size network tests, and bench.py compares every run's pipelined frames with the serial loop.
// ABL (diagnostic builds only, -DSIVO_DIAG; results are wrong by construction): 1 no MFMAs, 2 no Softmax / sum at the end of a sample,
// 4 no patch DMA after the first stage, 8 no fragment reads after the first tap.
template <int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void conv_cls_h3_kernel(ClsMcArgs a) {
    constexpr int W0 = C_STAGE;                 // the filter bank behind the patch buffer
    constexpr int NB = 16 / NW;                 // 16-pixel blocks per wave: block b = (row b / 2 of the wave's rows, half b % 2)
    static_assert(NW == 4 && C_NDMA % NW == 0, "every wave issues the same number of DMA instructions per stage (the vmcnt accounting)");
    constexpr int NDW = C_NDMA / NW;            // DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_c[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, lg = lane >> 4;

    // XCD-aware order (workgroup L runs on XCD L % 8): every XCD walks its own contiguous band of pixel tiles in row-major order
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) >> 3;
    const int bid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (bid >= P) return;
    const int tx = bid % a.tiles_x, ty = bid / a.tiles_x;
    const int x0 = tx * C_TW, y0 = ty * C_TH;
    const int64_t plane = (int64_t)a.H * a.W;
    const int nks = a.Cin / 32;
    const uint32_t lds_base = lds_addr_uniform(lds_c);

    // ---- the filter bank: nks x 18 pieces of 1 KiB, wave w copies pieces w, w + 4, ...
    {
        const unsigned char *wsrc = static_cast<const unsigned char *>(a.wt_h3);
        for (int pc = wave; pc < 18 * nks; pc += NW) lds_dma16_s(wsrc, (uint32_t)(pc * 1024 + lane * 16), lds_base + W0 + pc * 1024);
    }
    // ---- patch DMA plan: instruction j = wave + 4 i (i = 0 .. 10) fills pieces 64 j .. 64 j + 63 of the stage image
    // [plane][octet][352]: piece q = (plane * 4 + octet) * 352 + r, r < 340: patch pixel (r / 34, r % 34); pad pieces copy a
    // valid address.  Source: [n][C / 8][plane][Hp][Wp] pieces, patch row 0 = image row y0 - 1 = padded row y0.
    uint32_t voff[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        const int q0 = (wave + NW * i) * 64 + lane, q = q0 < 8 * C_OST ? q0 : 0;
        const int po = q / C_OST, r = q - po * C_OST;
        const int pl = po >> 2, o = po & 3;
        const int rr = r < C_NPX ? r : 0;
        const int py = rr / C_PW, px = rr - py * C_PW;
        voff[i] = (uint32_t)((((o * 2 + pl) * a.in_Hp + py) * a.in_Wp + px) * 16);
    }
    const int64_t ks_stride = (int64_t)8 * a.in_Hp * a.in_Wp * 16;           // 32 channels = 4 octets x 2 planes
    const unsigned char *tile_src = static_cast<const unsigned char *>(a.in_pk) + ((int64_t)y0 * a.in_Wp + x0) * 16;
    auto issue_stage = [&](int s, int ks) __attribute__((always_inline)) {
        const unsigned char *sb = tile_src + (int64_t)s * a.in_pk_sample_bytes + ks * ks_stride;
#pragma unroll
        for (int i = 0; i < NDW; ++i) lds_dma16_s(sb, voff[i], lds_base + (wave + NW * i) * 1024);
    };

    // ---- MFMA operands: A = weights [cout = lane & 15][octet = lane >> 4], B = patch [octet = lane >> 4][pixel = lane & 15]
    const uint32_t a_off = (uint32_t)(W0 + lane * 16);
    // block b = 2 rr + h of wave w: row (NB / 2) w + rr, columns 16 h .. 16 h + 15
    const uint32_t b_off = (uint32_t)((lg * C_OST + ((NB / 2) * wave) * C_PW + lp) * 16);
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double sum[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sum[c] = 0.0;

    // the pixel this lane post-processes: pixel lp of block lg of its wave
    const int prow = (NB / 2) * wave + (lg >> 1), pcol = 16 * (lg & 1) + lp;
    const int gy = y0 + prow, gx = x0 + pcol;
    const bool pix_ok = lg < NB && gy < a.H && gx < a.W;
    const int64_t pix = (int64_t)gy * a.W + gx;
    const float mscale = 1.f / (a.h3_vscale * a.h3_uscale);

    issue_stage(0, 0);
    const int total = a.T * nks;
    int s = 0, ks = 0;
    for (int g = 0; g < total; ++g) {
        // this wave's DMA of stage g (and, g = 0, of the filter bank) has landed; behind the barrier everybody's has
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int s2 = s, k2 = ks + 1;
        if (k2 == nks) { k2 = 0; ++s2; }
        const unsigned char *ps = lds_c + b_off, *ws = lds_c + a_off + ks * C_WKS;
        // fragments one tap ahead: [10 ds_read_b128 of tap t + 1][12 MFMAs of tap t].  With ONE wave per SIMD nothing else covers an
        // LDS round trip: left alone, hipcc placed every read right in front of its first use (35 lgkmcnt waits per stage in the
        // .s) and a stage took ~8 k cycles for ~1.8 k cycles of MFMAs.
        half8 A[2][2], B[2][NB][2];
        auto fetch = [&](int t, int par) __attribute__((always_inline)) {
            const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) A[par][pl] = *reinterpret_cast<const half8 *>(ws + (t * 2 + pl) * 1024);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    B[par][b][pl] = *reinterpret_cast<const half8 *>(ps + pl * (4 * C_OST * 16) + (((b >> 1) + ky) * C_PW + (b & 1) * 16 + kx) * 16);
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * NB, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9 && !(ABL & 8)) {
                fetch(t + 1, (t + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * NB, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NB, 0);
            // smallest terms first: (lo, hi) (hi, lo) (hi, hi); consecutive MFMAs on different accumulators
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (ABL & 1) acc[b][term] += (float)A[t & 1][PA[term]][0] + (float)B[t & 1][b][PB[term]][1];
                    else acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(ABL & 8) ? 0 : (t & 1)][PA[term]], B[(ABL & 8) ? 0 : (t & 1)][b][PB[term]], acc[b], 0, 0, 0);
                }
            }
        }
        if (g + 1 < total && !(ABL & 4)) {
            // the next stage goes into the buffer as soon as every wave has read this one — under this sample's Softmax when there
            // is one; the other workgroup of the CU covers the rest
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_stage(s2, k2);
        }
        if (ks == nks - 1 && !(ABL & 2)) {
            // ---- end of sample s: acc[b][j] = class 4 lg + j at pixel lp of block b.  4 x 4 transpose over the four 16-lane rows
            // (register index b <-> row lg), per j: afterwards w[c] = class 4 c + j at pixel lp of block lg.
            float x[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][j]), __float_as_uint(acc[1][j]), false, false);
                const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2][j]), __float_as_uint(acc[3][j]), false, false);
                const auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
                const auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
                x[0 + j] = __uint_as_float(t02[0]); x[4 + j] = __uint_as_float(t13[0]);
                x[8 + j] = __uint_as_float(t02[1]); x[12 + j] = __uint_as_float(t13[1]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float v = c < a.C ? x[c] * (a.ep_scale[c] * mscale) + a.ep_shift[c] : 0.f;
                if (a.relu) v = v > 0.f ? v : 0.f;
                x[c] = v;
            }
#include "conv_cls_mc_softmax.inc"
        }
        s = s2; ks = k2;
    }
    if (!pix_ok) return;
#include "conv_cls_mc_maps.inc"
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool cls_h3_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 3 && cin % 32 == 0 && cin >= 32 && cin <= 96 && cout >= 1 && cout <= 16 && H >= 1 && W >= 1;
}

void cls_h3_tile(int *th, int *tw) { *th = C_TH; *tw = C_TW; }

static inline uint16_t c_f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static inline float c_f16_value(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// Caffe (Cout,Cin,3,3) -> fp16 hi / lo planes [Cin / 32][tap][plane][octet][cout 16][8]; returns the power of two applied
float cls_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out) {
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * 9; ++i) wmax = std::fmax(wmax, std::fabs(W[i]));
    int ex = 0;
    if (wmax > 0.f) (void)std::frexp(wmax, &ex);
    const float scale = std::ldexp(1.f, 8 - ex);
    const int nks = cin / 32;
    out.assign((size_t)nks * C_WKS / 2, 0);
    for (int k = 0; k < nks; ++k)
        for (int t = 0; t < 9; ++t)
            for (int o = 0; o < 4; ++o)
                for (int co = 0; co < cout; ++co)
                    for (int e = 0; e < 8; ++e) {
                        const float x = W[((size_t)co * cin + k * 32 + o * 8 + e) * 9 + t] * scale;
                        const uint16_t hi = c_f16_bits(x);
                        const uint16_t lo = c_f16_bits(x - c_f16_value(hi));
                        uint16_t *img = out.data() + ((size_t)(k * 9 + t) * 2) * 512;          // hi plane: 512 halfs, then lo
                        const size_t at = (size_t)(o * 16 + co) * 8 + e;
                        img[at] = hi;
                        img[512 + at] = lo;
                    }
    return scale;
}

void launch_conv_cls_h3(const ClsMcArgs &a0, hipStream_t s) {
    ClsMcArgs a = a0;
    a.tiles_x = (a.W + C_TW - 1) / C_TW;
    a.tiles_y = (a.H + C_TH - 1) / C_TH;
    if (!a.in_pk || !a.wt_h3 || !(a.h3_vscale > 0.f) || !(a.h3_uscale > 0.f) || !cls_h3_supported(3, a.Cin, a.C, a.H, a.W) ||
        a.in_Hp < a.tiles_y * C_TH + 2 || a.in_Wp < a.tiles_x * C_TW + 2)
        throw std::invalid_argument("launch_conv_cls_h3: unsupported layer / packed input plane too small");
    if (a.sum_chunk <= 0 || a.sum_chunk > (int64_t)a.H * a.W) a.sum_chunk = (int64_t)a.H * a.W;
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_cls_h3_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int P = a.tiles_x * a.tiles_y, per = (P + 7) / 8;
    // 64 input channels: 80 KB, TWO workgroups of this kernel per CU (the two halves of a CU's LDS: what turned the sum of the kernel's
    // two phases into their maximum, NOTEBOOK 3.1g) — a full CU has no LDS left for anybody else, and what a half-filled CU (head / tail of
    // the launch) may host beside one of them is covered by tests/test_gpu_coresidency.py.  Any other width would leave a gap (98 KB at 96
    // channels): there the single workgroup claims the CU's whole LDS, like the other kernels that issue LDS-DMA in inline assembly.
    const size_t need = (size_t)C_STAGE + (size_t)(a.Cin / 32) * C_WKS;
    const size_t lds = 2 * need == (size_t)160 * 1024 ? need : (size_t)160 * 1024;
    lds_claim_note(LDS_CLAIM_CLS_H3, 2 * need == (size_t)160 * 1024 ? 2 * need : lds);
#ifdef SIVO_DIAG
    if (const char *ab = SIVO_DIAG_ENV("SIVO_CLS_ABL")) {                                     // diagnostic build: ablations of the default form
#define CLS_ABL_CASE(n)                                                                                                                             \
    case n:                                                                                                                                         \
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_cls_h3_kernel<4, n>), hipFuncAttributeMaxDynamicSharedMemorySize, C_STAGE + 3 * C_WKS)); \
        hipLaunchKernelGGL((conv_cls_h3_kernel<4, n>), dim3((unsigned)(8 * per)), dim3(256), lds, s, a);                            \
        return;
        switch (std::atoi(ab)) { CLS_ABL_CASE(1) CLS_ABL_CASE(2) CLS_ABL_CASE(3) CLS_ABL_CASE(4) CLS_ABL_CASE(6) CLS_ABL_CASE(8) CLS_ABL_CASE(9) CLS_ABL_CASE(11) CLS_ABL_CASE(15) default: break; }
#undef CLS_ABL_CASE
    }
#endif
    hipLaunchKernelGGL((conv_cls_h3_kernel<4>), dim3((unsigned)(8 * per)), dim3(256), lds, s, a);
}

}  // namespace sivo
