// conv7_x6.hip — direct 7x7 convolution (SegNet-Basic's 64 -> 64 layers) on the bf16 matrix cores: "bf16x6".
//
// SegNet-Basic (config/bayesian_segnet/basic/kitti/*.prototxt) is eight 7x7 convolutions; seven of them are 64 -> 64 and
// hold 97 % of its 1.22 TFLOP per frame (T = 6).  A 7x7 filter has no usable Winograd form in fp32 (F(2x2,7x7) needs the
// eight interpolation points of F(6x6,3x3), whose error was the reason not to use that one), so the fp32-MFMA direct
// kernel (conv_mfma_kernel<7,...>, 0.65 of the 157 TFLOP/s fp32 matrix peak) is MFMA-bound at its very instruction rate.
// The bf16 matrix cores are 16x faster, and an fp32 product is six bf16 products (conv_wino4.hip, "bf16x6": each operand the
// exact sum of three bf16 values up to 2^-24; smallest terms first; fp32 accumulate — the error of the fp32 FMA chain it
// replaces, no range restriction since bf16 has fp32's exponent): 6 x 16 cycles per 32 channels instead of 8 x 32.
//
//   workgroup = 8 x 32 output pixels x 64 couts, 8 waves (wave w: image row w, two 16-pixel m-tiles x four 16-cout
//   n-tiles = 32 accumulator VGPRs).  Per 32-channel half of the input:
//     * the 14 x 38 halo patch is read ONCE from HBM (buffer loads: an offset outside the image returns the zero padding),
//       split into three bf16 planes and stored as 16-byte pieces (8 channels of one pixel = the A fragment of a lane),
//       [plane][channel octet][pixel]: 104 KB of LDS;
//     * 49 taps = 49 stages: the tap's 32 x 64 weights arrive as the LDS image of their three planes (12 KiB, split on the
//       host) by LDS-DMA one stage ahead (inline asm, lds_dma.hpp; double-buffered); a wave reads 6 A pieces (the patch at
//       the tap's shift) + 12 B pieces and issues 48 MFMAs (v_mfma_f32_16x16x32_bf16); one fence-free barrier per stage.
//   Every ds_read_b128 covers 16 consecutive pieces = all 64 banks once (piece strides are multiples of 256 bytes).
// Bias + BN affine, ReLU and the Philox dropout in the epilogue as in the other kernels; float4 stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "lds_dma.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t c7_dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
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

constexpr int C7_TH = 8, C7_TW = 32;                       // output pixels per workgroup
constexpr int C7_PH = C7_TH + 6, C7_PW = C7_TW + 6;        // halo patch 14 x 38
constexpr int C7_NPIX = C7_PH * C7_PW;                     // 532
constexpr int C7_PIXP = 544;                               // padded: octet stride 544 * 16 B = 34 * 256 B
constexpr int C7_PLANE = 4 * C7_PIXP * 16;                 // bytes of one bf16 plane of the patch (4 channel octets)
constexpr int C7_PATCH = 3 * C7_PLANE;                     // 104448
constexpr int C7_WPLANE = 4 * 64 * 16;                     // bytes of one plane of a tap's weights (4 octets x 64 couts)
constexpr int C7_WSTAGE = 3 * C7_WPLANE;                   // 12288 = 12 KiB
constexpr int C7_LDS = C7_PATCH + 2 * C7_WSTAGE;           // 129024
constexpr int C7_NTHR = 512;
constexpr int C7_ITEMS = 4 * C7_NPIX;                      // (octet, pixel) staging items per half
constexpr int C7_IT = (C7_ITEMS + C7_NTHR - 1) / C7_NTHR;  // 5

// UNPOOL: `a.in` is the pooled tensor of an Upsample (scale 2) layer and `a.unpool_mask` its window codes: the patch loader
// reads the pooled value and its code where it would read the unpooled pixel (value at the recorded position of the 2 x 2
// window, zero elsewhere) — the unpooled tensor never exists and unpool2_kernel is not run.
template <bool UNPOOL>
__global__ __launch_bounds__(C7_NTHR, 1) void conv7_x6_kernel(ConvArgs a, const uint4 *__restrict__ Wx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds7[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // every XCD owns a contiguous band of pixel tiles (row-major): neighbouring tiles share their halo in one L2
    const int P = a.tiles_x * a.tiles_y * a.N, band = (P + 7) >> 3;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    int bid = xcd * band + slot;
    if (slot >= band || bid >= P) return;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * C7_TW, y0 = ty * C7_TH;
    const int64_t plane = (int64_t)a.H * a.W;
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;      // plane of the tensor that is actually read
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (int64_t)n * a.in_sample_stride), 0, (int)(a.Cin * plane_in * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mk_rsrc =
        UNPOOL ? __builtin_amdgcn_make_buffer_rsrc((void *)(a.unpool_mask + (int64_t)n * a.unpool_mask_stride), 0, (int)(a.Cin * plane_in), 0x00020000)
               : in_rsrc;
    constexpr uint32_t INV = 0xfffffff0u;

    // staging items of this thread: (octet g, patch pixel q); element offset of the pixel inside a plane, INV outside the image
    uint32_t s_off[C7_IT];
    int s_dst[C7_IT], s_code[C7_IT];
#pragma unroll
    for (int it = 0; it < C7_IT; ++it) {
        const int i = tid + it * C7_NTHR;
        const int g = i / C7_NPIX, q = i % C7_NPIX;
        const int py = q / C7_PW, px = q % C7_PW;
        const int gy = y0 + py - 3, gx = x0 + px - 3;
        const bool ok = i < C7_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        s_off[it] = !ok ? INV : UNPOOL ? (uint32_t)(g * 8 * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(g * 8 * plane + (int64_t)gy * a.W + gx);
        s_code[it] = (gy & 1) * 2 + (gx & 1);
        s_dst[it] = i < C7_ITEMS ? (g * C7_PIXP + q) * 16 : -1;
    }

    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: A piece of (m-tile j, tap shift 0) = pixel (wave, 16 j + li), octet lk; B piece = cout nt * 16 + li, octet lk
    const uint32_t a_base = (uint32_t)((lk * C7_PIXP + wave * C7_PW + li) * 16);
    const uint32_t b_base = (uint32_t)(C7_PATCH + (lk * 64 + li) * 16);
    const uint32_t w_lds = lds_addr_uniform(lds7 + C7_PATCH);
    const int nhalf = a.Cin / 32;

    auto dma_weights = [&](int stage, int buf) {            // stage = half * 49 + tap
        const uint4 *src = Wx + (int64_t)stage * (C7_WSTAGE / 16);
        // 12 KiB = 12 pieces: every wave one, waves 0..3 a second one
        lds_dma16(src + wave_u * 64 + lane, w_lds + (uint32_t)(buf * C7_WSTAGE + wave_u * 1024));
        if (wave_u < 4) lds_dma16(src + (8 + wave_u) * 64 + lane, w_lds + (uint32_t)(buf * C7_WSTAGE + (8 + wave_u) * 1024));
    };

    for (int half = 0; half < nhalf; ++half) {
        // ---- the half's patch: 8 channels of a pixel per item, split into three bf16 planes, one 16-byte piece per plane
        float v[C7_IT][8];
        const uint32_t hb = (uint32_t)(half * 32 * plane_in);
#pragma unroll
        for (int it = 0; it < C7_IT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t oe = s_off[it] == INV ? INV : s_off[it] + hb + (uint32_t)(e * plane_in);
                v[it][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(oe == INV ? INV : oe * 4u), 0, 0));
                if (UNPOOL) {
                    const int m = (int)(__builtin_amdgcn_raw_buffer_load_b8(mk_rsrc, (int)oe, 0, 0) & 0xffu);
                    v[it][e] = m == s_code[it] ? v[it][e] : 0.f;
                }
            }
        dma_weights(half * 49, 0);
#pragma unroll
        for (int it = 0; it < C7_IT; ++it) {
            bf16x8 p1, p2, p3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[it][e];
                const __bf16 x1 = (__bf16)x;
                const float r1 = x - (float)x1;
                const __bf16 x2 = (__bf16)r1;
                const float r2 = r1 - (float)x2;
                p1[e] = x1; p2[e] = x2; p3[e] = (__bf16)r2;
            }
            if (s_dst[it] >= 0) {
                *reinterpret_cast<bf16x8 *>(lds7 + s_dst[it]) = p1;
                *reinterpret_cast<bf16x8 *>(lds7 + C7_PLANE + s_dst[it]) = p2;
                *reinterpret_cast<bf16x8 *>(lds7 + 2 * C7_PLANE + s_dst[it]) = p3;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();

        // ---- 49 taps
#pragma unroll 1
        for (int tap = 0; tap < 49; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 49) dma_weights(half * 49 + tap + 1, buf ^ 1);
            const int dy = tap / 7, dx = tap - dy * 7;
            const unsigned char *ap = lds7 + a_base + (uint32_t)((dy * C7_PW + dx) * 16);
            const unsigned char *bp = lds7 + b_base + (uint32_t)(buf * C7_WSTAGE);
            bf16x8 af[2][3], bf[4][3];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *reinterpret_cast<const bf16x8 *>(bp + pl * C7_WPLANE + nt * 256);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[j][pl] = *reinterpret_cast<const bf16x8 *>(ap + pl * C7_PLANE + j * 256);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                // smallest terms first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][PA[term]], bf[nt][PB[term]], acc[j][nt], 0, 0, 0);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the next tap's weights
            lds_barrier();
        }
    }

    // ---- epilogue: acc[j][nt][r] = pixel (row wave, column 16 j + 4 lk + r), cout nt * 16 + li
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
    const int y = y0 + wave;
    if (y >= a.H) return;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = nt * 16 + li;
        if (co >= a.Cout) continue;
        const float sc = a.ep_scale[co], sh = a.ep_shift[co];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int x = x0 + 16 * j + 4 * lk;
            if (x >= a.W) continue;
            float v4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v4[r] = acc[j][nt][r] * sc + sh;
                if (a.relu) v4[r] = v4[r] > 0.f ? v4[r] : 0.f;
            }
            if (a.drop_site >= 0) {
                const uint32_t e = (uint32_t)((co * a.H + y) * a.W + x);
                const uint32_t w = c7_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = ((w >> r) & 1u) ? v4[r] * 2.f : 0.f;
            }
            *reinterpret_cast<float4 *>(out_n + (int64_t)co * plane + (int64_t)y * a.W + x) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
    }
}

// 7x7, Cin a multiple of 32, 64 couts, W a multiple of 4 (float4 stores, one Philox word per store), a sample below 2 GiB
bool conv7_x6_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 7 && cin >= 32 && cin % 32 == 0 && cout == 64 && (W % 4) == 0 && (int64_t)cin * H * W * 4 < (1ll << 31);
}

static inline uint16_t c7_bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float c7_bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
}

// Caffe (64, Cin, 7, 7) -> per stage (half, tap) the LDS image of the three bf16 planes: [plane][octet][cout][8 channels]
void conv7_x6_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out) {
    const int nhalf = cin / 32;
    out.assign((size_t)nhalf * 49 * (C7_WSTAGE / 2), 0);
    for (int half = 0; half < nhalf; ++half)
        for (int tap = 0; tap < 49; ++tap) {
            uint16_t *stage = out.data() + (size_t)(half * 49 + tap) * (C7_WSTAGE / 2);
            for (int g = 0; g < 4; ++g)
                for (int co = 0; co < cout; ++co)
                    for (int e = 0; e < 8; ++e) {
                        const float x = W[((size_t)co * cin + half * 32 + g * 8 + e) * 49 + tap];
                        const uint16_t x1 = c7_bf16_rne(x);
                        const float r1 = x - c7_bf16_to_float(x1);
                        const uint16_t x2 = c7_bf16_rne(r1);
                        const float r2 = r1 - c7_bf16_to_float(x2);
                        const size_t o = (size_t)(g * 64 + co) * 8 + e;
                        stage[o] = x1; stage[C7_WPLANE / 2 + o] = x2; stage[2 * (C7_WPLANE / 2) + o] = c7_bf16_rne(r2);
                    }
        }
}

void launch_conv7_x6(const ConvArgs &a0, hipStream_t s) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once)
    {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_x6_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C7_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_x6_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C7_LDS);
    }
    ConvArgs a = a0;
    a.tiles_x = (a.W + C7_TW - 1) / C7_TW;
    a.tiles_y = (a.H + C7_TH - 1) / C7_TH;
    const int P = a.tiles_x * a.tiles_y * a.N, band = (P + 7) / 8;
    if (a.unpool_mask) hipLaunchKernelGGL(conv7_x6_kernel<true>, dim3((unsigned)(8 * band)), dim3(C7_NTHR), C7_LDS, s, a, reinterpret_cast<const uint4 *>(a.wt_x6));
    else hipLaunchKernelGGL(conv7_x6_kernel<false>, dim3((unsigned)(8 * band)), dim3(C7_NTHR), C7_LDS, s, a, reinterpret_cast<const uint4 *>(a.wt_x6));
}

}  // namespace sivo
