// conv7_h3.hip — direct 7x7 convolution (SegNet-Basic's 64 -> 64 layers) on the fp16 matrix cores: "f16x3" (round 4).
//
// The same workgroup shape as conv7_x6.hip (8 x 32 output pixels x 64 couts, 8 waves, the 14 x 38 halo patch of a 32-channel half
// staged once as fragment pieces, the taps' weights streamed by LDS-DMA), with the arithmetic of conv_wino4_h3.hip / conv3_h3.hip:
// an fp32 operand (times a per-layer power of two) is fp16 hi + fp16 lo, exact to 2^-22, and a product is THREE
// v_mfma_f32_16x16x32_f16 (lo hi, hi lo, hi hi; fp32 accumulation) where bf16x6 needs six — half the matrix-core work for the
// layers that are 97 % of SegNet-Basic (BASELINE configs[1]; reference config/bayesian_segnet/basic/kitti/*.prototxt:19-393).
//   * patch: two fp16 planes instead of three bf16 planes: 69,632 bytes (104,448 before);
//   * weights: a tap is 8 KiB ([plane][octet][cout][8], split on the host, times a power of two); a STAGE is three taps (24 MFMAs
//     per tap and wave would otherwise stand against one barrier per tap): 24 KiB by LDS-DMA one stage ahead, double-buffered;
//   * range: the input's power of two comes from the calibration pass (segnet.cpp, as for conv3_h3.hip); a scaled input that
//     leaves the fp16 range raises the overflow flag and the frame is recomputed on the bf16x6 kernel, which stays resident.
// LDS: 69,632 + 2 x 24,576 = 118,784 bytes.  Error: 2^-22 of the split + the fp32 accumulation (no range restriction of bf16x6's
// kind is given up: the flag guards it).
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

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t h7_dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
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

constexpr int H7_TH = 8, H7_TW = 32;                       // output pixels per workgroup
constexpr int H7_PH = H7_TH + 6, H7_PW = H7_TW + 6;        // halo patch 14 x 38
constexpr int H7_NPIX = H7_PH * H7_PW;                     // 532
constexpr int H7_PIXP = 544;                               // padded: octet stride 544 * 16 B = 34 * 256 B
constexpr int H7_PLANE = 4 * H7_PIXP * 16;                 // bytes of one fp16 plane of the patch (4 channel octets)
constexpr int H7_PATCH = 2 * H7_PLANE;                     // 69,632
constexpr int H7_WPLANE = 4 * 64 * 16;                     // bytes of one plane of a tap's weights (4 octets x 64 couts)
constexpr int H7_WTAP = 2 * H7_WPLANE;                     // 8 KiB: one tap
constexpr int H7_TG = 3;                                   // taps per stage
constexpr int H7_NSTG = (49 + H7_TG - 1) / H7_TG;          // 17 stages per half (the last one holds one tap)
constexpr int H7_WSTAGE = H7_TG * H7_WTAP;                 // 24 KiB
constexpr int H7_LDS = H7_PATCH + 2 * H7_WSTAGE;           // 118,784
constexpr int H7_NTHR = 512;
constexpr int H7_ITEMS = 4 * H7_NPIX;                      // (octet, pixel) staging items per half
constexpr int H7_IT = (H7_ITEMS + H7_NTHR - 1) / H7_NTHR;  // 5

// UNPOOL: `a.in` is the pooled tensor of an Upsample (scale 2) layer and `a.unpool_mask` its window codes: the patch loader
// reads the pooled value and its code where it would read the unpooled pixel (value at the recorded position of the 2 x 2
// window, zero elsewhere) — the unpooled tensor never exists and unpool2_kernel is not run.
template <bool UNPOOL>
__global__ __launch_bounds__(H7_NTHR, 1) void conv7_h3_kernel(ConvArgs a, const uint4 *__restrict__ Wx) {
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
    const int x0 = tx * H7_TW, y0 = ty * H7_TH;
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
    uint32_t s_off[H7_IT];
    int s_dst[H7_IT], s_code[H7_IT];
#pragma unroll
    for (int it = 0; it < H7_IT; ++it) {
        const int i = tid + it * H7_NTHR;
        const int g = i / H7_NPIX, q = i % H7_NPIX;
        const int py = q / H7_PW, px = q % H7_PW;
        const int gy = y0 + py - 3, gx = x0 + px - 3;
        const bool ok = i < H7_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        s_off[it] = !ok ? INV : UNPOOL ? (uint32_t)(g * 8 * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(g * 8 * plane + (int64_t)gy * a.W + gx);
        s_code[it] = (gy & 1) * 2 + (gx & 1);
        s_dst[it] = i < H7_ITEMS ? (g * H7_PIXP + q) * 16 : -1;
    }

    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: A piece of (m-tile j, tap shift 0) = pixel (wave, 16 j + li), octet lk; B piece = cout nt * 16 + li, octet lk
    const uint32_t a_base = (uint32_t)((lk * H7_PIXP + wave * H7_PW + li) * 16);
    const uint32_t b_base = (uint32_t)(H7_PATCH + (lk * 64 + li) * 16);
    const uint32_t w_lds = lds_addr_uniform(lds7 + H7_PATCH);
    const int nhalf = a.Cin / 32;

    // stage = (half, tap group): taps 3 g .. 3 g + 2 (the last group: tap 48 alone); 8 pieces of 1 KiB per tap, wave w copies piece w of each
    auto dma_weights = [&](int half, int grp, int buf) {
        const uint4 *src = Wx + ((int64_t)half * 49 + grp * H7_TG) * (H7_WTAP / 16);
#pragma unroll
        for (int t = 0; t < H7_TG; ++t)
            if (grp * H7_TG + t < 49) lds_dma16(src + (t * 8 + wave_u) * 64 + lane, w_lds + (uint32_t)(buf * H7_WSTAGE + (t * 8 + wave_u) * 1024));
    };
    uint32_t ovf = 0u;          // largest |scaled input| seen by this lane, as bits << 1
    const float mscale = 1.f / (a.h3_vscale * a.h3_uscale);

    for (int half = 0; half < nhalf; ++half) {
        // ---- the half's patch: 8 channels of a pixel per item, split into three bf16 planes, one 16-byte piece per plane
        float v[H7_IT][8];
        const uint32_t hb = (uint32_t)(half * 32 * plane_in);
#pragma unroll
        for (int it = 0; it < H7_IT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t oe = s_off[it] == INV ? INV : s_off[it] + hb + (uint32_t)(e * plane_in);
                v[it][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(oe == INV ? INV : oe * 4u), 0, 0));
                if (UNPOOL) {
                    const int m = (int)(__builtin_amdgcn_raw_buffer_load_b8(mk_rsrc, (int)oe, 0, 0) & 0xffu);
                    v[it][e] = m == s_code[it] ? v[it][e] : 0.f;
                }
            }
        dma_weights(half, 0, 0);
#pragma unroll
        for (int it = 0; it < H7_IT; ++it) {
            uint32_t pr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xs = v[it][e] * a.h3_vscale;
                const _Float16 hi = (_Float16)xs;
                const _Float16 lo = (_Float16)(xs - (float)hi);
                const uint32_t mag = __float_as_uint(xs) << 1;
                ovf = mag > ovf ? mag : ovf;
                pr[e] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
            }
            if (s_dst[it] >= 0) {
                u32x4 hv, lv;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    hv[k] = __builtin_amdgcn_perm(pr[2 * k + 1], pr[2 * k], 0x05040100u);
                    lv[k] = __builtin_amdgcn_perm(pr[2 * k + 1], pr[2 * k], 0x07060302u);
                }
                *reinterpret_cast<u32x4 *>(lds7 + s_dst[it]) = hv;
                *reinterpret_cast<u32x4 *>(lds7 + H7_PLANE + s_dst[it]) = lv;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();

        // ---- 17 stages of three taps
#pragma unroll 1
        for (int grp = 0; grp < H7_NSTG; ++grp) {
            const int buf = grp & 1;
            if (grp + 1 < H7_NSTG) dma_weights(half, grp + 1, buf ^ 1);
#pragma unroll
            for (int t = 0; t < H7_TG; ++t) {
                const int tap = grp * H7_TG + t;
                if (tap < 49) {
                    const int dy = tap / 7, dx = tap - dy * 7;
                    const unsigned char *ap = lds7 + a_base + (uint32_t)((dy * H7_PW + dx) * 16);
                    const unsigned char *bp = lds7 + b_base + (uint32_t)(buf * H7_WSTAGE + t * H7_WTAP);
                    half8 af[2][2], bf[4][2];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) bf[nt][pl] = *reinterpret_cast<const half8 *>(bp + pl * H7_WPLANE + nt * 256);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) af[j][pl] = *reinterpret_cast<const half8 *>(ap + pl * H7_PLANE + j * 256);
                    // smallest terms first: (lo, hi) (hi, lo) (hi, hi); consecutive MFMAs on different accumulators
#pragma unroll
                    for (int term = 0; term < 3; ++term) {
                        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j][PA[term]], bf[nt][PB[term]], acc[j][nt], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the next stage's weights
            lds_barrier();
        }
    }
    if (ovf > (0x477fe000u << 1)) atomicOr(a.h3_flag, 1u);          // !(|xs| <= 65504): a scaled input left the fp16 range (or was not finite)

    // ---- epilogue: acc[j][nt][r] = pixel (row wave, column 16 j + 4 lk + r), cout nt * 16 + li
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
    const int y = y0 + wave;
    if (y >= a.H) return;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = nt * 16 + li;
        if (co >= a.Cout) continue;
        const float sc = a.ep_scale[co] * mscale, sh = a.ep_shift[co];
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
                const uint32_t w = h7_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = ((w >> r) & 1u) ? v4[r] * 2.f : 0.f;
            }
            *reinterpret_cast<float4 *>(out_n + (int64_t)co * plane + (int64_t)y * a.W + x) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
    }
}

// 7x7, Cin a multiple of 32, 64 couts, W a multiple of 4 (float4 stores, one Philox word per store), a sample below 2 GiB
bool conv7_h3_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 7 && cin >= 32 && cin % 32 == 0 && cout == 64 && (W % 4) == 0 && (int64_t)cin * H * W * 4 < (1ll << 31);
}

static inline uint16_t h7_f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static inline float h7_f16_value(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// Caffe (64, Cin, 7, 7) -> per (half, tap) the LDS image of the two fp16 planes: [plane][octet][cout][8 channels]; returns the
// power of two the weights were multiplied by (max |W| * scale in [2^7, 2^8))
float conv7_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out) {
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * 49; ++i) wmax = std::fmax(wmax, std::fabs(W[i]));
    int ex = 0;
    if (wmax > 0.f) (void)std::frexp(wmax, &ex);
    const float scale = std::ldexp(1.f, 8 - ex);
    const int nhalf = cin / 32;
    out.assign((size_t)nhalf * 49 * (H7_WTAP / 2), 0);
    for (int half = 0; half < nhalf; ++half)
        for (int tap = 0; tap < 49; ++tap) {
            uint16_t *img = out.data() + (size_t)(half * 49 + tap) * (H7_WTAP / 2);
            for (int g = 0; g < 4; ++g)
                for (int co = 0; co < cout; ++co)
                    for (int e = 0; e < 8; ++e) {
                        const float x = W[((size_t)co * cin + half * 32 + g * 8 + e) * 49 + tap] * scale;
                        const uint16_t hi = h7_f16_bits(x);
                        const uint16_t lo = h7_f16_bits(x - h7_f16_value(hi));
                        const size_t o = (size_t)(g * 64 + co) * 8 + e;
                        img[o] = hi; img[H7_WPLANE / 2 + o] = lo;
                    }
        }
    return scale;
}

void launch_conv7_h3(const ConvArgs &a0, hipStream_t s) {
    if (!a0.wt_h3 || !(a0.h3_vscale > 0.f) || !a0.h3_flag) throw std::invalid_argument("launch_conv7_h3: weights / scale / flag missing");
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_h3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv7_h3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    ConvArgs a = a0;
    a.tiles_x = (a.W + H7_TW - 1) / H7_TW;
    a.tiles_y = (a.H + H7_TH - 1) / H7_TH;
    const int P = a.tiles_x * a.tiles_y * a.N, band = (P + 7) / 8;
    // the kernel uses H7_LDS (116 KB); it claims the CU's whole LDS like every kernel that issues LDS-DMA in inline assembly (DESIGN 3.3:
    // no foreign workgroup beside it — it is a one-workgroup-per-CU kernel either way)
    const size_t lds = (size_t)160 * 1024;
    static_assert(H7_LDS <= 160 * 1024, "LDS");
    lds_claim_note(LDS_CLAIM_CONV7_H3, lds);
    if (a.unpool_mask) hipLaunchKernelGGL(conv7_h3_kernel<true>, dim3((unsigned)(8 * band)), dim3(H7_NTHR), lds, s, a, reinterpret_cast<const uint4 *>(a.wt_h3));
    else hipLaunchKernelGGL(conv7_h3_kernel<false>, dim3((unsigned)(8 * band)), dim3(H7_NTHR), lds, s, a, reinterpret_cast<const uint4 *>(a.wt_h3));
}

}  // namespace sivo
