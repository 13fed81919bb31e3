// segnet_kernels.hip — CDNA4 (gfx950) kernels of the Bayesian SegNet forward.
//
// Reference path: caffe::Net::Forward at src/bayesian_segnet/bayesian_segnet.cpp:310
// over the layer graphs of config/bayesian_segnet/{standard,basic}/kitti/*.prototxt;
// post-processing bayesian_segnet.cpp:180-203,262-297.
//
// Layout in HBM: every activation is planar NCHW fp32 (Caffe's blob layout, so
// blobs can be compared one-to-one with the oracle).  A blob that does not
// depend on the dropout sample ("shared": everything before the first Dropout)
// is stored once (N = 1) and broadcast with a zero sample stride.
//
// conv_mfma: implicit-GEMM convolution on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, 157 TFLOP/s peak).  A
// workgroup of 4 waves owns TH x TW output pixels x BN output channels.  Per
// K-chunk of KC input channels it stages, once, the (TH+k-1) x (TW+k-1) input
// halo patch and the k*k*KC x BN weight slab in LDS; the k*k taps then re-read
// the patch at shifted addresses, so the im2col matrix never exists.
//   A fragment (16 pixels x 4 k): lane l reads patch[c0 + (l>>4)][y+dy][x+dx+(l&15)]
//   B fragment (4 k x 16 cout):   lane l reads slab[tap*KC + c0 + (l>>4)][n0 + (l&15)]
// Channel stride and slab row stride are padded to 16 (mod 32) dwords so the
// two 32-lane halves of a ds_read_b32 hit 32 distinct banks.  All tap / chunk
// offsets are compile-time immediates of the ds_read.
// Epilogue (fused): y = scale[c]*acc + shift[c]  (bias and BN-inference folded),
// ReLU, test-time dropout (Philox bit x2), float4 stores along x.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "common.hpp"
#include "segnet_kernels.hpp"
#include "softmax.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ Philox4x32-10
// Salmon et al., SC'11.  Mask bit of element e at (site, sample):
// bit (e & 31) of word (e >> 5) & 3 of Philox(ctr = {e >> 7, site, sample, 0}, key = seed).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The 32-bit mask word holding element e's keep-bit.
__device__ __forceinline__ uint32_t dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
    uint32_t w[4];
    philox4x32_10(e >> 7, site, sample, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const uint32_t sel = (e >> 5) & 3u;
    return sel == 0 ? w[0] : sel == 1 ? w[1] : sel == 2 ? w[2] : w[3];
}

// ------------------------------------------------------------------ convolution
constexpr int pad16mod32(int n) { return n + ((16 - (n % 32)) + 32) % 32; }

template <int KS, int TH, int TW, int BN, int KC, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    constexpr int PADK = KS / 2;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
    constexpr int CS = pad16mod32(PH * PW);         // channel stride of the patch (dwords)
    constexpr int BNP = (BN % 32 == 0) ? BN + 16 : BN;  // slab row stride (dwords)
    constexpr int MTB = TH * (TW / 16);             // 16-pixel m-tiles per workgroup
    constexpr int MT = MTB / WM, NT = (BN / 16) / WN;
    constexpr int KROWS = KS * KS * KC;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MTB % WM == 0 && (BN / 16) % WN == 0, "tile split");
    static_assert(KC % 4 == 0, "K chunk is a multiple of the MFMA K");

    __shared__ float lds[KC * CS + KROWS * BNP];
    float *s_patch = lds;
    float *s_w = lds + KC * CS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lk = lane >> 4;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = blockIdx.y * BN;

    const float *in_n = a.in + (int64_t)n * a.in_sample_stride;
    const int64_t plane = (int64_t)a.H * a.W;

    // per-lane LDS base offsets (dwords)
    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int g = wm * MT + mt;
        const int row = g / (TW / 16), col = (g % (TW / 16)) * 16;
        a_base[mt] = lk * CS + row * PW + col + li;
    }
    const int b_base = lk * BNP + wn * NT * 16 + li;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging plan (chunk invariant): which global elements this thread brings in.
    // Loads of chunk k+1 are issued into registers BEFORE the MFMA phase of chunk k and land in
    // LDS after it (one LDS buffer, two barriers per chunk), so HBM/L2 latency hides under MFMA.
    constexpr int NPATCH = KC * PH * PW, PITER = (NPATCH + 255) / 256;
    constexpr int V = BN / 4, NWV = KROWS * V, WITER = (NWV + 255) / 256;
    int p_goff[PITER], p_dst[PITER];   // p_dst = LDS dword offset | channel-in-chunk << 24, or -1
#pragma unroll
    for (int it = 0; it < PITER; ++it) {
        const int idx = tid + it * 256;
        const int c = idx / (PH * PW), r = idx % (PH * PW);
        const int py = r / PW, px = r % PW;
        const int gy = y0 + py - PADK, gx = x0 + px - PADK;
        const bool inb = idx < NPATCH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        p_goff[it] = inb ? (int)(c * plane + (int64_t)gy * a.W + gx) : -1;
        p_dst[it] = idx < NPATCH ? ((c * CS + py * PW + px) | (c << 24)) : -1;
    }
    float pv[PITER];
    f32x4 wv[WITER];
    const int nchunks = (a.Cin + KC - 1) / KC;

    auto issue_loads = [&](int chunk) {
        if (a.variant & 4) return;
        const float *psrc = in_n + (int64_t)chunk * KC * plane;
        const int cleft = a.Cin - chunk * KC;   // channels of this chunk that exist
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            const bool ok = p_goff[it] >= 0 && (p_dst[it] >> 24) < cleft;
            const float v = psrc[ok ? p_goff[it] : 0];
            pv[it] = ok ? v : 0.f;
        }
        const float *wsrc = a.wt + (int64_t)chunk * KROWS * a.CoutPad + n0;
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / V, c4 = idx % V;
            const bool ok = (NWV % 256 == 0) || idx < NWV;
            wv[it] = *reinterpret_cast<const f32x4 *>(wsrc + (ok ? (int64_t)row * a.CoutPad + c4 * 4 : 0));
        }
    };
    auto commit_loads = [&]() {
        if (a.variant & 2) return;
#pragma unroll
        for (int it = 0; it < PITER; ++it)
            if (p_dst[it] >= 0) s_patch[p_dst[it] & 0xffffff] = pv[it];
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / V, c4 = idx % V;
            if ((NWV % 256 == 0) || idx < NWV) *reinterpret_cast<f32x4 *>(s_w + row * BNP + c4 * 4) = wv[it];
        }
    };

    issue_loads(0);
    commit_loads();
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        issue_loads(more ? chunk + 1 : chunk);   // unconditional (re-reads the last chunk once): keeps the staging registers out of scratch
        // ---- k*k taps x KC/4 MFMA k-steps
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int c4 = 0; c4 < KC / 4; ++c4) {
                float af[MT], bf[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = s_patch[a_base[mt] + c4 * 4 * CS + dy * PW + dx];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = s_w[b_base + (tap * KC + c4 * 4) * BNP + nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            commit_loads();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds pixels x..x+3 (rows 4*(lane>>4)+r of the m-tile) of channel n0+..+(lane&15)
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
    const bool vec_ok = (a.W & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + (wn * NT + nt) * 16 + li;
        if (co >= a.Cout) continue;
        const float sc = a.ep_scale[co], sh = a.ep_shift[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int g = wm * MT + mt;
            const int y = y0 + g / (TW / 16), x = x0 + (g % (TW / 16)) * 16 + lk * 4;
            if (y >= a.H || x >= a.W) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mt][nt][r] * sc + sh;
                if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            const uint32_t e = (uint32_t)((co * a.H + y) * a.W + x);
            if (a.drop_site >= 0) {
                if (vec_ok) {
                    const uint32_t w = dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? v[r] * 2.f : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t w = dropout_word(e + r, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed);
                        v[r] = ((w >> ((e + r) & 31)) & 1u) ? v[r] * 2.f : 0.f;
                    }
                }
            }
            float *dst = out_n + (int64_t)co * plane + (int64_t)y * a.W + x;
            if ((a.variant & 1) && v[0] != 12345.678f) continue;
            if (vec_ok) {
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (x + r < a.W) dst[r] = v[r];
            }
        }
    }
}

template <int KS, int TH, int TW, int BN, int KC, int WM, int WN>
static void launch_conv_cfg(const ConvArgs &a0, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N), (unsigned)(a.CoutPad / BN));
    hipLaunchKernelGGL((conv_mfma_kernel<KS, TH, TW, BN, KC, WM, WN>), grid, dim3(256), 0, s, a);
}

int conv_cout_tile(int ks, int cout) {
    (void)ks;
    if (cout <= 16) return 16;
    if (cout <= 64) return 64;
    return 128;
}
int conv_k_chunk(int ks, int cin) {
    if (ks == 7) return 4;
    return cin < 8 ? 4 : 8;
}

void launch_conv(const ConvArgs &a, int ks, hipStream_t s) {
    const int bn = conv_cout_tile(ks, a.Cout), kc = conv_k_chunk(ks, a.Cin);
    if (ks == 3) {
        if (bn == 16) return kc == 4 ? launch_conv_cfg<3, 8, 32, 16, 4, 4, 1>(a, s) : launch_conv_cfg<3, 8, 32, 16, 8, 4, 1>(a, s);
        if (bn == 64 && kc == 4) return launch_conv_cfg<3, 8, 32, 64, 4, 4, 1>(a, s);
        if (bn == 64) return launch_conv_cfg<3, 8, 32, 64, 8, 4, 1>(a, s);
        if (kc == 4) return launch_conv_cfg<3, 4, 32, 128, 4, 2, 2>(a, s);
        return launch_conv_cfg<3, 4, 32, 128, 8, 2, 2>(a, s);
    }
    if (ks == 7) {
        if (bn == 16) return launch_conv_cfg<7, 8, 32, 16, 4, 4, 1>(a, s);
        if (bn == 64) return launch_conv_cfg<7, 8, 32, 64, 4, 4, 1>(a, s);
        return launch_conv_cfg<7, 4, 32, 128, 4, 2, 2>(a, s);
    }
    // ks == 1
    if (bn == 16) return kc == 4 ? launch_conv_cfg<1, 8, 32, 16, 4, 4, 1>(a, s) : launch_conv_cfg<1, 8, 32, 16, 8, 4, 1>(a, s);
    if (bn == 64) return kc == 4 ? launch_conv_cfg<1, 8, 32, 64, 4, 4, 1>(a, s) : launch_conv_cfg<1, 8, 32, 64, 8, 4, 1>(a, s);
    return kc == 4 ? launch_conv_cfg<1, 4, 32, 128, 4, 2, 2>(a, s) : launch_conv_cfg<1, 4, 32, 128, 8, 2, 2>(a, s);
}

// ------------------------------------------------------------------ elementwise / pooling
// preprocessImage (bayesian_segnet.cpp:164-178): u8 BGR interleaved -> fp32 planes, raw 0..255.
__global__ void preprocess_kernel(const uint8_t *bgr, float *out, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    out[i] = (float)bgr[3 * i];
    out[hw + i] = (float)bgr[3 * i + 1];
    out[2 * hw + i] = (float)bgr[3 * i + 2];
}
void launch_preprocess(const uint8_t *bgr, float *out, int64_t hw, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, bgr, out, hw);
}

// Pooling MAX 2x2 stride 2 with argmax code (0..3 = dy*2+dx inside the window; first
// strict maximum in scan order wins, as Caffe's '>' update does) and optional dropout.
__global__ void maxpool2_kernel(PoolArgs a) {
    const int64_t total = (int64_t)a.N * a.C * a.Ho * a.Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pw = (int)(i % a.Wo);
    int64_t t = i / a.Wo;
    const int ph = (int)(t % a.Ho); t /= a.Ho;
    const int c = (int)(t % a.C);
    const int n = (int)(t / a.C);
    const float *ip = a.in + (int64_t)n * a.in_sample_stride + (int64_t)c * a.H * a.W;
    const int hs = ph * 2, ws = pw * 2;
    float best = -3.402823466e+38f;
    int code = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int h = hs + dy, w = ws + dx;
            if (h < a.H && w < a.W) {
                const float v = ip[(int64_t)h * a.W + w];
                if (v > best) { best = v; code = dy * 2 + dx; }
            }
        }
    const int64_t chw = (int64_t)a.C * a.Ho * a.Wo;
    const int64_t e = i - (int64_t)n * chw;
    if (n < a.mask_N) a.mask[(int64_t)n * chw + e] = (uint8_t)code;
    if (a.drop_site >= 0) {
        const uint32_t w = dropout_word((uint32_t)e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed);
        best = ((w >> (e & 31)) & 1u) ? best * 2.f : 0.f;
    }
    a.out[i] = best;
}
// The same for FOUR adjacent windows of a row per thread (W % 8 == 0, H even: every geometry of the reference nets at KITTI size):
// two 16-byte loads per input row, one 16-byte store, four codes in one dword, the four dropout bits from one Philox word
// (consecutive elements e .. e + 3 with e % 4 == 0 share it).  Window scan order and the strict '>' are those of the scalar kernel.
__global__ __launch_bounds__(256) void maxpool2x4_kernel(PoolArgs a) {
    const int Wq = a.Wo >> 2;
    const int64_t total = (int64_t)a.N * a.C * a.Ho * Wq;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pq = (int)(i % Wq);
    int64_t t = i / Wq;
    const int ph = (int)(t % a.Ho); t /= a.Ho;
    const int c = (int)(t % a.C);
    const int n = (int)(t / a.C);
    const float *ip = a.in + (int64_t)n * a.in_sample_stride + ((int64_t)c * a.H + 2 * ph) * a.W + 8 * pq;
    const float4 r0a = *reinterpret_cast<const float4 *>(ip), r0b = *reinterpret_cast<const float4 *>(ip + 4);
    const float4 r1a = *reinterpret_cast<const float4 *>(ip + a.W), r1b = *reinterpret_cast<const float4 *>(ip + a.W + 4);
    const float top[8] = {r0a.x, r0a.y, r0a.z, r0a.w, r0b.x, r0b.y, r0b.z, r0b.w};
    const float bot[8] = {r1a.x, r1a.y, r1a.z, r1a.w, r1b.x, r1b.y, r1b.z, r1b.w};
    float best[4];
    uint32_t codes = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float b = -3.402823466e+38f;
        int code = 0;
        if (top[2 * k] > b) { b = top[2 * k]; code = 0; }
        if (top[2 * k + 1] > b) { b = top[2 * k + 1]; code = 1; }
        if (bot[2 * k] > b) { b = bot[2 * k]; code = 2; }
        if (bot[2 * k + 1] > b) { b = bot[2 * k + 1]; code = 3; }
        best[k] = b;
        codes |= (uint32_t)code << (8 * k);
    }
    const int64_t chw = (int64_t)a.C * a.Ho * a.Wo;
    const int64_t e = ((int64_t)c * a.Ho + ph) * a.Wo + 4 * pq;          // element index inside the sample
    if (n < a.mask_N) *reinterpret_cast<uint32_t *>(a.mask + (int64_t)n * chw + e) = codes;
    if (a.drop_site >= 0) {
        const uint32_t w = dropout_word((uint32_t)e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
        for (int k = 0; k < 4; ++k) best[k] = ((w >> k) & 1u) ? best[k] * 2.f : 0.f;
    }
    *reinterpret_cast<float4 *>(a.out + (int64_t)n * chw + e) = make_float4(best[0], best[1], best[2], best[3]);
}
void launch_maxpool2(const PoolArgs &a, hipStream_t s) {
    if (a.W % 8 == 0 && a.H % 2 == 0 && a.Wo * 2 == a.W && a.Ho * 2 == a.H && a.in_sample_stride % 4 == 0) {
        const int64_t total = (int64_t)a.N * a.C * a.Ho * (a.Wo >> 2);
        hipLaunchKernelGGL(maxpool2x4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        return;
    }
    const int64_t total = (int64_t)a.N * a.C * a.Ho * a.Wo;
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// Upsample scale 2 (max-unpool): every input element writes its 2x2 output block
// (value at the recorded window position, zeros elsewhere).
__global__ void unpool2_kernel(UnpoolArgs a) {
    const int64_t total = (int64_t)a.N * a.C * a.H * a.W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % a.W);
    int64_t t = i / a.W;
    const int h = (int)(t % a.H); t /= a.H;   // t = n*C + c
    const int64_t chw = (int64_t)a.C * a.H * a.W;
    const int n = (int)(i / chw);
    const int64_t e = i - (int64_t)n * chw;
    const float v = a.in[i];
    const int code = a.mask[(int64_t)n * a.mask_sample_stride + e];
    const int Wo = a.W * 2;
    float *op = a.out + (t * (int64_t)(a.H * 2) + 2 * h) * Wo + 2 * w;
    *reinterpret_cast<float2 *>(op) = make_float2(code == 0 ? v : 0.f, code == 1 ? v : 0.f);
    *reinterpret_cast<float2 *>(op + Wo) = make_float2(code == 2 ? v : 0.f, code == 3 ? v : 0.f);
}
void launch_unpool2(const UnpoolArgs &a, hipStream_t s) {
    const int64_t total = (int64_t)a.N * a.C * a.H * a.W;
    hipLaunchKernelGGL(unpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// Standalone dropout, also the shared -> per-sample broadcast (in_sample_stride = 0).
__global__ void dropout_kernel(const float *in, int64_t in_sample_stride, float *out, int N, int64_t chw,
                               int site, int sample0, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * chw) return;
    const int n = (int)(i / chw);
    const int64_t e = i - (int64_t)n * chw;
    const float v = in[(int64_t)n * in_sample_stride + e];
    const uint32_t w = dropout_word((uint32_t)e, (uint32_t)site, (uint32_t)(sample0 + n), seed);
    out[i] = ((w >> (e & 31)) & 1u) ? v * 2.f : 0.f;
}
void launch_dropout(const float *in, int64_t in_sample_stride, float *out, int N, int64_t chw, int site,
                    int sample0, uint64_t seed, hipStream_t s) {
    const int64_t total = (int64_t)N * chw;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_sample_stride,
                       out, N, chw, site, sample0, seed);
}

// LRN ACROSS_CHANNELS: x / (1 + alpha/n * sum_window x^2)^beta, window clipped to [0,C).
__global__ void lrn_kernel(const float *in, float *out, int N, int C, int64_t hw, int local_size, float alpha,
                           float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * hw) return;
    const int n = (int)(i / hw);
    const int64_t p = i - (int64_t)n * hw;
    const float *ip = in + (int64_t)n * C * hw + p;
    float *op = out + (int64_t)n * C * hw + p;
    const int half = (local_size - 1) / 2;
    for (int c = 0; c < C; ++c) {
        const int c0 = c - half < 0 ? 0 : c - half, c1 = c + half >= C ? C - 1 : c + half;
        float ss = 0.f;
        for (int cc = c0; cc <= c1; ++cc) { const float v = ip[cc * hw]; ss = __fadd_rn(ss, __fmul_rn(v, v)); }
        const float scale = __fadd_rn(1.0f, __fmul_rn(alpha / (float)local_size, ss));
        op[c * hw] = ip[c * hw] * powf(scale, -beta);
    }
}
void launch_lrn(const float *in, float *out, int N, int C, int64_t hw, int local_size, float alpha, float beta,
                hipStream_t s) {
    const int64_t total = (int64_t)N * hw;
    hipLaunchKernelGGL(lrn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, N, C, hw,
                       local_size, alpha, beta);
}

// ------------------------------------------------------------------ Monte-Carlo reduction
// Softmax over the class axis (max-subtracted, fp32, as Caffe's SoftmaxLayer) of n samples,
// summed over the samples in f64 per pixel, written as the fp32 probability sum.
// VEC pixels per thread (2: 16 classes x 2 f64 sums fit the VGPR budget); every load is a
// coalesced VEC*4-byte access along the pixel axis.
template <int VEC, int CMAX>
__global__ void mc_reduce_kernel(const float *logits, int n, int C, int64_t hw, float *prob_sum, float *prob,
                                 int accumulate, int64_t chunk, double *prob_sum64) {
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (p >= hw) return;
    double sum[CMAX][VEC];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) sum[c][v] = 0.0;
    for (int s = 0; s < n; ++s) {
        const float *lp = logits + (int64_t)s * C * hw + p;
        float x[CMAX][VEC];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                if (VEC == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(lp + (int64_t)c * hw);
                    x[c][0] = t.x; x[c][1 % VEC] = t.y;
                } else {
                    x[c][0] = lp[(int64_t)c * hw];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float m = x[0][v];
#pragma unroll
            for (int c = 1; c < CMAX; ++c)
                if (c < C) m = x[c][v] > m ? x[c][v] : m;
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) { x[c][v] = softmax_exp(x[c][v] - m); den = __fadd_rn(den, x[c][v]); }
            const float rden = softmax_rcp(den);
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) { x[c][v] = __fmul_rn(x[c][v], rden); sum[c][v] += (double)x[c][v]; }
        }
        if (prob) {
            float *pp = prob + (int64_t)s * C * hw + p;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) {
                    if (VEC == 2) *reinterpret_cast<float2 *>(pp + (int64_t)c * hw) = make_float2(x[c][0], x[c][1 % VEC]);
                    else pp[(int64_t)c * hw] = x[c][0];
                }
        }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            // chunk == hw: [class][pixel]; otherwise pixel-chunk-major [pixel / chunk][class][pixel % chunk] (the layout a
            // reduce-scatter over pixel ranges needs; chunk is even when VEC == 2, so a pixel pair never straddles chunks)
            const int64_t at = ((p / chunk) * C + c) * chunk + (p % chunk);
            if (prob_sum64) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) prob_sum64[at + v] = accumulate ? prob_sum64[at + v] + sum[c][v] : sum[c][v];
            }
            if (!prob_sum) continue;
            float *dst = prob_sum + at;
            if (VEC == 2) {
                float2 o = make_float2((float)sum[c][0], (float)sum[c][1 % VEC]);
                if (accumulate) { const float2 old = *reinterpret_cast<float2 *>(dst); o.x += old.x; o.y += old.y; }
                *reinterpret_cast<float2 *>(dst) = o;
            } else {
                float o = (float)sum[c][0];
                if (accumulate) o += dst[0];
                dst[0] = o;
            }
        }
}

int launch_mc_reduce(const float *logits, int n, int C, int64_t hw, float *prob_sum, float *prob, int accumulate,
                     hipStream_t s, int64_t chunk, double *prob_sum64) {
    if (C > 16) return 1;
    if (chunk <= 0 || chunk > hw) chunk = hw;
    if ((hw & 1) == 0 && (chunk & 1) == 0) {
        const int64_t threads = hw / 2;
        hipLaunchKernelGGL((mc_reduce_kernel<2, 16>), dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, s, logits,
                           n, C, hw, prob_sum, prob, accumulate, chunk, prob_sum64);
    } else {
        hipLaunchKernelGGL((mc_reduce_kernel<1, 16>), dim3((unsigned)((hw + 127) / 128)), dim3(128), 0, s, logits, n, C,
                           hw, prob_sum, prob, accumulate, chunk, prob_sum64);
    }
    return 0;
}

// Single-device form of the whole Monte-Carlo post-processing (extractMeanConfidence + computeClasses +
// computeMaxConfidence + computeClassificationEntropy, bayesian_segnet.cpp:180-203, 262-297) in ONE pass over the logits:
// per pixel, softmax of each of the T samples (fp32, as the Softmax layer), the T float probabilities summed and divided
// by T in f64 exactly as the reference's f32 -> f64 cast + mean does, then argmax / max / entropy.  No probability sum
// goes through memory, so nothing is rounded to fp32 on the way.  One pixel per thread (360 k threads at 352 x 1024).
template <int CMAX>
__global__ __launch_bounds__(256) void mc_reduce_finalize_kernel(const float *__restrict__ logits, int T, int C, int64_t hw,
                                                                 uint8_t *classes, double *confidence, double *entropy) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    double sum[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) sum[c] = 0.0;
    for (int s = 0; s < T; ++s) {
        const float *lp = logits + (int64_t)s * C * hw + p;
        float x[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) x[c] = lp[(int64_t)c * hw];
        float m = x[0];
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c < C) m = x[c] > m ? x[c] : m;
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) { x[c] = softmax_exp(x[c] - m); den = __fadd_rn(den, x[c]); }
        const float rden = softmax_rcp(den);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) sum[c] += (double)__fmul_rn(x[c], rden);
    }
    const double dT = (double)T;
    int best = 0;
    double bv = sum[0] / dT;
    double ent = bv == 0 ? 0 : -1.0 * bv * log2(bv);
#pragma unroll
    for (int c = 1; c < CMAX; ++c)
        if (c < C) {
            const double v = sum[c] / dT;
            if (v > bv) { bv = v; best = c; }
            ent += v == 0 ? 0 : -1.0 * v * log2(v);
        }
    classes[p] = (uint8_t)best;
    confidence[p] = bv;
    entropy[p] = ent;
}
void launch_mc_reduce_finalize(const float *logits, int T, int C, int64_t hw, uint8_t *classes, double *confidence,
                               double *entropy, hipStream_t s) {
    hipLaunchKernelGGL((mc_reduce_finalize_kernel<16>), dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, logits, T, C, hw, classes,
                       confidence, entropy);
}

// mean = sum / T (f64); argmax with first-wins ties; max; entropy in bits with the
// exact-zero guard of computeEntropy (bayesian_segnet.cpp:38-44).
template <class SumT>
__global__ void mc_finalize_kernel(const SumT *prob_sum, int C, int64_t hw, int T, uint8_t *classes,
                                   double *confidence, double *entropy) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const double invT = (double)T;
    int best = 0;
    double bv = (double)prob_sum[p] / invT;
    double ent = bv == 0 ? 0 : -1.0 * bv * log2(bv);
    for (int c = 1; c < C; ++c) {
        const double v = (double)prob_sum[(int64_t)c * hw + p] / invT;
        if (v > bv) { bv = v; best = c; }
        ent += v == 0 ? 0 : -1.0 * v * log2(v);
    }
    if (classes) classes[p] = (uint8_t)best;
    if (confidence) confidence[p] = bv;
    if (entropy) entropy[p] = ent;
}
void launch_mc_finalize(const float *prob_sum, int C, int64_t hw, int T, uint8_t *classes, double *confidence,
                        double *entropy, hipStream_t s) {
    hipLaunchKernelGGL(mc_finalize_kernel<float>, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, prob_sum, C, hw, T,
                       classes, confidence, entropy);
}
// the same on f64 sums: with the sums of all samples exact to an f64 rounding, sum / T is the reference's f64 mean
// (bayesian_segnet.cpp:291-294) whatever the number of devices the samples were spread over
void launch_mc_finalize64(const double *prob_sum, int C, int64_t hw, int T, uint8_t *classes, double *confidence,
                          double *entropy, hipStream_t s) {
    hipLaunchKernelGGL(mc_finalize_kernel<double>, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, prob_sum, C, hw, T,
                       classes, confidence, entropy);
}
__global__ void add_f64_kernel(double *dst, const double *src, int64_t n, int init) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = init ? src[i] : dst[i] + src[i];
}
void launch_add_f64(double *dst, const double *src, int64_t n, bool init, hipStream_t s) {
    hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n, init ? 1 : 0);
}

// computeVariance (bayesian_segnet.cpp:205-260): sample variance over T of the winning class.
__global__ void mc_variance_kernel(const float *prob, int T, int C, int64_t hw, const uint8_t *classes,
                                   double *variance) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int c = classes[p];
    double avg = 0.0;
    for (int t = 0; t < T; ++t) avg += (double)prob[((int64_t)t * C + c) * hw + p];
    avg /= (double)T;
    double sum = 0.0;
    for (int t = 0; t < T; ++t) {
        const double d = (double)prob[((int64_t)t * C + c) * hw + p] - avg;
        sum += d * d;
    }
    variance[p] = sum / (double)(T - 1);
}
void launch_mc_variance(const float *prob, int T, int C, int64_t hw, const uint8_t *classes, double *variance,
                        hipStream_t s) {
    hipLaunchKernelGGL(mc_variance_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, s, prob, T, C, hw,
                       classes, variance);
}

// Pool-mask codes (u8) -> the flat input-plane index Caffe stores (as float), for sivo_segnet_blob.
__global__ void mask_to_index_kernel(const uint8_t *mask, float *out, int64_t total, int Ho, int Wo, int Win) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho);
    const int code = mask[i];
    out[i] = (float)((2 * ph + (code >> 1)) * Win + 2 * pw + (code & 1));
}
void launch_mask_to_index(const uint8_t *mask, float *out, int64_t total, int Ho, int Wo, int Win, hipStream_t s) {
    hipLaunchKernelGGL(mask_to_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mask, out, total,
                       Ho, Wo, Win);
}

// Row bands of the sample-invariant prefix (segnet.cpp PrefixBands).
// pack: the valid rows of a rank's band blobs -> its slot ([C][rows_max][W] per item, rows past the band's own count are padding);
// unpack: the slots of all ranks -> the full (C, H, W) blobs; rank r owns the rows [y0[r], y0[r + 1]) of an item.  The item flagged
// `drop` (the fork pooling's values) is not stored as it is: its per-sample dropout (the pooling kernel's own: same counter-based
// word per element and global sample) is applied on the way and the n per-sample copies are written.
// One launch each, one thread per 16 bytes of a row (W * elt % 16 == 0).
__global__ __launch_bounds__(256) void pack_bands_kernel(BandPack p, unsigned char *slot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = 0;
    while (k < p.n_items && i >= p.item[k].vecs) { i -= p.item[k].vecs; ++k; }
    if (k >= p.n_items) return;
    const BandPackItem &it = p.item[k];
    const int row_vec = it.W * it.elt / 16;
    const int v = (int)(i % row_vec);
    const int y = (int)((i / row_vec) % it.n_rows);
    const int c = (int)(i / ((int64_t)row_vec * it.n_rows));
    const unsigned char *src = it.src + ((size_t)((int64_t)c * it.src_H + it.row0 + y) * row_vec + v) * 16;
    *reinterpret_cast<uint4 *>(slot + it.off + ((size_t)((int64_t)c * it.rows_max + y) * row_vec + v) * 16) = *reinterpret_cast<const uint4 *>(src);
}
void launch_pack_bands(const BandPack &p, void *slot, hipStream_t s) {
    int64_t total = 0;
    for (int k = 0; k < p.n_items; ++k) total += p.item[k].vecs;
    if (total) hipLaunchKernelGGL(pack_bands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, (unsigned char *)slot);
}

__global__ __launch_bounds__(256) void unpack_bands_kernel(BandUnpack u, const unsigned char *slots) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = 0;
    while (k < u.n_items && i >= u.item[k].vecs) { i -= u.item[k].vecs; ++k; }
    if (k >= u.n_items) return;
    const BandUnpackItem &it = u.item[k];
    const int row_vec = it.W * it.elt / 16;
    const int v = (int)(i % row_vec);
    const int y = (int)((i / row_vec) % it.H);
    const int c = (int)(i / ((int64_t)row_vec * it.H));
    int r = 0;
    while (r + 1 < u.world && y >= it.y0[r + 1]) ++r;
    const uint4 q = *reinterpret_cast<const uint4 *>(slots + (size_t)r * u.slot_bytes + it.off + ((size_t)((int64_t)c * it.rows_max + (y - it.y0[r])) * row_vec + v) * 16);
    if (!it.drop) {
        *reinterpret_cast<uint4 *>(it.dst + (size_t)i * 16) = q;
        return;
    }
    // four consecutive fp32 elements e .. e + 3 of the (C, H, W) sample; out[n] = dropout of the SAME values with sample n's bits
    const int64_t chw = (int64_t)it.C * it.H * it.W;
    const uint32_t e = (uint32_t)(i * 4);
    const float x[4] = {__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
    for (int n = 0; n < u.n; ++n) {
        float o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t w = dropout_word(e + t, (uint32_t)u.site, (uint32_t)(u.sample0 + n), u.seed);
            o[t] = ((w >> ((e + t) & 31)) & 1u) ? x[t] * 2.f : 0.f;
        }
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(it.dst) + (int64_t)n * chw + e) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
void launch_unpack_bands(const BandUnpack &u, const void *slots, hipStream_t s) {
    int64_t total = 0;
    for (int k = 0; k < u.n_items; ++k) total += u.item[k].vecs;
    if (total) hipLaunchKernelGGL(unpack_bands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, u, (const unsigned char *)slots);
}


}  // namespace sivo

// ------------------------------------------------------------------ diagnostics
