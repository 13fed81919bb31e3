// conv_v2.hip — second-generation fp32-MFMA convolution (same math and tiling as
// conv_mfma_kernel in segnet_kernels.hip; see its header for the fragment mapping).
//
// What changed, and why (measured with tools/conv_probe.py on MI355X: the MFMA + LDS-read loop
// alone sustains 146 TFLOP/s, the v1 kernel 124 — the difference was the global->LDS staging):
//   * K-chunk of 4 input channels, LDS DOUBLE BUFFERED (2 x ~24.5 KB per workgroup, 3 workgroups
//     per CU): chunk k+1 is staged while chunk k feeds the matrix cores; ONE barrier per chunk
//     and no separate commit phase.
//   * Weight slab by LDS-DMA: the slab of a (chunk, Cout-tile) is stored in HBM already in its
//     LDS image (row stride padded to 16 mod 32 dwords, slab padded to whole KiB), so
//     `global_load_lds_dwordx4` copies it linearly with no VGPRs and no ds_write.
//   * Input halo patch by 16-byte loads: the interior of a patch row is 16-byte aligned in HBM
//     (x0 % 32 == 0) and in LDS (interior at dword offset 4 of a 40-dword row), one float4 per
//     thread; the 2*(k/2) halo columns per row are scalar.  v1 used one scalar load per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "segnet_kernels.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void philox4x32_10_v2(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ uint32_t dropout_word_v2(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
    uint32_t w[4];
    philox4x32_10_v2(e >> 7, site, sample, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const uint32_t sel = (e >> 5) & 3u;
    return sel == 0 ? w[0] : sel == 1 ? w[1] : sel == 2 ? w[2] : w[3];
}

constexpr int c2_pad16mod32(int n) { return n + ((16 - (n % 32)) + 32) % 32; }
constexpr int c2_bnp(int bn) { return (bn % 32 == 0) ? bn + 16 : bn; }
constexpr int c2_slab(int ks, int bn) { return ((ks * ks * 4 * c2_bnp(bn)) + 255) / 256 * 256; }   // floats, whole KiB

template <int KS, int TH, int TW, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 ? 4 : 3)) void conv_mfma2_kernel(ConvArgs a) {
    constexpr int KC = 4;
    constexpr int NTHR = WM * WN * 64, NWAVE = WM * WN;
    constexpr int HALO = KS / 2;
    constexpr int OFF = HALO > 0 ? 4 : 0;                // dword offset of the interior inside an LDS patch row
    constexpr int PH = TH + KS - 1, PWp = TW + (HALO > 0 ? 8 : 0);
    constexpr int CS = c2_pad16mod32(PH * PWp);          // channel stride (dwords), multiple of 4
    constexpr int BNP = c2_bnp(BN);
    constexpr int WSLAB = c2_slab(KS, BN);
    constexpr int PATCH = (KC * CS + 3) / 4 * 4;
    constexpr int BUF = PATCH + WSLAB;                   // dwords per LDS buffer
    constexpr int MTB = TH * (TW / 16);
    constexpr int MT = MTB / WM, NT = (BN / 16) / WN;
    static_assert((NWAVE == 4 || NWAVE == 8) && MTB % WM == 0 && (BN / 16) % WN == 0, "tile split");
    static_assert(CS % 4 == 0 && PWp % 4 == 0, "16-byte aligned patch rows");

    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lk = lane >> 4;

    // XCD-aware tile order (see conv_wino.hip): workgroups of one XCD walk the Cout tiles of a pixel tile
    const int ntiles = a.CoutPad / BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = slot % ntiles;
    int bid = (slot / ntiles) * 8 + xcd;
    if (bid >= a.tiles_x * a.tiles_y * a.N) return;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = ntile * BN;

    const float *in_n = a.in + (int64_t)n * a.in_sample_stride;
    const int64_t plane = (int64_t)a.H * a.W;

    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int g = wm * MT + mt;
        const int row = g / (TW / 16), col = (g % (TW / 16)) * 16;
        a_base[mt] = lk * CS + row * PWp + col + li + (OFF - HALO);
    }
    const int b_base = PATCH + lk * BNP + wn * NT * 16 + li;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging plan of this thread (chunk invariant)
    constexpr int NV4 = KC * PH * (TW / 4), V4IT = (NV4 + NTHR - 1) / NTHR;       // interior float4s
    constexpr int H2 = HALO > 0 ? 2 * HALO : 1;
    constexpr int NSC = KC * PH * 2 * HALO, SCIT = (NSC + NTHR - 1) / NTHR;       // halo scalars
    const bool vec_ok = (a.W & 3) == 0;
    int v_goff[V4IT], v_dst[V4IT];     // v_dst: LDS dword offset | c << 24, or -1; v_goff: offset of element 0, or -1 if row out of image
    int v_nvalid[V4IT];                // how many of the 4 columns are inside the image
#pragma unroll
    for (int it = 0; it < V4IT; ++it) {
        const int idx = tid + it * NTHR;
        const int seg = idx % (TW / 4), r = idx / (TW / 4);
        const int py = r % PH, c = r / PH;
        const int gy = y0 + py - HALO, gx = x0 + seg * 4;
        const bool rowok = idx < NV4 && gy >= 0 && gy < a.H;
        v_nvalid[it] = rowok ? min(max(a.W - gx, 0), 4) : 0;
        v_goff[it] = (int)(c * plane + (int64_t)(rowok ? gy : 0) * a.W + min(gx, a.W - 1));
        v_dst[it] = idx < NV4 ? ((c * CS + py * PWp + OFF + seg * 4) | (c << 24)) : -1;
    }
    int s_goff[SCIT > 0 ? SCIT : 1], s_dst[SCIT > 0 ? SCIT : 1];
#pragma unroll
    for (int it = 0; it < SCIT; ++it) {
        const int idx = tid + it * NTHR;
        const int h = idx % H2, r = idx / H2;
        const int py = r % PH, c = r / PH;
        const int px = h < HALO ? OFF - HALO + h : OFF + TW + (h - HALO);   // LDS column
        const int gy = y0 + py - HALO, gx = x0 + px - OFF;
        const bool ok = idx < NSC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        s_goff[it] = ok ? (int)(c * plane + (int64_t)gy * a.W + gx) : -1;
        s_dst[it] = idx < NSC ? ((c * CS + py * PWp + px) | (c << 24)) : -1;
    }
    f32x4 pv4[V4IT];
    float psc[SCIT > 0 ? SCIT : 1];
    const int nchunks = (a.Cin + KC - 1) / KC;

    auto issue_patch = [&](int chunk) {
        const float *psrc = in_n + (int64_t)chunk * KC * plane;
        const int cleft = a.Cin - chunk * KC;
#pragma unroll
        for (int it = 0; it < V4IT; ++it) {
            const bool chok = (v_dst[it] >> 24) < cleft && v_dst[it] >= 0;
            const int nv = chok ? v_nvalid[it] : 0;
            if (vec_ok) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(psrc + (nv == 4 ? v_goff[it] : 0));
                pv4[it] = nv == 4 ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float t = psrc[j < nv ? v_goff[it] + j : 0]; v[j] = j < nv ? t : 0.f; }
                pv4[it] = v;
            }
        }
#pragma unroll
        for (int it = 0; it < SCIT; ++it) {
            const bool ok = s_goff[it] >= 0 && (s_dst[it] >> 24) < cleft;
            const float v = psrc[ok ? s_goff[it] : 0];
            psc[it] = ok ? v : 0.f;
        }
    };
    auto commit_patch = [&](int buf) {
        float *sp = lds + buf * BUF;
#pragma unroll
        for (int it = 0; it < V4IT; ++it)
            if (v_dst[it] >= 0) *reinterpret_cast<f32x4 *>(sp + (v_dst[it] & 0xffffff)) = pv4[it];
#pragma unroll
        for (int it = 0; it < SCIT; ++it)
            if (s_dst[it] >= 0) sp[s_dst[it] & 0xffffff] = psc[it];
    };
    // weight slab: linear LDS-DMA, 1 KiB per wave instruction; instruction i of a wave moves KiB i*NWAVE + wave
    constexpr int NDMA = (WSLAB / 256 + NWAVE - 1) / NWAVE;
    auto dma_one = [&](int chunk, int buf, int i) {
        const float *wsrc = a.wt + ((int64_t)chunk * ntiles + ntile) * WSLAB;
        float *dst = lds + buf * BUF + PATCH;
        const int kib = i * NWAVE + wave;
        if (kib < WSLAB / 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + kib * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(dst + kib * 256), 16, 0, 0);
    };

    // ---- prologue: chunk 0 into buffer 0
    issue_patch(0);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_one(0, 0, i);
    commit_patch(0);
    __syncthreads();

    // Staging instructions of chunk k+1 are spread through the MFMA stream of chunk k (one LDS-DMA
    // per tap, patch loads early, patch commit late): a burst at the chunk start costs ~60 issue
    // cycles per DMA with nothing to overlap, spread out each one hides behind 16 MFMAs.
    constexpr int TAPS = KS * KS;
    constexpr int T_LOAD = TAPS > 1 ? 1 : 0, T_COMMIT = TAPS > 2 ? TAPS - 2 : TAPS - 1;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        const int nxt = more ? chunk + 1 : chunk;
        const float *sp = lds + cur * BUF;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
            float af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = sp[a_base[mt] + dy * PWp + dx];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = sp[b_base + tap * KC * BNP + nt * 16];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
            if (tap == T_LOAD && !(a.variant & 2)) issue_patch(nxt);   // unconditional (re-reads the last chunk once)
            if (more && !(a.variant & 4)) {
                if (TAPS >= NDMA) { if (tap < NDMA) dma_one(chunk + 1, cur ^ 1, tap); }
                else if (tap == 0) {
#pragma unroll
                    for (int i = 0; i < NDMA; ++i) dma_one(chunk + 1, cur ^ 1, i);
                }
            }
            if (tap == T_COMMIT && more && !(a.variant & 2)) commit_patch(cur ^ 1);   // the other buffer: nobody reads it this chunk
        }
        __syncthreads();
    }

    // ---- epilogue (identical to v1)
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
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
                    const uint32_t w = dropout_word_v2(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? v[r] * 2.f : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t w = dropout_word_v2(e + r, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed);
                        v[r] = ((w >> ((e + r) & 31)) & 1u) ? v[r] * 2.f : 0.f;
                    }
                }
            }
            float *dst = out_n + (int64_t)co * plane + (int64_t)y * a.W + x;
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

template <int KS, int TH, int TW, int BN, int WM, int WN>
static void launch2_cfg(const ConvArgs &a0, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const int ptiles = a.tiles_x * a.tiles_y * a.N;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8 * (a.CoutPad / BN)));
    hipLaunchKernelGGL((conv_mfma2_kernel<KS, TH, TW, BN, WM, WN>), grid, dim3(WM * WN * 64), 0, s, a);
}

int conv2_slab_floats(int ks, int cout) {
    const int bn = conv_cout_tile(ks, cout);
    return ((ks * ks * 4 * c2_bnp(bn)) + 255) / 256 * 256;
}

// Caffe (Cout,Cin,k,k) -> [ceil(Cin/4)][CoutPad/BN][slab], slab row (tap*4 + c%4) has stride BNP.
void conv2_pack_weights(const float *W, int ks, int cin, int cout, std::vector<float> &out, int *cout_pad) {
    const int bn = conv_cout_tile(ks, cout), bnp = c2_bnp(bn), slab = conv2_slab_floats(ks, cout);
    const int ntiles = (cout + bn - 1) / bn, nchunks = (cin + 3) / 4, taps = ks * ks;
    *cout_pad = ntiles * bn;
    out.assign((size_t)nchunks * ntiles * slab, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) {
                const size_t base = ((size_t)(ci / 4) * ntiles + co / bn) * slab;
                out[base + (size_t)(t * 4 + ci % 4) * bnp + co % bn] = W[((size_t)co * cin + ci) * taps + t];
            }
}

bool conv2_supported(int ks) { return ks == 3 || ks == 1; }   // 7x7 slabs do not fit two LDS buffers: v1 handles them

void launch_conv2(const ConvArgs &a, int ks, hipStream_t s) {
    const int bn = conv_cout_tile(ks, a.Cout);
    if (ks == 3) {
        if (bn == 16) return launch2_cfg<3, 8, 32, 16, 4, 1>(a, s);
        if (bn == 64) return launch2_cfg<3, 8, 32, 64, 4, 1>(a, s);
        if (a.variant & 32) return launch2_cfg<3, 8, 32, 128, 4, 2>(a, s);
        return launch2_cfg<3, 4, 32, 128, 2, 2>(a, s);
    }
    if (bn == 16) return launch2_cfg<1, 8, 32, 16, 4, 1>(a, s);
    if (bn == 64) return launch2_cfg<1, 8, 32, 64, 4, 1>(a, s);
    return launch2_cfg<1, 4, 32, 128, 2, 2>(a, s);
}

}  // namespace sivo
