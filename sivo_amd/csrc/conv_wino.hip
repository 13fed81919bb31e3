// conv_wino.hip — 3x3 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (CVPR 2016):
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 2x2 output tile, 4x4 input tile d, 3x3 filter g,
// summed over input channels BEFORE the output transform, i.e. for each of the 16 transform
// positions p an independent GEMM  M_p[tile][co] = sum_ci V_p[tile][ci] * U_p[ci][co]:
// 16 multiplies per 4 outputs instead of 36 (2.25x fewer MFMA flops), every operation in fp32
// (the transforms are +-, *0.5; measured |dlogit| stays ~1e-5, budget 1e-3).
//
// Mapping onto v_mfma_f32_16x16x4_f32 (A = 16 tiles x 4 channels, B = 4 channels x 16 couts):
//   * an m-tile = 16 horizontally consecutive 2x2 tiles (32 x 2 output pixels);
//   * lane l owns (tile i = l & 15, channel c0 + (l >> 4)): it reads that tile's 4x4 input patch
//     from the LDS halo patch (16 ds_read_b32), applies B^T d B in registers (32 adds) and so
//     holds exactly the A operand of all 16 positions — no cross-lane traffic;
//   * B operand of position p: U slab row (p*4 + (l >> 4)), column n0 + (l & 15);
//   * accumulators acc[p][nt]: lane holds, for its 4 tiles (rows 4*(l>>4)+r) and its cout, all 16
//     positions, so the output transform A^T M A is lane-local too (24 adds per tile), followed
//     by the fused epilogue (bias+BN affine, ReLU, Philox dropout) and two float4 stores per row.
// Staging is the v2 scheme (conv_v2.hip): K-chunks of 4 channels, double-buffered LDS, the
// pre-transformed weight slab (16 x 4 x BN, LDS image) by LDS-DMA, the patch by 16-byte loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <vector>

#include "lds_dma.hpp"
#include "common.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wino_dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
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

constexpr int wino_bnp(int bn) { return bn == 64 ? 80 : bn; }   // 64: 80-dword rows with odd rows shifted by 16; 32: dense rows (both conflict-free for ds_read_b64)
constexpr int wino_slab(int bn, int kc) { return (16 * kc * wino_bnp(bn) + 255) / 256 * 256; }   // floats, whole KiB

// Workgroup: WM m-tiles stacked vertically (2*WM output rows x 32 output columns) x BN = WN*NT*16 couts.
// ABL: ablation switches for tools/conv_probe.py only (0 in production): 1 no staging after the prologue,
// 2 additionally no barrier, 4 additionally no input transform (V = raw patch), 8 additionally B from a register.
// UNPOOL: `a.in` is the pooled tensor of an Upsample (scale 2) layer and `a.unpool_mask` its window codes; the patch
// loader reads 2 pooled values + 2 codes where it would read 4 unpooled pixels (the unpooled tensor never exists).
template <int WM, int WN, int NT, int KC, int ABL = 0, bool UNPOOL = false>
__global__ __launch_bounds__(WM * WN * 64, ((KC == 4 || WM * WN == 12) ? 3 : 2)) void conv_wino_kernel(ConvArgs a) {
    constexpr int NTHR = WM * WN * 64, NWAVE = WM * WN;
    constexpr int BN = WN * NT * 16, BNP = wino_bnp(BN);
    constexpr int TH = 2 * WM, TW = 32;
    // Patch rows are stored DE-INTERLEAVED: with q = x - x0 + 1 in [0, 34), odd q go to O[q/2] at row offset 0..16 and
    // even q to E[q/2] at row offset 19..35.  The 4x4 patch of tile t is then E[t], O[t], E[t+1], O[t+1] per row:
    // the 16 lanes of an m-tile read CONSECUTIVE dwords (an interleaved row gives stride 2 = 2-way bank conflicts on
    // every one of the 16 patch reads, 25 % of the LDS time of this LDS-bound kernel).  Channel stride = 16 (mod 32).
    constexpr int PH = TH + 2, PWp = 36, EOFF = 19;
    constexpr int CS = PH * PWp + ((16 - (PH * PWp) % 32) + 32) % 32;
    constexpr int WSLAB = wino_slab(BN, KC);
    constexpr int PATCH = KC * CS;
    constexpr int BUF = PATCH + WSLAB;
    static_assert((NWAVE == 4 || NWAVE == 12) && NT == 2, "4 or 12 waves, paired n-tiles");

    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lk = lane >> 4;

    // XCD-aware tile order: workgroup L runs on XCD L % 8 (observed dispatch rule; speed only, never
    // correctness).  Inside an XCD consecutive workgroups walk the Cout tiles of ONE pixel tile, so the
    // input patch chunks are fetched once into that XCD's L2 and hit there for the other Cout tiles
    // (before: re-fetched once per Cout tile — 3-6x the algorithmic input traffic in the PMC counters).
    const int ntiles = a.CoutPad / BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = slot % ntiles;
    int bid = (slot / ntiles) * 8 + xcd;                    // pixel-tile index
    if (bid >= a.tiles_x * a.tiles_y * a.N) return;         // grid is padded to a multiple of 8 pixel tiles
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = ntile * BN;

    const float *in_n = a.in + (int64_t)n * a.in_sample_stride;
    const int64_t plane = (int64_t)a.H * a.W;
    // UNPOOL: geometry of the pooled source (H/2 x W/2) and the codes of this sample
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;
    const uint8_t *mk_n = UNPOOL ? a.unpool_mask + (int64_t)n * a.unpool_mask_stride : nullptr;

    // 4x4 input patch of tile (row wm, column li): LDS rows 2*wm .. 2*wm+3; columns q = 2*li .. 2*li+3
    const int a_base = lk * CS + (2 * wm) * PWp + li;
    // B pair: the two 16-cout n-tiles of this wave's 32 couts are interleaved in the slab (one ds_read_b64 per position)
    // (odd slab rows are shifted by 16 dwords into the row padding: the two rows a 32-lane half reads then sit on
    //  disjoint bank halves of the 64-bank ds_read_b64 — conflict-free with the 80-dword stride)
    const int b_base = PATCH + lk * BNP + (BN == 64 ? (lk & 1) * 16 : 0) + wn * 32 + 2 * li;

    f32x4 acc[16][NT];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[p][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging plan (as conv_v2): interior float4s + 2 halo scalars per patch row
    constexpr int NV4 = KC * PH * (TW / 4), V4IT = (NV4 + NTHR - 1) / NTHR;
    constexpr int NSC = KC * PH * 2, SCIT = (NSC + NTHR - 1) / NTHR;
    int v_goff[V4IT], v_dst[V4IT];
    bool v_ok[V4IT];
#pragma unroll
    for (int it = 0; it < V4IT; ++it) {
        const int idx = tid + it * NTHR;
        const int seg = idx % (TW / 4), r = idx / (TW / 4);
        const int py = r % PH, c = r / PH;
        const int gy = y0 + py - 1, gx = x0 + seg * 4;
        v_ok[it] = idx < NV4 && gy >= 0 && gy < a.H && gx + 3 < a.W;
        if (UNPOOL) v_goff[it] = v_ok[it] ? (int)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) | ((gy & 1) << 30) : 0;   // bit 30: window row
        else v_goff[it] = v_ok[it] ? (int)(c * plane + (int64_t)gy * a.W + gx) : 0;
        // pixels x0+4s..+3 are q = 4s+1..4s+4: (v0, v2) -> O[2s], O[2s+1]; (v1, v3) -> E[2s+1], E[2s+2]
        v_dst[it] = idx < NV4 ? ((c * CS + py * PWp + 2 * seg) | (c << 24)) : -1;
    }
    int s_goff[SCIT], s_dst[SCIT];
#pragma unroll
    for (int it = 0; it < SCIT; ++it) {
        const int idx = tid + it * NTHR;
        const int h = idx % 2, r = idx / 2;
        const int py = r % PH, c = r / PH;
        const int px = h == 0 ? EOFF : 16;            // x = x0-1 is q = 0 -> E[0]; x = x0+32 is q = 33 -> O[16]
        const int gy = y0 + py - 1, gx = h == 0 ? x0 - 1 : x0 + TW;
        const bool ok = idx < NSC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if (UNPOOL) s_goff[it] = ok ? (int)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) | ((gy & 1) << 30) | ((gx & 1) << 29) : -1;
        else s_goff[it] = ok ? (int)(c * plane + (int64_t)gy * a.W + gx) : -1;
        s_dst[it] = idx < NSC ? ((c * CS + py * PWp + px) | (c << 24)) : -1;
    }
    f32x4 pv4[V4IT];
    float psc[SCIT];
    const int nchunks = (a.Cin + KC - 1) / KC;

    auto issue_patch = [&](int chunk) {
        const float *psrc = in_n + (int64_t)chunk * KC * plane_in;
        const int cleft = a.Cin - chunk * KC;
        if (UNPOOL) {
            const uint8_t *msrc = mk_n + (int64_t)chunk * KC * plane_in;
#pragma unroll
            for (int it = 0; it < V4IT; ++it) {
                const bool ok = v_ok[it] && (v_dst[it] >> 24) < cleft;
                const int off = ok ? (v_goff[it] & 0x1fffffff) : 0, code0 = (v_goff[it] >> 30) << 1;   // window row * 2
                const float2 v = *reinterpret_cast<const float2 *>(psrc + off);                          // gx % 4 == 0: 8-byte aligned
                const uchar2 m = *reinterpret_cast<const uchar2 *>(msrc + off);
                pv4[it] = ok ? (f32x4){m.x == code0 ? v.x : 0.f, m.x == code0 + 1 ? v.x : 0.f, m.y == code0 ? v.y : 0.f, m.y == code0 + 1 ? v.y : 0.f}
                             : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < SCIT; ++it) {
                const bool ok = s_goff[it] >= 0 && (s_dst[it] >> 24) < cleft;
                const int off = ok ? (s_goff[it] & 0x1fffffff) : 0, code = ((s_goff[it] >> 30) & 1) * 2 + ((s_goff[it] >> 29) & 1);
                const float v = psrc[off];
                const int m = msrc[off];
                psc[it] = (ok && m == code) ? v : 0.f;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < V4IT; ++it) {
            const bool ok = v_ok[it] && (v_dst[it] >> 24) < cleft;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(psrc + (ok ? v_goff[it] : 0));
            pv4[it] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
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
            if (v_dst[it] >= 0) {
                float *q = sp + (v_dst[it] & 0xffffff);
                *reinterpret_cast<float2 *>(q) = make_float2(pv4[it][0], pv4[it][2]);                 // O[2s], O[2s+1]
                *reinterpret_cast<float2 *>(q + EOFF + 1) = make_float2(pv4[it][1], pv4[it][3]);      // E[2s+1], E[2s+2]
            }
#pragma unroll
        for (int it = 0; it < SCIT; ++it)
            if (s_dst[it] >= 0) sp[s_dst[it] & 0xffffff] = psc[it];
    };
    constexpr int NDMA = (WSLAB / 256 + NWAVE - 1) / NWAVE;
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + ((int64_t)chunk * ntiles + ntile) * WSLAB;
        float *dst = lds + buf * BUF + PATCH;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int kib = i * NWAVE + wave;
            if (kib < WSLAB / 256)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + kib * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + kib * 256), 16, 0, 0);
        }
    };

    issue_patch(0);
    dma_weights(0, 0);
    commit_patch(0);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        if (!(ABL & 1)) issue_patch(more ? chunk + 1 : chunk);
        if (more && !(ABL & 1)) dma_weights(chunk + 1, cur ^ 1);
        const float *sp = lds + cur * BUF;
#pragma unroll
        for (int c4 = 0; c4 < KC / 4; ++c4) {
            // ---- input transform V = B^T d B of this lane's (tile, channel c4*4 + lk)
            float d[4][4], t[4][4], V[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = sp[a_base + c4 * 4 * CS + r * PWp + ((c & 1) ? 0 : EOFF) + (c >> 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
            if (ABL & 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[r * 4 + c] = d[r][c];
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[r * 4 + 0] = t[r][0] - t[r][2];
                V[r * 4 + 1] = t[r][1] + t[r][2];
                V[r * 4 + 2] = t[r][2] - t[r][1];
                V[r * 4 + 3] = t[r][1] - t[r][3];
            }
            // ---- 16 positions x 2 MFMAs
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float2 bf = (ABL & 8) ? make_float2(V[(p + 1) & 15], V[(p + 2) & 15]) : *reinterpret_cast<const float2 *>(sp + b_base + (p * KC + c4 * 4) * BNP);
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf.x, acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf.y, acc[p][1], 0, 0, 0);
                if (c4 == KC / 4 - 1 && p == 10 && more && !(ABL & 1)) commit_patch(cur ^ 1);
            }
        }
        if (!(ABL & 2)) __syncthreads();
    }

    // ---- output transform Y = A^T M A (lane-local) + epilogue
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
    const int yb = y0 + 2 * wm;
    const int xb = x0 + 8 * lk;                 // this lane's 4 tiles = 8 consecutive output columns
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + (wn * NT + nt) * 16 + li;
        if (co >= a.Cout) continue;
        const float sc = a.ep_scale[co], sh = a.ep_shift[co];
        float y[2][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[0 + c][nt][r], m1 = acc[4 + c][nt][r], m2 = acc[8 + c][nt][r], m3 = acc[12 + c][nt][r];
                s[0][c] = m0 + m1 + m2;
                s[1][c] = m1 - m2 - m3;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[i][2 * r + 0] = s[i][0] + s[i][1] + s[i][2];
                y[i][2 * r + 1] = s[i][1] - s[i][2] - s[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = yb + i;
            if (yy >= a.H || xb >= a.W) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = y[i][j] * sc + sh;
                if (a.relu) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            const uint32_t e = (uint32_t)((co * a.H + yy) * a.W + xb);
            if (a.drop_site >= 0) {
                const uint32_t w = wino_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((w >> j) & 1u) ? v[j] * 2.f : 0.f;
            }
            float *dst = out_n + (int64_t)co * plane + (int64_t)yy * a.W + xb;
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel with its memory traffic in flight during the matrix-core phase (see lds_dma.hpp and conv_wino4f.hip):
// behind a builtin LDS-DMA hipcc drains vmcnt(0) in front of the first LDS read, i.e. at the top of every K-chunk a wave
// waits for what it has just requested for the next one.  Here the weight DMA goes through inline assembly, the patch
// comes by BUFFER loads (an offset beyond the descriptor returns 0 = the zero padding outside the image and beyond Cin; no
// select, no branch, the same number of vector-memory instructions in every wave), and iteration c writes the patch of
// chunk c + 1 (requested a whole iteration ago) to LDS, starts the DMA of its weights, requests the patch of chunk c + 2,
// computes chunk c and then waits with `s_waitcnt vmcnt(NL)` (NL = the loads issued behind the DMA: it has landed, they
// stay in flight) in front of a barrier without the vmcnt(0) drain of __syncthreads().
template <int WM, int WN, int NT, int KC, bool UNPOOL>
__global__ __launch_bounds__(WM * WN * 64, ((KC == 4 || WM * WN == 12) ? 3 : 2)) void conv_wino_p_kernel(ConvArgs a) {
    constexpr int NTHR = WM * WN * 64, NWAVE = WM * WN;
    constexpr int BN = WN * NT * 16, BNP = wino_bnp(BN);
    constexpr int TH = 2 * WM, TW = 32;
    constexpr int PH = TH + 2, PWp = 36, EOFF = 19;
    constexpr int CS = PH * PWp + ((16 - (PH * PWp) % 32) + 32) % 32;
    constexpr int WSLAB = wino_slab(BN, KC);
    constexpr int PATCH = KC * CS;
    constexpr int BUF = PATCH + WSLAB;
    static_assert((NWAVE == 4 || NWAVE == 12) && NT == 2, "4 or 12 waves, paired n-tiles");

    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lk = lane >> 4;

    const int ntiles = a.CoutPad / BN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ntile = slot % ntiles;
    int bid = (slot / ntiles) * 8 + xcd;                    // pixel-tile index
    if (bid >= a.tiles_x * a.tiles_y * a.N) return;         // grid is padded to a multiple of 8 pixel tiles
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = ntile * BN;

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

    const int a_base = lk * CS + (2 * wm) * PWp + li;
    const int b_base = PATCH + lk * BNP + (BN == 64 ? (lk & 1) * 16 : 0) + wn * 32 + 2 * li;

    f32x4 acc[16][NT];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[p][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging plan: interior float4s + 2 halo scalars per patch row.  v_idx / s_idx: element index inside the chunk's KC
    // planes (UNPOOL: of the pooled plane), INV when the item has no source
    constexpr int NV4 = KC * PH * (TW / 4), V4IT = (NV4 + NTHR - 1) / NTHR;
    constexpr int NSC = KC * PH * 2, SCIT = (NSC + NTHR - 1) / NTHR;
    uint32_t v_idx[V4IT], s_idx[SCIT];
    int v_dst[V4IT], s_dst[SCIT], v_code0[V4IT], s_code[SCIT];
#pragma unroll
    for (int it = 0; it < V4IT; ++it) {
        const int idx = tid + it * NTHR;
        const int seg = idx % (TW / 4), r = idx / (TW / 4);
        const int py = r % PH, c = r / PH;
        const int gy = y0 + py - 1, gx = x0 + seg * 4;
        const bool ok = idx < NV4 && gy >= 0 && gy < a.H && gx + 3 < a.W;
        v_idx[it] = !ok ? INV : UNPOOL ? (uint32_t)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(c * plane + (int64_t)gy * a.W + gx);
        v_code0[it] = (gy & 1) << 1;
        v_dst[it] = idx < NV4 ? (c * CS + py * PWp + 2 * seg) : -1;
    }
#pragma unroll
    for (int it = 0; it < SCIT; ++it) {
        const int idx = tid + it * NTHR;
        const int h = idx % 2, r = idx / 2;
        const int py = r % PH, c = r / PH;
        const int px = h == 0 ? EOFF : 16;            // x = x0-1 is q = 0 -> E[0]; x = x0+32 is q = 33 -> O[16]
        const int gy = y0 + py - 1, gx = h == 0 ? x0 - 1 : x0 + TW;
        const bool ok = idx < NSC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        s_idx[it] = !ok ? INV : UNPOOL ? (uint32_t)(c * plane_in + (int64_t)(gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(c * plane + (int64_t)gy * a.W + gx);
        s_code[it] = (gy & 1) * 2 + (gx & 1);
        s_dst[it] = idx < NSC ? (c * CS + py * PWp + px) : -1;
    }
    constexpr int NL = (UNPOOL ? 2 : 1) * (V4IT + SCIT);     // vector-memory loads per issue_patch, in every wave
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x4 pv4[V4IT];
    unsigned psc[SCIT], pvm[V4IT], psm[SCIT];
#pragma unroll
    for (int it = 0; it < V4IT; ++it) { pv4[it] = (u32x4){0u, 0u, 0u, 0u}; pvm[it] = 0; }
#pragma unroll
    for (int it = 0; it < SCIT; ++it) { psc[it] = 0; psm[it] = 0; }
    const int nchunks = (a.Cin + KC - 1) / KC;

    auto issue_patch = [&](int chunk) {
        const uint32_t cb = (uint32_t)(chunk * KC * plane_in);
#pragma unroll
        for (int it = 0; it < V4IT; ++it) {
            const uint32_t vi = v_idx[it] == INV ? INV : v_idx[it] + cb;
            if (UNPOOL) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)(vi == INV ? INV : vi * 4), 0, 0);
                pv4[it][0] = v[0]; pv4[it][1] = v[1];
                pvm[it] = __builtin_amdgcn_raw_buffer_load_b16(mk_rsrc, (int)vi, 0, 0);
            } else {
                pv4[it] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(vi == INV ? INV : vi * 4), 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < SCIT; ++it) {
            const uint32_t si = s_idx[it] == INV ? INV : s_idx[it] + cb;
            psc[it] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(si == INV ? INV : si * 4), 0, 0);
            if (UNPOOL) psm[it] = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc, (int)si, 0, 0);
        }
    };
    auto commit_patch = [&](int buf) {
        float *sp = lds + buf * BUF;
#pragma unroll
        for (int it = 0; it < V4IT; ++it)
            if (v_dst[it] >= 0) {
                float v0, v1, v2, v3;
                if (UNPOOL) {
                    const float vx = __uint_as_float(pv4[it][0]), vy = __uint_as_float(pv4[it][1]);
                    const int mx = (int)(pvm[it] & 0xffu), my = (int)((pvm[it] >> 8) & 0xffu), c0 = v_code0[it];
                    v0 = mx == c0 ? vx : 0.f; v1 = mx == c0 + 1 ? vx : 0.f; v2 = my == c0 ? vy : 0.f; v3 = my == c0 + 1 ? vy : 0.f;
                } else {
                    v0 = __uint_as_float(pv4[it][0]); v1 = __uint_as_float(pv4[it][1]); v2 = __uint_as_float(pv4[it][2]); v3 = __uint_as_float(pv4[it][3]);
                }
                float *q = sp + v_dst[it];
                *reinterpret_cast<float2 *>(q) = make_float2(v0, v2);                 // O[2s], O[2s+1]
                *reinterpret_cast<float2 *>(q + EOFF + 1) = make_float2(v1, v3);      // E[2s+1], E[2s+2]
            }
#pragma unroll
        for (int it = 0; it < SCIT; ++it)
            if (s_dst[it] >= 0) sp[s_dst[it]] = (!UNPOOL || (int)(psm[it] & 0xffu) == s_code[it]) ? __uint_as_float(psc[it]) : 0.f;
        // every wave (also one without a staging item) is done with the registers here: the compiler's own wait for these
        // loads comes now, before the DMA it cannot see is in flight, and not at a later reuse of the registers
#pragma unroll
        for (int it = 0; it < V4IT; ++it) asm volatile("" ::"v"(pv4[it]), "v"(pvm[it]));
#pragma unroll
        for (int it = 0; it < SCIT; ++it) asm volatile("" ::"v"(psc[it]), "v"(psm[it]));
    };
    constexpr int NDMA = (WSLAB / 256 + NWAVE - 1) / NWAVE;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t slab_lds = lds_addr_uniform(lds + PATCH);
    auto dma_weights = [&](int chunk, int buf) {
        const float *wsrc = a.wt + ((int64_t)chunk * ntiles + ntile) * WSLAB;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int kib = i * NWAVE + wave_u;
            if (kib < WSLAB / 256) lds_dma16(wsrc + kib * 256 + lane * 4, slab_lds + (uint32_t)(buf * BUF * 4 + kib * 1024));
        }
    };
    // DMAs per wave: uniform over the waves only when the slab is a whole number of KiB per wave
    static_assert((WSLAB / 256) % NWAVE == 0, "every wave issues the same number of DMAs (vmcnt is counted per wave)");

    issue_patch(0);
    dma_weights(0, 0);
    commit_patch(0);
    if (nchunks > 1) issue_patch(1);
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks, more2 = chunk + 2 < nchunks;
        if (more) {
            commit_patch(cur ^ 1);                   // chunk + 1: requested a whole iteration ago
            dma_weights(chunk + 1, cur ^ 1);
            asm volatile("" ::: "memory");           // the DMA stays ahead of the loads in program order
            if (more2) issue_patch(chunk + 2);
        }
        const float *sp = lds + cur * BUF;
#pragma unroll
        for (int c4 = 0; c4 < KC / 4; ++c4) {
            float d[4][4], t[4][4], V[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = sp[a_base + c4 * 4 * CS + r * PWp + ((c & 1) ? 0 : EOFF) + (c >> 1)];
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
                const float2 bf = *reinterpret_cast<const float2 *>(sp + b_base + (p * KC + c4 * 4) * BNP);
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf.x, acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p], bf.y, acc[p][1], 0, 0, 0);
            }
        }
        if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }

    // ---- output transform Y = A^T M A (lane-local) + epilogue
    float *out_n = a.out + (int64_t)n * a.Cout * plane;
    const int yb = y0 + 2 * wm;
    const int xb = x0 + 8 * lk;                 // this lane's 4 tiles = 8 consecutive output columns
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + (wn * NT + nt) * 16 + li;
        if (co >= a.Cout) continue;
        const float sc = a.ep_scale[co], sh = a.ep_shift[co];
        float y[2][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[0 + c][nt][r], m1 = acc[4 + c][nt][r], m2 = acc[8 + c][nt][r], m3 = acc[12 + c][nt][r];
                s[0][c] = m0 + m1 + m2;
                s[1][c] = m1 - m2 - m3;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[i][2 * r + 0] = s[i][0] + s[i][1] + s[i][2];
                y[i][2 * r + 1] = s[i][1] - s[i][2] - s[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = yb + i;
            if (yy >= a.H || xb >= a.W) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = y[i][j] * sc + sh;
                if (a.relu) v[j] = v[j] > 0.f ? v[j] : 0.f;
            }
            const uint32_t e = (uint32_t)((co * a.H + yy) * a.W + xb);
            if (a.drop_site >= 0) {
                const uint32_t w = wino_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((w >> j) & 1u) ? v[j] * 2.f : 0.f;
            }
            float *dst = out_n + (int64_t)co * plane + (int64_t)yy * a.W + xb;
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Winograd is used for 3x3 layers whose geometry keeps every access aligned: W a multiple of 8
// (float4 stores, one Philox word per 8 outputs), H even, Cout a multiple of the Cout tile.
// Two tilings: cfg 0 = 4 x 32 px x 64 couts, K-chunk 4 (3 workgroups/CU); cfg 1 = 8 x 32 px x 32 couts,
// K-chunk 8 (2 workgroups/CU, half as many barriers per MFMA); cfg 2 = 12 x 32 px x 64 couts, K-chunk 4, 12 waves
// (one workgroup per CU at 3 waves/SIMD: the weight slab is staged once for 3x as many MFMAs).
static int wino_bn(int cfg) { return cfg == 1 ? 32 : 64; }
static int wino_kc(int cfg) { return cfg == 1 ? 8 : 4; }
bool wino_supported(int ks, int cin, int cout, int H, int W) {
    return ks == 3 && cin >= 4 && cout % 64 == 0 && (W % 8) == 0 && (H % 2) == 0;
}
int wino_slab_floats(int cfg) { return wino_slab(wino_bn(cfg), wino_kc(cfg)); }
int wino_chunks(int cfg, int cin) { return (cin + wino_kc(cfg) - 1) / wino_kc(cfg); }
int wino_cout_tile(int cfg) { return wino_bn(cfg); }

// Caffe (Cout,Cin,3,3) -> U = G g G^T, laid out [ceil(Cin/KC)][Cout/BN][slab], slab row p*KC + ci%KC.
void wino_pack_weights(const float *W, int cin, int cout, int cfg, std::vector<float> &out, int *cout_pad) {
    const int bn = wino_bn(cfg), kc = wino_kc(cfg), bnp = wino_bnp(bn), slab = wino_slab(bn, kc);
    const int ntiles = cout / bn, nchunks = (cin + kc - 1) / kc;
    *cout_pad = cout;
    out.assign((size_t)nchunks * ntiles * slab, 0.f);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = W + ((size_t)co * cin + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) tmp[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const size_t base = ((size_t)(ci / kc) * ntiles + co / bn) * slab;
            // within each 32-cout group the two 16-cout n-tiles are interleaved (one ds_read_b64 feeds both MFMAs)
            const int cl = co % bn, col = (cl / 32) * 32 + 2 * (cl % 16) + (cl % 32) / 16;
            const int c = ci % kc, shift = bn == 64 ? (c & 1) * 16 : 0;
            for (int p = 0; p < 16; ++p) out[base + (size_t)(p * kc + c) * bnp + shift + col] = (float)U[p / 4][p % 4];
        }
}

template <int WM, int WN, int NT, int KC, int ABL = 0, bool UNPOOL = false>
static void launch_wino_cfg(const ConvArgs &a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int BN = WN * NT * 16;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 2 * WM - 1) / (2 * WM);
    const int ptiles = a.tiles_x * a.tiles_y * a.N;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8 * (a.CoutPad / BN)));
    hipLaunchKernelGGL((conv_wino_kernel<WM, WN, NT, KC, ABL, UNPOOL>), grid, dim3(WM * WN * 64), 0, s, a);
}

template <bool UNPOOL>
static void launch_wino_p(const ConvArgs &a0, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 3) / 4;
    const int ptiles = a.tiles_x * a.tiles_y * a.N;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8 * (a.CoutPad / 64)));
    hipLaunchKernelGGL((conv_wino_p_kernel<2, 2, 2, 4, UNPOOL>), grid, dim3(256), 0, s, a);
}

void launch_conv_wino(const ConvArgs &a, int cfg, hipStream_t s) {
    // default tiling (cfg 0), no probe variant: the form with its memory traffic in flight (SIVO_WINO_PIPE=0: the older one)
    static const bool pipe_env = !(SIVO_DIAG_ENV("SIVO_WINO_PIPE") && std::atoi(SIVO_DIAG_ENV("SIVO_WINO_PIPE")) == 0);
    if (pipe_env && cfg == 0 && (a.variant >> 8) == 0 && (int64_t)a.Cin * a.H * a.W * 4 < (1ll << 31))
        return a.unpool_mask ? launch_wino_p<true>(a, s) : launch_wino_p<false>(a, s);
    if (a.unpool_mask) return launch_wino_cfg<2, 2, 2, 4, 0, true>(a, s);     // Upsample fused into the patch loader (cfg 0 only)
    if (cfg == 1) return launch_wino_cfg<4, 1, 2, 8>(a, s);
    if (cfg == 2) return launch_wino_cfg<6, 2, 2, 4>(a, s);   // 12 waves: 12 x 32 px x 64 couts, one workgroup per CU
    switch (a.variant >> 8) {   // ablations, probe only
        case 1: return launch_wino_cfg<2, 2, 2, 4, 1>(a, s);
        case 3: return launch_wino_cfg<2, 2, 2, 4, 3>(a, s);
        case 5: return launch_wino_cfg<2, 2, 2, 4, 5>(a, s);
        case 9: return launch_wino_cfg<2, 2, 2, 4, 9>(a, s);
        case 15: return launch_wino_cfg<2, 2, 2, 4, 15>(a, s);
        default: return launch_wino_cfg<2, 2, 2, 4>(a, s);
    }
}

}  // namespace sivo
