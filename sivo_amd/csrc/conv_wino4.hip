// conv_wino4.hip — 3x3 convolution by Winograd F(4x4, 3x3) as three kernels around a batched fp32 MFMA GEMM.
//
// Lavin & Gray (CVPR 2016), interpolation points {0, +-1, +-2, inf}:
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 4x4 output tile, 6x6 input tile d;
// for each of the 36 transform positions xi an independent GEMM over the input channels
//   M_xi[tile][cout] = sum_c V_xi[tile][c] * U_xi[c][cout]
// i.e. 36 multiplies per 16 outputs instead of 144: 4x fewer MFMA flops than the direct convolution
// (F(2x2,3x3) in conv_wino.hip saves 2.25x).  Everything stays fp32; the larger transform constants cost
// accuracy (measured through the whole Standard net: |dlogit| ~2e-4 against the 1e-3 budget, F(2x2) ~3e-5), so
// this path is used only where it pays: the wide (>= 128 channel) layers, where a transform-position GEMM has
// enough arithmetic intensity (C*K / (2 (C + K)) flop/byte >= 32) to sit on the matrix cores.
//
// A fused F(4x4) kernel would need 36 positions x 4 accumulator registers per 16x16 MFMA block = 144 VGPRs
// with no register blocking left, i.e. two LDS operand reads per MFMA — LDS-bound.  Splitting the work lets
// the GEMM use 64x64 register tiles (0.5 LDS reads per MFMA) while the two transforms are plain streaming
// kernels; the price is the V and M round trip (2.25x the activation size each), which is why the work is
// issued in groups of samples small enough for V and M to stay in the 256 MB memory-side cache.
//
//   wino4_input_kernel   x (n,C,H,W)            -> V [36][C][Pp]      thread = (channel, tile), 6 float4 loads +
//                                                                     neighbour columns by wave shuffle
//   wino4_gemm_kernel    V, U [36][C][Kp]       -> M [36][Kp][Pp]     128 tiles x 128 couts per workgroup, 4 waves
//                                                                     x (64 x 64), K-chunks of 16 channels by
//                                                                     LDS-DMA into a double buffer, XOR-swizzled
//   wino4_output_kernel  M                      -> y (n,K,H,W)        thread = (cout, tile): A^T M A, bias+BN,
//                                                                     ReLU, Philox dropout, 4 float4 stores
// P = tiles of the sample group (n * ceil(H/4) * W/4), Pp / Kp padded to multiples of 128.
#include <hip/hip_runtime.h>
#include <map>
#include <cstdio>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

// wino4_bridge_kernel is compiled WITHOUT packed-FP32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), DESIGN 3.3: with
// them, a bridge workgroup that shares a CU with a workgroup of the f16x3 GEMM (wino4_gemm_h3_kernel<128, 256>: fp16 MFMAs + LDS-DMA) now
// and then stores a V' word computed from other values than its registers held — LDS, M and the window reads verified clean, the same
// arithmetic in scalar instructions never differs (tools/coresident_probe.py HZ5 / HZ7 / HZ8 / HZ9, profiles/r05_coresident_hazard_*).
// tests/test_codeobj.py checks the product's code object for it.  The diagnostic build can put the packed form back
// (-DSIVO_BRIDGE_PACKED_FP32, Makefile diag_pkbridge): the reproducer.
#if (defined(SIVO_DIAG) && defined(SIVO_BRIDGE_PACKED_FP32)) || !defined(__HIP_DEVICE_COMPILE__)      // (a device feature: the host pass has no use for it)
#define W4_BRIDGE_NO_PK
#else
#define W4_BRIDGE_NO_PK __attribute__((target("no-packed-fp32-ops")))
#endif

#include "common.hpp"
#include "h3_split.hpp"
#include "segnet_kernels.hpp"
#include "wino4_transforms.hpp"

namespace sivo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wino4_dropout_word(uint32_t e, uint32_t site, uint32_t sample, uint64_t seed) {
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

struct Wino4Args {
    const float *in; int64_t in_sample_stride;   // (n or 1, C, H, W); pooled (n or 1, C, H/2, W/2) when mask != null
    const uint8_t *mask; int64_t mask_sample_stride;
    float *V, *M;                                // workspace
    const float *U;                              // [36][C][Kp]
    const float *ep_scale, *ep_shift;
    float *out;                                  // (n, K, H, W)
    int n, C, K, Kp, H, W, th, tw, P, Pp;
    int relu, drop_site, sample0;
    int in_drop, in_drop_site;                   // in_drop != 0: dropout of site in_drop_site applied to the input as it is read (ConvArgs::in_drop_site)
    uint64_t seed;
    // output transform fused with the MAX 2x2 pooling that consumes this layer (the 4x4 tile holds four whole windows):
    float *pool_out;            // (n, K, Ho, Wo) or null
    uint8_t *pool_mask;         // window codes, same shape
    int pool_drop_site, Ho, Wo;
    // f16x3 GEMM (conv_wino4_h3.hip): the transform kernels instantiated with PACK write V as (hi | lo << 16) fp16 pairs of
    // V * vscale instead of fp32 (same 4 bytes per element) and raise *h3_flag when a value leaves the fp16 range; the
    // output transform multiplies M by mscale = 1 / (vscale * uscale) inside its per-channel affine (powers of two: exact)
    float vscale, mscale;
    uint32_t *h3_flag;
    uint32_t *vmax;             // calibration pass: atomicMax of the bit pattern of |V| (the layer's largest transformed value), or null
};

// end of a transform thread: report an overflow / the calibration maximum (rare / calibration only)
__device__ __forceinline__ void wino4_report(const Wino4Args &a, bool bad, float vmax) {
    if (bad) atomicOr(a.h3_flag, 1u);
    if (a.vmax) {
        const uint32_t b = __float_as_uint(vmax);        // non-negative floats order like their bit patterns
        if (b > *a.vmax) atomicMax(a.vmax, b);
    }
}

// (the 1-D transforms B^T d, A^T m and the weight matrix G: wino4_transforms.hpp)

constexpr int W4_TIN = 256;

// grid: (ceil(P / 256), C).  Lanes run over consecutive tiles (x fastest), so V stores are fully coalesced.
// UNPOOL: the input is read through a max-unpool (Upsample scale 2): 4 x 4 pooled values + window codes per tile
// instead of 6 x 6 unpooled values, and the unpooled tensor never exists in HBM.
template <bool UNPOOL, bool PACK>
__global__ __launch_bounds__(W4_TIN) void wino4_input_kernel(Wino4Args a) {
    const int p = blockIdx.x * W4_TIN + threadIdx.x, c = blockIdx.y;
    if (p >= a.P) return;       // whole waves leave together except in the last block; shuffles below only pair live lanes
    const int tx = p % a.tw, ty = (p / a.tw) % a.th, n = p / (a.tw * a.th);
    const int lane = threadIdx.x & 63;
    // the left / right neighbour tile is the previous / next lane when it exists in this wave and in this tile row
    const bool left_lane = lane > 0 && tx > 0, right_lane = lane < 63 && tx < a.tw - 1 && p + 1 < a.P;
    float d[6][6];
    if (UNPOOL) {
        const int Hp = a.H >> 1, Wp = a.W >> 1;
        const float *src = a.in + (int64_t)n * a.in_sample_stride + (int64_t)c * Hp * Wp;
        const uint8_t *msk = a.mask + (int64_t)n * a.mask_sample_stride + (int64_t)c * Hp * Wp;
        const int px0 = 2 * tx;
        float pv[4][4];
        int pc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int py = 2 * ty - 1 + r;
            const bool row_ok = py >= 0 && py < Hp;
            float v0 = 0.f, v1 = 0.f;
            int m0 = -1, m1 = -1;
            if (row_ok) {
                const float2 v = *reinterpret_cast<const float2 *>(src + (int64_t)py * Wp + px0);     // Wp even: aligned
                const uchar2 m = *reinterpret_cast<const uchar2 *>(msk + (int64_t)py * Wp + px0);
                v0 = v.x; v1 = v.y; m0 = m.x; m1 = m.y;
            }
            const float vl = __shfl_up(v1, 1, 64), vr = __shfl_down(v0, 1, 64);
            const int ml = __shfl_up(m1, 1, 64), mr = __shfl_down(m0, 1, 64);
            float l = 0.f, rr = 0.f;
            int lm = -1, rm = -1;
            if (row_ok) {
                if (left_lane) { l = vl; lm = ml; }
                else if (px0 > 0) { l = src[(int64_t)py * Wp + px0 - 1]; lm = msk[(int64_t)py * Wp + px0 - 1]; }
                if (right_lane) { rr = vr; rm = mr; }
                else if (px0 + 2 < Wp) { rr = src[(int64_t)py * Wp + px0 + 2]; rm = msk[(int64_t)py * Wp + px0 + 2]; }
            }
            pv[r][0] = l; pv[r][1] = v0; pv[r][2] = v1; pv[r][3] = rr;
            pc[r][0] = lm; pc[r][1] = m0; pc[r][2] = m1; pc[r][3] = rm;
        }
        // unpooled pixel (4ty - 1 + i, 4tx - 1 + j): pooled index ((i + 1) >> 1, (j + 1) >> 1), window position ((i + 1) & 1, (j + 1) & 1)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int pi = (i + 1) >> 1, pj = (j + 1) >> 1, code = (((i + 1) & 1) << 1) | ((j + 1) & 1);
                d[i][j] = pc[pi][pj] == code ? pv[pi][pj] : 0.f;
            }
    } else {
        const float *src = a.in + (int64_t)n * a.in_sample_stride + (int64_t)c * a.H * a.W;
        const int x0 = 4 * tx, y0 = 4 * ty - 1;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = y0 + r;
            const bool row_ok = y >= 0 && y < a.H;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row_ok) v = *reinterpret_cast<const f32x4 *>(src + (int64_t)y * a.W + x0);     // W % 4 == 0: aligned, in bounds
            // in_drop: the four elements e .. e + 3 of the (C, H, W) sample share one dropout word (e % 4 == 0); dropped BEFORE the
            // shuffles, so a neighbour receives the dropped value
            const uint32_t e = (uint32_t)((c * a.H + y) * a.W + x0);
            if (a.in_drop && row_ok) {
                const uint32_t w = wino4_dropout_word(e, (uint32_t)a.in_drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
                v.x = (w & 1u) ? v.x * 2.f : 0.f; v.y = (w & 2u) ? v.y * 2.f : 0.f; v.z = (w & 4u) ? v.z * 2.f : 0.f; v.w = (w & 8u) ? v.w * 2.f : 0.f;
            }
            // every lane takes part in the shuffles (rows outside the image contribute zeros)
            const float from_left = __shfl_up(v.w, 1, 64), from_right = __shfl_down(v.x, 1, 64);
            float l = 0.f, rr = 0.f;
            if (row_ok) {
                if (left_lane) l = from_left;
                else if (x0 > 0) {
                    l = src[(int64_t)y * a.W + x0 - 1];
                    if (a.in_drop) l = ((wino4_dropout_word(e - 1u, (uint32_t)a.in_drop_site, (uint32_t)(a.sample0 + n), a.seed) >> ((e - 1u) & 31)) & 1u) ? l * 2.f : 0.f;
                }
                if (right_lane) rr = from_right;
                else if (x0 + 4 < a.W) {
                    rr = src[(int64_t)y * a.W + x0 + 4];
                    if (a.in_drop) rr = ((wino4_dropout_word(e + 4u, (uint32_t)a.in_drop_site, (uint32_t)(a.sample0 + n), a.seed) >> ((e + 4u) & 31)) & 1u) ? rr * 2.f : 0.f;
                }
            }
            d[r][0] = l; d[r][1] = v.x; d[r][2] = v.y; d[r][3] = v.z; d[r][4] = v.w; d[r][5] = rr;
        }
    }
    // columns, then rows
    float t[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col[6];
        wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], col);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][j] = col[i];
    }
    float *dst = a.V + (int64_t)c * a.Pp + p;
    const int64_t xi_stride = (int64_t)a.C * a.Pp;
    bool bad = false;
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float row[6];
        wino4_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], row);
        if (a.vmax)
#pragma unroll
            for (int j = 0; j < 6; ++j) vmax = fmaxf(vmax, fabsf(row[j]));
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (PACK) reinterpret_cast<uint32_t *>(dst)[(int64_t)(i * 6 + j) * xi_stride] = wino4_pack_h3(row[j], a.vscale, bad);
            else dst[(int64_t)(i * 6 + j) * xi_stride] = row[j];
        }
    }
    if (PACK || a.vmax) wino4_report(a, bad, vmax);
}

// ---------------------------------------------------------------------------------------------------
// batched GEMM  M_xi[k][p] = sum_c U_xi[c][k] * V_xi[c][p]
// ---------------------------------------------------------------------------------------------------
constexpr int G_BM = 128, G_BN = 128, G_KC = 16;          // largest tile: tiles x couts x channels per stage (Pp, Kp are padded to these)
constexpr int G_STAGE = G_KC * (G_BM + G_BN);             // floats per LDS stage of the largest tile (16 KB)

// Workgroup tile BM tiles x BN couts (128x128, 64x128 or 64x64: the smaller ones keep the CUs busy when a launch has
// few tiles — the 22x64 layers, or one or two samples per GPU), 4 waves as 2 x 2, wave tile (BM/2) x (BN/2).
// Stage rows are BM / BN floats (multiples of the 32 banks): the four k-rows a wave reads together would collide, so
// row c stores its 16-float groups XOR-swizzled by (c & 1): even rows as they are, odd rows with neighbouring
// groups exchanged.  LDS-DMA fixes the destination (wave base + lane * 16 B), so the swizzle is applied to the SOURCE.
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wino4_gemm_kernel(Wino4Args a, int ptiles, int ktiles) {
    extern __shared__ float lds[];
    constexpr int TM = BM / 2, TN = BN / 2, MTF = TM / 16, NTF = TN / 16;     // fragments per lane: 4 or 2
    constexpr int STAGE = G_KC * (BM + BN);
    constexpr int NV = BM / 16, NU = BN / 16;           // 1 KiB DMA instructions per stage for the V / U part
    static_assert((NV + NU) % 4 == 0, "whole DMA instructions per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    // XCD-aware order: (position, p-tile) pairs round-robin over the 8 XCDs, the k-tiles of a pair back to back on one
    // XCD so the V tile is fetched into that L2 once; U_xi (C*Kp*4 B <= 1 MB) stays resident in every L2.
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int kt = j % ktiles, pair = (j / ktiles) * 8 + xcd;
    if (pair >= 36 * ptiles) return;
    const int xi = pair / ptiles, pt = pair % ptiles;
    const float *Vg = a.V + ((int64_t)xi * a.C) * a.Pp + (int64_t)pt * BM;
    const float *Ug = a.U + ((int64_t)xi * a.C) * a.Kp + (int64_t)kt * BN;

    // DMA: a stage = 16 V rows then 16 U rows; one instruction = 1 KiB = 256 / L rows of L floats.
    // Lane -> (row within the instruction, 16-byte chunk); logical chunk = physical ^ (4 * (row & 1)).
    auto dma = [&](int chunk, int buf) {
        float *dstb = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < (NV + NU) / 4; ++i) {
            const int inst = wave * ((NV + NU) / 4) + i;
            const bool isU = inst >= NV;
            const int L4 = (isU ? BN : BM) / 4;                  // 16-byte chunks per row
            const int il = isU ? inst - NV : inst;
            const int row = il * (64 / L4) + lane / L4, ch = lane % L4;
            const int lchunk = ch ^ ((row & 1) << 2);
            const float *src = (isU ? Ug + (int64_t)(chunk * G_KC + row) * a.Kp : Vg + (int64_t)(chunk * G_KC + row) * a.Pp) + lchunk * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dstb + inst * 256), 16, 0, 0);
        }
    };

    f32x4 acc[MTF][NTF];
#pragma unroll
    for (int mt = 0; mt < MTF; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTF; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.C / G_KC;
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    __syncthreads();
    const int sw = (lk & 1) << 4;            // swizzle of this lane's k-row (c = 4*step + lk)
    // MFMA block mt takes the tiles {MTF*i + mt}, block nt the couts {NTF*j + nt} of the wave's sub-tile: the A (B)
    // fragments of a lane are then MTF (NTF) consecutive floats of a stage row — ONE ds_read_b128 / b64 each.
    const int a_off = (wm * TM + MTF * li) ^ sw, b_off = (wn * TN + NTF * li) ^ sw;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        if (chunk + 1 < nchunks) dma(chunk + 1, cur ^ 1);
        const float *Vs = lds + cur * STAGE, *Us = Vs + G_KC * BM;
#pragma unroll
        for (int s = 0; s < G_KC / 4; ++s) {
            float af[MTF], bf[NTF];
            if (MTF == 4) { const f32x4 q = *reinterpret_cast<const f32x4 *>(Vs + (4 * s + lk) * BM + a_off); af[0] = q[0]; af[1] = q[1]; af[MTF - 2] = q[2]; af[MTF - 1] = q[3]; }
            else { const float2 q = *reinterpret_cast<const float2 *>(Vs + (4 * s + lk) * BM + a_off); af[0] = q.x; af[1] = q.y; }
            if (NTF == 4) { const f32x4 q = *reinterpret_cast<const f32x4 *>(Us + (4 * s + lk) * BN + b_off); bf[0] = q[0]; bf[1] = q[1]; bf[NTF - 2] = q[2]; bf[NTF - 1] = q[3]; }
            else { const float2 q = *reinterpret_cast<const float2 *>(Us + (4 * s + lk) * BN + b_off); bf[0] = q.x; bf[1] = q.y; }
#pragma unroll
            for (int mt = 0; mt < MTF; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
    // acc[mt][nt][r] = M[tile p0 + wm*TM + MTF*(4*lk + r) + mt][cout k0 + wn*TN + NTF*li + nt]: per cout a lane owns the
    // 4*MTF consecutive tiles from 4*MTF*lk, MTF floats per store; the four lk lanes of a cout write 64*MTF contiguous bytes.
    float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * BN) * a.Pp + (int64_t)pt * BM;
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) {
        float *row = Mg + (int64_t)(wn * TN + NTF * li + nt) * a.Pp + wm * TM + 4 * MTF * lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MTF == 4) *reinterpret_cast<f32x4 *>(row + 4 * r) = f32x4{acc[0][nt][r], acc[1][nt][r], acc[MTF - 2][nt][r], acc[MTF - 1][nt][r]};
            else *reinterpret_cast<float2 *>(row + 2 * r) = make_float2(acc[0][nt][r], acc[1][nt][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same batched GEMM on the bf16 matrix cores, fp32 in / fp32 out: "bf16x6".
//
// fp32 MFMA runs at 1/16 of the bf16 rate, and this GEMM is the one kernel of the path that sits on the MFMA
// roofline.  An fp32 value is the exact sum of three bf16 values up to 2^-24 relative (x = x1 + x2 + x3,
// x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); bf16 keeps fp32's exponent, so no range problem), and
// a product of two bf16 values is exact in fp32.  So
//     v u  =  v1 u1 + (v1 u2 + v2 u1) + (v1 u3 + v2 u2 + v3 u1)  +  O(2^-23 |v u|)
// is six v_mfma_f32_16x16x32_bf16 with fp32 accumulation per 32 channels (6 x ~17 cycles) instead of eight
// v_mfma_f32_16x16x4_f32 (8 x 32 cycles): 2.5x fewer matrix-core cycles at the accuracy of the fp32 chain (emulated on
// Winograd-domain data: rms error 5.0e-6 vs 8.0e-6 for the sequential fp32 FMA chain it replaces; three products
// only — "bf16x3" — would be 15x worse and is not used).  The logits tests state the tolerance.
//
// V stays what the transform kernels write (fp32 [36][C][Pp]) and is split into its three bf16 planes inside the GEMM; U is
// split once on the host and stored as the LDS image of its stages (LDS-DMA).  The operand unit is a 16-byte piece = 8
// channels of one row (tile or cout) = the A / B fragment of one lane; pieces are placed (x6_slot) so that every
// ds_read_b128 / ds_write_b64 is conflict-free.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X6_KC = 32;                       // channels per stage (= K of one MFMA)
constexpr int X6_PLANE = 128 * X6_KC * 2;       // bytes of one bf16 plane of a 128-row operand stage (8 KB)

// 16-byte piece (row r of the 128-row operand tile, channel octet kg) -> slot of the plane.  Rows are grouped by
// MFMA block (r & 3) so that a block's 16 rows x 4 octets are 64 consecutive pieces; inside, the octet is XOR-ed /
// rotated so that the four 16-lane groups of a ds_read_b128 and the eight 8-lane groups of a ds_write_b128 each cover
// distinct bank quads.
__host__ __device__ __forceinline__ int x6_slot(int r, int kg) {
    const int mt = r & 3, q = r >> 2;
    return 4 * (mt * 32 + q) + (((kg ^ ((q & 8) ? 3 : 0)) + mt) & 3);
}

// ---------------------------------------------------------------------------------------------------
// bf16x6 GEMM, producer / consumer form.  Since round 3 this is the FALLBACK of the three-kernel F(4x4) path: the default
// GEMM is the f16x3 one of conv_wino4_h3.hip (half the matrix-core work); this kernel runs the layers whose operands do
// not fit the fp16 range (overflow flag of the transform kernels), calibration passes, and SIVO_GEMM=x6.
//
//   waves 0-3  CONSUMERS  one per SIMD: per stage 24 ds_read_b128 + 96 back-to-back independent MFMAs, nothing else;
//                         at the end of a work item the 64 accumulators go to M and are cleared;
//   waves 4-7  PRODUCERS  one per SIMD, beside a consumer: everything they bring in comes by LDS-DMA a whole stage ahead
//                         (the U image of the next stage, the raw fp32 V rows each wave then splits itself into the three
//                         bf16 planes of the other V buffer).
// LDS: bf16 V planes x 2, U planes x 3, raw fp32 V x 2 = 152 KB; ONE barrier per stage hands a stage over in both directions.
// The workgroup is persistent: it walks a list of (position, tile block, cout block) items, so the producers run ahead
// across item boundaries and a consumer's epilogue stores overlap the next item's first stages.  Items are dealt so
// that the cout blocks of one (position, tile block) pair run at the same time on CUs of ONE XCD (V tile fetched into
// that L2 once), and an XCD stays on one position for many items (U_xi resident in its L2).
// Measured in round 2 (NOTEBOOK 3.1b; the variants tried there — flat two-workgroup form, 8 consumer / 8 producer waves, U
// fragments from global memory, two stages per barrier, wave priorities, per-role cycle stamps — are in the git history of
// this file, their numbers in DESIGN): a stage costs 3250 cycles against 1536 of its MFMAs and the producers' chain
// (ten DMA pieces + the V split per wave and stage) is the critical one.
// ---------------------------------------------------------------------------------------------------
// Workgroup barrier without the release / acquire fences of __syncthreads(): those wait for vmcnt(0), i.e. for the LDS-DMA a
// producer has just issued for the NEXT stage and for a consumer's epilogue stores — exactly what has to stay in flight.
__device__ __forceinline__ void x6p_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int X6P_VRAW = 128 * X6_KC * 4;        // one fp32 V stage as the LDS-DMA leaves it (16 KB)
constexpr int X6P_LDS = (2 + 3) * 3 * X6_PLANE + 2 * X6P_VRAW;       // V planes x 2, U planes x 3, raw V x 2 = 152 KB

__global__ __launch_bounds__(512, 1) void wino4_gemm_x6p_kernel(Wino4Args a, const uint4 *__restrict__ Ux, int ptiles, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunks = a.C / X6_KC;
    // item list of this XCD: pairs xcd, xcd + 8, ... (pair = position * ptiles + tile block), each with its ktiles cout
    // blocks back to back; the workgroups of the XCD (blockIdx.x >> 3 = 0 .. per_xcd - 1) take the items round-robin
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int npairs = 36 * ptiles, pairs_x = (npairs - xcd + 7) / 8;          // pairs of this XCD
    const int nitems = pairs_x * ktiles;
    if (wg >= nitems) return;
    const int my_items = (nitems - wg + per_xcd - 1) / per_xcd;
    const int nstages = my_items * nchunks;
    auto item_of = [&](int k, int &xi, int &pt, int &kt) {       // k-th item of this workgroup
        const int it = wg + k * per_xcd;
        kt = it % ktiles;
        const int pair = (it / ktiles) * 8 + xcd;
        xi = pair / ptiles; pt = pair % ptiles;
    };
    auto Vl = [&](int buf) { return lds6 + 3 * X6_PLANE * buf; };
    auto Ul = [&](int buf) { return lds6 + 3 * X6_PLANE * (2 + buf); };
    auto Vraw = [&](int buf) { return lds6 + 3 * X6_PLANE * 5 + X6P_VRAW * buf; };

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        // wave w copies exactly the 8 channel rows it splits itself (4 x 1 KiB), so no producer depends on another one
        const int w = wave - 4, vh = lane >> 5, vtq = lane & 31;
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        int k_item = 0, chunk = 0;                  // the stage the NEXT DMA batch belongs to
        int xi, pt, kt;
        item_of(0, xi, pt, kt);
        auto issue_stage = [&](int s) {             // DMA of stage s = (xi, pt, kt, chunk); then step to the next stage
            const uint4 *usrc = Ux + ((int64_t)(xi * ktiles + kt) * nchunks + chunk) * (3 * 512);
            unsigned char *udst = Ul(s % 3);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int kib = w * 6 + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(usrc + kib * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(udst + kib * 1024), 16, 0, 0);
            }
            const float *Vg = a.V + ((int64_t)xi * a.C) * a.Pp + (int64_t)pt * 128;
            unsigned char *vdst = Vraw(s & 1) + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)          // row pair i of the octet: channels 2 i, 2 i + 1
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(Vg + (int64_t)(chunk * X6_KC + w * 8 + 2 * i + vh) * a.Pp + 4 * vtq),
                    (__attribute__((address_space(3))) void *)(vdst + i * 1024), 16, 0, 0);
            if (++chunk == nchunks) {
                chunk = 0;
                if (++k_item < my_items) item_of(k_item, xi, pt, kt);
            }
        };
        auto split_stage = [&](int s) {             // raw V rows of this wave -> three bf16 planes of V buffer s & 1
            const unsigned char *src = Vraw(s & 1) + w * 4096 + lane * 16;
            f32x4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vr[i] = *reinterpret_cast<const f32x4 *>(src + i * 1024);
            unsigned char *Vb = Vl(s & 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                __bf16 p1[4], p2[4], p3[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = vr[i][t];
                    const __bf16 x1 = (__bf16)x;
                    const float r1 = x - (float)x1;
                    const __bf16 x2 = (__bf16)r1;
                    const float r2 = r1 - (float)x2;
                    p1[i] = x1; p2[i] = x2; p3[i] = (__bf16)r2;
                }
                // element 4 vh + i of the piece is channel 2 i + vh
                unsigned char *dst = Vb + x6_slot(4 * vtq + t, w) * 16 + 8 * vh;
                *reinterpret_cast<bf16x4 *>(dst) = bf16x4{p1[0], p1[1], p1[2], p1[3]};
                *reinterpret_cast<bf16x4 *>(dst + X6_PLANE) = bf16x4{p2[0], p2[1], p2[2], p2[3]};
                *reinterpret_cast<bf16x4 *>(dst + 2 * X6_PLANE) = bf16x4{p3[0], p3[1], p3[2], p3[3]};
            }
        };
        if (0 < nstages) issue_stage(0);
        for (int s = 0; s <= nstages; ++s) {
            if (s < nstages) {
                if (s + 1 < nstages) {
                    issue_stage(s + 1);              // lands during the consumers' stage s
                    // all but that batch: stage s has landed (vector-memory operations complete in issue order)
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                split_stage(s);
            }
            x6p_barrier();                          // stage s handed to the consumers, stage s - 1's buffers free again
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave & 1, wn = (wave >> 1) & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int a_off[4], b_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_off[t] = x6_slot(64 * wm + 4 * li + t, lk) * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) b_off[t] = x6_slot(64 * wn + 4 * li + t, lk) * 16;
    int k_item = 0, chunk = 0;
    x6p_barrier();                                  // stage 0 ready
    bf16x8 bfrag[4][3], af[3];
    for (int s = 0; s < nstages; ++s) {
        const unsigned char *Vs = Vl(s & 1), *Us = Ul(s % 3);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bfrag[nt][pl] = *reinterpret_cast<const bf16x8 *>(Us + pl * X6_PLANE + b_off[nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8 *>(Vs + pl * X6_PLANE + a_off[mt]);
            // smallest terms first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[term]], bfrag[nt][PB[term]], acc[mt][nt], 0, 0, 0);
            }
        }
        if (++chunk == nchunks) {
            chunk = 0;
            int xi, pt, kt;
            item_of(k_item++, xi, pt, kt);
            float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * 128) * a.Pp + (int64_t)pt * 128;
            // M stores in 64-byte runs: a lane holds 16 consecutive tiles of one cout row as four 16-byte pieces
            // (piece 4 lk + r); stored as they are, one instruction writes pieces 64 bytes apart.  A 4 x 4 transpose of
            // the pieces over the four lanes li, li + 16, li + 32, li + 48 (v_permlane16_swap / v_permlane32_swap, two
            // stages, no LDS) gives lane lk the pieces 4 j + lk: instruction j then writes 64 contiguous bytes per row.
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float *row = Mg + (int64_t)(wn * 64 + 4 * li + nt) * a.Pp + wm * 64 + 4 * lk;
                f32x4 y[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mt][nt][0]), __float_as_uint(acc[mt][nt][1]), false, false);
                    const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mt][nt][2]), __float_as_uint(acc[mt][nt][3]), false, false);
                    const auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
                    const auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
                    y[0][mt] = __uint_as_float(t02[0]); y[2][mt] = __uint_as_float(t02[1]);
                    y[1][mt] = __uint_as_float(t13[0]); y[3][mt] = __uint_as_float(t13[1]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(row + 16 * j) = y[j];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        x6p_barrier();                              // done with stage s; stage s + 1 ready
    }
}

// grid: (ceil(P / 256), K)
// POOL: instead of the 4x4 outputs the kernel writes the 2x2 pooled values (first strict maximum in scan order, as
// maxpool2_kernel / Caffe), their window codes and the pooling layer's dropout — the convolution output is not stored.
template <bool POOL>
__global__ __launch_bounds__(W4_TIN) void wino4_output_kernel(Wino4Args a) {
    const int p = blockIdx.x * W4_TIN + threadIdx.x, co = blockIdx.y;
    if (p >= a.P) return;
    const int tx = p % a.tw, ty = (p / a.tw) % a.th, n = p / (a.tw * a.th);
    const float *src = a.M + (int64_t)co * a.Pp + p;
    const int64_t xi_stride = (int64_t)a.Kp * a.Pp;
    float t[4][6];      // A^T M (columns)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = src[(int64_t)(i * 6 + j) * xi_stride];
        float s[4];
        wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], s);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i][j] = s[i];
    }
    const float sc = a.ep_scale[co] * a.mscale, sh = a.ep_shift[co];
    float *dst = a.out + ((int64_t)n * a.K + co) * a.H * a.W;
    float y4[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 4 * ty + i;
        float v[4];
        wino4_at(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], v);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = v[r] * sc + sh;
            if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
        const int x = 4 * tx;
        if (a.drop_site >= 0 && y < a.H) {
            const uint32_t e = (uint32_t)((co * a.H + y) * a.W + x);
            const uint32_t w = wino4_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? v[r] * 2.f : 0.f;
        }
        if (POOL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y4[i][r] = v[r];
        } else if (y < a.H) {
            *reinterpret_cast<f32x4 *>(dst + (int64_t)y * a.W + x) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    if (POOL) {
        const int64_t chw = (int64_t)a.K * a.Ho * a.Wo;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int py = 2 * ty + wy;
            if (py >= a.Ho) break;
            float pv[2];
            int pc[2];
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                float best = -3.402823466e+38f;
                int code = 0;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int yy = 4 * ty + 2 * wy + dy;
                        const float v = y4[2 * wy + dy][2 * wx + dx];
                        if (yy < a.H && v > best) { best = v; code = dy * 2 + dx; }
                    }
                pv[wx] = best; pc[wx] = code;
            }
            const int px = 2 * tx;
            const int64_t e = ((int64_t)co * a.Ho + py) * a.Wo + px;       // element index inside the sample
            if (a.pool_drop_site >= 0) {
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const uint32_t ee = (uint32_t)(e + wx);
                    const uint32_t w = wino4_dropout_word(ee, (uint32_t)a.pool_drop_site, (uint32_t)(a.sample0 + n), a.seed);
                    pv[wx] = ((w >> (ee & 31)) & 1u) ? pv[wx] * 2.f : 0.f;
                }
            }
            *reinterpret_cast<float2 *>(a.pool_out + (int64_t)n * chw + e) = make_float2(pv[0], pv[1]);
            *reinterpret_cast<uchar2 *>(a.pool_mask + (int64_t)n * chw + e) = make_uchar2((unsigned char)pc[0], (unsigned char)pc[1]);
        }
    }
}

// Output transform of layer A fused with the input transform of layer B = the next convolution (same geometry,
// C_B = K_A): the activation between two F(4x4) convolutions never goes to HBM.  One workgroup per (sample, channel)
// plane: every thread turns its tiles' 36 M values into the 4x4 outputs (+ A's epilogue) and puts them into an LDS
// image of the plane with a zero border; after one barrier it reads its 6x6 window back and writes B's 36 V values.
// Reads 2.25 y + writes 2.25 y instead of (2.25 y + y) + (y + 2.25 y).   grid: (n, K_A), dynamic LDS = plane.
// PACK: the next layer runs the f16x3 GEMM: its V is written as packed fp16 pairs scaled by next_vscale.
template <bool PACK>
__global__ W4_BRIDGE_NO_PK __launch_bounds__(1024) void wino4_bridge_kernel(Wino4Args a, float *Vnext, float next_vscale, uint32_t *next_vmax) {
    extern __shared__ float plane_raw[];        // (4*th + 2) rows x (W + 4) floats; image pixel (y, x) at [y + 1][x + 1]
    float *plane = plane_raw;
    const int n = blockIdx.x, co = blockIdx.y;
    const int RS = a.W + 4, rows = 4 * a.th + 2, ntile = a.th * a.tw;
    for (int i = threadIdx.x; i < rows * RS; i += blockDim.x) plane[i] = 0.f;
    __syncthreads();
    const float sc = a.ep_scale[co] * a.mscale, sh = a.ep_shift[co];
    const int64_t xs_m = (int64_t)a.Kp * a.Pp, xs_v = (int64_t)a.K * a.Pp;
    bool bad = false;
    float vmax = 0.f;
    // the 4 x 4 output pixels of tile t of (sample n, cout co) after the epilogue (BN, ReLU, dropout); rows 4 ty + i >= H do not exist
    auto tile_values = [&](int t, float (&vv)[4][4]) __attribute__((always_inline)) {
        const int tx = t % a.tw, ty = t / a.tw;
        const float *src = a.M + (int64_t)co * a.Pp + (int64_t)n * ntile + t;
        float tt[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                m[i] = src[(int64_t)(i * 6 + j) * xs_m];
            }
            float s4[4];
            wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], s4);
#pragma unroll
            for (int i = 0; i < 4; ++i) tt[i][j] = s4[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 4 * ty + i;
            float v[4];
            wino4_at(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = v[r] * sc + sh;
                if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            const int x = 4 * tx;
            if (a.drop_site >= 0 && y < a.H) {
                const uint32_t e = (uint32_t)((co * a.H + y) * a.W + x);
                const uint32_t w = wino4_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? v[r] * 2.f : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[i][r] = v[r];
        }
    };
    for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
        const int tx = t % a.tw, ty = t / a.tw;
        float vv[4][4];
        tile_values(t, vv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 4 * ty + i;
            if (y >= a.H) break;                 // rows below the image stay zero
            float *dst = plane + (y + 1) * RS + 4 * tx + 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = vv[i][r];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
        const int tx = t % a.tw, ty = t / a.tw;
        const float *win = plane + (4 * ty) * RS + 4 * tx;       // 16-byte aligned: RS % 4 == 0
        float d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(win + i * RS);
            const float2 r2 = *reinterpret_cast<const float2 *>(win + i * RS + 4);
            d[i][0] = q.x; d[i][1] = q.y; d[i][2] = q.z; d[i][3] = q.w; d[i][4] = r2.x; d[i][5] = r2.y;
        }
        float tb[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float col[6];
            wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], col);
#pragma unroll
            for (int i = 0; i < 6; ++i) tb[i][j] = col[i];
        }
        float *dst = Vnext + (int64_t)co * a.Pp + (int64_t)n * ntile + t;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float row[6];
            wino4_bt(tb[i][0], tb[i][1], tb[i][2], tb[i][3], tb[i][4], tb[i][5], row);
            if (next_vmax)
#pragma unroll
                for (int j = 0; j < 6; ++j) vmax = fmaxf(vmax, fabsf(row[j]));
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (PACK) reinterpret_cast<uint32_t *>(dst)[(int64_t)(i * 6 + j) * xs_v] = wino4_pack_h3(row[j], next_vscale, bad);
                else dst[(int64_t)(i * 6 + j) * xs_v] = row[j];
            }
        }
    }
    if (PACK || next_vmax) {
        Wino4Args r = a;
        r.vmax = next_vmax;
        wino4_report(r, bad, vmax);
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool wino4_supported(int ks, int cin, int cout, int H, int W) {
    // below 128 channels a transform-position GEMM is HBM-bound on V and M (C*K / (2 (C + K)) < 32 flop/byte; couts are
    // padded to 128 as well) and the fused F(2x2) kernel wins — measured on the Standard shapes: 256->128 0.69 vs 0.92 ms,
    // 128->128 1.52 vs 1.74 ms, but 128->64 1.37 vs 0.95 ms and 64->64 3.2 vs 1.9 ms.
    const int minc = 128;
    return ks == 3 && cin % G_KC == 0 && cin >= minc && cout >= minc && W % 4 == 0 && H >= 4 && W >= 4;
}

int wino4_cout_pad(int cout) { return (cout + G_BN - 1) / G_BN * G_BN; }

// Caffe (Cout,Cin,3,3) -> U [36][Cin][Kp] = G g G^T, evaluated in double and rounded once
void wino4_pack_weights(const float *W, int cin, int cout, std::vector<float> &out, int *cout_pad) {
    const auto &G = WINO4_G;
    const int Kp = wino4_cout_pad(cout);
    *cout_pad = Kp;
    out.assign((size_t)36 * cin * Kp, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = W + ((size_t)co * cin + ci) * 9;
            double t[6][3];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    out[((size_t)(i * 6 + j) * cin + ci) * Kp + co] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

// bf16x6: U [36][Cin][Kp] fp32 -> three bf16 planes per (position, 128-cout tile, 32-channel chunk), in the order the
// kernel's LDS stage has them: [xi][kt][chunk][plane][x6_slot(cout row, channel octet)][8 channels]
bool wino4_x6_supported(int cin, int cout_pad) { return cin % X6_KC == 0 && cout_pad % 128 == 0; }

static inline uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    std::memcpy(&x, &u, 4);
    return x;
}

void wino4_x6_pack_weights(const std::vector<float> &U, int cin, int Kp, std::vector<uint16_t> &out) {
    const int ktiles = Kp / 128, nchunks = cin / X6_KC;
    out.assign((size_t)36 * ktiles * nchunks * 3 * 512 * 8, 0);
    for (int xi = 0; xi < 36; ++xi)
        for (int kt = 0; kt < ktiles; ++kt)
            for (int ch = 0; ch < nchunks; ++ch) {
                uint16_t *stage = out.data() + ((size_t)((xi * ktiles + kt) * nchunks + ch)) * (3 * 512 * 8);
                for (int r = 0; r < 128; ++r)
                    for (int kg = 0; kg < 4; ++kg)
                        for (int e = 0; e < 8; ++e) {
                            // element e of octet kg <-> channel 8 kg + 2 (e & 3) + (e >> 2): the order the kernel's V loads produce
                            const float x = U[((size_t)xi * cin + ch * X6_KC + kg * 8 + 2 * (e & 3) + (e >> 2)) * Kp + kt * 128 + r];
                            const uint16_t x1 = bf16_rne(x);
                            const float r1 = x - bf16_to_float(x1);
                            const uint16_t x2 = bf16_rne(r1);
                            const float r2 = r1 - bf16_to_float(x2);
                            const size_t o = (size_t)x6_slot(r, kg) * 8 + e;
                            stage[o] = x1; stage[512 * 8 + o] = x2; stage[2 * 512 * 8 + o] = bf16_rne(r2);
                        }
            }
}

// samples per group so that V + M of a group stay within `budget` bytes (memory-side cache), at least 1
int wino4_group(int N, int cin, int cout, int H, int W, size_t budget) {
    const int64_t tiles = (int64_t)((H + 3) / 4) * (W / 4);
    const int64_t per_sample = 36 * tiles * 4 * ((int64_t)cin + wino4_cout_pad(cout));
    int g = (int)(budget / (size_t)per_sample);
    if (g < 1) g = 1;
    if (g > N) g = N;
    return g;
}

size_t wino4_workspace_floats(int group, int cin, int cout, int H, int W) {
    const int64_t P = (int64_t)group * ((H + 3) / 4) * (W / 4), Pp = (P + G_BM - 1) / G_BM * G_BM;
    return (size_t)(36 * Pp * ((int64_t)cin + wino4_cout_pad(cout)));
}

#ifdef SIVO_DIAG
// diagnostic build, SIVO_W4_VERIFY=1: a layer's GEMM and bridge are run a second time into scratch buffers, under the same
// concurrent conditions, and compared word for word: diag word [4] counts M words that differ, [5] V' words, [6] layers compared
__global__ void diag_compare_kernel(const uint32_t *x, const uint32_t *y, int64_t n, uint32_t *count, int Pp, int P, uint32_t *rec) {
    unsigned bad = 0;          // (columns P .. Pp of a row of Pp words are padding nobody writes)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if ((i % Pp) < P && x[i] != y[i]) {
            ++bad;
            if (rec) {                               // the first twelve differing words: index, first run, second run
                const unsigned k = atomicAdd(rec, 1u);
                if (k < 12) { rec[1 + 3 * k] = (uint32_t)i; rec[2 + 3 * k] = x[i]; rec[3 + 3 * k] = y[i]; }
            }
        }
    if (bad) atomicAdd(count, bad);
}
#endif

size_t wino4_bridge_lds_bytes(int H, int W) { return (size_t)(4 * ((H + 3) / 4) + 2) * (W + 4) * sizeof(float); }

// One F(4x4,3x3) layer.  `group` samples per pass over the workspace.  plan (optional) chains layers without going
// through HBM with the activation: V / M / Vnext are caller-chosen disjoint buffers, skip_input says V already holds
// this layer's transformed input (written by the previous layer's bridge), bridge replaces the output transform by
// wino4_bridge_kernel writing the NEXT layer's V into Vnext (the plain output `c.out` is then not produced).  A plan
// requires all samples in one group.
// GEMM: f16x3 (conv_wino4_h3.hip) when the layer carries split fp16 weights and a V scale (c.wt_h3, c.h3_vscale > 0), else
// bf16x6 when it carries the bf16 planes (c.wt_x6) and the launch is not tiny, else the fp32-MFMA kernel.
// ev (optional, profiling): 4 events per group, recorded before the input transform, after it, after the GEMM and
// after the output transform / bridge (gemm_only_events: only the two around the GEMM).
void launch_conv_wino4(const ConvArgs &c, float *workspace, int group, hipStream_t s, hipEvent_t *ev, bool gemm_only_events,
                       const Wino4Plan *plan) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        // a refused opt-in would otherwise only show as an opaque launch failure of the first frame
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_gemm_x6p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS));
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_gemm_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_STAGE * 4));
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_bridge_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_bridge_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    Wino4Args a{};
    a.C = c.Cin; a.K = c.Cout; a.Kp = c.CoutPad; a.H = c.H; a.W = c.W;
    a.th = (c.H + 3) / 4; a.tw = c.W / 4;
    a.U = c.wt; a.ep_scale = c.ep_scale; a.ep_shift = c.ep_shift;
    a.relu = c.relu; a.drop_site = c.drop_site; a.seed = c.seed; a.in_sample_stride = c.in_sample_stride;
    a.in_drop = c.in_drop_site >= 0 ? 1 : 0; a.in_drop_site = c.in_drop_site;
    if (a.in_drop && (c.unpool_mask || c.in_sample_stride != 0)) throw std::invalid_argument("launch_conv_wino4: in_drop_site needs a plain sample-invariant input");
    a.pool_out = c.pool_out; a.pool_mask = c.pool_mask; a.pool_drop_site = c.pool_drop_site; a.Ho = (c.H + 1) / 2; a.Wo = (c.W + 1) / 2;
    const bool h3 = c.wt_h3 && c.h3_vscale > 0.f;
    a.vscale = h3 ? c.h3_vscale : 0.f;
    a.mscale = h3 ? 1.f / (c.h3_vscale * c.h3_uscale) : 1.f;
    a.h3_flag = c.h3_flag; a.vmax = c.vmax;
    if (h3 && !c.h3_flag) throw std::invalid_argument("f16x3 GEMM without an overflow flag");
    if (plan) group = c.N;
    for (int n0 = 0; n0 < c.N; n0 += group) {
        a.n = c.N - n0 < group ? c.N - n0 : group;
        a.P = a.n * a.th * a.tw;
        a.Pp = (a.P + G_BM - 1) / G_BM * G_BM;
        a.in = c.in + (int64_t)n0 * c.in_sample_stride;
        a.mask = c.unpool_mask ? c.unpool_mask + (int64_t)n0 * c.unpool_mask_stride : nullptr;
        a.mask_sample_stride = c.unpool_mask_stride;
        a.out = c.out ? c.out + (int64_t)n0 * c.Cout * c.H * c.W : nullptr;
        if (c.pool_out) {
            a.pool_out = c.pool_out + (int64_t)n0 * c.Cout * a.Ho * a.Wo;
            a.pool_mask = c.pool_mask + (int64_t)n0 * c.Cout * a.Ho * a.Wo;
        }
        a.sample0 = c.sample0 + n0;
        a.V = plan ? plan->V : workspace;
        a.M = plan ? plan->M : workspace + (size_t)36 * a.C * a.Pp;
        const unsigned pblocks = (unsigned)((a.P + W4_TIN - 1) / W4_TIN);
        hipEvent_t *e = ev ? ev + 4 * (n0 / group) : nullptr;
        if (e && !gemm_only_events) (void)hipEventRecord(e[0], s);
        if (!(plan && plan->skip_input)) {
            const dim3 gi(pblocks, (unsigned)a.C), bi(W4_TIN);
            if (a.mask && h3) hipLaunchKernelGGL((wino4_input_kernel<true, true>), gi, bi, 0, s, a);
            else if (a.mask) hipLaunchKernelGGL((wino4_input_kernel<true, false>), gi, bi, 0, s, a);
            else if (h3) hipLaunchKernelGGL((wino4_input_kernel<false, true>), gi, bi, 0, s, a);
            else hipLaunchKernelGGL((wino4_input_kernel<false, false>), gi, bi, 0, s, a);
        }
        if (e) (void)hipEventRecord(e[1], s);
        auto nblocks = [&](int bm, int bn) { return (int64_t)36 * ((a.P + bm - 1) / bm) * (a.Kp / bn); };
        // bf16x6 (128 x 128 items): whenever the layer has the split weights and the launch is not tiny
        const int x6_min_blocks = 128;
        if (h3) {
            launch_wino4_gemm_h3(reinterpret_cast<const uint32_t *>(a.V), c.wt_h3, a.M, a.C, a.Kp, a.P, a.Pp, s);
        } else if (c.wt_x6 && nblocks(128, 128) >= x6_min_blocks) {
            const int pt6 = (a.P + 127) / 128, kt6 = a.Kp / 128;
            static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
            const dim3 gp((unsigned)((n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8));        // one persistent workgroup per CU, a multiple of the 8 XCDs
            hipLaunchKernelGGL(wino4_gemm_x6p_kernel, gp, dim3(512), X6P_LDS, s, a, reinterpret_cast<const uint4 *>(c.wt_x6), pt6, kt6);
        } else {
            // fp32 MFMA: the largest tile that still gives every CU ~4 workgroups (256 CUs)
            const int min_blocks = 1024;
            const int tile = nblocks(128, 128) >= min_blocks ? 0 : nblocks(64, 128) >= min_blocks ? 1 : 2;
            const int bm = tile == 0 ? 128 : 64, bn = tile == 2 ? 64 : 128;
            const int pt_n = (a.P + bm - 1) / bm, kt_n = a.Kp / bn, pairs8 = (36 * pt_n + 7) / 8;
            const dim3 ggrid((unsigned)(pairs8 * kt_n * 8));
            const size_t glds = (size_t)2 * G_KC * (bm + bn) * 4;
            if (tile == 0) hipLaunchKernelGGL((wino4_gemm_kernel<128, 128>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
            else if (tile == 1) hipLaunchKernelGGL((wino4_gemm_kernel<64, 128>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
            else hipLaunchKernelGGL((wino4_gemm_kernel<64, 64>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
        }
        if (e) (void)hipEventRecord(e[2], s);
        if (plan && plan->bridge) {
            const int ntile = a.th * a.tw;
            const int nthr = ntile >= 1024 ? 1024 : (ntile + 63) / 64 * 64;
            const dim3 gb((unsigned)a.n, (unsigned)a.K);
            const size_t lb = wino4_bridge_lds_bytes(a.H, a.W);
            if (plan->next_vscale > 0.f) hipLaunchKernelGGL(wino4_bridge_kernel<true>, gb, dim3(nthr), lb, s, a, plan->Vnext, plan->next_vscale, plan->next_vmax);
            else hipLaunchKernelGGL(wino4_bridge_kernel<false>, gb, dim3(nthr), lb, s, a, plan->Vnext, 0.f, plan->next_vmax);
        } else {
            if (a.pool_out) hipLaunchKernelGGL(wino4_output_kernel<true>, dim3(pblocks, (unsigned)a.K), dim3(W4_TIN), 0, s, a);
            else hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(pblocks, (unsigned)a.K), dim3(W4_TIN), 0, s, a);
        }
        if (e && !gemm_only_events) (void)hipEventRecord(e[3], s);
#ifdef SIVO_DIAG
        if (h3 && plan && plan->bridge && SIVO_DIAG_ENV("SIVO_W4_VERIFY")) {
            struct Scratch { float *m2 = nullptr, *v2 = nullptr; size_t cap = 0; };
            static std::map<hipStream_t, Scratch> per_stream;           // (the lanes of a handle are enqueued by one host thread)
            Scratch &sc = per_stream[s];
            const size_t need = (size_t)36 * std::max(a.Kp, a.C) * a.Pp;
            if (need > sc.cap) {
                SIVO_HIP(hipDeviceSynchronize());
                if (sc.m2) { (void)hipFree(sc.m2); (void)hipFree(sc.v2); }
                SIVO_HIP(hipMalloc((void **)&sc.m2, need * 4)); SIVO_HIP(hipMalloc((void **)&sc.v2, need * 4));
                sc.cap = need;
            }
            float *m2 = sc.m2, *v2 = sc.v2;
            launch_wino4_gemm_h3(reinterpret_cast<const uint32_t *>(a.V), c.wt_h3, m2, a.C, a.Kp, a.P, a.Pp, s);
            hipLaunchKernelGGL(diag_compare_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(a.M), reinterpret_cast<const uint32_t *>(m2),
                               (int64_t)36 * a.Kp * a.Pp, diag_words() + 4, (int)a.Pp, (int)a.Pp, (uint32_t *)nullptr);
            const int ntile = a.th * a.tw, nthr = ntile >= 1024 ? 1024 : (ntile + 63) / 64 * 64;
            const size_t lb = wino4_bridge_lds_bytes(a.H, a.W);
            if (plan->next_vscale > 0.f) hipLaunchKernelGGL(wino4_bridge_kernel<true>, dim3((unsigned)a.n, (unsigned)a.K), dim3(nthr), lb, s, a, v2, plan->next_vscale, plan->next_vmax);
            else hipLaunchKernelGGL(wino4_bridge_kernel<false>, dim3((unsigned)a.n, (unsigned)a.K), dim3(nthr), lb, s, a, v2, 0.f, plan->next_vmax);
            // (the padding columns P .. Pp of V' are written by nobody: compare sample by sample, the tiles that exist)
            hipLaunchKernelGGL(diag_compare_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(plan->Vnext), reinterpret_cast<const uint32_t *>(v2),
                               (int64_t)36 * a.K * a.Pp, diag_words() + 5, (int)a.Pp, (int)a.P, diag_words() + 20);
            ++diag_words()[6];
            if (diag_words()[20] && !diag_words()[19]) {        // geometry of the first layer that showed a difference (host side, after the fact: approximate)
                diag_words()[16] = (uint32_t)a.K; diag_words()[17] = (uint32_t)a.Pp; diag_words()[18] = (uint32_t)(a.th * a.tw); diag_words()[19] = (uint32_t)a.tw;
            }
        }
#endif
    }
}

}  // namespace sivo
