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
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "segnet_kernels.hpp"

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
    uint64_t seed;
    // output transform fused with the MAX 2x2 pooling that consumes this layer (the 4x4 tile holds four whole windows):
    float *pool_out;            // (n, K, Ho, Wo) or null
    uint8_t *pool_mask;         // window codes, same shape
    int pool_drop_site, Ho, Wo;
};

// 1-D input transform B^T d (points 0, +-1, +-2, inf)
__device__ __forceinline__ void wino4_bt(const float d0, const float d1, const float d2, const float d3, const float d4,
                                         const float d5, float *t) {
    const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    t[0] = 4.f * d0 - 5.f * d2 + d4;
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d1 - 5.f * d3 + d5;
}

constexpr int W4_TIN = 256;

// grid: (ceil(P / 256), C).  Lanes run over consecutive tiles (x fastest), so V stores are fully coalesced.
// UNPOOL: the input is read through a max-unpool (Upsample scale 2): 4 x 4 pooled values + window codes per tile
// instead of 6 x 6 unpooled values, and the unpooled tensor never exists in HBM.
template <bool UNPOOL>
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
            // every lane takes part in the shuffles (rows outside the image contribute zeros)
            const float from_left = __shfl_up(v.w, 1, 64), from_right = __shfl_down(v.x, 1, 64);
            float l = 0.f, rr = 0.f;
            if (row_ok) {
                if (left_lane) l = from_left; else if (x0 > 0) l = src[(int64_t)y * a.W + x0 - 1];
                if (right_lane) rr = from_right; else if (x0 + 4 < a.W) rr = src[(int64_t)y * a.W + x0 + 4];
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
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float row[6];
        wino4_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], row);
#pragma unroll
        for (int j = 0; j < 6; ++j) dst[(int64_t)(i * 6 + j) * xi_stride] = row[j];
    }
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
// V stays what the transform kernels write (fp32 [36][C][Pp]): every thread loads its 2 x 8 channel values of the
// next K-chunk with coalesced dword loads one chunk ahead, splits them in registers (v_cvt_pk_bf16_f32) and stores
// three 16-byte pieces per (tile, channel octet) into LDS.  U is split once on the host and stored as the LDS image of
// its stage (LDS-DMA).  Operand pieces are laid out so that every ds_read_b128 / ds_write_b128 is conflict-free.
// Workgroup 128 tiles x 128 couts, 4 waves x (64 x 64), K-chunk 32: V single-buffered (24 KB, it goes through
// registers anyway), U double-buffered (2 x 24 KB) -> 72 KB, two workgroups per CU whose phases interleave.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X6_KC = 32;                       // channels per stage (= K of one MFMA)
constexpr int X6_PLANE = 128 * X6_KC * 2;       // bytes of one bf16 plane of a 128-row operand stage (8 KB)
constexpr int X6_LDS = 3 * X6_PLANE * 3;        // V + 2 x U

// 16-byte piece (row r of the 128-row operand tile, channel octet kg) -> slot of the plane.  Rows are grouped by
// MFMA block (r & 3) so that a block's 16 rows x 4 octets are 64 consecutive pieces; inside, the octet is XOR-ed /
// rotated so that the four 16-lane groups of a ds_read_b128 and the eight 8-lane groups of a ds_write_b128 each cover
// distinct bank quads.
__host__ __device__ __forceinline__ int x6_slot(int r, int kg) {
    const int mt = r & 3, q = r >> 2;
    return 4 * (mt * 32 + q) + (((kg ^ ((q & 8) ? 3 : 0)) + mt) & 3);
}

// ABL (diagnostics, tools/x6_probe.py): 1 no V loads after the first chunk, 2 no split / LDS stores after it, 4 no U DMA
// after it, 8 no MFMAs, 16 no epilogue stores.  0 = the production kernel.
template <int ABL>
__global__ __launch_bounds__(256, 2) void wino4_gemm_x6_kernel(Wino4Args a, const uint4 *__restrict__ Ux, int ptiles, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int kt = j % ktiles, pair = (j / ktiles) * 8 + xcd;
    if (pair >= 36 * ptiles) return;
    const int xi = pair / ptiles, pt = pair % ptiles;
    const int nchunks = a.C / X6_KC;
    const float *Vg = a.V + ((int64_t)xi * a.C) * a.Pp + (int64_t)pt * 128;
    // U image: [xi][kt][chunk][plane][512 pieces]
    const uint4 *Ug = Ux + ((int64_t)(xi * ktiles + kt) * nchunks) * (3 * 512);

    unsigned char *Vl = lds6;                                   // 3 planes
    auto Ul = [&](int buf) { return lds6 + 3 * X6_PLANE * (1 + buf); };

    // V of a stage = 32 channel rows x 128 tiles fp32 (16 KB): four 16-byte loads per thread, a wave-load covering two whole
    // 512-byte channel rows (lanes 0-31 / 32-63).  Thread (wave w, half h = lane >> 5, tile quad tq = lane & 31) holds
    // x[i][t] = V[32 chunk + 8 w + 2 i + h][4 tq + t]: for each of its four tiles the 4 channels {8 w + 2 i + h} — half of
    // the (tile, octet w) piece.  The order of the 8 channels inside a piece is free as long as U uses the same one
    // (x6_channel_of): element e of octet kg <-> channel 8 kg + 2 (e & 3) + (e >> 2).
    const int vh = lane >> 5, vtq = lane & 31;
    f32x4 vreg[4];
    auto load_v = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vreg[i] = *reinterpret_cast<const f32x4 *>(Vg + (int64_t)(chunk * X6_KC + wave * 8 + 2 * i + vh) * a.Pp + 4 * vtq);
    };
    auto dma_u = [&](int chunk, int buf) {
        const uint4 *src = Ug + (int64_t)chunk * (3 * 512);
        unsigned char *dst = Ul(buf);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int kib = wave * 6 + i;                        // 24 KiB: six 1 KiB copies per wave
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kib * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(dst + kib * 1024), 16, 0, 0);
        }
    };
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto split_store_v = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x4 p1, p2, p3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = vreg[i][t];
                const __bf16 x1 = (__bf16)x;
                const float r1 = x - (float)x1;
                const __bf16 x2 = (__bf16)r1;
                const float r2 = r1 - (float)x2;
                p1[i] = x1; p2[i] = x2; p3[i] = (__bf16)r2;
            }
            unsigned char *dst = Vl + x6_slot(4 * vtq + t, wave) * 16 + 8 * vh;
            *reinterpret_cast<bf16x4 *>(dst) = p1;
            *reinterpret_cast<bf16x4 *>(dst + X6_PLANE) = p2;
            *reinterpret_cast<bf16x4 *>(dst + 2 * X6_PLANE) = p3;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // MFMA block mt of this wave takes the tiles 64 wm + 4 i + mt (i = A row), block nt the couts 64 wn + 4 j + nt:
    // the accumulators of a lane are then runs of 4 consecutive tiles per cout (f32x4 stores, as in the fp32 kernel)
    int a_off[4], b_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a_off[t] = x6_slot(64 * wm + 4 * li + t, lk) * 16;
        b_off[t] = x6_slot(64 * wn + 4 * li + t, lk) * 16;
    }

    // Pipeline: at the start of chunk c's MFMA phase U(c + 1) goes by LDS-DMA into the other U buffer and V(c + 1) into
    // registers; both have the whole phase to arrive and are waited for (vmcnt(0)) at the top of the next chunk.
    dma_u(0, 0);
    load_v(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this chunk's V values in registers, its U stage in LDS
        __syncthreads();                             // every wave is done reading the previous chunk's V stage
        if (!(ABL & 2) || chunk == 0) split_store_v();
        __syncthreads();                             // V stage + U stage visible
        if (chunk + 1 < nchunks) {
            if (!(ABL & 4)) dma_u(chunk + 1, cur ^ 1);
            if (!(ABL & 1)) load_v(chunk + 1);
        }
        const unsigned char *Us = Ul(cur);
        bf16x8 bfrag[4][3];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bfrag[nt][pl] = *reinterpret_cast<const bf16x8 *>(Us + pl * X6_PLANE + b_off[nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            bf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8 *>(Vl + pl * X6_PLANE + a_off[mt]);
            // smallest terms first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (ABL & 8) acc[mt][nt][0] += (float)af[PA[term]][0] + (float)bfrag[nt][PB[term]][1];
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[term]], bfrag[nt][PB[term]], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }
    float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * 128) * a.Pp + (int64_t)pt * 128;
    if ((ABL & 16) && acc[0][0][0] != 12345.678f) return;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float *row = Mg + (int64_t)(wn * 64 + 4 * li + nt) * a.Pp + wm * 64 + 16 * lk;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4 *>(row + 4 * r) = f32x4{acc[0][nt][r], acc[1][nt][r], acc[2][nt][r], acc[3][nt][r]};
    }
}

// ---------------------------------------------------------------------------------------------------
// bf16x6 GEMM, producer / consumer form (the production kernel).
//
// Measured on the kernel above (rocprofv3 PMC, conv4_2 at T = 12): matrix cores busy 43 % of the time; with all staging
// removed 69 %, i.e. the MFMA phase itself loses a quarter to the two barriers per chunk and to two waves per SIMD taking
// turns, and none of the staging (V loads + split, U DMA, epilogue stores) hides behind the other workgroup's MFMA phase.
// So the roles are separated inside one 512-thread workgroup per CU:
//   waves 0-3  CONSUMERS  one per SIMD: per stage 24 ds_read_b128 + 96 back-to-back independent MFMAs, nothing else;
//                         at the end of a work item the 64 accumulators go to M and are cleared;
//   waves 4-7  PRODUCERS  one per SIMD, beside a consumer: wait for the V values of the NEXT stage (loaded one stage
//                         earlier), split them into the three bf16 planes of the other V buffer, start the LDS-DMA of the
//                         next stage's U image, issue the V loads of the stage after that.
// LDS: bf16 V planes x 2, U planes x 3, raw fp32 V x 2 = 152 KB; ONE barrier per stage hands a stage over in both directions.
// The workgroup is persistent: it walks a list of (position, tile block, cout block) items, so the producers run ahead
// across item boundaries and a consumer's epilogue stores overlap the next item's first stages.  Items are dealt so
// that the cout blocks of one (position, tile block) pair run at the same time on CUs of ONE XCD (V tile fetched into
// that L2 once), and an XCD stays on one position for many items (U_xi resident in its L2).
// ---------------------------------------------------------------------------------------------------
// Workgroup barrier without the release / acquire fences of __syncthreads(): those wait for vmcnt(0), i.e. for the LDS-DMA a
// producer has just issued for the NEXT stage and for a consumer's epilogue stores — exactly what has to stay in flight.
// What the hand-over needs: this wave's LDS writes and reads done (lgkmcnt(0)); a producer additionally waits for the
// DMA of the stage it hands over with an explicit vmcnt before calling this.
__device__ __forceinline__ void x6p_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic (template flag TS, SIVO_X6_STAMPS=1, read with sivo_debug_x6_stamps): shader-clock cycles every wave spends
// between its hand-overs (work: everything up to its own LDS operations being done) and inside them (wait: from there until
// the slowest wave of the workgroup arrives), summed over the waves of a role and over all workgroups since the last reset.
// [0] consumers work, [1] consumers wait, [2] producers work, [3] producers wait, [4] hand-overs counted (all waves); the
// producers' work split further: [5] issuing the LDS-DMA batch, [6] s_waitcnt vmcnt for the stage to land, [7] the split
// (LDS reads, conversions, LDS stores, until lgkmcnt(0)).
__device__ unsigned long long x6p_stamps[8];
struct X6pClock {
    unsigned long long work = 0, wait = 0, last = 0, n = 0, issue = 0, vmwait = 0, split = 0;
};
template <bool TS>
__device__ __forceinline__ unsigned long long x6p_now() {
    if constexpr (TS) {
        asm volatile("" ::: "memory");
        const unsigned long long t = __builtin_readcyclecounter();
        asm volatile("" ::: "memory");
        return t;
    } else {
        return 0;
    }
}
template <bool TS>
__device__ __forceinline__ void x6p_barrier_t(X6pClock &c) {
    if constexpr (!TS) {
        x6p_barrier();
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        asm volatile("s_barrier" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        c.work += t0 - c.last; c.wait += t1 - t0; c.last = t1; ++c.n;
    }
}
template <bool TS>
__device__ __forceinline__ void x6p_clock_flush(const X6pClock &c, int role, int lane) {
    if constexpr (TS) {
        if (lane == 0) {
            atomicAdd(&x6p_stamps[2 * role], c.work);
            atomicAdd(&x6p_stamps[2 * role + 1], c.wait);
            atomicAdd(&x6p_stamps[4], c.n);
            if (role == 1) {
                atomicAdd(&x6p_stamps[5], c.issue);
                atomicAdd(&x6p_stamps[6], c.vmwait);
                atomicAdd(&x6p_stamps[7], c.split);
            }
        }
    }
}

constexpr int X6P_VRAW = 128 * X6_KC * 4;        // one fp32 V stage as the LDS-DMA leaves it (16 KB)
constexpr int X6P_LDS = (2 + 3) * 3 * X6_PLANE + 2 * X6P_VRAW;       // V planes x 2, U planes x 3, raw V x 2 = 152 KB
constexpr int X6P_LDS_SB2 = 4 * 3 * X6_PLANE + 4 * X6P_VRAW;          // BG, SB = 2: V planes x 4, raw V x 4 = 160 KB

// ABL (diagnostics): 1 no V DMA after stage 0, 2 no U DMA after stage 0, 4 no split after stage 0, 8 no MFMAs, 16 operand
// fragments read from LDS in stage 0 only, 32 no M stores.
// NCW: consumer waves, 4 (one per SIMD, 64 x 64 each) or 8 (two per SIMD, 64 tiles x 32 couts each: the two MFMA row classes
// 2h, 2h + 1 of a 64-cout range).  One wave per SIMD cannot keep the matrix pipe full — the MFMA-only ablation of the
// 4-consumer form tops out at 1.22 PFLOP/s while conv7_x6.hip, two MFMA waves per SIMD, executes 1.42 with all its staging.
// NPW: producer waves, 4 (one channel octet each) or 8 (half an octet each: two of the octet's four row pairs; the bf16 pieces
// are then written as 4-byte halves).  A stage waits for the slower of the two chains (tools/x6_probe.py ablate: consumers
// alone 0.44 ms, everything but the MFMAs 0.43 ms, together 0.57 ms on conv4_2): eight producers halve the latency of theirs.
// BG (experiment, SIVO_X6_BGLOBAL=1): the consumers take their U fragments straight from global memory (the U image is
// stored in fragment order: the 64 pieces of one (cout block, plane) are 1 KiB contiguous), three loads per 16-cout block
// issued as soon as the block's MFMAs of the current stage are issued — no U DMA, no U bytes through LDS (per stage 24 KB
// written + 48 KB read of the 176 KB the LDS moves).
// SB (with BG only, SIVO_X6_BGLOBAL=2): stages per workgroup barrier.  SB = 2: the consumers multiply two stages between
// barriers while the producers split the next two (four V-plane buffers + the ring of four raw buffers = 160 KB): half as
// many hand-overs, each covering twice the work of both roles.
// AP (with SB = 2, SIVO_X6_BGLOBAL=3; written at the end of round 2, compiled, NOT yet run on a GPU): the V fragments of an
// interval's second stage are read under the MFMAs of its first stage (second register set), so only every other stage
// starts with the twelve ds_read_b128 in front of its first MFMA.
// A2 (SIVO_X6_AFRAG=1; compiled, not yet run): the V fragments of the four MFMA blocks of a stage in four register sets, all
// twelve read at the top of the stage — instead of one set refilled immediately in front of each block's first MFMA
// (four exposed LDS latencies per stage).
template <int ABL, int NCW = 4, int NPW = 4, bool BG = false, int SB = 1, bool AP = false, bool TS = false, bool A2 = false>
__global__ __launch_bounds__((NCW + NPW) * 64, 1) void wino4_gemm_x6p_kernel(Wino4Args a, const uint4 *__restrict__ Ux, int ptiles, int ktiles_prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ktiles_prio: cout blocks | wave priorities << 16 (experiment SIVO_X6_PRIO: bits 0-1 consumers, bits 2-3 producers).
    // s_setprio ignores EXEC, so the role test is made on a scalar.
    const int ktiles = ktiles_prio & 0xffff;
    {
        const int prio = ktiles_prio >> 16;
        const int want = __builtin_amdgcn_readfirstlane(tid >> 6) >= NCW ? (prio >> 2) & 3 : prio & 3;
        if (want == 1) __builtin_amdgcn_s_setprio(1);
        else if (want == 2) __builtin_amdgcn_s_setprio(2);
        else if (want == 3) __builtin_amdgcn_s_setprio(3);
    }
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
    // BG: no U stages in LDS, so the raw V rows get a ring of PD + 1 buffers in that space and are requested PD stages ahead
    static_assert(SB == 1 || (BG && SB == 2), "SB = 2 needs the LDS the U stages occupy");
    constexpr int PD = BG ? 3 : 1, NRAW = PD + 1, NVB = 2 * SB;
    auto Vraw = [&](int buf) { return lds6 + 3 * X6_PLANE * (BG ? NVB : 5) + X6P_VRAW * buf; };

    if (wave >= NCW) {
        // ------------------------------------------------------------------ producers
        // Everything a producer brings in comes by LDS-DMA, issued a whole stage ahead and waited for with vmcnt only:
        // the U image of the next stage (24 KB, 6 x 1 KiB per wave) and the raw fp32 V rows of the next stage — wave w
        // copies exactly the 8 channel rows it splits itself (4 x 1 KiB), so no producer depends on another one.
        const int w = wave - NCW, vh = lane >> 5, vtq = lane & 31;
        constexpr int UPW = BG ? 0 : 24 / NPW, VPW = 16 / NPW;         // 1 KiB DMA pieces per wave and stage: U planes, raw V rows
        const int oct = NPW == 8 ? w >> 1 : w, ih = NPW == 8 ? (w & 1) : 0;      // channel octet, half of it (row pairs 2 ih, 2 ih + 1)
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        int k_item = 0, chunk = 0;                  // the stage the NEXT DMA batch belongs to
        int xi, pt, kt;
        item_of(0, xi, pt, kt);
        auto issue_stage = [&](int s) {             // DMA of stage s = (xi, pt, kt, chunk); then step to the next stage
            const uint4 *usrc = Ux + ((int64_t)(xi * ktiles + kt) * nchunks + chunk) * (3 * 512);
            unsigned char *udst = Ul(s % 3);
#pragma unroll
            for (int i = 0; i < ((ABL & 2) && s > 2 ? 0 : UPW); ++i) {
                const int kib = w * UPW + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(usrc + kib * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(udst + kib * 1024), 16, 0, 0);
            }
            const float *Vg = a.V + ((int64_t)xi * a.C) * a.Pp + (int64_t)pt * 128;
            unsigned char *vdst = Vraw(s % NRAW) + oct * 4096;
            if (!((ABL & 1) && s > 1))
#pragma unroll
            for (int i = 0; i < VPW; ++i) {
                const int ii = VPW * ih + i;        // row pair of the octet: channels 2 ii, 2 ii + 1
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(Vg + (int64_t)(chunk * X6_KC + oct * 8 + 2 * ii + vh) * a.Pp + 4 * vtq),
                    (__attribute__((address_space(3))) void *)(vdst + ii * 1024), 16, 0, 0);
            }
            if (++chunk == nchunks) {
                chunk = 0;
                if (++k_item < my_items) item_of(k_item, xi, pt, kt);
            }
        };
        auto split_stage = [&](int s) {             // raw V rows of this wave -> three bf16 planes of V buffer s & 1
            const unsigned char *src = Vraw(s % NRAW) + oct * 4096 + lane * 16;
            f32x4 vr[VPW];
#pragma unroll
            for (int i = 0; i < VPW; ++i) vr[i] = *reinterpret_cast<const f32x4 *>(src + (VPW * ih + i) * 1024);
            unsigned char *Vb = Vl(s % NVB);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                __bf16 p1[VPW], p2[VPW], p3[VPW];
#pragma unroll
                for (int i = 0; i < VPW; ++i) {
                    const float x = vr[i][t];
                    const __bf16 x1 = (__bf16)x;
                    const float r1 = x - (float)x1;
                    const __bf16 x2 = (__bf16)r1;
                    const float r2 = r1 - (float)x2;
                    p1[i] = x1; p2[i] = x2; p3[i] = (__bf16)r2;
                }
                // element 4 vh + ii of the piece is channel 2 ii + vh: this wave's VPW elements start at 4 vh + VPW ih
                unsigned char *dst = Vb + x6_slot(4 * vtq + t, oct) * 16 + 8 * vh + 2 * VPW * ih;
                if (NPW == 8) {
                    *reinterpret_cast<bf16x2 *>(dst) = bf16x2{p1[0], p1[1]};
                    *reinterpret_cast<bf16x2 *>(dst + X6_PLANE) = bf16x2{p2[0], p2[1]};
                    *reinterpret_cast<bf16x2 *>(dst + 2 * X6_PLANE) = bf16x2{p3[0], p3[1]};
                } else {
                    *reinterpret_cast<bf16x4 *>(dst) = bf16x4{p1[0], p1[1], p1[VPW - 2], p1[VPW - 1]};
                    *reinterpret_cast<bf16x4 *>(dst + X6_PLANE) = bf16x4{p2[0], p2[1], p2[VPW - 2], p2[VPW - 1]};
                    *reinterpret_cast<bf16x4 *>(dst + 2 * X6_PLANE) = bf16x4{p3[0], p3[1], p3[VPW - 2], p3[VPW - 1]};
                }
            }
        };
        if constexpr (SB == 2) {
            // step f: request the raw rows of stages 2f + 2, 2f + 3 (into the ring slots of 2f - 2, 2f - 1, split in the
            // previous step), wait for those of 2f, 2f + 1 (requested a step ago), split them, hand over
            if (0 < nstages) issue_stage(0);
            if (1 < nstages) issue_stage(1);
            const int nsteps = (nstages + 1) / 2;
            for (int f = 0; f <= nsteps; ++f) {
                if (f < nsteps) {
                    const int younger = (2 * f + 2 < nstages) + (2 * f + 3 < nstages);
                    if (2 * f + 2 < nstages) issue_stage(2 * f + 2);
                    if (2 * f + 3 < nstages) issue_stage(2 * f + 3);
                    if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VPW) : "memory");
                    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    split_stage(2 * f);
                    if (2 * f + 1 < nstages) split_stage(2 * f + 1);
                }
                x6p_barrier();
            }
            return;
        }
        static_assert(!TS || (!BG && NCW == 4 && NPW == 4), "TS: the production form only");
        X6pClock clk;
        if constexpr (TS) clk.last = __builtin_readcyclecounter();
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d < nstages) issue_stage(d);
        for (int s = 0; s <= nstages; ++s) {
            if (s < nstages) {
                const unsigned long long ta = x6p_now<TS>();
                if (s + PD < nstages) issue_stage(s + PD);       // lands during the consumers' stages s - 1 .. s + PD - 1
                const unsigned long long tb = x6p_now<TS>();
                const int younger = nstages - 1 - s < PD ? nstages - 1 - s : PD;       // batches issued behind stage s's
                // all but those batches: stage s has landed (vector-memory operations complete in issue order)
                if (ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (UPW + VPW)) : "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (UPW + VPW)) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW + VPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long tc = x6p_now<TS>();
                if (!((ABL & 4) && s > 1)) split_stage(s);
                if constexpr (TS) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long td = x6p_now<TS>();
                    clk.issue += tb - ta; clk.vmwait += tc - tb; clk.split += td - tc;
                }
            }
            x6p_barrier_t<TS>(clk);                 // stage s handed to the consumers, stage s - 1's buffers free again
        }
        x6p_clock_flush<TS>(clk, 1, lane);
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int li = lane & 15, lk = lane >> 4;
    constexpr int NT = NCW == 8 ? 2 : 4;         // 16-cout blocks per consumer
    const int wm = wave & 1, wn = (wave >> 1) & 1, nh = NCW == 8 ? 2 * (wave >> 2) : 0;      // nh: first MFMA row class of this wave's couts
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int a_off[4], b_off[NT];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_off[t] = x6_slot(64 * wm + 4 * li + t, lk) * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) b_off[t] = x6_slot(64 * wn + 4 * li + nh + t, lk) * 16;
    int k_item = 0, chunk = 0;
    if constexpr (BG) {
        static_assert(NCW == 4 && !(ABL & 16), "BG: four consumers, no fragment ablation");
        int kb_item = 0, chunk_b = 0, bxi, bpt, bkt;        // the stage the NEXT U fragment loads belong to
        item_of(0, bxi, bpt, bkt);
        bf16x8 bq[NT][3], afr[4][3];
        auto load_b = [&](int nt) {                 // U fragments of block nt for the stage at the cursor
            const unsigned char *usrc = reinterpret_cast<const unsigned char *>(Ux + ((int64_t)(bxi * ktiles + bkt) * nchunks + chunk_b) * (3 * 512));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[nt][pl] = *reinterpret_cast<const bf16x8 *>(usrc + pl * X6_PLANE + b_off[nt]);
        };
        // the cursor stops at the last stage: the loads are issued unconditionally (in the last stage they fetch that stage
        // again and nobody uses them), so the loop body is straight-line code and hipcc's vmcnt waits are exact
        // (behind an `if` every wait degraded to the loads just issued)
        auto step_b = [&]() {
            if (chunk_b + 1 == nchunks && kb_item + 1 == my_items) return;
            if (++chunk_b == nchunks) {
                chunk_b = 0;
                item_of(++kb_item, bxi, bpt, bkt);
            }
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_b(nt);
        step_b();
        x6p_barrier();                            // stage 0 ready
        if constexpr (AP) {
            static_assert(SB == 2, "AP: the second stage of an interval is what can be read early");
            bf16x8 afr2[4][3];
            auto read_a = [&](bf16x8 (&fr)[4][3], int st) {
                const unsigned char *Vs = Vl(st % NVB);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fr[mt][pl] = *reinterpret_cast<const bf16x8 *>(Vs + pl * X6_PLANE + a_off[mt]);
            };
            auto item_end = [&]() {
                if (++chunk == nchunks) {
                    chunk = 0;
                    int xi, pt, kt;
                    item_of(k_item++, xi, pt, kt);
                    float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * 128) * a.Pp + (int64_t)pt * 128;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float *row = Mg + (int64_t)(wn * 64 + 4 * li + nh + nt) * a.Pp + wm * 64 + 4 * lk;
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
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            // one stage on the fragments fr; early: the OTHER set is filled with the next stage's fragments behind the
            // first block's MFMAs (an odd stage count reads an unused buffer there, harmlessly)
            auto stage = [&](bf16x8 (&fr)[4][3], bf16x8 (&other)[4][3], int st, bool early) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int term = 0; term < 6; ++term) {
                        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[mt][PA[term]], bq[nt][PB[term]], acc[mt][nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_b(nt);
                    if (nt == 0 && early) read_a(other, st + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                step_b();
                item_end();
            };
            for (int s = 0; s < nstages; s += 2) {
                read_a(afr, s);
                stage(afr, afr2, s, true);
                if (s + 1 < nstages) stage(afr2, afr, s + 1, false);
                x6p_barrier();                    // done with stages s, s + 1; the next two are ready
            }
            return;
        }
        for (int s = 0; s < nstages; ++s) {
            const unsigned char *Vs = Vl(s % NVB);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afr[mt][pl] = *reinterpret_cast<const bf16x8 *>(Vs + pl * X6_PLANE + a_off[mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // per accumulator the same order of terms as the LDS form: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[mt][PA[term]], bq[nt][PB[term]], acc[mt][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks all twelve loads to the end of the stage)
                load_b(nt);                       // next stage's fragments of this block: in flight for the rest of the stage
                __builtin_amdgcn_sched_barrier(0);
            }
            step_b();
            if (++chunk == nchunks) {
                chunk = 0;
                int xi, pt, kt;
                item_of(k_item++, xi, pt, kt);
                float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * 128) * a.Pp + (int64_t)pt * 128;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float *row = Mg + (int64_t)(wn * 64 + 4 * li + nh + nt) * a.Pp + wm * 64 + 4 * lk;
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
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (SB == 1 || (s & 1) || s + 1 == nstages) x6p_barrier();       // done with stage s (and s - 1); the next ones are ready
        }
        return;
    }
    X6pClock clk;
    if constexpr (TS) clk.last = __builtin_readcyclecounter();
    x6p_barrier_t<TS>(clk);                       // stage 0 ready
    bf16x8 bfrag[NT][3], afr[4][3];
    for (int s = 0; s < nstages; ++s) {
        const unsigned char *Vs = Vl(s & 1), *Us = Ul(s % 3);
        if (!(ABL & 16) || s == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bfrag[nt][pl] = *reinterpret_cast<const bf16x8 *>(Us + pl * X6_PLANE + b_off[nt]);
        }
        if constexpr (A2) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afr[mt][pl] = *reinterpret_cast<const bf16x8 *>(Vs + pl * X6_PLANE + a_off[mt]);
            __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks every read back in front of its first use)
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (!A2 && (!(ABL & 16) || s == 0)) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afr[(ABL & 16) ? mt : 0][pl] = *reinterpret_cast<const bf16x8 *>(Vs + pl * X6_PLANE + a_off[mt]);
            }
            const bf16x8 *af = afr[(A2 || (ABL & 16)) ? mt : 0];
            // smallest terms first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (ABL & 8) acc[mt][nt][0] += (float)af[PA[term]][0] + (float)bfrag[nt][PB[term]][1];
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[term]], bfrag[nt][PB[term]], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        if (++chunk == nchunks) {
            chunk = 0;
            int xi, pt, kt;
            item_of(k_item++, xi, pt, kt);
            float *Mg = a.M + ((int64_t)xi * a.Kp + (int64_t)kt * 128) * a.Pp + (int64_t)pt * 128;
            if (ABL & 64) {
                // M stores in 64-byte runs: a lane holds 16 consecutive tiles of one cout row as four 16-byte pieces
                // (piece 4 lk + r); stored as they are, one instruction writes pieces 64 bytes apart.  A 4 x 4 transpose of
                // the pieces over the four lanes li, li + 16, li + 32, li + 48 (v_permlane16_swap / v_permlane32_swap, two
                // stages, no LDS) gives lane lk the pieces 4 j + lk: instruction j then writes 64 contiguous bytes per row.
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float *row = Mg + (int64_t)(wn * 64 + 4 * li + nh + nt) * a.Pp + wm * 64 + 4 * lk;
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
            } else
            if (!(ABL & 32) || acc[0][0][0] == 12345.678f)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float *row = Mg + (int64_t)(wn * 64 + 4 * li + nh + nt) * a.Pp + wm * 64 + 16 * lk;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4 *>(row + 4 * r) = f32x4{acc[0][nt][r], acc[1][nt][r], acc[2][nt][r], acc[3][nt][r]};
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        x6p_barrier_t<TS>(clk);                   // done with stage s; stage s + 1 ready
    }
    x6p_clock_flush<TS>(clk, 0, lane);
}

// per-role cycle totals of the TS form since the last reset (segnet_kernels.hpp)
void x6p_read_stamps(unsigned long long out[8], bool reset) {
    SIVO_HIP(hipDeviceSynchronize());
    SIVO_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(x6p_stamps), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        SIVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(x6p_stamps), zero, sizeof zero));
    }
}

// 1-D output transform A^T m
__device__ __forceinline__ void wino4_at(const float m0, const float m1, const float m2, const float m3, const float m4,
                                         const float m5, float *s) {
    const float p12 = m1 + m2, q12 = m1 - m2, p34 = m3 + m4, q34 = m3 - m4;
    s[0] = m0 + p12 + p34;
    s[1] = q12 + 2.f * q34;
    s[2] = p12 + 4.f * p34;
    s[3] = q12 + 8.f * q34 + m5;
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
    const float sc = a.ep_scale[co], sh = a.ep_shift[co];
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
__global__ __launch_bounds__(1024) void wino4_bridge_kernel(Wino4Args a, float *Vnext) {
    extern __shared__ float plane[];            // (4*th + 2) rows x (W + 4) floats; image pixel (y, x) at [y + 1][x + 1]
    const int n = blockIdx.x, co = blockIdx.y;
    const int RS = a.W + 4, rows = 4 * a.th + 2, ntile = a.th * a.tw;
    for (int i = threadIdx.x; i < rows * RS; i += blockDim.x) plane[i] = 0.f;
    __syncthreads();
    const float sc = a.ep_scale[co], sh = a.ep_shift[co];
    const int64_t xs_m = (int64_t)a.Kp * a.Pp, xs_v = (int64_t)a.K * a.Pp;
    for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
        const int tx = t % a.tw, ty = t / a.tw;
        const float *src = a.M + (int64_t)co * a.Pp + (int64_t)n * ntile + t;
        float tt[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = src[(int64_t)(i * 6 + j) * xs_m];
            float s4[4];
            wino4_at(m[0], m[1], m[2], m[3], m[4], m[5], s4);
#pragma unroll
            for (int i = 0; i < 4; ++i) tt[i][j] = s4[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 4 * ty + i;
            if (y >= a.H) break;                 // rows below the image stay zero
            float v[4];
            wino4_at(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = v[r] * sc + sh;
                if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            const int x = 4 * tx;
            if (a.drop_site >= 0) {
                const uint32_t e = (uint32_t)((co * a.H + y) * a.W + x);
                const uint32_t w = wino4_dropout_word(e, (uint32_t)a.drop_site, (uint32_t)(a.sample0 + n), a.seed) >> (e & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? v[r] * 2.f : 0.f;
            }
            float *dst = plane + (y + 1) * RS + x + 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = v[r];
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
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[(int64_t)(i * 6 + j) * xs_v] = row[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool wino4_supported(int ks, int cin, int cout, int H, int W) {
    // below 128 channels a transform-position GEMM is HBM-bound on V and M (C*K / (2 (C + K)) < 32 flop/byte; couts are
    // padded to 128 as well) and the fused F(2x2) kernel wins — measured on the Standard shapes: 256->128 0.69 vs 0.92 ms,
    // 128->128 1.52 vs 1.74 ms, but 128->64 1.37 vs 0.95 ms and 64->64 3.2 vs 1.9 ms.  SIVO_WINO4_MINC moves the threshold.
    static const int minc = std::getenv("SIVO_WINO4_MINC") ? std::atoi(std::getenv("SIVO_WINO4_MINC")) : 128;
    return ks == 3 && cin % G_KC == 0 && cin >= minc && cout >= minc && W % 4 == 0 && H >= 4 && W >= 4;
}

int wino4_cout_pad(int cout) { return (cout + G_BN - 1) / G_BN * G_BN; }

// Caffe (Cout,Cin,3,3) -> U [36][Cin][Kp] = G g G^T, evaluated in double and rounded once
void wino4_pack_weights(const float *W, int cin, int cout, std::vector<float> &out, int *cout_pad) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
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

size_t wino4_bridge_lds_bytes(int H, int W) { return (size_t)(4 * ((H + 3) / 4) + 2) * (W + 4) * sizeof(float); }

// One F(4x4,3x3) layer.  `group` samples per pass over the workspace.  plan (optional) chains layers without going
// through HBM with the activation: V / M / Vnext are caller-chosen disjoint buffers, skip_input says V already holds
// this layer's transformed input (written by the previous layer's bridge), bridge replaces the output transform by
// wino4_bridge_kernel writing the NEXT layer's V into Vnext (the plain output `c.out` is then not produced).  A plan
// requires all samples in one group.
// ev (optional, profiling): 4 events per group, recorded before the input transform, after it, after the GEMM and
// after the output transform / bridge (gemm_only_events: only the two around the GEMM).
void launch_conv_wino4(const ConvArgs &c, float *workspace, int group, hipStream_t s, hipEvent_t *ev, bool gemm_only_events,
                       const Wino4Plan *plan) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        for (const void *f : {(const void *)wino4_gemm_x6p_kernel<0>, (const void *)wino4_gemm_x6p_kernel<1>, (const void *)wino4_gemm_x6p_kernel<2>,
                              (const void *)wino4_gemm_x6p_kernel<3>, (const void *)wino4_gemm_x6p_kernel<4>, (const void *)wino4_gemm_x6p_kernel<7>,
                              (const void *)wino4_gemm_x6p_kernel<8>, (const void *)wino4_gemm_x6p_kernel<16>, (const void *)wino4_gemm_x6p_kernel<23>,
                              (const void *)wino4_gemm_x6p_kernel<32>, (const void *)wino4_gemm_x6p_kernel<64>, (const void *)wino4_gemm_x6p_kernel<64, 8>, (const void *)wino4_gemm_x6p_kernel<64, 4, 8>, (const void *)wino4_gemm_x6p_kernel<64, 4, 4, true>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS);
        (void)hipFuncSetAttribute((const void *)wino4_gemm_x6p_kernel<64, 4, 4, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS_SB2);
        (void)hipFuncSetAttribute((const void *)wino4_gemm_x6p_kernel<64, 4, 4, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS_SB2);
        (void)hipFuncSetAttribute((const void *)wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS);
        (void)hipFuncSetAttribute((const void *)wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS);
        (void)hipFuncSetAttribute((const void *)wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X6P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_gemm_x6_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS);
        for (const void *f : {(const void *)wino4_gemm_x6_kernel<1>, (const void *)wino4_gemm_x6_kernel<2>, (const void *)wino4_gemm_x6_kernel<4>,
                              (const void *)wino4_gemm_x6_kernel<8>, (const void *)wino4_gemm_x6_kernel<16>, (const void *)wino4_gemm_x6_kernel<7>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_gemm_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_STAGE * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_bridge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    Wino4Args a{};
    a.C = c.Cin; a.K = c.Cout; a.Kp = c.CoutPad; a.H = c.H; a.W = c.W;
    a.th = (c.H + 3) / 4; a.tw = c.W / 4;
    a.U = c.wt; a.ep_scale = c.ep_scale; a.ep_shift = c.ep_shift;
    a.relu = c.relu; a.drop_site = c.drop_site; a.seed = c.seed; a.in_sample_stride = c.in_sample_stride;
    a.pool_out = c.pool_out; a.pool_mask = c.pool_mask; a.pool_drop_site = c.pool_drop_site; a.Ho = (c.H + 1) / 2; a.Wo = (c.W + 1) / 2;
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
            if (a.mask) hipLaunchKernelGGL(wino4_input_kernel<true>, dim3(pblocks, (unsigned)a.C), dim3(W4_TIN), 0, s, a);
            else hipLaunchKernelGGL(wino4_input_kernel<false>, dim3(pblocks, (unsigned)a.C), dim3(W4_TIN), 0, s, a);
        }
        if (e) (void)hipEventRecord(e[1], s);
        // tile choice: the largest one that still gives every CU ~4 workgroups (256 CUs; SIVO_WINO4_TILE forces 0/1/2)
        static const int force_tile = std::getenv("SIVO_WINO4_TILE") ? std::atoi(std::getenv("SIVO_WINO4_TILE")) : -1;
        auto nblocks = [&](int bm, int bn) { return (int64_t)36 * ((a.P + bm - 1) / bm) * (a.Kp / bn); };
        static const int min_blocks = std::getenv("SIVO_WINO4_MINBLOCKS") ? std::atoi(std::getenv("SIVO_WINO4_MINBLOCKS")) : 1024;
        int tile = nblocks(128, 128) >= min_blocks ? 0 : nblocks(64, 128) >= min_blocks ? 1 : 2;
        if (force_tile >= 0) tile = force_tile;
        // bf16x6 (128 x 128 tiles only): whenever the layer has the split weights and the launch is not tiny
        static const int x6_min_blocks = std::getenv("SIVO_X6_MINBLOCKS") ? std::atoi(std::getenv("SIVO_X6_MINBLOCKS")) : 128;
        if (c.wt_x6 && nblocks(128, 128) >= x6_min_blocks) {
            const int pt6 = (a.P + 127) / 128, kt6 = a.Kp / 128, pairs6 = (36 * pt6 + 7) / 8;
            const dim3 g6((unsigned)(pairs6 * kt6 * 8));
            const uint4 *u6 = reinterpret_cast<const uint4 *>(c.wt_x6);
            // SIVO_X6=flat: the two-workgroups-per-CU kernel without role separation (kept for comparison and ablations)
            static const bool x6_flat = std::getenv("SIVO_X6") && std::string(std::getenv("SIVO_X6")) == "flat";
            static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
            const int abl = (c.variant >> 12) & 31;
            (void)abl;
            if (!x6_flat) {
                const dim3 gp((unsigned)((n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8));        // one persistent workgroup per CU, a multiple of the 8 XCDs
                // M stores transposed into 64-byte runs (ABL bit 64): GEMM time of a frame 5.51 -> 5.17 ms; SIVO_X6_MSTORE=0: as held
                static const bool mstore64 = !(std::getenv("SIVO_X6_MSTORE") && std::atoi(std::getenv("SIVO_X6_MSTORE")) == 0);
                // SIVO_X6_CONSUMERS=8: two consumer waves per SIMD (12-wave workgroup)
                static const bool x6_consumers8 = std::getenv("SIVO_X6_CONSUMERS") && std::atoi(std::getenv("SIVO_X6_CONSUMERS")) == 8;
                // SIVO_X6_PRODUCERS=8: eight producer waves (12-wave workgroup)
                static const bool x6_producers8 = std::getenv("SIVO_X6_PRODUCERS") && std::atoi(std::getenv("SIVO_X6_PRODUCERS")) == 8;
                switch (((c.variant >> 12) & 63) == 0 && mstore64 ? 64 : ((c.variant >> 12) & 63)) {
                    case 64:
                        if (x6_producers8) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 8>), gp, dim3(768), X6P_LDS, s, a, u6, pt6, kt6);
                        else if (x6_consumers8) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 8>), gp, dim3(768), X6P_LDS, s, a, u6, pt6, kt6);
                        else {
                            const char *pe = std::getenv("SIVO_X6_PRIO");          // experiment: wave priorities of the two roles
                            const char *bg = std::getenv("SIVO_X6_BGLOBAL");       // experiment: U fragments from global memory
                            const int kt_prio = kt6 | ((pe ? std::atoi(pe) & 15 : 0) << 16);
                            const char *ts = std::getenv("SIVO_X6_STAMPS");        // diagnostic: per-role work / wait cycles (sivo_debug_x6_stamps)
                            const char *a2 = std::getenv("SIVO_X6_AFRAG");         // experiment: four V-fragment register sets
                            const bool a2on = a2 && std::atoi(a2) == 1;
                            if (ts && std::atoi(ts) == 1 && a2on) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, true, true>), gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt_prio);
                            else if (ts && std::atoi(ts) == 1) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, true>), gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt_prio);
                            else if (a2on) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, false, 1, false, false, true>), gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt_prio);
                            else if (bg && std::atoi(bg) == 3) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, true, 2, true>), gp, dim3(512), X6P_LDS_SB2, s, a, u6, pt6, kt_prio);
                            else if (bg && std::atoi(bg) == 2) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, true, 2>), gp, dim3(512), X6P_LDS_SB2, s, a, u6, pt6, kt_prio);
                            else if (bg && std::atoi(bg) == 1) hipLaunchKernelGGL((wino4_gemm_x6p_kernel<64, 4, 4, true>), gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt_prio);
                            else hipLaunchKernelGGL(wino4_gemm_x6p_kernel<64>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt_prio);
                        }
                        break;
                    case 1: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<1>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 2: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<2>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 3: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<3>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 4: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<4>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 7: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<7>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 8: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<8>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 16: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<16>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 23: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<23>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    case 32: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<32>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6); break;
                    default: hipLaunchKernelGGL(wino4_gemm_x6p_kernel<0>, gp, dim3(512), X6P_LDS, s, a, u6, pt6, kt6);
                }
            } else
            switch (abl) {          // ablations for tools/x6_probe.py; 0 in production
                case 1: hipLaunchKernelGGL(wino4_gemm_x6_kernel<1>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                case 2: hipLaunchKernelGGL(wino4_gemm_x6_kernel<2>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                case 4: hipLaunchKernelGGL(wino4_gemm_x6_kernel<4>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                case 7: hipLaunchKernelGGL(wino4_gemm_x6_kernel<7>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                case 8: hipLaunchKernelGGL(wino4_gemm_x6_kernel<8>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                case 16: hipLaunchKernelGGL(wino4_gemm_x6_kernel<16>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6); break;
                default: hipLaunchKernelGGL(wino4_gemm_x6_kernel<0>, g6, dim3(256), X6_LDS, s, a, u6, pt6, kt6);
            }
            tile = -1;
        }
        const int bm = tile == 0 ? 128 : 64, bn = tile == 2 ? 64 : 128;
        const int pt_n = (a.P + bm - 1) / bm, kt_n = a.Kp / bn, pairs8 = (36 * pt_n + 7) / 8;
        const dim3 ggrid((unsigned)(pairs8 * kt_n * 8));
        const size_t glds = (size_t)2 * G_KC * (bm + bn) * 4;
        if (tile < 0) {}
        else if (tile == 0) hipLaunchKernelGGL((wino4_gemm_kernel<128, 128>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
        else if (tile == 1) hipLaunchKernelGGL((wino4_gemm_kernel<64, 128>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
        else hipLaunchKernelGGL((wino4_gemm_kernel<64, 64>), ggrid, dim3(256), glds, s, a, pt_n, kt_n);
        if (e) (void)hipEventRecord(e[2], s);
        if (plan && plan->bridge) {
            const int ntile = a.th * a.tw;
            const int nthr = ntile >= 1024 ? 1024 : (ntile + 63) / 64 * 64;
            hipLaunchKernelGGL(wino4_bridge_kernel, dim3((unsigned)a.n, (unsigned)a.K), dim3(nthr), wino4_bridge_lds_bytes(a.H, a.W), s, a, plan->Vnext);
        } else {
            if (a.pool_out) hipLaunchKernelGGL(wino4_output_kernel<true>, dim3(pblocks, (unsigned)a.K), dim3(W4_TIN), 0, s, a);
            else hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(pblocks, (unsigned)a.K), dim3(W4_TIN), 0, s, a);
        }
        if (e && !gemm_only_events) (void)hipEventRecord(e[3], s);
    }
}

}  // namespace sivo
