// conv_wino4_h3.hip — the batched GEMM of the three-kernel F(4x4,3x3) path on the fp16 matrix cores: "f16x3".
//
//   M_xi[k][p] = sum_c U_xi[c][k] * V_xi[c][p]      36 independent GEMMs, fp32 in / fp32 out (conv_wino4.hip)
//
// An fp32 value x (scaled by a power of two so that it sits well inside the fp16 range) is hi + lo with hi = fp16(x),
// lo = fp16(x - hi), exact to 2^-22 |x|; a product of two fp16 values is exact in fp32.  So
//     v u  =  v_lo u_hi + v_hi u_lo + v_hi u_hi  +  O(2^-22 |v u|)
// is THREE v_mfma_f32_32x32x16_f16 per 16 channels with fp32 accumulation — half the matrix-core work of the bf16x6 form
// (three bf16 planes, six products) at the same accuracy: emulated on Winograd-domain data (K = 512) the rms error is
// 1.9e-7 relative against 2.4e-7 for bf16x6 and 4.0e-7 for the sequential fp32 FMA chain (fewer roundings of the
// accumulator).  What fp16 does not have is bf16's range: the transform kernels therefore write V already multiplied by a
// per-layer power of two chosen at load time (calibration frame, 2^8 of headroom to 65504, an overflow flag that sends the
// frame to the bf16x6 path), the weights are scaled per layer on the host, and the output transform multiplies the two
// powers back out inside its per-channel affine — all exact.
//
// Data.  V' [36][C][Pp] has the layout the transform kernels always wrote, with 4-byte elements that are now the pair
// (hi | lo << 16) instead of one fp32: no extra HBM bytes, no split in this kernel.  U' is split once on the host and
// stored as the LDS image of its stages.  M stays fp32 [36][Kp][Pp].
//
// Kernel.  One persistent 512-thread workgroup per CU walks (position, tile group, cout group) items in an XCD-aware
// order; workgroup tile BM tiles x BN couts (256 x 256; 128 x 256 for launches with few tiles; 256 x 128 for Kp = 128),
// 8 waves = 2 per SIMD, wave tile (BM / WT) tiles x 64 couts as 32 x 32 MFMA blocks (128 / 64 accumulator registers).
// A stage is 32 channels: 2 k-steps of v_mfma_f32_32x32x16_f16, 48 (24) MFMAs per wave against 24 (16) ds_read_b128.
// Staging (two V' and three U' buffers in LDS, loads two stages deep — see the kernel), ONE barrier per stage:
//   * U': LDS-DMA issued in inline assembly (lds_dma.hpp: invisible to hipcc's waitcnt pass), 1 KiB pieces that are
//     already in fragment order — the 64 lanes of a ds_read_b128 read 1 KiB contiguous;
//   * V': every lane loads the 8 (16) channel values of its tile with buffer loads (row offset in an SGPR, no address
//     arithmetic), de-interleaves hi and lo with v_perm_b32 (one per register) and writes the two fragment pieces of a
//     channel octet with ds_write_b128 — two stages behind the loads, right after the barrier ("write late, re-issue at
//     once"): the loads of stages s + 2, s + 3 and the DMA of stages s + 1, s + 2 are in flight while stage s is multiplied.
// Fragment order = [32-row block][plane][channel octet][row]: every ds_read_b128 / ds_write_b128 touches consecutive
// 16-byte pieces in lane order, conflict-free without any swizzle.
// C/D mapping of the 32 x 32 MFMA: column = lane & 31 = tile, so one accumulator register of a wave is two 128-byte
// runs of M; rows = couts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "lds_dma.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int H3_KC = 32;                    // channels per stage

struct H3Args {
    const uint32_t *V;       // packed (hi | lo << 16) [36][C][Pp]
    const unsigned char *U;  // [36][Kp / 32][C / 32][plane][octet][row][8] fp16
    float *M;                // [36][Kp][Pp]
    int C, Kp, P, Pp;
    int ptiles, ktiles;      // tile groups (BM) and cout groups (BN) of the launch
};

// LDS-DMA with a scalar base: lane l copies the 16 bytes at sbase + voff to LDS address lds_byte_addr + 16 l.
__device__ __forceinline__ void h3_dma16(const void *sbase, uint32_t voff, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

// Pipeline, two stages deep (round 3, after the ablations of tools/h3_probe.py: with ONE stage of loads in flight the memory
// side of a stage — 64 KB per CU — took as long as its MFMAs and the two did not overlap: 0.31 ms as built against 0.20 ms
// with the loads removed and 0.21 ms with the MFMAs removed, conv4_2; bytes in flight / latency is what a CU gets):
//   iteration s:  wait until only the operations issued in iteration s - 1 are still in flight (vector-memory operations
//                 complete in issue order)  ->  barrier  ->  M stores of an item that ended with stage s - 1  ->  V'(s + 1)
//                 registers -> LDS  ->  issue V'(s + 3) loads into those registers  ->  issue U'(s + 2) DMA  ->  multiply stage s.
// LDS: two V' buffers + THREE U' buffers ((2 BM + 3 BN) * 128 bytes: 160 KB for 256 x 256); two V' register sets, indexed
// statically (the stage loop is unrolled by two).  The M stores of an item are issued at the top of the NEXT iteration,
// ahead of its loads, so that they have a whole stage to drain and never stand between a wait and the loads it leaves in flight.
// ABL (diagnostic builds only, -DSIVO_DIAG -> libsivo_hip_diag.so, tools/h3_probe.py; results are wrong by construction):
// 1 no V' loads after the prologue, 2 no U' DMA after the prologue, 4 no M stores, 8 no MFMAs, 16 V' by LDS-DMA: what the kernel would
// cost if the transform kernels wrote V' as the LDS image of its stages ([position][32-tile block][stage][plane][octet][tile][8] fp16,
// as U' is) — the wave of a tile block copies four 1 KiB pieces per stage into V' buffer (s + 1) & 1 right behind the barrier of stage s
// (two V' buffers: one stage of cover), no V' registers, no v_perm, no ds_write; the same bytes of the same slab in another order.
// 32 start skew: workgroup w of an XCD sleeps (w & 3) quarter items (~ nst / 4 stage times) before its first stage.
// 64 M stores with the non-temporal hint, 128 V' loads with it (results stay right under these two).
// 512 no stage barrier (FORM 3), 1024 no fragment reads after the first two (FORM 3): with 7 they split the MFMA + LDS time into its parts.
// FORM 1 (round 6, the product's): the memory side of a stage is issued INSIDE its multiply phase.  In the phased form (FORM 0, kept for
// A/B in the diagnostic build) every wave did, behind the barrier, registers -> LDS (16 v_perm, 4 ds_write_b128), 16 buffer loads, 4 LDS-DMA
// (each with its M0 save / restore) and only then its first fragment reads — all eight waves at once, so the matrix cores of the CU stood
// idle for that head of every stage (the ablations' "stage = MFMA time + ingest time").  Now: barrier -> fragment reads of k-step 0 ->
// registers -> LDS under the reads' latency -> MFMA, two loads, MFMA, two loads ... MFMA, DMA ...; the loads and DMA are issued
// unconditionally (the cursors clamp at the last item: the last three stages fetch bytes nobody consumes) so that the hot path has no
// branch around a filler and the wait at the top is always vmcnt(NV + NU).  Same MFMAs in the same order per accumulator: M bit-identical.
// FORM 2 (round 6, second half; the product's): FORM 1 with the V' loads of a 256-tile item as EIGHT 8-byte loads per lane and stage
// instead of sixteen 4-byte ones — lane = (tile pair m of 16, channel octet o of 4) of its wave's 32-tile block: one load per channel of the
// octet brings tiles 2m, 2m + 1; the same 16 v_perm and 4 ds_write_b128 (two adjacent fragment pieces per plane).  The fragment order, the
// MFMAs and their order are FORM 1's: M bit-identical.  (The V' register path costs four times the U' LDS-DMA for the same bytes,
// half the instructions on it — worth 0 - 2 %, so the instruction count is not what makes it expensive.)  128-tile and
// 256 x 128 items keep FORM 1's loads.
template <int BM, int BN, int ABL = 0, int FORM = 3>
__global__ __launch_bounds__(512, 2) void wino4_gemm_h3_kernel(H3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_h3[];
    constexpr int WC = BN / 64, WT = 8 / WC;                     // wave grid: couts (64 per wave) x tiles
    constexpr int TB = BM / WT / 32;                             // 32-tile MFMA blocks per wave (4 or 2); 2 cout blocks
    constexpr int VBYTES = BM * 128, UBYTES = BN * 128;
    constexpr int U0 = 2 * VBYTES;                               // LDS: V' buffers 0, 1 then U' buffers 0, 1, 2
    constexpr int NQ = BM == 256 ? 2 : 1;                        // channel octets each lane brings in per stage
    constexpr int NU = BN / 64;                                  // 1 KiB U pieces each wave copies per stage (4 or 2)
    constexpr bool V2 = FORM >= 2 && BM == 256 && BN == 256;     // 8-byte V' loads (two tiles per lane); 256 x 128 items measured 1 % slower with them
    constexpr int NV = V2 ? 8 : NQ * 8;                          // V' load instructions per lane and stage
    static_assert(BM == 256 || BM == 128, "tile");
    static_assert(BN == 256 || BN == 128, "tile");
    static_assert(TB == 4 || TB == 2, "wave tile");
    static_assert(2 * VBYTES + 3 * UBYTES <= 160 * 1024, "LDS");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int nst = a.C / H3_KC;

    // items of this XCD: a contiguous range of the (position, tile group, cout group) list (cout groups of one V tile
    // adjacent, positions in order: U_xi stays in this L2 while the XCD works through its tile groups); the workgroups of
    // the XCD take them round-robin
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int per_pos = a.ptiles * a.ktiles, nitems = 36 * per_pos;
    const int lo_it = (int)(((int64_t)nitems * xcd) >> 3), hi_it = (int)(((int64_t)nitems * (xcd + 1)) >> 3);
    if (lo_it + wg >= hi_it) return;
    const int my_items = (hi_it - lo_it - wg + per_xcd - 1) / per_xcd;
    const int total = my_items * nst;
    if constexpr ((ABL & 32) != 0) {          // start skew: the workgroups of an XCD in four phases a quarter item apart (are the M stores expensive because all CUs issue them at once?)
        for (int i = 0; i < (wg & 3) * nst / 4; ++i) __builtin_amdgcn_s_sleep(110);
    }

    struct Cursor {          // a stage = (item, chunk); everything here is wave-uniform
        int k = 0, chunk = 0, xi = 0, pt = 0, kt = 0;
    };
    auto locate = [&](Cursor &c) __attribute__((always_inline)) {
        const int it = lo_it + wg + c.k * per_xcd;
        c.xi = it / per_pos;
        const int rem = it - c.xi * per_pos;
        c.pt = rem / a.ktiles;
        c.kt = rem - c.pt * a.ktiles;
    };
    auto advance = [&](Cursor &c) __attribute__((always_inline)) {
        if (++c.chunk == nst) {
            c.chunk = 0;
            if (++c.k < my_items) locate(c);
        }
    };

    // ---- V' staging: lane (tile ln of tile block vtb, octets vo0 + 2 q + lh) -------------------------------------------
    const int vtb = BM == 256 ? wave : wave >> 1, vo0 = BM == 256 ? 0 : 2 * (wave & 1);
    const uint32_t v_lane_off = (uint32_t)(((int64_t)(8 * (vo0 + lh)) * a.Pp + vtb * 32 + ln) * 4);
    const uint32_t v_slab_bytes = (uint32_t)((int64_t)a.C * a.Pp * 4);
    typedef uint32_t VSet[NQ][8];
    // The V' loads are issued in inline assembly: hipcc's waitcnt pass cannot keep two register sets of loads in flight
    // beside the (to it invisible) LDS-DMA — it drained every outstanding load, the other set's included, in front of the
    // first use of a set (seen in the .s of the first two-stage form of this kernel: vmcnt(14) ... vmcnt(0) where vmcnt(20) was
    // right) — so the counting is done by hand: vector-memory operations complete in issue order, this wave issues per
    // iteration [M stores of an item end] [NV loads] [NU DMA], and the one wait at the top of an iteration leaves exactly the
    // previous iteration's loads + DMA in flight.  The destinations are the elements of the register set themselves (no
    // temporaries: a compiler copy between load and wait would copy stale bytes — checked in the .s: no v_mov reads them).
    auto load_v = [&](const Cursor &c, VSet &r) __attribute__((always_inline)) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.V + (int64_t)c.xi * a.C * a.Pp);
        const i32x4 rs = {(int)(uint32_t)base, (int)(uint32_t)((base >> 32) & 0xffffu), (int)v_slab_bytes, 0x00020000};
        const uint32_t vo = v_lane_off + (uint32_t)c.pt * (BM * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t so = (uint32_t)((int64_t)(c.chunk * H3_KC + 16 * q + e) * a.Pp * 4);
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(r[q][e]) : "v"(vo), "s"(rs), "s"(so) : "memory");
            }
    };
    // the loads of a set have landed (the caller's s_waitcnt): from here on its registers may be read
    auto landed = [&](VSet &r) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(r[q][e]));
    };
    const uint32_t v_lds_off = (uint32_t)(vtb * 4096 + (vo0 + lh) * 512 + ln * 16);
    auto write_v = [&](int buf, const VSet &r) __attribute__((always_inline)) {
        unsigned char *dst = lds_h3 + buf * VBYTES + v_lds_off;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            u32x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = __builtin_amdgcn_perm(r[q][2 * j + 1], r[q][2 * j], 0x05040100u);
                lo[j] = __builtin_amdgcn_perm(r[q][2 * j + 1], r[q][2 * j], 0x07060302u);
            }
            *reinterpret_cast<u32x4 *>(dst + q * 1024) = hi;
            *reinterpret_cast<u32x4 *>(dst + q * 1024 + 2048) = lo;
        }
    };

    // ---- V' staging, V2: lane = (tile pair v2_m, octet v2_o) of tile block `wave`; register e of the set = channel e of the octet, .x / .y = tiles 2m / 2m + 1
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 V2Set[8];
    const int v2_m = lane & 15, v2_o = lane >> 4;
    const uint32_t v2_lane_off = (uint32_t)(((int64_t)(8 * v2_o) * a.Pp + wave * 32 + 2 * v2_m) * 4);
    const uint32_t v2_lds_off = (uint32_t)(wave * 4096 + v2_o * 512 + 2 * v2_m * 16);
    auto load_v2_one = [&](const Cursor &c, V2Set &r, const int e) __attribute__((always_inline)) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.V + (int64_t)c.xi * a.C * a.Pp);
        const i32x4 rs = {(int)(uint32_t)base, (int)(uint32_t)((base >> 32) & 0xffffu), (int)v_slab_bytes, 0x00020000};
        const uint32_t vo = v2_lane_off + (uint32_t)c.pt * (BM * 4);
        const uint32_t so = (uint32_t)((int64_t)(c.chunk * H3_KC + e) * a.Pp * 4);
        if constexpr ((ABL & 128) != 0) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=&v"(r[e]) : "v"(vo), "s"(rs), "s"(so) : "memory");
        else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(r[e]) : "v"(vo), "s"(rs), "s"(so) : "memory");
    };
    auto landed2 = [&](V2Set &r) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(r[e]));
    };
    auto write_v2 = [&](int buf, const V2Set &r) __attribute__((always_inline)) {
        unsigned char *dst = lds_h3 + buf * VBYTES + v2_lds_off;
        u32x4 hi0, lo0, hi1, lo1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi0[j] = __builtin_amdgcn_perm(r[2 * j + 1].x, r[2 * j].x, 0x05040100u);
            lo0[j] = __builtin_amdgcn_perm(r[2 * j + 1].x, r[2 * j].x, 0x07060302u);
            hi1[j] = __builtin_amdgcn_perm(r[2 * j + 1].y, r[2 * j].y, 0x05040100u);
            lo1[j] = __builtin_amdgcn_perm(r[2 * j + 1].y, r[2 * j].y, 0x07060302u);
        }
        *reinterpret_cast<u32x4 *>(dst) = hi0;
        *reinterpret_cast<u32x4 *>(dst + 16) = hi1;
        *reinterpret_cast<u32x4 *>(dst + 2048) = lo0;
        *reinterpret_cast<u32x4 *>(dst + 2048 + 16) = lo1;
    };
    // one ds_write_b128 of a set at a time (FORM 3 deals them over MFMA shadows): NPARTS pieces
    constexpr int NPARTS = V2 ? 4 : 2 * NQ;
    auto write_part = [&](int buf, const auto &r, const int j) __attribute__((always_inline)) {
        if constexpr (V2) {
            unsigned char *dst = lds_h3 + buf * VBYTES + v2_lds_off + (j >> 1) * 2048 + (j & 1) * 16;
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t x1 = (j & 1) ? r[2 * i + 1].y : r[2 * i + 1].x, x0 = (j & 1) ? r[2 * i].y : r[2 * i].x;
                w[i] = __builtin_amdgcn_perm(x1, x0, (j >> 1) ? 0x07060302u : 0x05040100u);
            }
            *reinterpret_cast<u32x4 *>(dst) = w;
        } else {
            const int q = j >> 1;
            unsigned char *dst = lds_h3 + buf * VBYTES + v_lds_off + q * 1024 + (j & 1) * 2048;
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_perm(r[q][2 * i + 1], r[q][2 * i], (j & 1) ? 0x07060302u : 0x05040100u);
            *reinterpret_cast<u32x4 *>(dst) = w;
        }
    };
    // the register set of a stage and its operations, by form
    using Set = std::conditional_t<V2, V2Set, VSet>;
    auto load_set = [&](const Cursor &c, Set &r) __attribute__((always_inline)) {
        if constexpr (V2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) load_v2_one(c, r, e);
        } else load_v(c, r);
    };
    auto landed_set = [&](Set &r) __attribute__((always_inline)) { if constexpr (V2) landed2(r); else landed(r); };
    auto write_set = [&](int buf, const Set &r) __attribute__((always_inline)) { if constexpr (V2) write_v2(buf, r); else write_v(buf, r); };

    // ---- U' staging: piece g = wave * NU + j of the stage: cout block g >> 2, quarter g & 3 ------------------------------
    const uint32_t lds_base = lds_addr_uniform(lds_h3);
    uint32_t u_voff[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int g = wave * NU + j;
        u_voff[j] = (uint32_t)((g >> 2) * nst * 4096 + (g & 3) * 1024 + lane * 16);
    }
    auto dma_u = [&](const Cursor &c, int ubuf) __attribute__((always_inline)) {
        const unsigned char *sb = a.U + ((int64_t)(c.xi * (a.Kp / 32) + c.kt * (BN / 32)) * nst + c.chunk) * 4096;
#pragma unroll
        for (int j = 0; j < NU; ++j) h3_dma16(sb, u_voff[j], lds_base + U0 + ubuf * UBYTES + (wave * NU + j) * 1024);
    };

    // ---- MFMA phase ----------------------------------------------------------------------------------------------------
    const int wc = wave % WC, wt = wave / WC;
    f32x16 acc[2][TB];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
    };
    clear_acc();
    const uint32_t a_off = (uint32_t)(wc * 2 * 4096 + lane * 16), b_off = (uint32_t)(wt * TB * 4096 + lane * 16);
    // accumulator register r of block (c, t) is M[cout 32 (2 wc + c) + 8 (r >> 2) + 4 lh + (r & 3)][tile 32 (TB wt + t) + ln]
    auto store_item = [&](int xi, int pt, int kt) __attribute__((always_inline)) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.M + (int64_t)xi * a.Kp * a.Pp, 0, (int)((int64_t)a.Kp * a.Pp * 4), 0x00020000);
        const uint32_t vo = (uint32_t)(((int64_t)(4 * lh) * a.Pp + ln) * 4);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int p0 = pt * BM + (wt * TB + t) * 32;
            if (p0 < a.Pp && (!(ABL & 4) || acc[0][0][0] == 12345.678f)) {          // (a tile group may reach beyond the padded tile count: nothing to store there)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = kt * BN + (2 * wc + c) * 32 + 8 * (r >> 2) + (r & 3);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[c][t][r]), rs, vo, (uint32_t)(((int64_t)k * a.Pp + p0) * 4), (ABL & 64) ? 2 : 0);
                    }
            }
        }
        clear_acc();
    };

    // ABL 4096 (timing only, M wrong): what an item would cost if it left in 8 TB 16-byte stores of WHOLE lines (a lane quad transposed so
    // that a lane holds four consecutive tiles of one cout: 8 lanes = a 128-byte run, 8 rows per instruction) that the counter can leave in
    // flight until the second barrier behind them.  The registers go out as they are, to the addresses such a form would write.
    auto store_item_x4 = [&](int xi, int pt, int kt) __attribute__((always_inline)) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.M + (int64_t)xi * a.Kp * a.Pp, 0, (int)((int64_t)a.Kp * a.Pp * 4), 0x00020000);
        const uint32_t vo = (uint32_t)(((int64_t)(4 * lh + (ln >> 3)) * a.Pp + 4 * (ln & 7)) * 4);
        int issued = 0;
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int p0 = pt * BM + (wt * TB + t) * 32;
            if (p0 < a.Pp) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k0 = kt * BN + (2 * wc + c) * 32 + 8 * q;
                        const u32x4 w = {__float_as_uint(acc[c][t][4 * q]), __float_as_uint(acc[c][t][4 * q + 1]), __float_as_uint(acc[c][t][4 * q + 2]), __float_as_uint(acc[c][t][4 * q + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(w, rs, vo, (uint32_t)(((int64_t)k0 * a.Pp + p0) * 4), 0);
                    }
                issued += 8;
            }
        }
        clear_acc();
        return issued;
    };

    // one load / one DMA piece at a time (FORM 1 spreads them over the MFMAs of k-step 0)
    auto load_v_one = [&](const Cursor &c, VSet &r, const int q, const int e) __attribute__((always_inline)) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.V + (int64_t)c.xi * a.C * a.Pp);
        const i32x4 rs = {(int)(uint32_t)base, (int)(uint32_t)((base >> 32) & 0xffffu), (int)v_slab_bytes, 0x00020000};
        const uint32_t vo = v_lane_off + (uint32_t)c.pt * (BM * 4);
        const uint32_t so = (uint32_t)((int64_t)(c.chunk * H3_KC + 16 * q + e) * a.Pp * 4);
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(r[q][e]) : "v"(vo), "s"(rs), "s"(so) : "memory");
    };
    auto dma_u_one = [&](const Cursor &c, int ubuf, const int j) __attribute__((always_inline)) {
        const unsigned char *sb = a.U + ((int64_t)(c.xi * (a.Kp / 32) + c.kt * (BN / 32)) * nst + c.chunk) * 4096;
        h3_dma16(sb, u_voff[j], lds_base + U0 + ubuf * UBYTES + (wave * NU + j) * 1024);
    };

    constexpr bool VD = (ABL & 16) != 0;
    auto dma_v_one = [&](const Cursor &c, int vbuf, const int j) __attribute__((always_inline)) {
        const int nblk = a.Pp >> 5, blk = c.pt * (BM / 32) + (BM == 256 ? wave : wave >> 1);
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(a.V) + (int64_t)c.xi * a.C * a.Pp * 4 + ((int64_t)(blk < nblk ? blk : nblk - 1) * nst + c.chunk) * 4096;
        h3_dma16(sb, (uint32_t)(j * 1024 + lane * 16), lds_base + vbuf * VBYTES + (BM == 256 ? wave : wave >> 1) * 4096 + j * 1024);
    };
    Cursor cc, cu, cv;          // compute; U' DMA (two stages ahead); V' loads (three stages ahead)
    locate(cc);
    cu = cc; cv = cc;
    Set vA, vB;                 // even iterations: vB holds V'(s + 1) and is refilled with V'(s + 3); odd iterations: vA
    // prologue: V'(0) -> LDS; V'(1) in vB, V'(2) in flight into vA; U'(0), U'(1) in flight
    if constexpr (VD) {
        for (int j = 0; j < 4; ++j) dma_v_one(cv, 0, j);
        advance(cv);
        dma_u(cu, 0); advance(cu);
        if (1 < total) { dma_u(cu, 1); advance(cu); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    load_set(cv, vA); advance(cv);
    dma_u(cu, 0); advance(cu);
    if (1 < total) { load_set(cv, vB); advance(cv); dma_u(cu, 1); advance(cu); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed_set(vA); landed_set(vB);
    write_set(0, vA);
    if (2 < total) { load_set(cv, vA); advance(cv); }
    }
    int prev_ops = 2 < total ? NV : 0;              // vector-memory operations this wave issued behind the last full wait
    bool pend = false;                              // an item ended with the previous stage: its M stores are due
    int pxi = 0, ppt = 0, pkt = 0;

    int ub_cur = 0, ub_next2 = 2;                   // U' buffers: of stage s, and the one the DMA of stage s + 2 fills (s % 3, (s + 2) % 3)
    auto iteration = [&](const int s, Set &r) __attribute__((always_inline)) {
        // Everything but what the previous iteration issued has landed: U'(s) (DMA of iteration s - 2) and V'(s + 1) (loads of
        // iteration s - 2, in r).  This wave's V'(s) pieces are written (lgkmcnt).  Behind the barrier nobody reads V' buffer
        // (s + 1) & 1 or U' buffer (s + 2) % 3 (stage s - 1) any more.
        if (prev_ops == NV + NU) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV + NU) : "memory");
        else if (prev_ops == NV) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV) : "memory");
        else if (prev_ops == NU) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NU) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        prev_ops = 0;
        landed_set(r);
        if (pend) { store_item(pxi, ppt, pkt); pend = false; }        // (older than this iteration's loads: a whole stage to drain)
        if (s + 1 < total) write_set((s + 1) & 1, r);
        if (s + 3 < total) {
            if (!(ABL & 1)) { load_set(cv, r); prev_ops += NV; }
            advance(cv);
        }
        if (s + 2 < total) {
            if (!(ABL & 2)) { dma_u(cu, ub_next2); prev_ops += NU; }
            advance(cu);
        }
        const unsigned char *vs = lds_h3 + (s & 1) * VBYTES, *us = lds_h3 + U0 + ub_cur * UBYTES;
        ub_cur = ub_cur == 2 ? 0 : ub_cur + 1;
        ub_next2 = ub_next2 == 2 ? 0 : ub_next2 + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 A[2][2], B[TB][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) A[c][pl] = *reinterpret_cast<const half8 *>(us + a_off + c * 4096 + pl * 2048 + kk * 1024);
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) B[t][pl] = *reinterpret_cast<const half8 *>(vs + b_off + t * 4096 + pl * 2048 + kk * 1024);
            // smallest terms first: (lo, hi) (hi, lo) (hi, hi); consecutive MFMAs on different accumulators
            // (Tried in round 3 and removed: the second k-step's fragments under the first one's MFMAs — tile blocks in pairs, the
            // V' registers of a pair refilled behind its last use, a second U' set, pinned by sched_group_barrier; 234 VGPRs,
            // bit-identical, the lgkmcnt(0) waits behind freshly issued reads gone from the .s — and the GEMM took 2.65 instead of
            // 2.63 ms per frame: with two waves per SIMD the other wave already covers those waits.)
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        if (ABL & 8) acc[c][t][term] += (float)A[c][PA[term]][0] + (float)B[t][PB[term]][1];
                        else acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][PA[term]], B[t][PB[term]], acc[c][t], 0, 0, 0);
                    }
            }
        }
        if (++cc.chunk == nst) {
            pend = true; pxi = cc.xi; ppt = cc.pt; pkt = cc.kt;
            cc.chunk = 0;
            if (++cc.k < my_items) locate(cc);
        }
    };
    // FORM 1: see the comment above the kernel
    auto iteration1 = [&](const int s, Set &r) __attribute__((always_inline)) {
        // all but the NV + NU operations of the previous iteration have landed: U'(s), and V'(s + 1) in r; this wave's V'(s) pieces are written
        if constexpr (VD) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NU) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV + NU) : "memory");
        if constexpr (!VD) landed_set(r);
        if (pend) { store_item(pxi, ppt, pkt); pend = false; }
        const unsigned char *vs = lds_h3 + (s & 1) * VBYTES, *us = lds_h3 + U0 + ub_cur * UBYTES;
        const int ub_fill = ub_next2;
        ub_cur = ub_cur == 2 ? 0 : ub_cur + 1;
        ub_next2 = ub_next2 == 2 ? 0 : ub_next2 + 1;
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};            // smallest terms first: (lo, hi) (hi, lo) (hi, hi)
        {
            half8 A[2][2], B[TB][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) A[c][pl] = *reinterpret_cast<const half8 *>(us + a_off + c * 4096 + pl * 2048);
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) B[t][pl] = *reinterpret_cast<const half8 *>(vs + b_off + t * 4096 + pl * 2048);
            if constexpr (!VD) write_set((s + 1) & 1, r);               // V'(s + 1): registers -> LDS, under the latency of the fragment reads
            __builtin_amdgcn_sched_barrier(0);
            int slot = 0;
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        if (ABL & 8) acc[c][t][term] += (float)A[c][PA[term]][0] + (float)B[t][PB[term]][1];
                        else acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][PA[term]], B[t][PB[term]], acc[c][t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        // fillers behind MFMA number `slot` of the k-step: first the loads of V'(s + 3) (one 8-byte or two 4-byte loads per slot), then
                        // the NU DMA pieces of U'(s + 2)
                        constexpr int LSLOTS = VD ? 4 : V2 ? NV : NV / 2;
                        if (slot < LSLOTS) {
                            if constexpr (VD) {
                                dma_v_one(cv, (s + 1) & 1, slot);
                            } else if (!(ABL & 1)) {
                                if constexpr (V2) load_v2_one(cv, r, slot);
                                else { load_v_one(cv, r, (2 * slot) / 8, (2 * slot) % 8); load_v_one(cv, r, (2 * slot + 1) / 8, (2 * slot + 1) % 8); }
                            }
                        } else if (slot < LSLOTS + NU) {
                            if (!(ABL & 2)) dma_u_one(cu, ub_fill, slot - LSLOTS);
                        }
                        if (slot < LSLOTS + NU) __builtin_amdgcn_sched_barrier(0);
                        ++slot;
                    }
            static_assert((VD ? 4 : V2 ? NV : NV / 2) + NU <= 6 * TB, "more fillers than MFMAs in a k-step");
        }
        advance(cv);
        advance(cu);
        {
            half8 A[2][2], B[TB][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) A[c][pl] = *reinterpret_cast<const half8 *>(us + a_off + c * 4096 + pl * 2048 + 1024);
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) B[t][pl] = *reinterpret_cast<const half8 *>(vs + b_off + t * 4096 + pl * 2048 + 1024);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        if (ABL & 8) acc[c][t][term] += (float)A[c][PA[term]][0] + (float)B[t][PB[term]][1];
                        else acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][PA[term]], B[t][PB[term]], acc[c][t], 0, 0, 0);
                    }
        }
        if (++cc.chunk == nst) {
            pend = true; pxi = cc.xi; ppt = cc.pt; pkt = cc.kt;
            cc.chunk = 0;
            if (++cc.k < my_items) locate(cc);
        }
    };
    // FORM 3: the fragment reads as a rotating software pipeline over the SAME 48 fragment registers.  PMC on FORM 2 (tools/h3_pmc.sh,
    // profiles/r06_h3_pmc_conv4_2*.txt; the cycle picture is the same on all-zero operands, where the clock is not throttled): the matrix
    // cores are busy 0.49 of the cycles as built and only 0.65 with NO memory side at all — all eight waves pass the barrier together, issue
    // the twelve fragment reads of k-step 0 together (96 KiB through a 128 B / cycle LDS: ~770 cycles in which no MFMA can issue), and the
    // compiler, short of registers, re-reads three fragments of k-step 1 right in front of their use (`ds_read ... s_waitcnt lgkmcnt(0) ...
    // v_mfma` twice per stage).  The three products of a k-step use (A lo, B hi), (A hi, B lo), (A hi, B hi): A lo dies after the first
    // eight MFMAs, B lo after the second eight — so the next k-step's fragments can be read into registers as they die:
    //     k0.t0 [read A lo']  k0.t1 [read B hi']  k0.t2 [read A hi', B lo']  k1.t0  k1.t1  [BARRIER of stage s + 1; read A lo'', B hi'' of its k-step 0]  k1.t2
    // every read is issued at least eight MFMAs (~260 cycles) ahead of its first use, the stage barrier moves one term forward (the
    // fragments k1.t2 multiplies are in registers by then; all LDS reads of stage s are complete), and the M stores of an item stay
    // behind its last MFMA.  Same MFMAs in the same order per accumulator: M bit-identical to FORM 0 / 1 / 2.
    half8 fAl[2], fBh[TB];          // (A lo, B hi) of k-step 0 of the stage about to be multiplied: loop-carried
    half8 frozen[2];                 // ABL 1024: two fragments read once
    if constexpr ((ABL & 1024) != 0) { frozen[0] = *reinterpret_cast<const half8 *>(lds_h3 + a_off); frozen[1] = *reinterpret_cast<const half8 *>(lds_h3 + b_off); }
    auto rdA = [&](const unsigned char *us, int c, int pl, int kk) __attribute__((always_inline)) {
        if constexpr ((ABL & 1024) != 0) { half8 x = frozen[0]; asm volatile("" : "+v"(x)); return x; }
        else return *reinterpret_cast<const half8 *>(us + a_off + c * 4096 + pl * 2048 + kk * 1024);
    };
    auto rdB = [&](const unsigned char *vs, int t, int pl, int kk) __attribute__((always_inline)) {
        if constexpr ((ABL & 1024) != 0) { half8 x = frozen[1]; asm volatile("" : "+v"(x)); return x; }
        else return *reinterpret_cast<const half8 *>(vs + b_off + t * 4096 + pl * 2048 + kk * 1024);
    };
    auto mma = [&](f32x16 &d, const half8 &x, const half8 &y, const int term) __attribute__((always_inline)) {
        if (ABL & 8) d[term] += (float)x[0] + (float)y[1];
        else d = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, d, 0, 0, 0);
    };
    auto iteration3 = [&](const int s, Set &r) __attribute__((always_inline)) {
        // entry: the barrier of stage s is behind us (U'(s) landed, everybody's V'(s) pieces written, V'(s + 1) landed in r), fAl / fBh issued
        landed_set(r);
        int x4_full = 0;
        if (pend) {
            if constexpr ((ABL & 4096) != 0) x4_full = store_item_x4(pxi, ppt, pkt) == 8 * TB ? 1 : 0;
            else store_item(pxi, ppt, pkt);
            pend = false;
        }
        const unsigned char *vs = lds_h3 + (s & 1) * VBYTES, *us = lds_h3 + U0 + ub_cur * UBYTES;
        const int ub_fill = ub_next2;
        ub_cur = ub_cur == 2 ? 0 : ub_cur + 1;
        ub_next2 = ub_next2 == 2 ? 0 : ub_next2 + 1;
        half8 Ah0[2], Bl0[TB], Al1[2], Bh1[TB], Ah1[2], Bl1[TB];
#pragma unroll
        for (int c = 0; c < 2; ++c) Ah0[c] = rdA(us, c, 0, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) Bl0[t] = rdB(vs, t, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        int slot = 0;
        constexpr int PPS = TB == 4 ? 1 : 2;                           // write pieces per slot
        constexpr int WSLOTS = VD ? 0 : NPARTS / PPS;                  // V'(s + 1): registers -> LDS, a ds_write_b128 (+ its four v_perm) per piece
        constexpr int LSLOTS = VD ? 4 : V2 ? NV : NV / 2;
        static_assert(WSLOTS + LSLOTS + NU <= 6 * TB, "more fillers than MFMAs in a k-step");
        auto filler = [&]() __attribute__((always_inline)) {
            // behind MFMA number `slot` of k-step 0: first the pieces of V'(s + 1) out of r, then the loads of V'(s + 3) INTO r (one 8-byte or two
            // 4-byte loads per slot), then the NU DMA pieces of U'(s + 2)
            if (slot < WSLOTS) {
#pragma unroll
                for (int j = 0; j < PPS; ++j) write_part((s + 1) & 1, r, slot * PPS + j);
            } else if (slot < WSLOTS + LSLOTS) {
                const int ls = slot - WSLOTS;
                if constexpr (VD) {
                    dma_v_one(cv, (s + 1) & 1, ls);
                } else if (!(ABL & 1)) {
                    if constexpr (V2) load_v2_one(cv, r, ls);
                    else { load_v_one(cv, r, (2 * ls) / 8, (2 * ls) % 8); load_v_one(cv, r, (2 * ls + 1) / 8, (2 * ls + 1) % 8); }
                }
            } else if (slot < WSLOTS + LSLOTS + NU) {
                if (!(ABL & 2)) dma_u_one(cu, ub_fill, slot - WSLOTS - LSLOTS);
            }
            if (slot < WSLOTS + LSLOTS + NU) __builtin_amdgcn_sched_barrier(0);
            ++slot;
        };
        // k-step 0, term 0: (A lo, B hi)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) { mma(acc[c][t], fAl[c], fBh[t], 0); __builtin_amdgcn_sched_barrier(0); filler(); }
#pragma unroll
        for (int c = 0; c < 2; ++c) Al1[c] = rdA(us, c, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        // term 1: (A hi, B lo)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) { mma(acc[c][t], Ah0[c], Bl0[t], 1); __builtin_amdgcn_sched_barrier(0); filler(); }
#pragma unroll
        for (int t = 0; t < TB; ++t) Bh1[t] = rdB(vs, t, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // term 2: (A hi, B hi)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) { mma(acc[c][t], Ah0[c], fBh[t], 2); __builtin_amdgcn_sched_barrier(0); filler(); }
#pragma unroll
        for (int c = 0; c < 2; ++c) Ah1[c] = rdA(us, c, 0, 1);
#pragma unroll
        for (int t = 0; t < TB; ++t) Bl1[t] = rdB(vs, t, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        // k-step 1, terms 0 and 1
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                mma(acc[c][t], Al1[c], Bh1[t], 0);
                __builtin_amdgcn_sched_barrier(0);
                // the scalar bookkeeping of the stage in the shadow of an MFMA each (both waves of a SIMD run this code at the same time: outside
                // a shadow the matrix pipe would stand idle for it)
                if (c == 0 && t == 0) { advance(cv); __builtin_amdgcn_sched_barrier(0); }
                if (c == 1 && t == 0) { advance(cu); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) { mma(acc[c][t], Ah1[c], Bl1[t], 1); __builtin_amdgcn_sched_barrier(0); }
        // the barrier of stage s + 1 (after the last stage: of nothing — the reads fetch bytes nobody multiplies): all but this iteration's
        // NV + NU operations have landed; this wave's V'(s + 1) pieces are written and its reads of stage s complete (lgkmcnt)
        if constexpr (VD) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NU) : "memory");
        else if constexpr ((ABL & 512) != 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NV + NU) : "memory");
        else if ((ABL & 4096) && x4_full) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV + NU + 8 * TB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV + NU) : "memory");
        {
            const unsigned char *vs1 = lds_h3 + ((s + 1) & 1) * VBYTES, *us1 = lds_h3 + U0 + ub_cur * UBYTES;
#pragma unroll
            for (int c = 0; c < 2; ++c) fAl[c] = rdA(us1, c, 1, 0);
#pragma unroll
            for (int t = 0; t < TB; ++t) fBh[t] = rdB(vs1, t, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // term 2 of k-step 1
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                mma(acc[c][t], Ah1[c], Bh1[t], 2);
                __builtin_amdgcn_sched_barrier(0);
                if (c == 0 && t == 0) {
                    if (++cc.chunk == nst) {
                        pend = true; pxi = cc.xi; ppt = cc.pt; pkt = cc.kt;
                        cc.chunk = 0;
                        if (++cc.k < my_items) locate(cc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    if (FORM >= 3) {
        // the barrier of stage 0 and the first reads
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV + NU) : "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c) fAl[c] = rdA(lds_h3 + U0, c, 1, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) fBh[t] = rdB(lds_h3, t, 0, 0);
        for (int s = 0; s < total; s += 2) {
            iteration3(s, vB);
            if (s + 1 < total) iteration3(s + 1, vA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (FORM >= 1) {
        for (int s = 0; s < total; s += 2) {
            iteration1(s, vB);
            if (s + 1 < total) iteration1(s + 1, vA);
        }
        // the last stages issued loads and DMA nobody consumes: they must have landed before the LDS goes to the next workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int s = 0; s < total; s += 2) {
            iteration(s, vB);
            if (s + 1 < total) iteration(s + 1, vA);
        }
    }
    if (pend) store_item(pxi, ppt, pkt);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool wino4_h3_supported(int cin, int cout_pad) { return cin % H3_KC == 0 && cout_pad % 128 == 0; }

static inline uint16_t f16_rne_bits(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static inline float f16_bits_to_float(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// packed (hi | lo << 16) of x * scale — what the transform kernels write (host mirror, for tests and sivo_debug_h3_gemm)
uint32_t wino4_h3_pack_value(float x, float scale) {
    const float xs = x * scale;
    const uint16_t hi = f16_rne_bits(xs);
    const uint16_t lo = f16_rne_bits(xs - f16_bits_to_float(hi));
    return (uint32_t)hi | ((uint32_t)lo << 16);
}

// U [36][Cin][Kp] fp32 -> fp16 hi / lo planes in stage order; returns the power of two the values were multiplied by
// (max |U| * scale in [2^7, 2^8): 2^8 of headroom, full hi / lo precision down to max / 2^10)
float wino4_h3_pack_weights(const std::vector<float> &U, int cin, int Kp, std::vector<uint16_t> &out) {
    float umax = 0.f;
    for (float v : U) umax = std::fmax(umax, std::fabs(v));
    int e = 0;
    if (umax > 0.f) (void)std::frexp(umax, &e);          // umax = m 2^e, m in [0.5, 1)
    const float scale = std::ldexp(1.f, 8 - e);
    const int nst = cin / H3_KC, ncb = Kp / 32;
    out.assign((size_t)36 * ncb * nst * 2048, 0);
    for (int xi = 0; xi < 36; ++xi)
        for (int cb = 0; cb < ncb; ++cb)
            for (int s = 0; s < nst; ++s) {
                uint16_t *img = out.data() + ((size_t)(xi * ncb + cb) * nst + s) * 2048;
                for (int o = 0; o < 4; ++o)
                    for (int r = 0; r < 32; ++r)
                        for (int el = 0; el < 8; ++el) {
                            const float x = U[((size_t)xi * cin + s * H3_KC + 8 * o + el) * Kp + cb * 32 + r];
                            const uint32_t p = wino4_h3_pack_value(x, scale);
                            const size_t at = (size_t)(o * 32 + r) * 8 + el;
                            img[at] = (uint16_t)(p & 0xffffu);
                            img[1024 + at] = (uint16_t)(p >> 16);
                        }
            }
    return scale;
}

// Workgroup tile (tiles x couts): 256 x 256; 128-tile groups when the launch would otherwise leave the CUs fewer than ~3
// items each (the 22 x 64 layers, short shards); 256 x 128 when the layer has 128 couts.  (Items holding all 512 couts of a
// layer — 128 x 512 tiles, the V' tile read from HBM once instead of twice — were built, verified and measured in round 3:
// 3.19 against 3.00 ms of GEMM per frame; twice the U' bytes per stage through a CU's load path cost more than the
// halved V' traffic saves.  Removed; numbers in NOTEBOOK 3.1e.)
struct H3Tile { int bm, bn; };
static H3Tile h3_tile(int64_t P, int Kp) {
    if (Kp % 256) return {256, 128};
    const int64_t items_big = 36 * ((P + 255) / 256) * (Kp / 256);
    return {items_big >= 3 * 256 ? 256 : 128, 256};
}

void launch_wino4_gemm_h3(const uint32_t *V, const void *U, float *M, int C, int Kp, int P, int Pp, hipStream_t s) {
    // The persistent workgroup claims the CU's whole LDS (160 KB) whatever its five stage buffers need (112 - 160 KB): with
    // the exact size (96 - 128 KB in the first form of the kernel) another lane's small-LDS workgroups (wino4_bridge_kernel:
    // 7 - 24 KB) were placed beside it and the frame was no longer reproducible run to run (measured, round 3: three lanes !=
    // one lane, the same handle twice != itself; with every workgroup alone on its CU, or with no LDS user beside it,
    // bit-identical; the GEMM alone beside such workgroups stays bit-exact: the victim is the neighbour.  Round 5, DESIGN 3.3: the
    // bridge's packed-FP32 instructions go wrong beside this kernel's MFMAs; the bridge is compiled without them now, the claim stays
    // because the trigger is only known for that kernel).  SIVO_H3_LDS_ALL=0 requests the exact size (diagnostic build).
    static const bool lds_all = !(SIVO_DIAG_ENV("SIVO_H3_LDS_ALL") && std::atoi(SIVO_DIAG_ENV("SIVO_H3_LDS_ALL")) == 0);
    // SIVO_H3_TILE=0/1 forces the larger / smaller tile count per item (tests)
    static const int force_tile = SIVO_DIAG_ENV("SIVO_H3_TILE") ? std::atoi(SIVO_DIAG_ENV("SIVO_H3_TILE")) : -1;
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        for (const void *f : {(const void *)wino4_gemm_h3_kernel<256, 256>, (const void *)wino4_gemm_h3_kernel<128, 256>, (const void *)wino4_gemm_h3_kernel<256, 128>})
            SIVO_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    const dim3 grid((unsigned)((n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8));      // one persistent workgroup per CU, a multiple of the 8 XCDs
    H3Args a{};
    a.V = V; a.U = static_cast<const unsigned char *>(U); a.M = M; a.C = C; a.Kp = Kp; a.P = P; a.Pp = Pp;
    H3Tile t = h3_tile(P, Kp);
    if (t.bn == 256 && (force_tile == 0 || force_tile == 1)) t.bm = force_tile == 0 ? 256 : 128;
    a.ptiles = (int)((P + t.bm - 1) / t.bm); a.ktiles = Kp / t.bn;
    const size_t lds = lds_all ? (size_t)160 * 1024 : (size_t)(2 * t.bm + 3 * t.bn) * 128;
#ifdef SIVO_DIAG
    if (const char *ab = SIVO_DIAG_ENV("SIVO_H3_ABL")) {          // diagnostic build: ablations of the 256 x 256 kernel
        a.ptiles = (P + 255) / 256; a.ktiles = Kp / 256;
#define H3_ABL_CASE(n)                                                                                                              \
    case n:                                                                                                                         \
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_gemm_h3_kernel<256, 256, n>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 256, n>), grid, dim3(512), (size_t)160 * 1024, s, a);                         \
        return;
        switch (std::atoi(ab)) {
            H3_ABL_CASE(1) H3_ABL_CASE(2) H3_ABL_CASE(3) H3_ABL_CASE(4) H3_ABL_CASE(7) H3_ABL_CASE(8) H3_ABL_CASE(12) H3_ABL_CASE(16) H3_ABL_CASE(24) H3_ABL_CASE(32) H3_ABL_CASE(36) H3_ABL_CASE(64) H3_ABL_CASE(128) H3_ABL_CASE(192) H3_ABL_CASE(4096) H3_ABL_CASE(519) H3_ABL_CASE(1031) H3_ABL_CASE(1543)
            default: break;
        }
#undef H3_ABL_CASE
    }
#endif
    lds_claim_note(LDS_CLAIM_GEMM_H3, lds);
#ifdef SIVO_DIAG
    if (const char *f = SIVO_DIAG_ENV("SIVO_H3_FORM"); f && std::atoi(f) == 1) {          // diagnostic build: FORM 1 (4-byte V' loads), for A/B
        static int attr1[64] = {0};
        if (FirstUse once(attr1); once)
            for (const void *fn : {(const void *)wino4_gemm_h3_kernel<256, 256, 0, 1>, (const void *)wino4_gemm_h3_kernel<256, 128, 0, 1>})
                SIVO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (t.bm == 256 && t.bn == 256) { hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 256, 0, 1>), grid, dim3(512), lds, s, a); return; }
        if (t.bm == 256 && t.bn == 128) { hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 128, 0, 1>), grid, dim3(512), lds, s, a); return; }
    }
    if (const char *f = SIVO_DIAG_ENV("SIVO_H3_FORM"); f && std::atoi(f) == 2) {          // diagnostic build: FORM 2 (the fragment reads where the compiler puts them), for A/B
        static int attr2[64] = {0};
        if (FirstUse once(attr2); once)
            for (const void *fn : {(const void *)wino4_gemm_h3_kernel<256, 256, 0, 2>, (const void *)wino4_gemm_h3_kernel<128, 256, 0, 2>, (const void *)wino4_gemm_h3_kernel<256, 128, 0, 2>})
                SIVO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (t.bm == 256 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 256, 0, 2>), grid, dim3(512), lds, s, a);
        else if (t.bm == 128 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<128, 256, 0, 2>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 128, 0, 2>), grid, dim3(512), lds, s, a);
        return;
    }
    if (const char *f = SIVO_DIAG_ENV("SIVO_H3_FORM"); f && std::atoi(f) == 0) {          // diagnostic build: the phased form of round 3 - 5, for A/B
        static int attr0[64] = {0};
        if (FirstUse once(attr0); once)
            for (const void *fn : {(const void *)wino4_gemm_h3_kernel<256, 256, 0, 0>, (const void *)wino4_gemm_h3_kernel<128, 256, 0, 0>, (const void *)wino4_gemm_h3_kernel<256, 128, 0, 0>})
                SIVO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (t.bm == 256 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 256, 0, 0>), grid, dim3(512), lds, s, a);
        else if (t.bm == 128 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<128, 256, 0, 0>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 128, 0, 0>), grid, dim3(512), lds, s, a);
        return;
    }
#endif
    if (t.bm == 256 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 256>), grid, dim3(512), lds, s, a);
    else if (t.bm == 128 && t.bn == 256) hipLaunchKernelGGL((wino4_gemm_h3_kernel<128, 256>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((wino4_gemm_h3_kernel<256, 128>), grid, dim3(512), lds, s, a);
}

}  // namespace sivo
