// conv3_h3.hip — DIRECT 3x3 convolution on the fp16 matrix cores with fp32 operands split as fp16 hi + lo ("f16x3", the
// arithmetic of conv_wino4_h3.hip) for the narrow layers (<= 128 channels at >= 176 x 512: conv2_2_D, conv2_1_D, conv1_2_D).
//
// Why direct.  A fused Winograd F(4x4) kernel has to keep 36 transform positions of a tile on chip: 2.25 accumulators per
// output and 36 weight matrices per channel chunk; conv_wino4f.hip does that on the fp32 matrix pipe (k = 4 per MFMA: a
// 36 KB weight slab per 4 channels).  The fp16 instructions take k = 16: the same slab would be 147 KB per step.  On the
// fp16 pipe a direct product costs 3 MFMA flops where the fp32 pipe pays 16, so the 4x of F(4x4) is no longer needed to
// get under the HBM time of these layers — and a direct kernel has ONE accumulator per output (a workgroup owns four times
// the pixels), 9 weight matrices per chunk, no transforms, no exchange stage, and no Winograd error: its results carry the
// 2^-22 of the split and the fp32 accumulation only.  The three-kernel F(4x4) path moved 5.7 GB through HBM for conv2_2_D
// (V and M, 2.25x the activation each way); this kernel moves the activation once (0.7 GB).
//
//   out[n][co][y][x] = act( ep_scale[co] * sum_{ci,ky,kx} W[co][ci][ky][kx] * in[n][ci][y+ky-1][x+kx-1] + ep_shift[co] )
//
// Implicit GEMM per tap: D[cout][pixel] += A_tap[cout][ci] * B[ci][pixel shifted by the tap].  The input patch of an item
// (8 x 64 output pixels: 10 x 66 with its halo) is staged ONCE per 16-channel chunk as fp16 hi / lo fragments, pixel-major;
// the nine taps are nine shifted ds_read_b128 windows of the same LDS image.
//
// Kernel.  One persistent 512-thread workgroup per CU walks (sample, pixel tile, 64-cout group) items in an XCD-aware
// order (each XCD a contiguous range of the pixel-tile-major list: halos and the cout groups of a tile meet in one L2);
// 8 waves = 4 row pairs x 2 column halves, wave tile 2 rows x 32 px x 64 couts = 2 x 2 MFMA blocks of 32 x 32 (64
// accumulator registers).  A stage is 16 channels = one k-step per tap: 9 x 12 = 108 v_mfma_f32_32x32x16_f16 per wave
// against 9 x 8 ds_read_b128.  (item, chunk) form one stream of stages, LDS double-buffered:
//   iteration s:  barrier -> output stage of an item that ended with stage s - 1 (its stores have a whole stage to drain)
//                 -> issue the patch loads (registers) and the weight LDS-DMA of stage s + 1 -> multiply stage s ->
//                 s_waitcnt vmcnt(0) -> split the patch of stage s + 1 into the other buffer.
//   * weights: split once on the host, stored as the LDS image of a stage ([cout group][chunk][tap][32-cout block][plane]
//     [octet][cout][8 ch]: 36 KiB), copied by LDS-DMA issued in inline assembly (lds_dma.hpp);
//   * patch: lane = (pixel, channel octet): 8 buffer loads (one per channel; out-of-image pixels carry an offset beyond the
//     descriptor and read 0 — the zero padding), x vscale, hi = fp16(x), lo = fp16(x - hi), v_perm_b32 to separate the
//     planes, two ds_write_b128.  UNPOOL: the loads go to the pooled tensor and its window codes (the Upsample in front of
//     the layer is never materialised).  The loads are inline assembly as in conv_wino4_h3.hip (hipcc's waitcnt pass would
//     otherwise drain them, and the DMA, in front of the first LDS read).
// LDS: 2 x 42,240 (patch) + 2 x 36,864 (weights) + 1 KiB (epilogue affine of two items) = 159,232 bytes.
// Fragment order [plane][octet][patch row][patch column] / [plane][octet][cout]: every ds_read_b128 / ds_write_b128 touches
// consecutive 16-byte pieces in lane order (conflict-free, any tap shift).  C/D of the 32 x 32 MFMA: column = lane & 31 =
// pixel, so an accumulator register of a wave is two 128-byte runs of an output row.
//
// Packed activations (round 4).  Between two layers of this kernel the activation does not have to be fp32 NCHW: the
// producer holds every output beside its channel neighbours and knows the consumer's power of two, so its output stage can
// write the consumer's LDS pieces directly —
//     P[n][C / 8][plane: hi, lo][Hp][Wp][8 halfs]      (the same 4 bytes per element as fp32)
// with the image at rows / columns 1 .. of a ZERO-BORDERED (Hp, Wp) plane (Hp >= 8 tiles_y + 2, Wp >= 64 tiles_x + 2: the
// halo and the overhang of partial items read zeros that are simply there; producers write the interior only).  A value is
// then split ONCE (by its producer) instead of once per (consumer workgroup, cout group, halo overlap), and
//   * IN_PK: the consumer's patch staging is 42 LDS-DMA pieces per workgroup and stage (1 KiB each: 64 consecutive pieces
//     of a plane-octet row run) and no VALU at all;
//   * IN_PK_UNPOOL (the layer reads through an Upsample): the POOLED tensor is packed, and the window codes come as one dword
//     per (pooled pixel, channel octet) whose byte k has bit e set when channel e's maximum sat at window position k
//     (pool_bits_kernel, pk_format.hip).  A lane loads one pooled piece (hi, lo: two 16-byte loads) and its dword, and
//     writes up to four unpooled pieces: piece & LUT[byte k] (256 x 16-byte table in LDS: bit e -> 0xffff in half e).
//   * OUT_PK: the output stage multiplies by the consumer's power of two, splits, and stores 8-byte half pieces (a lane
//     holds four consecutive channels of an octet); it also raises the overflow flag the consumer can no longer raise.
// The split is the same arithmetic in the same order as the fp32 form's (x * scale, hi = fp16, lo = fp16(rest)), so a
// chain of packed layers computes bit for bit what the chain of fp32 blobs computes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "h3_split.hpp"
#include "lds_dma.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int D_TH = 8, D_TW = 64;                 // output pixels of an item
constexpr int D_PR = D_TH + 2, D_PW = D_TW + 2;    // patch rows / columns (halo of one)
constexpr int D_KC = 16;                           // channels per stage
constexpr int D_NPX = D_PR * D_PW;                 // 660 patch pixels
constexpr int D_PLANE = 2 * D_NPX * 16;            // bytes of the hi (or lo) plane of a patch stage: two channel octets
constexpr int D_PBYTES = 2 * D_PLANE;              // 42,240
constexpr int D_UBYTES = 9 * 4096;                 // 36,864: nine taps x (2 cout blocks x 2 planes x 2 octets x 32 couts x 16 B)
constexpr int D_U0 = 2 * D_PBYTES;
constexpr int D_EP0 = D_U0 + 2 * D_UBYTES;         // epilogue affine: [item parity][scale 64 | shift 64] floats
constexpr int D_LDS = D_EP0 + 1024;
constexpr int D_NIT = 3;                           // patch pieces per thread: ceil(2 * 660 / 512)
constexpr int D_NPIECE = 2 * D_NPX;                // 1320
constexpr uint32_t D_INV = 0xfffffff0u;            // beyond any descriptor: loads return 0, stores are dropped
static_assert(D_LDS <= 160 * 1024, "LDS");
// packed input by LDS-DMA: 2 * D_NPIECE = 2640 pieces per stage in 42 DMA instructions of 64; the 48 pieces the last one
// writes beyond the patch land in a pad behind each buffer
constexpr int D_PK_DMA = 42;
constexpr int D_PBYTES_DMA = D_PK_DMA * 1024;      // 43,008
// packed input through an Upsample: pooled pieces of a stage = 2 octets x 6 rows x 34 columns, one per thread
constexpr int D_QR = D_TH / 2 + 2, D_QC = D_TW / 2 + 2, D_NQ = 2 * D_QR * D_QC;      // 408
static_assert(D_NQ <= 512, "one pooled piece per thread");
static_assert(2 * D_PBYTES_DMA + 2 * D_UBYTES + 1024 <= 160 * 1024, "LDS (IN_PK)");
static_assert(D_LDS + 4096 <= 160 * 1024, "LDS (IN_PK_UNPOOL)");

enum : int { IN_F32 = 0, IN_F32_UNPOOL = 1, IN_PK = 2, IN_PK_UNPOOL = 3 };

// Two forms of the stage loop (FORM; bit-identical results — same products, same order):
//   0  "phased": barrier -> output stage -> issue loads / DMA of stage s + 1 -> multiply stage s -> wait -> split stage s + 1.
//      Everything but the multiply runs with the matrix cores idle (all eight waves are in the same phase).  Measured on
//      conv1_2_D (tools/d3_probe.py): 1.03 ms as built, 0.65 ms with loads, split and stores removed.
//   1  "interleaved" (default): the multiply of stage s is cut into 36 slots of 3 MFMAs, and the other work of the iteration
//      is dealt over the slots in source order, pinned by __builtin_amdgcn_sched_barrier(0): slots 0-4 the weight DMA of
//      stage s + 1, slots 5-16 the patch loads of stage s + 2 (a second register set: they land during the rest of this
//      multiply and the next top-of-iteration wait finds them done), slots 17-34 the split of stage s + 1 (loaded during the
//      previous iteration) into the other patch buffer.  The VALU / VMEM / LDS-write instructions issue in the shadow of the
//      MFMAs (an MFMA occupies the pipe for 32 cycles, its issue 4).
// ABL (diagnostic builds only, -DSIVO_DIAG; results are wrong by construction): 1 no patch loads after the prologue,
// 2 no weight DMA after the prologue, 4 no output stores, 8 no MFMAs, 16 no patch split / LDS writes after the prologue.
template <int IN, bool OUT_PK, int FORM, int ABL = 0>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_h3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_d[];
    constexpr bool F32IN = IN == IN_F32 || IN == IN_F32_UNPOOL;
    constexpr bool UNPOOL = IN == IN_F32_UNPOOL;               // (the fp32 staging code below; the packed forms have their own)
    static_assert(F32IN || FORM == 1, "the packed inputs exist in the interleaved form only");
    // LDS map: two patch buffers, two weight buffers, the epilogue affine of two items, (IN_PK_UNPOOL) the 256 x 16-byte mask table
    constexpr int PB = IN == IN_PK ? D_PBYTES_DMA : D_PBYTES;
    constexpr int U0 = 2 * PB, EP0 = U0 + 2 * D_UBYTES, LUT0 = EP0 + 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, lh = lane >> 5;
    const int nst = a.Cin / D_KC;
    const int ngroups = a.CoutPad / 64;

    // items of this XCD: a contiguous range of the (sample, tile row, tile column, cout group) list; the workgroups of the
    // XCD take them round-robin (neighbouring workgroups work on neighbouring tiles at the same time)
    const int nitems = a.tiles_x * a.tiles_y * a.N * ngroups;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int lo_it = (int)(((int64_t)nitems * xcd) >> 3), hi_it = (int)(((int64_t)nitems * (xcd + 1)) >> 3);
    if (lo_it + wg >= hi_it) return;
    const int my_items = (hi_it - lo_it - wg + per_xcd - 1) / per_xcd;
    const int total = my_items * nst;

    const int64_t plane = (int64_t)a.H * a.W;
    const int Wh = a.W >> 1;
    const int64_t plane_in = UNPOOL ? (int64_t)(a.H >> 1) * Wh : plane;
    const float mscale = 1.f / (a.h3_vscale * a.h3_uscale);

    struct Cursor {          // a stage = (item, chunk); wave-uniform
        int k = 0, chunk = 0, n = 0, ty = 0, tx = 0, g = 0;
    };
    auto locate = [&](Cursor &c) __attribute__((always_inline)) {
        const int it = lo_it + wg + c.k * per_xcd;
        const int pt = it / ngroups;
        c.g = it - pt * ngroups;
        const int q = pt / a.tiles_x;
        c.tx = pt - q * a.tiles_x;
        c.n = q / a.tiles_y;
        c.ty = q - c.n * a.tiles_y;
    };
    auto advance = [&](Cursor &c) __attribute__((always_inline)) {
        if (++c.chunk == nst) {
            c.chunk = 0;
            if (++c.k < my_items) locate(c);
        }
    };

    // ---- patch staging: piece p = tid + 512 r = (octet o, patch row py, patch column px); LDS byte p * 16 in each plane ----
    int p_py[D_NIT], p_px[D_NIT], p_o[D_NIT];
    bool p_valid[D_NIT];
#pragma unroll
    for (int r = 0; r < D_NIT; ++r) {
        const int p = tid + 512 * r;
        p_valid[r] = p < D_NPIECE;
        const int pc = p_valid[r] ? p : 0;
        p_o[r] = pc / D_NPX;
        const int rem = pc - p_o[r] * D_NPX;
        p_py[r] = rem / D_PW;
        p_px[r] = rem - p_py[r] * D_PW;
    }
    typedef uint32_t PSet[D_NIT][8];
    struct LoadPlan {        // what the loads of one stage need: descriptors (wave-uniform) and this lane's offsets
        i32x4 rs, mrs;
        uint32_t vo[D_NIT], mo[D_NIT];
        int chunk;
    };
    auto plan_loads = [&](const Cursor &c, LoadPlan &lp) __attribute__((always_inline)) {
        const uint64_t base = (uint64_t)(uintptr_t)(a.in + (int64_t)c.n * a.in_sample_stride);
        lp.rs = (i32x4){(int)(uint32_t)base, (int)(uint32_t)((base >> 32) & 0xffffu), (int)(a.Cin * plane_in * 4), 0x00020000};
        const uint64_t mbase = UNPOOL ? (uint64_t)(uintptr_t)(a.unpool_mask + (int64_t)c.n * a.unpool_mask_stride) : base;
        lp.mrs = (i32x4){(int)(uint32_t)mbase, (int)(uint32_t)((mbase >> 32) & 0xffffu), (int)(a.Cin * plane_in), 0x00020000};
        lp.chunk = c.chunk;
        const int y0 = c.ty * D_TH, x0 = c.tx * D_TW;
#pragma unroll
        for (int r = 0; r < D_NIT; ++r) {
            const int gy = y0 + p_py[r] - 1, gx = x0 + p_px[r] - 1;
            const bool inside = p_valid[r] && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const uint32_t idx = (uint32_t)(p_o[r] * 8 * plane_in) + (UNPOOL ? (uint32_t)((gy >> 1) * Wh + (gx >> 1)) : (uint32_t)(gy * a.W + gx));
            lp.vo[r] = inside ? idx * 4u : D_INV;
            lp.mo[r] = inside ? idx : D_INV;
        }
    };
    // load number L (0 .. 23) of a stage: channel e = L % 8 of piece r = L / 8 (value, and with UNPOOL its window code)
    auto load_one = [&](const LoadPlan &lp, int L, PSet &pv, PSet &pm) __attribute__((always_inline)) {
        const int r = L >> 3, e = L & 7;
        // (readfirstlane: the value is wave-uniform, but hipcc may have computed it on the vector ALU, and an "s" operand of an
        // asm statement is not legalised — the assembler then rejects a VGPR in the soffset position)
        const uint32_t sm = (uint32_t)__builtin_amdgcn_readfirstlane((lp.chunk * D_KC + e) * (int)plane_in);
        const uint32_t so = sm * 4u;
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(pv[r][e]) : "v"(lp.vo[r]), "s"(lp.rs), "s"(so) : "memory");
        if (UNPOOL) {
            asm volatile("buffer_load_ubyte %0, %1, %2, %3 offen" : "=&v"(pm[r][e]) : "v"(lp.mo[r]), "s"(lp.mrs), "s"(sm) : "memory");
        }
    };
    // the loads of a set have landed (the caller's s_waitcnt): from here on its registers may be read.  UNPOOL: the value
    // stays where its window code says this pixel is the window's maximum, else 0 (the codes' registers are free again)
    auto landed = [&](PSet &pv, PSet &pm) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < D_NIT; ++r) {
            // window code of this piece's pixel (item origins are even): rows / columns of the image alternate 0, 1
            const uint32_t code = (uint32_t)((((p_py[r] + 1) & 1) << 1) | ((p_px[r] + 1) & 1));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                asm volatile("" : "+v"(pv[r][e]));
                if (UNPOOL) {
                    asm volatile("" : "+v"(pm[r][e]));
                    pv[r][e] = pm[r][e] == code ? pv[r][e] : 0u;
                }
            }
        }
    };
    uint32_t ovf = 0u;          // largest |scaled input| seen by this lane, as bits << 1
    // split of piece r, in steps: 0..3 -> channels 2 step, 2 step + 1 become (hi | lo << 16) in place; 4 -> the hi plane's
    // piece is written; 5 -> the lo plane's
    auto split_step = [&](int buf, PSet &pv, int r, int step) __attribute__((always_inline)) {
        if (r == D_NIT - 1 && !p_valid[r]) return;
        unsigned char *dst = lds_d + buf * PB + tid * 16 + r * 8192;
        if (step < 4) {
#pragma unroll
            for (int e = 2 * step; e < 2 * step + 2; ++e) {
                // as wino4_pack_h3 (h3_split.hpp); the range check is a running maximum of the bit pattern of |xs| (sign shifted
                // out; inf and NaN order above every finite value) instead of a compare + mask update per value
                const float xs = __uint_as_float(pv[r][e]) * a.h3_vscale;
                const _Float16 hi = (_Float16)xs;
                const _Float16 lo = (_Float16)(xs - (float)hi);
                const uint32_t mag = __float_as_uint(xs) << 1;
                ovf = mag > ovf ? mag : ovf;
                pv[r][e] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
            }
        } else {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_perm(pv[r][2 * j + 1], pv[r][2 * j], step == 4 ? 0x05040100u : 0x07060302u);
            *reinterpret_cast<u32x4 *>(dst + (step == 4 ? 0 : D_PLANE)) = w;
        }
    };
    auto split_all = [&](int buf, PSet &pv) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < D_NIT; ++r)
#pragma unroll
            for (int step = 0; step < 6; ++step) split_step(buf, pv, r, step);
    };

    // ---- weights: 36 pieces of 1 KiB per stage, wave w copies pieces w, w + 8, ... ------------------------------------------
    const uint32_t lds_base = lds_addr_uniform(lds_d);
    auto dma_piece = [&](const Cursor &c, int buf, int j) __attribute__((always_inline)) {
        const unsigned char *sb = static_cast<const unsigned char *>(a.wt_h3) + ((int64_t)c.g * nst + c.chunk) * D_UBYTES;
        const int piece = wave + 8 * j;
        if (piece < 36) lds_dma16_s(sb, (uint32_t)(piece * 1024 + lane * 16), lds_base + U0 + buf * D_UBYTES + piece * 1024);
    };
    // the same with the stage's base computed once by the caller (the packed forms: a slot has nothing else to do, so the 64-bit
    // address arithmetic per piece was most of its scalar work); pieces 0 .. 31 exist for every wave: no branch for j < 4
    const uint32_t w_voff = (uint32_t)(wave * 1024 + lane * 16);
    auto dma_piece_at = [&](const unsigned char *sb, int buf, int j) __attribute__((always_inline)) {
        if (j < 4 || wave < 4) lds_dma16_s(sb + j * 8192, w_voff, lds_base + U0 + buf * D_UBYTES + (wave + 8 * j) * 1024);
    };

    // ---- MFMA phase: wave (rp, ch) owns output rows 2 rp, 2 rp + 1, columns 32 ch .. 32 ch + 31 of the item, all 64 couts ----
    const int rp = wave >> 1, ch = wave & 1;
    f32x16 acc[2][2];           // [cout block][row]
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][sg][r] = 0.f;
    };
    clear_acc();
    const uint32_t b_off = (uint32_t)((lh * D_NPX + (2 * rp) * D_PW + ch * 32 + ln) * 16), a_off = (uint32_t)(lane * 16);
    // [fragment set][block / row][plane]; sets 0 / 1 = the tap's parity; set 2 (packed forms): tap 0 of a stage, read under the LAST tap of the
    // stage before it (which sits in set 0) — see the packed iterations
    half8 A[3][2][2], B[3][2][2];
    auto fetch = [&](int buf, int t, int par) __attribute__((always_inline)) {
        const unsigned char *ps = lds_d + buf * PB + b_off, *us = lds_d + U0 + buf * D_UBYTES + a_off;
        const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) A[par][b][pl] = *reinterpret_cast<const half8 *>(us + t * 4096 + b * 2048 + pl * 1024);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) B[par][sg][pl] = *reinterpret_cast<const half8 *>(ps + pl * D_PLANE + ((sg + ky) * D_PW + kx) * 16);
    };
    // MFMAs 3 q .. 3 q + 2 of a tap's twelve: smallest terms first — (lo, hi) (hi, lo) (hi, hi) — consecutive MFMAs on
    // different accumulators
    auto mfma3 = [&](int par, int q) __attribute__((always_inline)) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int m = 3 * q; m < 3 * q + 3; ++m) {
            const int term = m >> 2, b = (m >> 1) & 1, sg = m & 1;
            if (ABL & 8) acc[b][sg][term] += (float)A[par][b][PA[term]][0] + (float)B[par][sg][PB[term]][1];
            else acc[b][sg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[par][b][PA[term]], B[par][sg][PB[term]], acc[b][sg], 0, 0, 0);
        }
    };

    // ---- output stage: register r of block (b, sg) is out[cout 64 g + 32 b + 8 (r >> 2) + 4 lh + (r & 3)][row y0 + 2 rp + sg][col x0 + 32 ch + ln]
    // (Tried in round 3, slower, removed: a 4 x 4 transpose of the four cout registers of a pixel across the lanes of a quad —
    // select + DPP quad_perm + two selects per register pair, twice — so that a lane holds four consecutive pixels of one cout
    // and the item takes 16 buffer_store_dwordx4 instead of 64 dword stores: output stage 2.2k instead of 1.4k cycles per
    // stage on conv1_2_D, tools/d3_stamps.py.  The dword stores already write whole 128-byte runs; what the stage waits for
    // is the write path's bytes, not its instruction count.)
    // (relu_tag: the layer's ReLU flag as a compile-time constant — tested per element it cost a mask update and a wait state
    // in front of every select)
    auto store_item_as = [&](int n, int ty, int tx, int g, int par, auto relu_tag) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_tag)::value;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)n * a.Cout * plane, 0, (int)((int64_t)a.Cout * plane * 4), 0x00020000);
        const float *epl = reinterpret_cast<const float *>(lds_d + EP0) + par * 128;
        const int x = tx * D_TW + ch * 32 + ln;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int y = ty * D_TH + 2 * rp + sg;
            const uint32_t vo = (y < a.H && x < a.W && (!(ABL & 4) || acc[0][0][0] == 12345.678f)) ? (uint32_t)(((int64_t)(4 * lh) * plane + (int64_t)y * a.W + x) * 4) : D_INV;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(epl + b * 32 + 8 * q + 4 * lh);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(epl + 64 + b * 32 + 8 * q + 4 * lh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[b][sg][4 * q + i] * sc[i] + sh[i];
                        if (RELU) v = v > 0.f ? v : 0.f;
                        const int co = g * 64 + b * 32 + 8 * q + i;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, vo, (uint32_t)((int64_t)co * plane * 4), 0);
                    }
                }
        }
        clear_acc();
    };
    // OUT_PK: the consumer's pieces.  A lane holds channels 4 lh .. 4 lh + 3 of octet 8 g + 4 b + q for its pixel: the 8 bytes at
    // offset 8 lh of that pixel's hi piece and of its lo piece; the 32 pixels of a wave row are 512 contiguous bytes per store.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    auto store_item_pk_as = [&](int n, int ty, int tx, int g, int par, auto relu_tag) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_tag)::value;
        const int64_t opl = (int64_t)a.out_Hp * a.out_Wp * 16;          // bytes of one plane of one channel octet
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned char *>(a.out_pk) + (int64_t)n * (a.Cout / 8) * 2 * opl, 0, (int)((a.Cout / 8) * 2 * opl), 0x00020000);
        const float *epl = reinterpret_cast<const float *>(lds_d + EP0) + par * 128;
        const int x = tx * D_TW + ch * 32 + ln;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int y = ty * D_TH + 2 * rp + sg;
            const uint32_t vo = (y < a.H && x < a.W && (!(ABL & 4) || acc[0][0][0] == 12345.678f)) ? (uint32_t)((((y + 1) * a.out_Wp) + x + 1) * 16 + lh * 8) : D_INV;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(epl + b * 32 + 8 * q + 4 * lh);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(epl + 64 + b * 32 + 8 * q + 4 * lh);
                    uint32_t pr[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[b][sg][4 * q + i] * sc[i] + sh[i];
                        if (RELU) v = v > 0.f ? v : 0.f;
                        const float xs = v * a.out_vscale;              // as the consumer's split_step does with the fp32 blob
                        const _Float16 hi = (_Float16)xs;
                        const _Float16 lo = (_Float16)(xs - (float)hi);
                        const uint32_t mag = __float_as_uint(xs) << 1;
                        ovf = mag > ovf ? mag : ovf;
                        pr[i] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
                    }
                    const u32x2 hv = {__builtin_amdgcn_perm(pr[1], pr[0], 0x05040100u), __builtin_amdgcn_perm(pr[3], pr[2], 0x05040100u)};
                    const u32x2 lv = {__builtin_amdgcn_perm(pr[1], pr[0], 0x07060302u), __builtin_amdgcn_perm(pr[3], pr[2], 0x07060302u)};
                    const uint32_t so = (uint32_t)((int64_t)((g * 8 + b * 4 + q) * 2) * opl);
                    __builtin_amdgcn_raw_buffer_store_b64(hv, rs, vo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(lv, rs, vo, so + (uint32_t)opl, 0);
                }
        }
        clear_acc();
    };
    auto store_item = [&](int n, int ty, int tx, int g, int par) __attribute__((always_inline)) {
        if constexpr (OUT_PK) {
            if (a.relu) store_item_pk_as(n, ty, tx, g, par, std::true_type{});
            else store_item_pk_as(n, ty, tx, g, par, std::false_type{});
        } else {
            if (a.relu) store_item_as(n, ty, tx, g, par, std::true_type{});
            else store_item_as(n, ty, tx, g, par, std::false_type{});
        }
    };

    // ---- the stream of stages ----------------------------------------------------------------------------------------------
    Cursor cc;                  // compute
    locate(cc);
    float epv = 0.f;            // waves 0 / 1: the epilogue scale / shift of an item, on its way to LDS
    const float *ep_src = wave == 0 ? a.ep_scale : a.ep_shift;
    bool pend = false;          // an item ended with the previous stage: its output stage is due
    int pn = 0, pty = 0, ptx = 0, pg = 0, ppar = 0;
    auto end_of_stage = [&]() __attribute__((always_inline)) {
        if (++cc.chunk == nst) {
            pend = true; pn = cc.n; pty = cc.ty; ptx = cc.tx; pg = cc.g; ppar = cc.k & 1;
            cc.chunk = 0;
            if (++cc.k < my_items) locate(cc);
        }
    };

    // ABL & 64 (diagnostic builds): shader-clock stamps around the parts of an iteration, summed per wave into a.vmax[0..4]
    // (wait for the previous iteration's memory traffic, barrier, output stage, multiply + interleaved work, iterations)
    auto stamp = [&]() __attribute__((always_inline)) -> uint32_t {
        uint64_t t = 0;
        if (ABL & 64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return (uint32_t)t;
    };
    uint32_t st_wait = 0, st_bar = 0, st_out = 0, st_mul = 0;
    const uint32_t st_begin = stamp();
    if constexpr (FORM == 0) {
        Cursor cl = cc;         // loads, one stage ahead
        PSet pv, pm;
        bool l_first = false;   // the stage in the registers is the first of its item ...
        int l_par = 0;          // ... of this parity
        auto issue = [&](int buf, bool with_loads, bool with_dma) __attribute__((always_inline)) {
            l_first = cl.chunk == 0;
            l_par = cl.k & 1;
            if (l_first && wave < 2) epv = ep_src[cl.g * 64 + lane];        // (used in commit: no wait here)
            if (with_loads) {
                LoadPlan lp;
                plan_loads(cl, lp);
#pragma unroll
                for (int L = 0; L < 8 * D_NIT; ++L) load_one(lp, L, pv, pm);
            }
            if (with_dma)
#pragma unroll
                for (int j = 0; j < 5; ++j) dma_piece(cl, buf, j);
            advance(cl);
        };
        auto commit = [&](int buf, bool with_split) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed(pv, pm);
            if (with_split) split_all(buf, pv);
            if (l_first && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[l_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
        };
        issue(0, true, true);
        commit(0, true);
        for (int s = 0; s < total; ++s) {
            // this wave's pieces of stage s are written (lgkmcnt), its DMA has landed (vmcnt(0) in commit); behind the barrier
            // everybody's are, and nobody reads the buffers of stage s - 1 any more
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (pend) { store_item(pn, pty, ptx, pg, ppar); pend = false; }
            const bool more = s + 1 < total;
            if (more) issue((s + 1) & 1, !(ABL & 1), !(ABL & 2));
            // fragments one tap ahead: [8 ds_read_b128 of tap t + 1][12 MFMAs of tap t] (hipcc on its own places each read right
            // in front of its first use — seen in the .s — and the wave then waits out the LDS latency four times per tap)
            fetch(s & 1, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) {
                    fetch(s & 1, t + 1, (t + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) mfma3(t & 1, q);
            }
            end_of_stage();
            if (more) commit((s + 1) & 1, !(ABL & 16));
        }
    } else if constexpr (F32IN) {
        Cursor cu = cc, cl = cc;    // weight DMA (one stage ahead of the multiply); patch loads (two stages ahead)
        PSet vA, vB, pm;            // even iterations split vA (stage s + 1) and load stage s + 2 into vB; odd ones the reverse
        bool ep_due = false;        // epv holds the affine of an item whose first stage's DMA was issued in the previous iteration
        int ep_par = 0;
        LoadPlan lp;
        // prologue: patch(0) -> LDS, weights(0) in flight, patch(1) in flight into vA
        plan_loads(cl, lp);
#pragma unroll
        for (int L = 0; L < 8 * D_NIT; ++L) load_one(lp, L, vB, pm);
        advance(cl);
        if (wave < 2) { epv = ep_src[cu.g * 64 + lane]; }
        ep_due = true; ep_par = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma_piece(cu, 0, j);
        advance(cu);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(vB, pm);
        split_all(0, vB);
        if (1 < total) {
            plan_loads(cl, lp);
#pragma unroll
            for (int L = 0; L < 8 * D_NIT; ++L) load_one(lp, L, vA, pm);
            advance(cl);
        }
        auto iteration = [&](const int s, PSet &vs, PSet &vl) __attribute__((always_inline)) {
            // Everything this wave issued in the previous iteration has had a whole multiply to complete: the weights of stage s
            // (DMA), the patch of stage s + 1 (registers vs), the stores of an output stage.  This wave's pieces of the patch of
            // stage s are written (lgkmcnt).  Behind the barrier nobody reads the buffers of stage s - 1 any more.
            const uint32_t t0 = stamp();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t t1 = stamp();
            if (ep_due && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[ep_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
            ep_due = false;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const uint32_t t2 = stamp();
            if (pend) { store_item(pn, pty, ptx, pg, ppar); pend = false; }
            const uint32_t t3 = stamp();
            const bool more = s + 1 < total, more2 = s + 2 < total;
            if (more) landed(vs, pm);
            const int nb = (s + 1) & 1;
            if (more2) plan_loads(cl, lp);
            if (more && cu.chunk == 0) {
                ep_due = true; ep_par = cu.k & 1;
                if (wave < 2) epv = ep_src[cu.g * 64 + lane];
            }
            fetch(s & 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int slot = 0; slot < 36; ++slot) {
                const int t = slot >> 2, q = slot & 3;
                if (q == 0 && t + 1 < 9) fetch(s & 1, t + 1, (t + 1) & 1);
                mfma3(t & 1, q);
                if (slot < 5) {
                    if (more && !(ABL & 2)) dma_piece(cu, nb, slot);
                } else if (slot < 17) {
                    if (more2 && !(ABL & 1)) {
                        load_one(lp, 2 * (slot - 5), vl, pm);
                        load_one(lp, 2 * (slot - 5) + 1, vl, pm);
                    }
                } else if (slot < 35) {
                    if (more && !(ABL & 16)) split_step(nb, vs, (slot - 17) / 6, (slot - 17) % 6);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ABL & 64) {
                const uint32_t t4 = stamp();
                st_wait += (t1 - t0) & 0xfffffu; st_bar += (t2 - t1) & 0xfffffu; st_out += (t3 - t2) & 0xfffffu; st_mul += (t4 - t3) & 0xfffffu;
            }
            if (more) advance(cu);
            if (more2) advance(cl);
            end_of_stage();
        };
        for (int s = 0; s < total; s += 2) {
            iteration(s, vA, vB);
            if (s + 1 < total) iteration(s + 1, vB, vA);
        }
    } else if constexpr (IN == IN_PK) {
        // ---- packed input, no Upsample: the patch of a stage is 42 LDS-DMA pieces, one stage ahead like the weights ----------
        // piece j = wave + 8 i; lane l copies piece q = 64 j + l of the stage image [plane][octet][patch row][patch column]
        // (q >= 2640: the pad behind the buffer, from a valid address)
        Cursor cu = cc;
        bool ep_due = false;
        int ep_par = 0;
        uint32_t pk_voff[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = (wave + 8 * i) * 64 + lane;
            const int qq = q < 2 * D_NPIECE ? q : 0;
            const int pl = qq / D_NPIECE, rem = qq - pl * D_NPIECE, o = rem / D_NPX, r2 = rem - o * D_NPX, py = r2 / D_PW, px = r2 - py * D_PW;
            pk_voff[i] = (uint32_t)((((o * 2 + pl) * a.in_Hp + py) * a.in_Wp + px) * 16);
        }
        // patch row 0 = image row y0 - 1 = padded row y0, patch column 0 = padded column x0
        // running source pointers of the DMA cursor's stage (patch origin, weight image): one addition per stage, the full
        // address arithmetic only when the item changes
        const unsigned char *sbp = nullptr, *sbw = nullptr;
        const int64_t p_stride = (int64_t)4 * a.in_Hp * a.in_Wp * 16;           // one 16-channel chunk of the packed input
        auto run_locate = [&](const Cursor &c) __attribute__((always_inline)) {
            sbp = static_cast<const unsigned char *>(a.in_pk) + (int64_t)c.n * a.in_pk_sample_bytes + ((int64_t)(c.ty * D_TH) * a.in_Wp + c.tx * D_TW) * 16;
            sbw = static_cast<const unsigned char *>(a.wt_h3) + (int64_t)c.g * nst * D_UBYTES;
        };
        auto run_advance = [&](Cursor &c) __attribute__((always_inline)) {
            if (++c.chunk == nst) {
                c.chunk = 0;
                if (++c.k < my_items) { locate(c); run_locate(c); }
            } else {
                sbp += p_stride; sbw += D_UBYTES;
            }
        };
        auto dma_patch = [&](const unsigned char *sb, int buf, int i) __attribute__((always_inline)) {      // (pieces 0 .. 39 exist for every wave)
            if (i < 5 || wave < D_PK_DMA - 40) lds_dma16_s(sb, pk_voff[i], lds_base + buf * PB + (wave + 8 * i) * 1024);
        };
        // prologue: patch(0) and weights(0) in flight (the top of iteration 0 waits for them)
        run_locate(cu);
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_patch(sbp, 0, i);
        if (wave < 2) { epv = ep_src[cu.g * 64 + lane]; }
        ep_due = true; ep_par = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma_piece_at(sbw, 0, j);
        run_advance(cu);
        // MORE: stage s + 1 exists (every iteration but the last: peeled, so that the loop body is one straight block)
        auto iteration = [&](const int s, auto more_tag) __attribute__((always_inline)) {
            constexpr bool MORE = decltype(more_tag)::value;
            // everything this wave issued in the previous iteration has had a whole multiply to complete: patch and weights of
            // stage s (DMA), the stores of an output stage.  Behind the barrier nobody reads the buffers of stage s - 1 any more.
            // Entry: the barrier of stage s is behind us and tap 0's fragments are on their way into set 2 — both happened in front of the LAST tap
            // of stage s - 1 (the prologue for s = 0): all eight waves used to pass the barrier together, issue tap 0's eight reads together
            // (64 KiB through a 128 B / cycle LDS) and wait for them with idle matrix cores, once per stage (PMC, zero operands: conv1_2_D
            // 519 us with nothing but MFMAs and fragment reads, 423 us of MFMA cycles; the same finding as conv_wino4_h3.hip FORM 3).
            const uint32_t t0 = stamp(), t1 = t0, t2 = t0;
            if (pend) { store_item(pn, pty, ptx, pg, ppar); pend = false; }
            const uint32_t t3 = stamp();
            const int nb = (s + 1) & 1;
            if (MORE && cu.chunk == 0) {
                ep_due = true; ep_par = cu.k & 1;
                if (wave < 2) epv = ep_src[cu.g * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int slot = 0; slot < 36; ++slot) {
                const int t = slot >> 2, q = slot & 3;
                if (q == 0 && t + 1 < 9) fetch(s & 1, t + 1, (t + 1) & 1);
                if (MORE && slot == 32) {
                    // the barrier of stage s + 1: everything this wave issued in this iteration has had eight taps to complete (patch and weights
                    // of stage s + 1, the stores of an output stage); all reads of stage s are issued (tap 8's went out under tap 7) and complete
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ep_due && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[ep_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
                    ep_due = false;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    fetch(nb, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma3(t == 0 ? 2 : (t & 1), q);
                if (MORE) {
                    if (slot < 5) { if (!(ABL & 2)) dma_piece_at(sbw, nb, slot); }
                    else if (slot < 11) { if (!(ABL & 1)) dma_patch(sbp, nb, slot - 5); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ABL & 64) {
                const uint32_t t4 = stamp();
                st_wait += (t1 - t0) & 0xfffffu; st_bar += (t2 - t1) & 0xfffffu; st_out += (t3 - t2) & 0xfffffu; st_mul += (t4 - t3) & 0xfffffu;
            }
            if (MORE) run_advance(cu);
            end_of_stage();
        };
        // the barrier of stage 0 and its tap 0 (what every iteration does for its successor in front of its last tap)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ep_due && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[ep_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
        ep_due = false;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fetch(0, 0, 2);
        for (int s = 0; s + 1 < total; ++s) iteration(s, std::true_type{});
        iteration(total - 1, std::false_type{});
    } else {
        // ---- packed input through an Upsample: pooled pieces through registers, one stage ahead --------------------------------
        // thread t < 408 = (octet o, pooled row r of 6, pooled column c of 34): pooled pixel (y0 / 2 - 1 + r, x0 / 2 - 1 + c),
        // i.e. padded row y0 / 2 + r, padded column x0 / 2 + c; its window position (dy, dx) is patch pixel (2 r - 1 + dy, 2 c - 1 + dx)
        const bool q_valid = tid < D_NQ;
        const int qt = q_valid ? tid : 0;
        const int q_o = qt / (D_QR * D_QC), q_rem = qt - q_o * (D_QR * D_QC), q_r = q_rem / D_QC, q_c = q_rem - q_r * D_QC;
        // (lanes without a piece load from offset 0 — the padded tensors make every address of the patch valid — and store nothing)
        const uint32_t q_voff = (uint32_t)((((q_o * 2) * a.in_Hp + q_r) * a.in_Wp + q_c) * 16);                          // hi plane
        const uint32_t q_voff_lo = q_voff + (uint32_t)(a.in_Hp * a.in_Wp * 16);                                       // lo plane: one plane further
        const uint32_t q_moff = (uint32_t)(((q_o * a.in_Hp + q_r) * a.in_Wp + q_c) * 4);
        uint32_t q_dst[4];          // LDS byte offset of window position k's piece inside a patch buffer's hi plane; q_ok bit k: it exists
        uint32_t q_ok = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int py = 2 * q_r - 1 + (k >> 1), px = 2 * q_c - 1 + (k & 1);
            const bool ok = q_valid && (unsigned)py < (unsigned)D_PR && (unsigned)px < (unsigned)D_PW;
            q_dst[k] = ok ? (uint32_t)((q_o * D_NPX + py * D_PW + px) * 16) : 0u;
            q_ok |= ok ? (1u << k) : 0u;
        }
        struct QSet { u32x4 hi, lo; uint32_t m; };
        // running scalar bases of the load cursor's stage: pooled patch origin in the packed tensor, the same in the mask tensor,
        // the weight image; one addition per stage, the full address arithmetic only when the item changes
        const unsigned char *sbq = nullptr, *sbm = nullptr, *sbw = nullptr;
        const int64_t q_plane = (int64_t)a.in_Hp * a.in_Wp;              // pooled pieces of one plane of one octet
        auto run_locate = [&](const Cursor &c) __attribute__((always_inline)) {
            const int64_t org = (int64_t)(c.ty * (D_TH / 2)) * a.in_Wp + c.tx * (D_TW / 2);
            sbq = static_cast<const unsigned char *>(a.in_pk) + (int64_t)c.n * a.in_pk_sample_bytes + org * 16;
            sbm = reinterpret_cast<const unsigned char *>(a.unpool_bits + (int64_t)c.n * a.unpool_bits_stride) + org * 4;
            sbw = static_cast<const unsigned char *>(a.wt_h3) + (int64_t)c.g * nst * D_UBYTES;
        };
        auto run_advance = [&](Cursor &c) __attribute__((always_inline)) {
            if (++c.chunk == nst) {
                c.chunk = 0;
                if (++c.k < my_items) { locate(c); run_locate(c); }
            } else {
                sbq += 4 * q_plane * 16; sbm += 2 * q_plane * 4; sbw += D_UBYTES;
            }
        };
        // load L (0 .. 2) of a stage: hi piece, lo piece, mask dword (global loads with a scalar base: two SGPRs each where a
        // buffer descriptor takes four).  (s_nop 4: a scalar operand may come straight out of a v_readlane, and nothing pads a
        // VALU-written SGPR -> VMEM hazard inside an asm statement)
        auto load_q = [&](int L, QSet &qs) __attribute__((always_inline)) {
            if (L == 0) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(qs.hi) : "v"(q_voff), "s"(sbq) : "memory");
            else if (L == 1) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(qs.lo) : "v"(q_voff_lo), "s"(sbq) : "memory");
            else asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(qs.m) : "v"(q_moff), "s"(sbm) : "memory");
        };
        // (the wait is part of the statement that makes the registers readable: nothing can be scheduled between the two)
        auto landed_q = [&](QSet &qs) __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(qs.hi), "+v"(qs.lo), "+v"(qs.m)::"memory"); };
        // step 2 k + plane: window position k's piece of that plane = pooled piece & mask of the channels whose maximum sat at k
        auto expand_step = [&](int buf, const QSet &qs, int step) __attribute__((always_inline)) {
            const int k = step >> 1, plane = step & 1;
            const uint32_t byte = (qs.m >> (8 * k)) & 0xffu;
            const u32x4 msk = *reinterpret_cast<const u32x4 *>(lds_d + LUT0 + byte * 16);
            if (q_ok & (1u << k)) {
                const u32x4 v = plane == 0 ? qs.hi : qs.lo;
                *reinterpret_cast<u32x4 *>(lds_d + buf * PB + plane * D_PLANE + q_dst[k]) = v & msk;
            }
        };
        // the mask table: entry b = halves e of the piece are 0xffff where bit e of b is set
        if (tid < 256) {
            u32x4 e;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (((uint32_t)tid >> (2 * j)) & 1u ? 0xffffu : 0u) | (((uint32_t)tid >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u);
            *reinterpret_cast<u32x4 *>(lds_d + LUT0 + tid * 16) = e;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        Cursor cu = cc;             // weight DMA and pooled loads, one stage ahead of the multiply
        QSet qs;
        bool ep_due = false;
        int ep_par = 0;
        // prologue: patch(0) -> LDS, weights(0) in flight
        run_locate(cu);
#pragma unroll
        for (int L = 0; L < 3; ++L) load_q(L, qs);
        if (wave < 2) { epv = ep_src[cu.g * 64 + lane]; }
        ep_due = true; ep_par = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) dma_piece_at(sbw, 0, j);
        run_advance(cu);
        landed_q(qs);
#pragma unroll
        for (int st = 0; st < 8; ++st) expand_step(0, qs, st);
        // Iteration s: slots 0-4 the weight DMA of stage s + 1, slots 5-7 its three pooled loads; slot 23 waits for everything this
        // wave has in flight (the loads have had 16 slots of MFMAs to land; the DMA and an item's output stores are older), slots
        // 24-31 expand the pieces into the other patch buffer.  ONE register set: the second one of the fp32 form (loads two
        // stages ahead) bought SGPR spills here and nothing else — a pooled stage is 3 loads per lane, not 24.
        auto iteration = [&](const int s, auto more_tag) __attribute__((always_inline)) {
            constexpr bool MORE = decltype(more_tag)::value;
            // Entry: the barrier of stage s is behind us and tap 0's fragments are on their way into set 2 — both happened in front of the LAST tap
            // of stage s - 1 (the prologue for s = 0): all eight waves used to pass the barrier together, issue tap 0's eight reads together
            // (64 KiB through a 128 B / cycle LDS) and wait for them with idle matrix cores, once per stage (PMC, zero operands: conv1_2_D
            // 519 us with nothing but MFMAs and fragment reads, 423 us of MFMA cycles; the same finding as conv_wino4_h3.hip FORM 3).
            const uint32_t t0 = stamp(), t1 = t0, t2 = t0;
            if (pend) { store_item(pn, pty, ptx, pg, ppar); pend = false; }
            const uint32_t t3 = stamp();
            const int nb = (s + 1) & 1;
            if (MORE && cu.chunk == 0) {
                ep_due = true; ep_par = cu.k & 1;
                if (wave < 2) epv = ep_src[cu.g * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int slot = 0; slot < 36; ++slot) {
                const int t = slot >> 2, q = slot & 3;
                if (q == 0 && t + 1 < 9) fetch(s & 1, t + 1, (t + 1) & 1);
                if (MORE && slot == 32) {
                    // the barrier of stage s + 1: everything this wave issued in this iteration has had eight taps to complete (patch and weights
                    // of stage s + 1, the stores of an output stage); all reads of stage s are issued (tap 8's went out under tap 7) and complete
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ep_due && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[ep_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
                    ep_due = false;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    fetch(nb, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma3(t == 0 ? 2 : (t & 1), q);
                if (MORE) {
                    if (slot < 5) { if (!(ABL & 2)) dma_piece_at(sbw, nb, slot); }
                    else if (slot < 8) { if (!(ABL & 1)) load_q(slot - 5, qs); }
                    else if (slot == 23) landed_q(qs);
                    else if (slot >= 24 && slot < 32) { if (!(ABL & 16)) expand_step(nb, qs, slot - 24); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ABL & 64) {
                const uint32_t t4 = stamp();
                st_wait += (t1 - t0) & 0xfffffu; st_bar += (t2 - t1) & 0xfffffu; st_out += (t3 - t2) & 0xfffffu; st_mul += (t4 - t3) & 0xfffffu;
            }
            if (MORE) run_advance(cu);
            end_of_stage();
        };
        // the barrier of stage 0 and its tap 0 (what every iteration does for its successor in front of its last tap)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ep_due && wave < 2) reinterpret_cast<float *>(lds_d + EP0)[ep_par * 128 + tid] = wave == 0 ? epv * mscale : epv;
        ep_due = false;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fetch(0, 0, 2);
        for (int s = 0; s + 1 < total; ++s) iteration(s, std::true_type{});
        iteration(total - 1, std::false_type{});
    }
    if (pend) store_item(pn, pty, ptx, pg, ppar);
    if (ovf > (0x477fe000u << 1)) atomicOr(a.h3_flag, 1u);          // !(|xs| <= 65504): a scaled input left the fp16 range (or was not finite)
    if ((ABL & 64) && a.vmax && lane == 0) {
        atomicAdd(a.vmax + 0, st_wait >> 4); atomicAdd(a.vmax + 1, st_bar >> 4); atomicAdd(a.vmax + 2, st_out >> 4); atomicAdd(a.vmax + 3, st_mul >> 4);
        atomicAdd(a.vmax + 4, (uint32_t)total);
        atomicMax(a.vmax + 5, stamp() - st_begin);
        if (wave == 0) { a.vmax[8 + 2 * blockIdx.x] = stamp() - st_begin; a.vmax[9 + 2 * blockIdx.x] = (uint32_t)total; }      // per workgroup: cycles, stages
    }
}

// largest |x| of a tensor (calibration passes only): atomicMax on the bit pattern (non-negative floats order like their bits)
__global__ __launch_bounds__(256) void absmax_kernel(const float *x, int64_t n, uint32_t *out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) {
        const uint32_t b = __float_as_uint(m);
        if (b > *out) atomicMax(out, b);
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool conv3_h3_supported(int ks, int cin, int cout, int H, int W, bool unpool) {
    if (ks != 3 || cin % D_KC || cin < 2 * D_KC || cout % 64 || H < 1 || W < 1) return false;
    if (unpool && ((H | W) & 1)) return false;
    return (int64_t)cin * H * W * 4 < (1ll << 31) && (int64_t)cout * H * W * 4 < (1ll << 31);
}

static inline uint16_t d3_f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static inline float d3_f16_value(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// Caffe (Cout,Cin,3,3) -> fp16 hi / lo planes in stage order [cout / 64][Cin / 16][tap][32-cout block][plane][octet][cout][8];
// returns the power of two the weights were multiplied by (max |W| * scale in [2^7, 2^8))
float conv3_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out) {
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * 9; ++i) wmax = std::fmax(wmax, std::fabs(W[i]));
    int ex = 0;
    if (wmax > 0.f) (void)std::frexp(wmax, &ex);
    const float scale = std::ldexp(1.f, 8 - ex);
    const int nst = cin / D_KC, ng = cout / 64;
    out.assign((size_t)ng * nst * 9 * 2048, 0);
    for (int g = 0; g < ng; ++g)
        for (int c = 0; c < nst; ++c)
            for (int t = 0; t < 9; ++t)
                for (int b = 0; b < 2; ++b) {
                    uint16_t *img = out.data() + ((((size_t)g * nst + c) * 9 + t) * 2 + b) * 1024;      // hi plane: 512 halfs, then lo
                    for (int o = 0; o < 2; ++o)
                        for (int r = 0; r < 32; ++r)
                            for (int e = 0; e < 8; ++e) {
                                const float x = W[((size_t)(g * 64 + b * 32 + r) * cin + c * D_KC + o * 8 + e) * 9 + t] * scale;
                                const uint16_t hi = d3_f16_bits(x);
                                const uint16_t lo = d3_f16_bits(x - d3_f16_value(hi));
                                const size_t at = (size_t)(o * 32 + r) * 8 + e;
                                img[at] = hi;
                                img[512 + at] = lo;
                            }
                }
    return scale;
}

// Which instantiation a launch takes: IN from the input form (a0.in_pk set: packed, with a0.unpool_bits through an Upsample;
// else fp32, with a0.unpool_mask through an Upsample), OUT from a0.out_pk.
template <int IN, bool OUT_PK>
static void launch_d3_form(const ConvArgs &a, int form, dim3 grid, size_t lds, hipStream_t s) {
    if constexpr (IN == IN_F32 || IN == IN_F32_UNPOOL) {
        if (form == 0 && !OUT_PK) {
            hipLaunchKernelGGL((conv3_h3_kernel<IN, false, 0>), grid, dim3(512), lds, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((conv3_h3_kernel<IN, OUT_PK, 1>), grid, dim3(512), lds, s, a);
}

void launch_conv3_h3(const ConvArgs &a0, hipStream_t s) {
    const bool pk_in = a0.in_pk != nullptr, pk_out = a0.out_pk != nullptr;
    const bool unpool = pk_in ? a0.unpool_bits != nullptr : a0.unpool_mask != nullptr;
    if (!a0.wt_h3 || !(a0.h3_vscale > 0.f) || !a0.h3_flag || a0.drop_site >= 0 || a0.pool_out || a0.CoutPad != a0.Cout ||
        !conv3_h3_supported(3, a0.Cin, a0.Cout, a0.H, a0.W, unpool))
        throw std::invalid_argument("launch_conv3_h3: unsupported layer");
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once) {
        for (const void *f : {(const void *)conv3_h3_kernel<IN_F32, false, 0>, (const void *)conv3_h3_kernel<IN_F32_UNPOOL, false, 0>,
                              (const void *)conv3_h3_kernel<IN_F32, false, 1>, (const void *)conv3_h3_kernel<IN_F32_UNPOOL, false, 1>,
                              (const void *)conv3_h3_kernel<IN_F32, true, 1>,
                              (const void *)conv3_h3_kernel<IN_PK, false, 1>, (const void *)conv3_h3_kernel<IN_PK_UNPOOL, false, 1>,
                              (const void *)conv3_h3_kernel<IN_PK, true, 1>, (const void *)conv3_h3_kernel<IN_PK_UNPOOL, true, 1>})
            SIVO_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    const dim3 grid((unsigned)((n_cu / 8) * 8 > 0 ? (n_cu / 8) * 8 : 8));      // one persistent workgroup per CU, a multiple of the 8 XCDs
    ConvArgs a = a0;
    a.tiles_x = (a.W + D_TW - 1) / D_TW;
    a.tiles_y = (a.H + D_TH - 1) / D_TH;
    // packed tensors: the zero border and the overhang of partial items must exist (see the file header)
    if (pk_in) {
        const int need_h = unpool ? a.tiles_y * (D_TH / 2) + 2 : a.tiles_y * D_TH + 2, need_w = unpool ? a.tiles_x * (D_TW / 2) + 2 : a.tiles_x * D_TW + 2;
        if (a.in_Hp < need_h || a.in_Wp < need_w || (int64_t)a.Cin * a.in_Hp * a.in_Wp * 4 >= (1ll << 31))
            throw std::invalid_argument("launch_conv3_h3: packed input plane too small for the layer's tiling");
    }
    if (pk_out && (a.out_Hp < a.H + 2 || a.out_Wp < a.W + 2 || !(a.out_vscale > 0.f) || (int64_t)a.Cout * a.out_Hp * a.out_Wp * 4 >= (1ll << 31)))
        throw std::invalid_argument("launch_conv3_h3: packed output plane too small / no scale");
    // the whole LDS of the CU, as conv_wino4_h3.hip (no other workgroup beside a persistent one)
    const size_t lds = (size_t)160 * 1024;
    lds_claim_note(LDS_CLAIM_CONV3_H3, lds);
    // SIVO_D3_FORM=0: the phased stage loop of the fp32-input forms (read at every launch: tests compare the two forms bit for bit)
    const int form = SIVO_DIAG_ENV("SIVO_D3_FORM") && std::atoi(SIVO_DIAG_ENV("SIVO_D3_FORM")) == 0 ? 0 : 1;
#ifdef SIVO_DIAG
    if (const char *ab = SIVO_DIAG_ENV("SIVO_D3_ABL")) {
#define D3_ABL_CASE(n)                                                                                                                      \
    case n:                                                                                                                                 \
        if (form == 0) {                                                                                                                    \
            SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_h3_kernel<IN_F32, false, 0, n>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            hipLaunchKernelGGL((conv3_h3_kernel<IN_F32, false, 0, n>), grid, dim3(512), lds, s, a);                                         \
        } else {                                                                                                                            \
            SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_h3_kernel<IN_F32, false, 1, n>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            hipLaunchKernelGGL((conv3_h3_kernel<IN_F32, false, 1, n>), grid, dim3(512), lds, s, a);                                         \
        }                                                                                                                                   \
        return;
#define D3PK_ABL_CASE(IN_, OUT_, n)                                                                                                         \
    case n:                                                                                                                                 \
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_h3_kernel<IN_, OUT_, 1, n>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((conv3_h3_kernel<IN_, OUT_, 1, n>), grid, dim3(512), lds, s, a);                                                 \
        return;
#define D3PK_ABL_SET(IN_, OUT_) switch (std::atoi(ab)) { D3PK_ABL_CASE(IN_, OUT_, 1) D3PK_ABL_CASE(IN_, OUT_, 2) D3PK_ABL_CASE(IN_, OUT_, 3) D3PK_ABL_CASE(IN_, OUT_, 4) D3PK_ABL_CASE(IN_, OUT_, 7) D3PK_ABL_CASE(IN_, OUT_, 8) D3PK_ABL_CASE(IN_, OUT_, 16) D3PK_ABL_CASE(IN_, OUT_, 23) D3PK_ABL_CASE(IN_, OUT_, 64) default: break; }
        if (pk_in && !unpool && pk_out) D3PK_ABL_SET(IN_PK, true)
        if (pk_in && unpool && !pk_out) D3PK_ABL_SET(IN_PK_UNPOOL, false)
        if (pk_in && unpool && pk_out) D3PK_ABL_SET(IN_PK_UNPOOL, true)
#undef D3PK_ABL_SET
#undef D3PK_ABL_CASE
        if (!unpool && !pk_in && !pk_out) switch (std::atoi(ab)) {
            D3_ABL_CASE(1) D3_ABL_CASE(2) D3_ABL_CASE(3) D3_ABL_CASE(4) D3_ABL_CASE(8) D3_ABL_CASE(16) D3_ABL_CASE(19) D3_ABL_CASE(23) D3_ABL_CASE(64) D3_ABL_CASE(65) D3_ABL_CASE(68)
            default: break;
        }
#undef D3_ABL_CASE
    }
#endif
    const int in = pk_in ? (unpool ? IN_PK_UNPOOL : IN_PK) : (unpool ? IN_F32_UNPOOL : IN_F32);
    switch (in * 2 + (pk_out ? 1 : 0)) {
        case 0: launch_d3_form<IN_F32, false>(a, form, grid, lds, s); break;
        case 1: launch_d3_form<IN_F32, true>(a, form, grid, lds, s); break;
        case 2: launch_d3_form<IN_F32_UNPOOL, false>(a, form, grid, lds, s); break;
        case 3: throw std::invalid_argument("launch_conv3_h3: fp32 input through an Upsample with packed output is not built (it spills)");
        case 4: launch_d3_form<IN_PK, false>(a, form, grid, lds, s); break;
        case 5: launch_d3_form<IN_PK, true>(a, form, grid, lds, s); break;
        case 6: launch_d3_form<IN_PK_UNPOOL, false>(a, form, grid, lds, s); break;
        default: launch_d3_form<IN_PK_UNPOOL, true>(a, form, grid, lds, s); break;
    }
}

// the load-time accuracy guard (segnet.cpp accuracy_guard): largest |a - b| and |b| of two tensors (bit patterns, atomicMax) and
// the sums of (a - b)^2 and b^2 in f64
__global__ __launch_bounds__(256) void absdiff_max_kernel(const float *a, const float *b, int64_t n, uint32_t *out_bits, double *out_sums) {
    float md = 0.f, mb = 0.f;
    double sd = 0.0, sb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[i], y = b[i], d = x - y;
        md = fmaxf(md, fabsf(d)); mb = fmaxf(mb, fabsf(y));
        if (!(fabsf(d) <= 3.0e38f)) md = 3.0e38f;              // NaN / inf in the layer under test: an error of any size
        sd += (double)d * d; sb += (double)y * y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        md = fmaxf(md, __shfl_xor(md, off)); mb = fmaxf(mb, __shfl_xor(mb, off));
        sd += __shfl_xor(sd, off); sb += __shfl_xor(sb, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out_bits + 0, __float_as_uint(md));
        atomicMax(out_bits + 1, __float_as_uint(mb));
        atomicAdd(out_sums + 0, sd);
        atomicAdd(out_sums + 1, sb);
    }
}

void launch_absdiff_max(const float *a, const float *b, int64_t n, uint32_t *out_bits, double *out_sums, hipStream_t s) {
    const int blocks = (int)std::min<int64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(absdiff_max_kernel, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, s, a, b, n, out_bits, out_sums);
}

void launch_absmax(const float *x, int64_t n, uint32_t *out_bits, hipStream_t s) {
    const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, s, x, n, out_bits);
}

}  // namespace sivo
