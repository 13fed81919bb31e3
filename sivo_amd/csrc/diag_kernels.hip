// diag_kernels.hip — diagnostic build only (libsivo_hip_diag.so; empty in the product): a kernel that does nothing but hold a chosen
// amount of every CU's LDS for a chosen time, optionally with LDS traffic of its own, so that other kernels' workgroups are placed
// BESIDE it — at LDS bases they never see when the CU is theirs (tools/coresident_probe.py occupant).
#ifdef SIVO_DIAG
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "h3_split.hpp"
#include "lds_dma.hpp"

namespace sivo {

// mode 0: sleeps; 1: ds_read / ds_write over its own LDS; 2: LDS-DMA (global_load_lds_dwordx4) from `src` into its own LDS
__global__ __launch_bounds__(512) void occupy_kernel(int lds_bytes, int mode, long long cycles, const uint32_t *src, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char occ_lds[];
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();        // 100 MHz
    uint32_t acc = 0;
    const int words = lds_bytes / 4;
    if (mode == 1) for (int i = threadIdx.x; i < words; i += 512) reinterpret_cast<uint32_t *>(occ_lds)[i] = (uint32_t)i;
    __syncthreads();
    const uint32_t base = lds_addr_uniform(occ_lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int k = 0;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < cycles) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else if (mode == 1) {
            const int i = (threadIdx.x * 4 + 2048 * k) % (words - 4);
            const uint4 v = *reinterpret_cast<const uint4 *>(occ_lds + (size_t)(i & ~3) * 4);
            acc += v.x ^ v.y ^ v.z ^ v.w;
            reinterpret_cast<uint32_t *>(occ_lds)[(i + 1) % words] = acc;
        } else if (mode >= 3) {
            // a stage of the f16x3 GEMM without (3) / with (4) its matrix-core work: 4 LDS-DMA pieces per wave two stages ahead,
            // 24 ds_read_b128 + 4 ds_write_b128 per wave, one barrier
            const int pieces = lds_bytes / 1024;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int piece = (wave * 4 + q + 32 * k) % pieces;
                lds_dma16_s(src, (uint32_t)(((piece * 64 + lane) * 16) % (1 << 20)), (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)piece * 1024)));
            }
            uint4 sum = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(occ_lds + (size_t)(((wave * 24 + q + 7 * k) % pieces) * 1024 + lane * 16));
                sum.x ^= v.x; sum.y += v.y; sum.z ^= v.z; sum.w += v.w;
            }
            if (mode == 4) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                typedef float f16v __attribute__((ext_vector_type(16)));
                f16v c = {0};
                h8 x, y;
                for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(float)(sum.x & 3u); y[e] = (_Float16)(float)(sum.y & 3u); }
#pragma unroll
                for (int q = 0; q < 16; ++q) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
                sum.x ^= __float_as_uint(c[0]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4 *>(occ_lds + (size_t)(((wave * 4 + q + 11 * k) % pieces) * 1024 + lane * 16)) = sum;
            acc += sum.x ^ sum.w;
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        } else {
            const int pieces = lds_bytes / 1024;
            const int piece = (wave + 8 * k) % pieces;
            lds_dma16_s(src, (uint32_t)(((piece * 64 + lane) * 16) % (1 << 20)), (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)piece * 1024)));
            if ((k & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ++k;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (mode && sink && acc == 0x12345678u) sink[0] = acc + occ_lds[threadIdx.x];
}

void launch_occupy(int lds_bytes, int mode, int microseconds, const uint32_t *src, uint32_t *sink, hipStream_t s) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once)
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(occupy_kernel, dim3(256), dim3(512), (size_t)lds_bytes, s, lds_bytes, mode, (long long)microseconds * 100, src, sink);   // s_memrealtime: 100 MHz
    SIVO_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// The LDS access pattern of wino4_bridge_kernel (conv_wino4.hip) on SYNTHETIC, self-checking data (tools/coresident_repro.py):
// zero-fill of the plane by all threads -> barrier -> every thread writes the 4 x 4 pixels of its tiles at [y + 1][4 tx + 1 ..]
// (odd dword alignment: ds_write2_b32 pairs, as the bridge compiles) -> barrier -> every thread reads its 6 x 6 window back as
// ds_read_b128 + ds_read_b64 per row and compares all 36 words with what they MUST be (a function of round, workgroup and pixel;
// zero on the border).  `jitter`: dwords each thread loads from `src` in front of its writes (the bridge's M loads: the waves
// reach the LDS phase at different times).  Report words (rep): [0] workgroups run, [1] workgroups whose LDS allocation starts
// at >= 112 KB (s_getreg LDS_ALLOC: they ran BESIDE a large LDS user on their CU), [2] window words that differed, [3] rounds run by
// co-resident workgroups; first difference: [4] round, [5] window word (row * 6 + col), [6] expected bits, [7] bits read,
// [8] LDS_ALLOC register, [9] workgroup, [10] tile, [11] what the same word reads a second time (after another barrier);
// [12 .. 47] per window word (36): differences at that word; [48] / [49] LDS_ALLOC of a co-resident / a lone workgroup.
// PK (tools/coresident_repro.py "pk"): every thread also runs the bridge's ARITHMETIC on its (verified) window — the two 1-D input
// transforms (as wino4_bt, conv_wino4.hip; the compiler turns them into v_pk_mul_f32 / v_pk_add_f32) and the fp16 split of the 36 results —
// TWICE from the same registers, and compares the packed words row by row: [51] rows of six words whose two computations differ,
// first such row: [52] row, [53] / [54] hash of the first / second computation, [55] thread, [56] LDS_ALLOC, [57] rows computed / 2^20
__device__ __forceinline__ void victim_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float *t) {
    const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    t[0] = 4.f * d0 - 5.f * d2 + d4; t[1] = a + b; t[2] = a - b; t[3] = c + e; t[4] = c - e; t[5] = 4.f * d1 - 5.f * d3 + d5;
}
__device__ __forceinline__ void victim_rows(const float (&d)[6][6], float scale, uint32_t (&h)[6]) {
    float tb[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col[6];
        victim_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], col);
#pragma unroll
        for (int i = 0; i < 6; ++i) tb[i][j] = col[i];
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float row[6];
        victim_bt(tb[i][0], tb[i][1], tb[i][2], tb[i][3], tb[i][4], tb[i][5], row);
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x = x * 0x9E3779B1u + wino4_pack_h3(row[j], scale, bad);
        h[i] = x;
    }
}

template <bool PK>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PK ? 4 : 7, 8))) void lds_victim_kernel(int H, int W, int rounds, int jitter, const uint32_t *src, uint32_t *rep) {
    extern __shared__ float vplane[];
    const int th = (H + 3) / 4, tw = W / 4, ntile = th * tw, RS = W + 4, rows = 4 * th + 2;
    const uint32_t alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);          // HW_REG_LDS_ALLOC: [7:0] base, [20:12] size
    // (measured on gfx950: [11:0] base, [20:12] size, both in 256-byte granules)  beside = the workgroup sits above >= 112 KB of somebody
    // else's LDS — an occupant / GEMM workgroup, or a stack of its own kind; [50] counts bases of exactly 112 / 128 KB (one big neighbour)
    const uint32_t base_b = (alloc & 0xfffu) * 256u;
    const bool beside = base_b >= 112u * 1024u;
    uint32_t bad = 0;
    auto pix = [&](int r, int y, int x) -> float {        // the value of image pixel (y, x) in round r: never 0, never a NaN pattern
        return __uint_as_float(0x3f000000u | ((uint32_t)(r & 0x3f) << 17) | ((uint32_t)(blockIdx.x & 0xf) << 13) | (uint32_t)(y * W + x + 1) % 8191u + 1u);
    };
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < rows * RS; i += blockDim.x) vplane[i] = 0.f;
        __syncthreads();
        uint32_t acc = 0;
        for (int k = 0; k < jitter; ++k) acc ^= src[((blockIdx.x * 131 + r * 17 + k) * 1024 + threadIdx.x) & ((1 << 18) - 1)];
        for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
            const int tx = t % tw, ty = t / tw;
            float vv[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) vv[i][q] = pix(r, 4 * ty + i, 4 * tx + q);
            if (acc == 0x9e3779b9u) vv[0][0] = 1.f;      // (keeps the loads alive)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = 4 * ty + i;
                if (y >= H) break;
                float *dst = vplane + (y + 1) * RS + 4 * tx + 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = vv[i][q];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
            const int tx = t % tw, ty = t / tw;
            const float *win = vplane + (4 * ty) * RS + 4 * tx;
            float d[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 q = *reinterpret_cast<const float4 *>(win + i * RS);
                const float2 r2 = *reinterpret_cast<const float2 *>(win + i * RS + 4);
                d[i][0] = q.x; d[i][1] = q.y; d[i][2] = q.z; d[i][3] = q.w; d[i][4] = r2.x; d[i][5] = r2.y;
            }
            if (PK) {
                uint32_t h1[6], h2[6];
                victim_rows(d, 16.f, h1);
                asm volatile("" ::: "memory");          // the window is read from LDS once more (it was verified by the plain variant): each
#pragma unroll                                          // computation is the bridge's own sequence, reads -> transforms -> split
                for (int i = 0; i < 6; ++i) {
                    const float4 q = *reinterpret_cast<const float4 *>(win + i * RS);
                    const float2 r2 = *reinterpret_cast<const float2 *>(win + i * RS + 4);
                    d[i][0] = q.x; d[i][1] = q.y; d[i][2] = q.z; d[i][3] = q.w; d[i][4] = r2.x; d[i][5] = r2.y;
                }
                victim_rows(d, 16.f, h2);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (h1[i] != h2[i] && atomicAdd(rep + 51, 1u) == 0u) {
                        rep[52] = (uint32_t)i; rep[53] = h1[i]; rep[54] = h2[i]; rep[55] = threadIdx.x; rep[56] = alloc;
                    }
            }
            if (!PK)          // (the PK variant leaves the window check to the plain one: the registers are needed for the arithmetic)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int y = 4 * ty + i - 1, x = 4 * tx + j - 1;
                    const float want = (y >= 0 && y < H && x >= 0 && x < W) ? pix(r, y, x) : 0.f;
                    if (__float_as_uint(want) != __float_as_uint(d[i][j])) {
                        atomicAdd(rep + 12 + i * 6 + j, 1u);
                        if (atomicAdd(rep + 2, 1u) == 0u) {
                            rep[4] = (uint32_t)r; rep[5] = (uint32_t)(i * 6 + j); rep[6] = __float_as_uint(want); rep[7] = __float_as_uint(d[i][j]);
                            rep[8] = alloc; rep[9] = blockIdx.x; rep[10] = (uint32_t)t;
                            bad = 1u + (uint32_t)(i * 6 + j);
                        }
                    }
                }
        }
        __syncthreads();
        if (bad) {          // the thread that recorded the first difference reads the same word once more
            const int t = (int)rep[10], tx = t % tw, ty = t / tw, w = (int)bad - 1;
            rep[11] = __float_as_uint(vplane[(4 * ty + w / 6) * RS + 4 * tx + w % 6]);
            bad = 0;
        }
    }
    if (threadIdx.x == 0) {
        atomicAdd(rep + 0, 1u);
        if (beside) { atomicAdd(rep + 1, 1u); atomicAdd(rep + 3, (uint32_t)rounds); rep[48] = alloc; }
        if (base_b == 112u * 1024u || base_b == 128u * 1024u) atomicAdd(rep + 50, 1u);
        else rep[49] = alloc;
    }
}

// One packed-FP32 instruction form in a loop on known operands (tools/pkform_repro.py).  The forms are the ones that occur
// in wino4_bridge_kernel's packed build and in no kernel that is known to be safe; the first computes, IN PLACE, {a - a, a - b} from the
// register pair {a, b}: its high half reads the LOW source element, which the same instruction overwrites with its low result.
//   form 0: v_pk_add_f32 p, p, p op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]      (in place)            expected {0, a - b}
//   form 1: the same into another register pair                                                       expected {0, a - b}
//   form 2: v_pk_add_f32 p, p, q (plain, in place)                                                    expected {a + c, b + d}
// rep: [0] workgroups, [1] instructions checked / 2^20, [2] wrong high halves, [3] wrong low halves, [4] of the wrong high halves those that
// equal -b (= low RESULT - b: the high half read the overwritten register), first wrong: [5] a, [6] b, [7] high found, [8] lane, [9] form
__global__ __launch_bounds__(1024) void pkform_victim_kernel(int form, int rounds, uint32_t *rep) {
    extern __shared__ float pkf_lds[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    pkf_lds[threadIdx.x] = 0.f;           // (a dynamic LDS allocation like the bridge's, so that the workgroup is placed like one)
    uint32_t st = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    unsigned bad_hi = 0, bad_lo = 0, bad_hi_is_minus_b = 0;
    for (int r = 0; r < rounds; ++r) {
        st = st * 1664525u + 1013904223u;
        const float a = __uint_as_float(0x3f800000u | (st >> 9));                 // [1, 2)
        st = st * 1664525u + 1013904223u;
        const float b = __uint_as_float(0x40000000u | (st >> 9));                 // [2, 4)
        f2 p = {a, b}, q = {b, a};
        float want_lo, want_hi;
        if (form == 0) {
            asm volatile("v_pk_add_f32 %0, %0, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p));
            want_lo = 0.f; want_hi = a - b;
        } else if (form == 1) {
            f2 d;
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d) : "v"(p));
            p = d; want_lo = 0.f; want_hi = a - b;
        } else {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
            want_lo = a + b; want_hi = b + a;
        }
        if (__float_as_uint(p.y) != __float_as_uint(want_hi)) {
            if (bad_hi++ == 0 && atomicAdd(rep + 2, 0u) == 0u) { rep[5] = __float_as_uint(a); rep[6] = __float_as_uint(b); rep[7] = __float_as_uint(p.y); rep[8] = threadIdx.x & 63; rep[9] = (uint32_t)form; }
            if (__float_as_uint(p.y) == __float_as_uint(0.f - b)) ++bad_hi_is_minus_b;
        }
        if (__float_as_uint(p.x) != __float_as_uint(want_lo)) ++bad_lo;
    }
    if (bad_hi) atomicAdd(rep + 2, bad_hi);
    if (bad_lo) atomicAdd(rep + 3, bad_lo);
    if (bad_hi_is_minus_b) atomicAdd(rep + 4, bad_hi_is_minus_b);
    if (threadIdx.x == 0) { atomicAdd(rep + 0, 1u); if (pkf_lds[0] == 1.f) rep[10] = 1; }
}
void launch_pkform_victim(int grid, int threads, int lds_bytes, int form, int rounds, uint32_t *rep, hipStream_t s) {
    hipLaunchKernelGGL(pkform_victim_kernel, dim3(grid), dim3(threads), (size_t)lds_bytes, s, form, rounds, rep);
    SIVO_HIP(hipGetLastError());
}

void launch_lds_victim(int grid, int H, int W, int rounds, int jitter, const uint32_t *src, uint32_t *rep, hipStream_t s, bool pk) {
    const int th = (H + 3) / 4, tw = W / 4, ntile = th * tw;
    const int nthr = ntile >= 1024 ? 1024 : (ntile + 63) / 64 * 64;             // as launch_conv_wino4 launches the bridge
    const size_t lds = (size_t)(4 * th + 2) * (W + 4) * sizeof(float);
    if (pk) hipLaunchKernelGGL(lds_victim_kernel<true>, dim3(grid), dim3(nthr), lds, s, H, W, rounds, jitter, src, rep);
    else hipLaunchKernelGGL(lds_victim_kernel<false>, dim3(grid), dim3(nthr), lds, s, H, W, rounds, jitter, src, rep);
    SIVO_HIP(hipGetLastError());
}

}  // namespace sivo
#endif
