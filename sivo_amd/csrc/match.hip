// match.hip — 256-bit Hamming matching on CDNA4.
//
// Stands behind ORBmatcher::DescriptorDistance (reference src/orbslam/ORBmatcher.cc:1582-1596:
// 8 x (xor, SWAR popcount) — v_bcnt_u32_b32 here) and the candidate-list
// best / second-best inner loop of the Search* routines (ORBmatcher.cc:78-104).
// These are integer, HBM/latency-bound kernels: descriptors are read as 2 x 16-byte
// loads per row, distances are produced with wave-wide coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace sivo {

__device__ __forceinline__ int ham256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Dense matrix: a workgroup owns 256 columns (one B row per thread, held in registers)
// x ROWS rows of A (staged once in LDS, read back as same-address broadcasts).
// Each output row segment is one coalesced 1 KiB store per workgroup.
constexpr int HM_ROWS = 32;
__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint4 *__restrict__ A, int nA,
                                                            const uint4 *__restrict__ B, int nB,
                                                            int32_t *__restrict__ out) {
    __shared__ uint4 sA[HM_ROWS * 2];
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * HM_ROWS;
    if (threadIdx.x < HM_ROWS * 2) {
        const int r = i0 + threadIdx.x / 2;
        sA[threadIdx.x] = r < nA ? A[(int64_t)r * 2 + (threadIdx.x & 1)] : make_uint4(0, 0, 0, 0);
    }
    uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
    if (j < nB) { b0 = B[(int64_t)j * 2]; b1 = B[(int64_t)j * 2 + 1]; }
    __syncthreads();
    if (j >= nB) return;
#pragma unroll 8
    for (int r = 0; r < HM_ROWS; ++r) {
        if (i0 + r >= nA) break;
        out[(int64_t)(i0 + r) * nB + j] = ham256(sA[2 * r], sA[2 * r + 1], b0, b1);
    }
}

// Top-2 of a (key, row) multiset with key = dist << 22 | list position: the reference's sequential scan
// (`dist < best` moves best to second, `else if dist < second`; ORBmatcher.cc:99-110) ends with best = first minimum
// and second = first minimum of the rest, which is exactly the two smallest keys.
struct Top2 { uint32_t k1, k2; int j1, j2; };
__device__ __forceinline__ void top2_push(Top2 &t, uint32_t k, int j) {
    if (k < t.k1) { t.k2 = t.k1; t.j2 = t.j1; t.k1 = k; t.j1 = j; }
    else if (k < t.k2) { t.k2 = k; t.j2 = j; }
}

// One wave per query: lanes stride over the candidate list; the butterfly merges the lanes' top-2 sets.
// cand_idx == nullptr: candidates are rows [0, nB).  Lists hold fewer than 2^22 candidates.
__global__ __launch_bounds__(256) void hamming_argmin2_kernel(const uint4 *__restrict__ A, int nA,
                                                             const uint4 *__restrict__ B, int nB,
                                                             const int32_t *__restrict__ cand_off,
                                                             const int32_t *__restrict__ cand_idx,
                                                             int32_t *best_idx, int32_t *best_dist,
                                                             int32_t *second_dist, int32_t *second_idx) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nA) return;
    const uint4 a0 = A[(int64_t)q * 2], a1 = A[(int64_t)q * 2 + 1];
    const int c0 = cand_off ? cand_off[q] : 0, c1 = cand_off ? cand_off[q + 1] : nB;
    Top2 t{0xffffffffu, 0xffffffffu, -1, -1};
    for (int c = c0 + lane; c < c1; c += 64) {
        const int j = cand_idx ? cand_idx[c] : c;
        const int d = ham256(a0, a1, B[(int64_t)j * 2], B[(int64_t)j * 2 + 1]);
        if (d < 256) top2_push(t, ((uint32_t)d << 22) | (uint32_t)(c - c0), j);     // `dist < 256` is the reference's initial test
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t ok1 = __shfl_xor(t.k1, off), ok2 = __shfl_xor(t.k2, off);
        const int oj1 = __shfl_xor(t.j1, off), oj2 = __shfl_xor(t.j2, off);
        top2_push(t, ok1, oj1);
        top2_push(t, ok2, oj2);
    }
    if (lane == 0) {
        best_idx[q] = t.j1;
        best_dist[q] = t.j1 >= 0 ? (int)(t.k1 >> 22) : 256;
        second_dist[q] = t.j2 >= 0 ? (int)(t.k2 >> 22) : 256;
        if (second_idx) second_idx[q] = t.j2;
    }
}

}  // namespace sivo

using namespace sivo;

extern "C" int sivo_hamming_matrix_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, int n_b, int32_t *d_out,
                                       void *stream) {
    return guarded([&] {
        if (n_a < 0 || n_b < 0) throw std::invalid_argument("negative size");
        if (n_a == 0 || n_b == 0) return SIVO_OK;
        if (!d_a || !d_b || !d_out) throw std::invalid_argument("null argument");
        dim3 grid((unsigned)cdiv(n_b, 256), (unsigned)cdiv(n_a, HM_ROWS));
        hipLaunchKernelGGL(hamming_matrix_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_a, n_a,
                           (const uint4 *)d_b, n_b, d_out);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_hamming_argmin2_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, const int32_t *d_cand_off,
                                        const int32_t *d_cand_idx, int32_t *d_best_idx, int32_t *d_best_dist,
                                        int32_t *d_second_dist, int32_t *d_second_idx, void *stream) {
    return guarded([&] {
        if (n_a == 0) return SIVO_OK;
        if (!d_a || !d_b || !d_cand_off || !d_cand_idx || !d_best_idx || !d_best_dist || !d_second_dist || n_a < 0)
            throw std::invalid_argument("null argument");
        hipLaunchKernelGGL(hamming_argmin2_kernel, dim3((unsigned)cdiv(n_a, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)d_a, n_a, (const uint4 *)d_b, 0, d_cand_off, d_cand_idx, d_best_idx,
                           d_best_dist, d_second_dist, d_second_idx);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_hamming_bruteforce_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, int n_b,
                                           int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist,
                                           void *stream) {
    return guarded([&] {
        if (n_a == 0) return SIVO_OK;
        if (!d_a || (!d_b && n_b) || !d_best_idx || !d_best_dist || !d_second_dist || n_a < 0 || n_b < 0)
            throw std::invalid_argument("null argument");
        hipLaunchKernelGGL(hamming_argmin2_kernel, dim3((unsigned)cdiv(n_a, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)d_a, n_a, (const uint4 *)d_b, n_b, nullptr, nullptr, d_best_idx, d_best_dist,
                           d_second_dist, nullptr);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

namespace {
template <class T>
struct DevBuf {
    T *p = nullptr;
    explicit DevBuf(size_t n) : p(dev_alloc<T>(n)) {}
    ~DevBuf() { (void)hipFree(p); }
};
}  // namespace

extern "C" int sivo_hamming_matrix(const uint8_t *a, int n_a, const uint8_t *b, int n_b, int32_t *out) {
    return guarded([&] {
        if (n_a < 0 || n_b < 0) throw std::invalid_argument("negative size");
        if (n_a == 0 || n_b == 0) return SIVO_OK;
        if (!a || !b || !out) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        DevBuf<uint8_t> da((size_t)n_a * 32), db((size_t)n_b * 32);
        DevBuf<int32_t> dout((size_t)n_a * n_b);
        SIVO_HIP(hipMemcpy(da.p, a, (size_t)n_a * 32, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(db.p, b, (size_t)n_b * 32, hipMemcpyHostToDevice));
        int rc = sivo_hamming_matrix_dev(da.p, n_a, db.p, n_b, dout.p, nullptr);
        if (rc) return rc;
        SIVO_HIP(hipMemcpy(out, dout.p, (size_t)n_a * n_b * 4, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}

extern "C" int sivo_hamming_argmin2(const uint8_t *a, int n_a, const uint8_t *b, int n_b, const int32_t *cand_off,
                                    const int32_t *cand_idx, int32_t *best_idx, int32_t *best_dist,
                                    int32_t *second_dist, int32_t *second_idx) {
    return guarded([&] {
        if (n_a < 0 || n_b < 0) throw std::invalid_argument("negative size");
        if (n_a == 0) return SIVO_OK;
        if (!a || !cand_off || !best_idx || !best_dist || !second_dist) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        const int ncand = cand_off[n_a];
        DevBuf<uint8_t> da((size_t)n_a * 32), db((size_t)n_b * 32);
        DevBuf<int32_t> doff((size_t)n_a + 1), didx((size_t)ncand), dbi(n_a), dbd(n_a), dsd(n_a), dsi(n_a);
        SIVO_HIP(hipMemcpy(da.p, a, (size_t)n_a * 32, hipMemcpyHostToDevice));
        if (n_b) SIVO_HIP(hipMemcpy(db.p, b, (size_t)n_b * 32, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(doff.p, cand_off, ((size_t)n_a + 1) * 4, hipMemcpyHostToDevice));
        if (ncand) SIVO_HIP(hipMemcpy(didx.p, cand_idx, (size_t)ncand * 4, hipMemcpyHostToDevice));
        int rc = sivo_hamming_argmin2_dev(da.p, n_a, db.p, doff.p, didx.p, dbi.p, dbd.p, dsd.p, dsi.p, nullptr);
        if (rc) return rc;
        SIVO_HIP(hipMemcpy(best_idx, dbi.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(best_dist, dbd.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(second_dist, dsd.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        if (second_idx) SIVO_HIP(hipMemcpy(second_idx, dsi.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}
