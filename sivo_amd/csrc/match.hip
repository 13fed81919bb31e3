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

// (dist, position) lexicographic merge of two partial (best, second) states.
struct Best2 { int best, pos, second; };
__device__ __forceinline__ Best2 merge(const Best2 x, const Best2 y) {
    Best2 r;
    const bool xwins = x.best < y.best || (x.best == y.best && x.pos <= y.pos);
    r.best = xwins ? x.best : y.best;
    r.pos = xwins ? x.pos : y.pos;
    const int loser = xwins ? y.best : x.best;
    const int s = x.second < y.second ? x.second : y.second;
    r.second = s < loser ? s : loser;
    return r;
}

// One wave per query: lanes stride over the candidate list; sequential semantics of the
// reference loop (first minimum wins, second = second smallest of the multiset) are kept by
// the (dist, list position) order.  cand_idx == nullptr: candidates are rows [0, nB).
__global__ __launch_bounds__(256) void hamming_argmin2_kernel(const uint4 *__restrict__ A, int nA,
                                                             const uint4 *__restrict__ B, int nB,
                                                             const int32_t *__restrict__ cand_off,
                                                             const int32_t *__restrict__ cand_idx,
                                                             int32_t *best_idx, int32_t *best_dist,
                                                             int32_t *second_dist) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nA) return;
    const uint4 a0 = A[(int64_t)q * 2], a1 = A[(int64_t)q * 2 + 1];
    const int c0 = cand_off ? cand_off[q] : 0, c1 = cand_off ? cand_off[q + 1] : nB;
    Best2 st{256, 0x7fffffff, 256};
    int my_j = -1;
    for (int c = c0 + lane; c < c1; c += 64) {
        const int j = cand_idx ? cand_idx[c] : c;
        const int d = ham256(a0, a1, B[(int64_t)j * 2], B[(int64_t)j * 2 + 1]);
        if (d < st.best) { st.second = st.best; st.best = d; st.pos = c; my_j = j; }
        else if (d < st.second) st.second = d;
    }
    // butterfly over the 64 lanes; carry the row index of the winner along with its position
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Best2 o;
        o.best = __shfl_xor(st.best, off);
        o.pos = __shfl_xor(st.pos, off);
        o.second = __shfl_xor(st.second, off);
        const int oj = __shfl_xor(my_j, off);
        const Best2 m = merge(st, o);
        if (m.pos != st.pos) my_j = oj;
        st = m;
    }
    if (lane == 0) {
        best_idx[q] = st.best < 256 || c1 > c0 ? my_j : -1;
        if (c1 <= c0) best_idx[q] = -1;
        best_dist[q] = st.best;
        second_dist[q] = st.second;
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
                                        int32_t *d_second_dist, void *stream) {
    return guarded([&] {
        if (n_a == 0) return SIVO_OK;
        if (!d_a || !d_b || !d_cand_off || !d_cand_idx || !d_best_idx || !d_best_dist || !d_second_dist || n_a < 0)
            throw std::invalid_argument("null argument");
        hipLaunchKernelGGL(hamming_argmin2_kernel, dim3((unsigned)cdiv(n_a, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const uint4 *)d_a, n_a, (const uint4 *)d_b, 0, d_cand_off, d_cand_idx, d_best_idx,
                           d_best_dist, d_second_dist);
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
                           d_second_dist);
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
                                    int32_t *second_dist) {
    return guarded([&] {
        if (n_a < 0 || n_b < 0) throw std::invalid_argument("negative size");
        if (n_a == 0) return SIVO_OK;
        if (!a || !cand_off || !best_idx || !best_dist || !second_dist) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        const int ncand = cand_off[n_a];
        DevBuf<uint8_t> da((size_t)n_a * 32), db((size_t)n_b * 32);
        DevBuf<int32_t> doff((size_t)n_a + 1), didx((size_t)ncand), dbi(n_a), dbd(n_a), dsd(n_a);
        SIVO_HIP(hipMemcpy(da.p, a, (size_t)n_a * 32, hipMemcpyHostToDevice));
        if (n_b) SIVO_HIP(hipMemcpy(db.p, b, (size_t)n_b * 32, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(doff.p, cand_off, ((size_t)n_a + 1) * 4, hipMemcpyHostToDevice));
        if (ncand) SIVO_HIP(hipMemcpy(didx.p, cand_idx, (size_t)ncand * 4, hipMemcpyHostToDevice));
        int rc = sivo_hamming_argmin2_dev(da.p, n_a, db.p, doff.p, didx.p, dbi.p, dbd.p, dsd.p, nullptr);
        if (rc) return rc;
        SIVO_HIP(hipMemcpy(best_idx, dbi.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(best_dist, dbd.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(second_dist, dsd.p, (size_t)n_a * 4, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}
