// ba_solve.hip — the optimisation loops behind SIVO::Optimizer once the graph is built.
//
// Stands behind g2o as Optimizer::PoseOptimization (reference src/orbslam/Optimizer.cc:273-491),
// Optimizer::LocalBundleAdjustment (:493-926) and Optimizer::BundleAdjustment (:49-271) drive it:
// OptimizationAlgorithmLevenberg over BlockSolver_6_3 (landmarks marginalised by a Schur complement),
// the chi2 re-classification schedules of the two callers, and computeMarginals for the 6x6 pose
// covariance (:482-487, :900-907).  SURVEY.md 8f-3.  g2o is an un-vendored submodule; its published
// algorithm is restated in oracle/ba_solve_oracle.c, which these kernels are checked against.
//
// Layout in HBM (fp64 throughout):
//   poses 12/keyframe (Rcw row-major, tcw), points 3/map point, SivoEdge 48 B/edge;
//   per edge:  err 3, Jp 18, Jx 9, W = w*Omega*Jp'Jx 18, Y = W*Hll^-1 18;
//   per point: Hll 9, bl 3, Hll^-1 9, dx 3;   per free pose: Hpp 36, bp 6;
//   reduced system S (6F x 6F, dense) + rhs; CSR edge lists by point and by free pose; a dense
//   (free pose x point) -> edge table that turns the Schur products into gathers with a fixed
//   summation order (no atomics: results are reproducible run to run).
//
// Pose-only problems (one vertex, <= a few thousand edges) run the WHOLE schedule — 4 rounds x 10
// LM iterations x up to 10 trials, the chi2 re-classification and the covariance — in ONE launch of a
// single persistent workgroup: the loop is latency-bound, so there is nothing to gain from more CUs
// and everything to gain from not crossing the host 100+ times per frame.
// The BA kernels are HBM-bound gathers/reductions; the LM accept/reject logic reads 3 doubles per trial.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <vector>

#include "common.hpp"

#pragma clang fp contract(off)

namespace sivo {

struct Intr { double fx, fy, cx, cy, bf; };

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
template <bool JAC>
__device__ __forceinline__ void edge_eval(const double *P, const double *X, const SivoEdge &ed, const Intr &K, double *er,
                                          double *jp, double *jx, bool &depth_ok) {
    double R[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) R[i] = P[i];
    const double X0 = X[0], X1 = X[1], X2 = X[2];
    const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + R[9];
    const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + R[10];
    const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + R[11];
    const double invz = 1.0 / z, z_2 = z * z;
    const bool st = ed.stereo != 0;
    er[0] = ed.obs[0] - (x * invz * K.fx + K.cx);
    er[1] = ed.obs[1] - (y * invz * K.fy + K.cy);
    er[2] = st ? ed.obs[2] - (x * invz * K.fx + K.cx - K.bf * invz) : 0.0;
    depth_ok = z > 0.0;
    if (JAC) {
        if (jx) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double j0 = -K.fx * R[j] / z + K.fx * x * R[6 + j] / z_2;
                jx[j] = j0;
                jx[3 + j] = -K.fy * R[3 + j] / z + K.fy * y * R[6 + j] / z_2;
                jx[6 + j] = st ? j0 - K.bf * R[6 + j] / z_2 : 0.0;
            }
        }
        jp[0] = x * y / z_2 * K.fx;  jp[1] = -(1 + (x * x / z_2)) * K.fx;  jp[2] = y / z * K.fx;
        jp[3] = -1. / z * K.fx;      jp[4] = 0;                            jp[5] = x / z_2 * K.fx;
        jp[6] = (1 + y * y / z_2) * K.fy;  jp[7] = -x * y / z_2 * K.fy;    jp[8] = -x / z * K.fy;
        jp[9] = 0;                   jp[10] = -1. / z * K.fy;              jp[11] = y / z_2 * K.fy;
        if (st) {
            jp[12] = jp[0] - K.bf * y / z_2;  jp[13] = jp[1] + K.bf * x / z_2;  jp[14] = jp[2];
            jp[15] = jp[3];                   jp[16] = 0;                       jp[17] = jp[5] - K.bf / z_2;
        } else {
#pragma unroll
            for (int j = 12; j < 18; ++j) jp[j] = 0.0;
        }
    }
}

__device__ __forceinline__ void huber(double c2, double delta, double &rho, double &w) {
    const double dsqr = delta * delta;
    if (c2 <= dsqr) { rho = c2; w = 1.0; }
    else { const double s = sqrt(c2); rho = 2 * s * delta - dsqr; w = delta / s; }
}

// T <- exp([omega, upsilon]) * T   (g2o SE3Quat::exp, VertexSE3Expmap::oplusImpl)
__host__ __device__ inline void se3_oplus(const double *Tin, const double *u, double *Tout) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double tn[3], Rn[9];
    for (int i = 0; i < 3; ++i) tn[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rn[3 * i + j] = R[3 * i] * Tin[j] + R[3 * i + 1] * Tin[3 + j] + R[3 * i + 2] * Tin[6 + j];
        tn[i] += R[3 * i] * Tin[9] + R[3 * i + 1] * Tin[10] + R[3 * i + 2] * Tin[11];
    }
    for (int i = 0; i < 9; ++i) Tout[i] = Rn[i];
    for (int i = 0; i < 3; ++i) Tout[9 + i] = tn[i];
}

// sqrt(x) and 1 / sqrt(x) for the pivots of the dense Cholesky (x > 0, far from the ends of the exponent range): v_rsq_f64 and
// Newton steps — ~15 dependent operations instead of the ~70 of an IEEE sqrt followed by an IEEE division, which is what the
// 6 x 6 diagonal block's factorisation (a serial chain every step of the blocked algorithm waits for) consists of.  Both results
// are within an ulp of the correctly rounded ones.
__device__ __forceinline__ void sqrt_and_rsqrt(double x, double &s, double &r) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    double g = x * y;
    g = fma(fma(-g, g, x), 0.5 * y, g);          // g += (x - g^2) y / 2
    r = fma(fma(-g, y, 1.0), y, y);              // y += y (1 - g y)
    s = g;
}

// in-place lower Cholesky of a 6x6 (row-major); false when not positive definite
__host__ __device__ inline bool chol6(double *A) {
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    return true;
}
__host__ __device__ inline void chol6_solve(const double *L, double *x) {
    for (int i = 0; i < 6; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k]; x[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
}
__host__ __device__ inline bool inv6_spd(const double *H, double *out) {
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = H[i];
    if (!chol6(L)) return false;
    for (int c = 0; c < 6; ++c) {
        double e[6] = {0, 0, 0, 0, 0, 0};
        e[c] = 1;
        chol6_solve(L, e);
        for (int r = 0; r < 6; ++r) out[6 * r + c] = e[r];
    }
    return true;
}

// x from lane (lane ^ M) for M = 1, 2, 8 on the VALU's data-parallel crossbar (no LDS traffic); 4, 16, 32 through ds_bpermute
template <int M>
__device__ __forceinline__ double lane_xor(double x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "");
    if constexpr (M == 1 || M == 2 || M == 8) {
        constexpr int ctrl = M == 1 ? 0xB1 /* quad_perm [1,0,3,2] */ : M == 2 ? 0x4E /* quad_perm [2,3,0,1] */ : 0x128 /* row_ror:8 */;
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl_xor(x, M, 64);
    }
}

// One step of the halving butterfly: a lane whose bit M is clear keeps v[0, N) and hands v[N, 2N) to its partner, the other way
// round for a set bit; N values are left.
template <int M, int N>
__device__ __forceinline__ void halve_step(double *v, bool bit) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double keep = bit ? v[N + j] : v[j], send = bit ? v[j] : v[N + j];
        v[j] = keep + lane_xor<M>(send);
    }
}

// The same sum as block_sum for K > ~8 values: the wave step is a butterfly that HALVES the values a lane carries at every
// exchange (reduce-scatter: P/2 + P/4 + ... + 1 exchanges for P = K rounded up to a power of two, instead of 6 K), after which
// lane l holds the wave's total of ONE slot; the rest of the butterfly adds that slot over the lanes that share it.  Waves in
// index order after that.  A fixed order: the result does not depend on timing.  v has P entries, entries >= K are zero.
template <int P>
__device__ __forceinline__ int halving_slot(int lane) {       // the slot whose total lane `lane` ends up with
    int s = 0;
#pragma unroll
    for (int b = 0; (1 << b) < P; ++b) s |= ((lane >> b) & 1) << ((P == 64 ? 5 : P == 32 ? 4 : 3) - b);
    return s;
}
template <int K, int P, int NT>
__device__ __forceinline__ void block_sum_h(const double (&v)[K], double *s_red /* [NT/64][P] */, double *s_out /* [K] */) {
    static_assert((P == 16 || P == 32 || P == 64) && K <= P && 2 * K > P, "P = K rounded up to a power of two");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double w[P / 2];
    {   // first step on the K values themselves (slots >= K are zero and are not kept in registers)
        const bool bit = lane & 1;
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            const double hi = P / 2 + j < K ? v[P / 2 + j < K ? P / 2 + j : 0] : 0.0;
            const double keep = bit ? hi : v[j], send = bit ? v[j] : hi;
            w[j] = keep + lane_xor<1>(send);
        }
    }
    halve_step<2, P / 4>(w, lane & 2);
    halve_step<4, P / 8>(w, lane & 4);
    halve_step<8, P / 16>(w, lane & 8);
    if constexpr (P >= 32) halve_step<16, P / 32>(w, lane & 16);
    if constexpr (P >= 64) halve_step<32, P / 64>(w, lane & 32);
    double t = w[0];
    if constexpr (P < 32) t += lane_xor<16>(t);
    if constexpr (P < 64) t += lane_xor<32>(t);
    if (lane < P) s_red[wave * P + halving_slot<P>(lane)] = t;
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0;
        for (int w_ = 0; w_ < NT / 64; ++w_) s += s_red[w_ * P + threadIdx.x];
        s_out[threadIdx.x] = s;
    }
    __syncthreads();
}

// Sum K per-thread values over the workgroup in a fixed order: wave butterfly, then waves in index order.
template <int K, int NT>
__device__ __forceinline__ void block_sum(double (&v)[K], double *s_red /* [NT/64][K] */, double *s_out /* [K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[k] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) s_red[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0;
        for (int w = 0; w < NT / 64; ++w) s += s_red[w * K + threadIdx.x];
        s_out[threadIdx.x] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// PoseOptimization: one persistent workgroup, no serial section
//
// 4 rounds x 10 Levenberg-Marquardt iterations x (system build + >= 1 trial) are ~100 dependent steps over <= ~2000 edges: a
// latency chain, not a bandwidth problem.  The kernel is built around that:
//   * the edges (observation, information, the map point's position, flags) are read ONCE into LDS, 57 B per edge (PO_CAP =
//     2560 edges = 143 KB of the CU's 160 KB; edges beyond that are read from memory in every pass);
//   * every thread carries the whole solver state (pose, H, b, lambda ...) in registers and takes every decision itself from
//     the same reduced sums — 6x6 Cholesky, SE3 exponential, gain ratio are computed redundantly by all 512 threads instead of
//     by thread 0 between two barriers: ONE barrier per reduction is all the synchronisation there is;
//   * the 28 sums of a system build (21 of H, 6 of b, chi2) are reduced by a butterfly that HALVES the values a lane carries
//     at every step (reduce-scatter: 16 + 8 + 4 + 2 + 1 + 1 exchanges instead of 28 x 6), waves in index order after that:
//     a fixed summation order, the result does not depend on timing;
//   * the error vectors g2o keeps inside its edges are not stored: the chi2 test after a round needs them at the pose of the
//     LAST TRIAL (accepted or not — g2o does not recompute after a rejected step, Optimizer.cc:432-467 reads e->chi2()), which is
//     12 doubles; the errors are recomputed from it, bit for bit what the trial pass computed;
//   * divisions: the Jacobian multiplies by 1/z and 1/z^2 (one division per edge instead of 14), Cholesky by the reciprocal of
//     each pivot (12 instead of 33), (2 rho - 1)^3 is two multiplications — rounding-level differences (1e-16) against the
//     oracle's forms, far inside the 1e-9 the solver tests ask for.
// ------------------------------------------------------------------------------------------------
constexpr int PO_THREADS = 512, PO_WAVES = PO_THREADS / 64, PO_CAP = 2560;
constexpr int PO_LDS_BYTES = PO_CAP * (7 * 8 + 1) + 2 * PO_WAVES * 32 * 8;

struct PoseOptArgs {
    const double *in;        // [0, 12) pose0; then 8 doubles per edge: obs[3], inv_sigma2, X, Y, Z, stereo (0 / 1)
    int n;
    Intr K;
    double delta_mono, delta_stereo;
    uint8_t *outlier;        // n   (Frame::mvbOutlier)
    double *out;             // [0, 12) pose  [12, 48) covariance  [48, 52) cov_ok, nBad, iterations, trials  [52, 52 + n) chi2 (want_chi2)
    int want_chi2;
};

struct PoseReducer {
    double *s_red;           // [2][PO_WAVES][32]
    int buf = 0;
    // v[0, 28) summed over the workgroup -> out[0, 28) in every thread.  Slot k ends in the lanes with bits (b0 .. b4) = the
    // binary digits of k, most significant first.
    __device__ __forceinline__ void sum28(double (&v)[32], double (&out)[28]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        halve_step<1, 16>(v, lane & 1);
        halve_step<2, 8>(v, lane & 2);
        halve_step<4, 4>(v, lane & 4);
        halve_step<8, 2>(v, lane & 8);
        halve_step<16, 1>(v, lane & 16);
        const double t = v[0] + lane_xor<32>(v[0]);
        const int slot = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
        double *r = s_red + buf * (PO_WAVES * 32);
        if (lane < 32) r[wave * 32 + slot] = t;
        __syncthreads();
        double s = 0;                                        // lane k (and k + 32): slot k over the waves, in wave order
#pragma unroll
        for (int w = 0; w < PO_WAVES; ++w) s += r[w * 32 + (lane & 31)];
        buf ^= 1;
#pragma unroll
        for (int k = 0; k < 28; ++k)
            out[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), k), __builtin_amdgcn_readlane(__double2loint(s), k));
    }
    __device__ __forceinline__ double sum1(double x) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        x += lane_xor<1>(x); x += lane_xor<2>(x); x += lane_xor<4>(x); x += lane_xor<8>(x); x += lane_xor<16>(x); x += lane_xor<32>(x);
        double *r = s_red + buf * (PO_WAVES * 32);
        if (lane == 0) r[wave] = x;
        __syncthreads();
        double s = 0;
#pragma unroll
        for (int w = 0; w < PO_WAVES; ++w) s += r[w];
        buf ^= 1;
        return s;
    }
};

// lower Cholesky factor of the 6x6 matrix A (row-major, lower triangle read), the reciprocals of its diagonal in rd
__device__ __forceinline__ bool chol6_recip(double (&A)[36], double (&rd)[6]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        ok = ok && (d > 0.0);
        d = sqrt(d);
        A[j * 6 + j] = d;
        rd[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * rd[j];
        }
    }
    return ok;
}
__device__ __forceinline__ void chol6_recip_solve(const double (&L)[36], const double (&rd)[6], double (&x)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * x[k];
        x[i] = s * rd[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s * rd[i];
    }
}

struct PoEdge { double ox, oy, orr, isig, X, Y, Z; bool st; };

__global__ __launch_bounds__(PO_THREADS) void pose_optimize_kernel(PoseOptArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char po_lds[];
    double *const sE = reinterpret_cast<double *>(po_lds);                 // [7][PO_CAP]
    double *const s_red = sE + 7 * PO_CAP;                                 // [2][PO_WAVES][32]
    uint8_t *const sF = reinterpret_cast<uint8_t *>(s_red + 2 * PO_WAVES * 32);   // [PO_CAP]: bit 0 stereo, bit 1 outlier
    const int tid = threadIdx.x, n = a.n;
    const Intr K = a.K;
    const double *const edges = a.in + 12;
    for (int e = tid; e < n && e < PO_CAP; e += PO_THREADS) {
        const double *q = edges + 8 * (int64_t)e;
#pragma unroll
        for (int k = 0; k < 7; ++k) sE[k * PO_CAP + e] = q[k];
        sF[e] = q[7] != 0.0 ? 1 : 0;
    }
    for (int e = PO_CAP + tid; e < n; e += PO_THREADS) a.outlier[e] = 0;
    // (edge e is always handled by thread e % PO_THREADS: the LDS copy and the flags need no barrier)
    auto edge = [&](int e) {
        PoEdge g;
        if (e < PO_CAP) {
            g.ox = sE[e]; g.oy = sE[PO_CAP + e]; g.orr = sE[2 * PO_CAP + e]; g.isig = sE[3 * PO_CAP + e];
            g.X = sE[4 * PO_CAP + e]; g.Y = sE[5 * PO_CAP + e]; g.Z = sE[6 * PO_CAP + e]; g.st = sF[e] & 1;
        } else {
            const double *q = edges + 8 * (int64_t)e;
            g.ox = q[0]; g.oy = q[1]; g.orr = q[2]; g.isig = q[3]; g.X = q[4]; g.Y = q[5]; g.Z = q[6]; g.st = q[7] != 0.0;
        }
        return g;
    };
    // flag byte of an edge (LDS, or a.outlier[e] beyond PO_CAP until the kernel's last loop): bit 0 stereo (LDS only), bit 1 outlier,
    // bit 2 outlier before the latest chi2 test
    auto flags = [&](int e) -> int { return e < PO_CAP ? sF[e] : a.outlier[e]; };
    auto is_outlier = [&](int e) -> bool { return (flags(e) & 2) != 0; };
    auto set_outlier = [&](int e, bool o) {
        const int f = flags(e), nf = (f & 1) | (o ? 2 : 0) | ((f & 2) ? 4 : 0);
        if (e < PO_CAP) sF[e] = (uint8_t)nf; else a.outlier[e] = (uint8_t)nf;
    };
    // the error vector of edge g at pose T (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose::computeError), camera-frame point out
    auto residual = [&](const double (&T)[12], const PoEdge &g, double (&er)[3], double &x, double &y, double &invz) {
        x = T[0] * g.X + T[1] * g.Y + T[2] * g.Z + T[9];
        y = T[3] * g.X + T[4] * g.Y + T[5] * g.Z + T[10];
        const double z = T[6] * g.X + T[7] * g.Y + T[8] * g.Z + T[11];
        invz = 1.0 / z;
        er[0] = g.ox - (x * invz * K.fx + K.cx);
        er[1] = g.oy - (y * invz * K.fy + K.cy);
        er[2] = g.st ? g.orr - (x * invz * K.fx + K.cx - K.bf * invz) : 0.0;
    };
    PoseReducer red{s_red};

    double P[12], Peval[12], H[21], b[6];
    double lambda = 0, ni = 2, current = 0;
    int iters = 0, trials = 0, nbad_total = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) { Peval[i] = a.in[i]; P[i] = a.in[i]; }
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = 0.0;
    for (int round = 0; round < 4; ++round) {
#pragma unroll
        for (int i = 0; i < 12; ++i) P[i] = a.in[i];          // vSE3->setEstimate(pFrame->mTcw) at every round (:419)
        for (int it = 0; it < 10; ++it) {
            // computeActiveErrors + buildSystem
            double acc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = 0.0;
            for (int e = tid; e < n; e += PO_THREADS) {
                const PoEdge g = edge(e);
                if (g.st && is_outlier(e)) continue;             // level 1
                double er[3], x, y, iz;
                residual(P, g, er, x, y, iz);
                const double iz2 = iz * iz;
                double jp[18];
                jp[0] = x * y * iz2 * K.fx;   jp[1] = -(1 + (x * x * iz2)) * K.fx;  jp[2] = y * iz * K.fx;
                jp[3] = -iz * K.fx;           jp[4] = 0;                            jp[5] = x * iz2 * K.fx;
                jp[6] = (1 + y * y * iz2) * K.fy;  jp[7] = -x * y * iz2 * K.fy;     jp[8] = -x * iz * K.fy;
                jp[9] = 0;                    jp[10] = -iz * K.fy;                  jp[11] = y * iz2 * K.fy;
                if (g.st) {
                    jp[12] = jp[0] - K.bf * y * iz2;  jp[13] = jp[1] + K.bf * x * iz2;  jp[14] = jp[2];
                    jp[15] = jp[3];                   jp[16] = 0;                       jp[17] = jp[5] - K.bf * iz2;
                } else {
#pragma unroll
                    for (int j = 12; j < 18; ++j) jp[j] = 0.0;
                }
                const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g.isig;
                double r = c2, w = 1.0;
                if (!g.st || round < 3) huber(c2, g.st ? a.delta_stereo : a.delta_mono, r, w);   // kernel dropped after it==2 (:462)
                const double wo = w * g.isig;
                int k = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = i; j < 6; ++j) acc[k++] += wo * (jp[i] * jp[j] + jp[6 + i] * jp[6 + j] + jp[12 + i] * jp[12 + j]);
#pragma unroll
                for (int i = 0; i < 6; ++i) acc[21 + i] -= wo * (jp[i] * er[0] + jp[6 + i] * er[1] + jp[12 + i] * er[2]);
                acc[27] += r;
            }
            double S[28];
            red.sum28(acc, S);
#pragma unroll
            for (int i = 0; i < 21; ++i) H[i] = S[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = S[21 + i];
            current = S[27];
            if (it == 0) {
                const double md = fmax(fmax(fmax(fabs(H[0]), fabs(H[6])), fmax(fabs(H[11]), fabs(H[15]))), fmax(fabs(H[18]), fabs(H[20])));
                lambda = 1e-5 * md; ni = 2;
            }
            int qmax = 0;
            bool cont, term;
            do {
                double A[36], rd[6], x6[6], Bk[12];
                {
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = i; j < 6; ++j) { A[6 * j + i] = H[k]; A[6 * i + j] = H[k]; ++k; }
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) { A[7 * i] += lambda; x6[i] = b[i]; }
#pragma unroll
                for (int i = 0; i < 12; ++i) Bk[i] = P[i];
                const bool ok = chol6_recip(A, rd);
                double scale = 0;
                if (ok) {
                    chol6_recip_solve(A, rd, x6);
                    se3_oplus(Bk, x6, P);
#pragma unroll
                    for (int i = 0; i < 6; ++i) scale += x6[i] * (lambda * x6[i] + b[i]);
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) Peval[i] = P[i];
                double chi = 0.0;
                for (int e = tid; e < n; e += PO_THREADS) {
                    const PoEdge g = edge(e);
                    if (g.st && is_outlier(e)) continue;
                    double er[3], x, y, iz;
                    residual(P, g, er, x, y, iz);
                    const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g.isig;
                    double r = c2, w;
                    if (!g.st || round < 3) huber(c2, g.st ? a.delta_stereo : a.delta_mono, r, w);
                    chi += r;
                }
                const double chi_sum = red.sum1(chi);
                const double temp = ok ? chi_sum : DBL_MAX;
                const double rho = (current - temp) / (scale + 1e-3);
                if (rho > 0 && isfinite(temp)) {
                    const double t = 2 * rho - 1;
                    double alpha = 1. - t * t * t;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2; current = temp;
                } else {
                    lambda *= ni; ni *= 2;
#pragma unroll
                    for (int i = 0; i < 12; ++i) P[i] = Bk[i];
                }
                ++qmax; ++trials;
                cont = (rho < 0 && qmax < 10);
                term = (qmax == 10 || rho == 0);
            } while (cont);
            ++iters;
            if (term) break;
        }
        // chi2 test on the stereo edges (:432-467); mono edges are not re-classified by the reference.  An inlier's error vector
        // is the one of the last trial (Peval), an outlier's is computed at the estimate (computeError, :441-444)
        double bad = 0.0;
        for (int e = tid; e < n; e += PO_THREADS) {
            const PoEdge g = edge(e);
            if (!g.st) continue;
            double er[3], x, y, iz;
            if (is_outlier(e)) residual(P, g, er, x, y, iz);
            else residual(Peval, g, er, x, y, iz);
            const float c2 = (float)((er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g.isig);
            const bool out = c2 > 7.815f;
            set_outlier(e, out);
            bad += out ? 1.0 : 0.0;
        }
        nbad_total = (int)red.sum1(bad);
        if (n < 10) break;                                     // optimizer.edges().size() < 10 (:469)
    }
    for (int e = tid; e < n; e += PO_THREADS) {
        const int f = flags(e);
        a.outlier[e] = (f & 2) ? 1 : 0;
        if (a.want_chi2) {
            // the chi2 g2o's edge holds when the reference returns: from the error vector last computed for it — the final round's test
            // above recomputed an outlier's at the estimate, left an inlier's (and every mono edge's) at the last trial
            const PoEdge g = edge(e);
            double er[3], x, y, iz;
            if (g.st && (f & 4)) residual(P, g, er, x, y, iz);
            else residual(Peval, g, er, x, y, iz);
            a.out[52 + e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g.isig;
        }
    }
    if (tid < 12) a.out[tid] = P[tid];
    if (tid == 0) {
        double Hf[36], C[36];
        int k = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { Hf[6 * i + j] = H[k]; Hf[6 * j + i] = H[k]; ++k; }
        const bool ok = inv6_spd(Hf, C);
        for (int i = 0; i < 36; ++i) a.out[12 + i] = ok ? C[i] : 0.0;
        a.out[48] = ok; a.out[49] = nbad_total; a.out[50] = iters; a.out[51] = trials;
    }
}

// ------------------------------------------------------------------------------------------------
// Bundle adjustment: kernels of one LM iteration
// ------------------------------------------------------------------------------------------------
constexpr int BA_T = 256;

struct BaDev {
    // problem
    const SivoEdge *edges; int64_t nE;
    const int32_t *slot;        // pose -> free slot or -1
    const uint8_t *level, *robust;
    int nP, nF, nX;
    Intr K; double delta_mono, delta_stereo;
    // CSR
    const int64_t *pt_off; const int32_t *pt_edges;     // by point
    const int64_t *ps_off; const int32_t *ps_edges;     // by free pose
    const int32_t *table;                               // [nF][nX] edge id or -1
    // per edge, one set per estimate buffer (set k belongs to poses[k] / points[k]); kernels pick theirs with ba_set()
    struct BaEdgeSet { double *err, *Jp, *Jx, *wo, *rchi, *W; } sets[2];
    // per point / pose
    double *Hll, *bl, *xl, *sc_pt;
    double *Hpp, *bp;
    // reduced system
    double *S, *xs;
    double *scal;               // [0] chi  [1] scale  [2] maxdiag  [3] ok(0/1)
    // Levenberg-Marquardt state and the two estimate buffers it alternates between (round 6: the trial decision is taken on the device)
    struct BaLm *lm;
    double *poses[2], *points[2];
    const int *abort;           // pinned host word: non-zero = the caller's stop flag went up (checked where the host loop checked *stop)
};

// The Levenberg-Marquardt loop of g2o::SparseOptimizer::optimize / OptimizationAlgorithmLevenberg::solve (the reference's
// Optimizer.cc:757-821 stands on it) as device state.  Until round 5 the host took the decision of every trial: a blocking 56-byte read
// per trial, 15 per LocalBundleAdjustment, each with its ~50 us of launch gap and a scheduling jitter that made sigma 2.5 ms on a 3 ms call.
// Now the host enqueues `steps` (linearise-if-due + one trial each) without looking, every kernel starts by reading this state, and the
// LAST workgroup of the trial's edge kernel takes the decision (the same arithmetic in the same order as the host loop it replaces):
//   step:  [need_lin: errors + Jacobians, H / b blocks, chi2 of the linearisation point; first: lambda = 1e-5 max |diag H|]
//          -> Y, Schur complement, dense solve -> trial estimates -> their errors, chi2 -> decide()
// A step after `done` is seven empty launches.  The host reads the state once per batch of steps.
struct BaLm {
    double lambda, ni, current, rho;
    int cur;                    // poses[cur] / points[cur] hold the current estimate, [cur ^ 1] the trial
    int need_lin;               // the next step linearises (first step of an iteration)
    int have_current;           // `current` holds the chi2 of this iteration's linearisation point or of an accepted trial
    int first;                  // first iteration of an optimize() call: computeLambdaInit
    int it, it_limit;           // iterations completed / allowed in this call
    int qmax;                   // trials of the running iteration
    int done;
    int trials;                 // trials of this call
    int last_set;               // the per-edge set written last (g2o's edges hold the errors of the last computeActiveErrors())
    int pad[2];
};
// the per-edge arrays of estimate buffer k (k is wave-uniform: six scalar selects; indexing the kernel argument with a run-time k made
// hipcc copy the whole 400-byte struct to scratch and every kernel twice as slow)
using BaEdgeSet = BaDev::BaEdgeSet;
__device__ __forceinline__ BaEdgeSet ba_set(const BaDev &d, int k) {
    BaEdgeSet s;
    s.err = k ? d.sets[1].err : d.sets[0].err; s.Jp = k ? d.sets[1].Jp : d.sets[0].Jp; s.Jx = k ? d.sets[1].Jx : d.sets[0].Jx;
    s.wo = k ? d.sets[1].wo : d.sets[0].wo; s.rchi = k ? d.sets[1].rchi : d.sets[0].rchi; s.W = k ? d.sets[1].W : d.sets[0].W;
    return s;
}
__device__ __forceinline__ double *ba_poses(const BaDev &d, int k) { return k ? d.poses[1] : d.poses[0]; }
__device__ __forceinline__ double *ba_points(const BaDev &d, int k) { return k ? d.points[1] : d.points[0]; }
__device__ __forceinline__ bool ba_lm_idle(const BaDev &d) { return d.lm->done != 0; }
__device__ __forceinline__ bool ba_lm_skip_lin(const BaDev &d) { return d.lm->done != 0 || d.lm->need_lin == 0; }
// one thread, behind the sums of a trial (r0 = chi2 of the trial, r4 = the points' share of computeScale)
__device__ __forceinline__ void ba_lm_decide(const BaDev &d, double r0, double r4, int landmarks) {
    BaLm &m = *d.lm;
    if (!m.have_current) { m.current = d.scal[6]; m.have_current = 1; }
    const bool ok = d.scal[3] != 0.0;
    const double temp = ok ? r0 : DBL_MAX;
    const double scale = ok ? d.scal[1] + (landmarks ? r4 : 0.0) : 0.0;
    const double rho = (m.current - temp) / (scale + 1e-3);
    if (rho > 0 && isfinite(temp)) {
        const double t = 2 * rho - 1;
        double alpha = 1. - t * t * t;                  // (as pose_optimize_kernel)
        alpha = fmin(alpha, 2. / 3.);
        m.lambda *= fmax(1. / 3., alpha);
        m.ni = 2; m.current = temp;
        m.cur ^= 1;                                     // (the trial kernel left errors, Jacobians and W of the new estimate in its set)
    } else {
        m.lambda *= m.ni; m.ni *= 2;
    }
    m.rho = rho;
    ++m.qmax; ++m.trials;
    const bool stop = d.abort && __atomic_load_n(d.abort, __ATOMIC_RELAXED) != 0;
    if (rho < 0 && m.qmax < 10 && !stop) { m.need_lin = 0; return; }          // another trial of the same iteration
    const bool brk = m.qmax == 10 || rho == 0;
    ++m.it;
    if (brk || stop || m.it >= m.it_limit) { m.done = 1; return; }
    m.need_lin = 1; m.qmax = 0; m.have_current = 0;
}

// errors (+ Jacobians, W) of the active edges at (poses, points)
// JAC = false (a trial estimate) with `partial`: the sums the host reads after a trial are formed here instead of by a launch
// of their own — every block leaves the sum of its edges' robustified chi2 in partial[block]; the block that finishes last
// (atomic counter) adds the partials in block order (scal[0]) and, with_points, the per-point shares of computeScale
// (sc_pt, written by the update kernel before this launch: scal[4]).  Fixed order of additions, whichever block is last.
// Errors, Jacobians and W of every active edge at an estimate, into the per-edge set of that estimate's buffer.
// LIN = true: the current estimate — launched ONCE per optimize() call (the flags may have changed since the last one).  LIN = false:
// a trial estimate — Jacobians and W are formed here as well (+1.5 us on 36 k edges), so that an accepted trial needs no second pass over
// the edges (until round 5 the next iteration began by evaluating every edge again at the very same estimate).  The set of the current
// estimate is therefore valid from the first step of a call on: trials write the other set, an accepted trial's set becomes the current one.
template <bool LIN>
__global__ __launch_bounds__(BA_T) void ba_edge_kernel(BaDev d, double *partial = nullptr, unsigned *counter = nullptr, int with_points = 0) {
    if (ba_lm_idle(d)) return;
    const int buf = __builtin_amdgcn_readfirstlane(LIN ? d.lm->cur : d.lm->cur ^ 1);
    const BaEdgeSet es = ba_set(d, buf);
    const double *poses = ba_poses(d, buf), *points = ba_points(d, buf);
    const int64_t e = (int64_t)blockIdx.x * BA_T + threadIdx.x;
    double v[1] = {0.0};
    if (e < d.nE) {
        if (d.level[e]) {
            es.rchi[e] = 0.0;
        } else {
            const SivoEdge ed = d.edges[e];
            double er[3], jp[18], jx[9];
            bool dok;
            edge_eval<true>(poses + 12 * (int64_t)ed.pose, points + 3 * (int64_t)ed.point, ed, d.K, er, jp, jx, dok);
            es.err[3 * e] = er[0]; es.err[3 * e + 1] = er[1]; es.err[3 * e + 2] = er[2];
            const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * ed.inv_sigma2;
            double r = c2, w = 1.0;
            if (d.robust[e]) huber(c2, ed.stereo ? d.delta_stereo : d.delta_mono, r, w);
            es.rchi[e] = r;
            v[0] = r;
            const double wo = w * ed.inv_sigma2;
            es.wo[e] = wo;
#pragma unroll
            for (int i = 0; i < 18; ++i) es.Jp[18 * e + i] = jp[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) es.Jx[9 * e + i] = jx[i];
            if (d.slot[ed.pose] >= 0) {
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        es.W[18 * e + 3 * a + b] = wo * (jp[a] * jx[b] + jp[6 + a] * jx[3 + b] + jp[12 + a] * jx[6 + b]);
            }
        }
    }
    if (LIN) {
        if (e == 0) d.lm->last_set = buf;
        return;
    }
    // The sums the decision needs are formed here instead of by a launch of their own — every block leaves the sum of its edges'
    // robustified chi2 in partial[block]; the block that finishes last (atomic counter) adds the partials in block order (scal[0]) and,
    // with_points, the per-point shares of computeScale (sc_pt, written by the update kernel before this launch: scal[4]).  Fixed order
    // of additions, whichever block is last.
    __shared__ double s_red[BA_T / 64], s_out[1];
    __shared__ int s_last;
    block_sum<1, BA_T>(v, s_red, s_out);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s_out[0];
        __threadfence();
        s_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double p[1] = {0.0};
    for (unsigned i = threadIdx.x; i < gridDim.x; i += BA_T) p[0] += __builtin_nontemporal_load(partial + i);
    block_sum<1, BA_T>(p, s_red, s_out);
    const double chi = s_out[0];
    double sc4 = 0.0;
    if (threadIdx.x == 0) { d.scal[0] = chi; *counter = 0u; }
    if (with_points) {
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        int i = threadIdx.x;
        for (; i + 3 * BA_T < d.nX; i += 4 * BA_T) { q0 += d.sc_pt[i]; q1 += d.sc_pt[i + BA_T]; q2 += d.sc_pt[i + 2 * BA_T]; q3 += d.sc_pt[i + 3 * BA_T]; }
        for (; i < d.nX; i += BA_T) q0 += d.sc_pt[i];
        double q[1] = {(q0 + q1) + (q2 + q3)};
        block_sum<1, BA_T>(q, s_red, s_out);
        sc4 = s_out[0];
        if (threadIdx.x == 0) d.scal[4] = sc4;
    }
    if (threadIdx.x == 0) { d.lm->last_set = buf; ba_lm_decide(d, chi, sc4, with_points); }
}

// out[0] = sum(in[0..n)) in a fixed order; one workgroup per sum (blockIdx.x = 1: the second sum, when given).  Four independent
// partial sums per thread: the loop is a chain of dependent additions behind L2 loads otherwise (9.4 us for 36 k values).
__global__ __launch_bounds__(1024) void ba_sum_kernel(const double *in, int64_t n, double *out, const double *in2 = nullptr, int64_t n2 = 0,
                                                      double *out2 = nullptr) {
    __shared__ double s_red[16], s_out[1];
    if (blockIdx.x == 1) { in = in2; n = n2; out = out2; }
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) { p0 += in[i]; p1 += in[i + 1024]; p2 += in[i + 2048]; p3 += in[i + 3072]; }
    for (; i < n; i += 1024) p0 += in[i];
    double v[1] = {(p0 + p1) + (p2 + p3)};
    block_sum<1, 1024>(v, s_red, s_out);
    if (threadIdx.x == 0) out[0] = s_out[0];
}

// Hll, bl of every point from its active edges (CSR order)
// (BA_PL = 4 lanes per point, each walking every fourth edge of the point's list, partial sums combined by a butterfly over
// the four lanes: the list is a chain of dependent loads — index, edge, Jacobian — and a point has ~12 edges)
constexpr int BA_PL = 4;
constexpr int BA_BUILD_T = 1024;     // ba_build_kernel: a free pose's ~2000 edges are two per thread, the chi2 sum 35 per thread
template <int K>
__device__ __forceinline__ void ba_lanes_sum(double (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] += __shfl_xor(v[k], 1, 64);
        v[k] += __shfl_xor(v[k], 2, 64);
    }
}
__device__ __forceinline__ void ba_point_body(const BaDev &d, const BaEdgeSet &es, int block) {
    const int t = block * BA_BUILD_T + threadIdx.x, sub = t & (BA_PL - 1);
    const int q = t / BA_PL < d.nX ? t / BA_PL : d.nX - 1;          // (surplus lanes repeat the last point and do not store: every lane takes part in the butterfly)
    const bool mine = t / BA_PL < d.nX && sub == 0;
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int64_t i = d.pt_off[q] + sub; i < d.pt_off[q + 1]; i += BA_PL) {
        const int e = d.pt_edges[i];
        if (d.level[e]) continue;
        const double wo = es.wo[e];
        const double *jx = es.Jx + 9 * (int64_t)e, *er = es.err + 3 * (int64_t)e;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = a; c < 3; ++c) H[k++] += wo * (jx[a] * jx[c] + jx[3 + a] * jx[3 + c] + jx[6 + a] * jx[6 + c]);
#pragma unroll
        for (int a = 0; a < 3; ++a) b[a] -= wo * (jx[a] * er[0] + jx[3 + a] * er[1] + jx[6 + a] * er[2]);
    }
    ba_lanes_sum<6>(H);
    ba_lanes_sum<3>(b);
    if (!mine) return;
    double *Hq = d.Hll + 9 * (int64_t)q;
    Hq[0] = H[0]; Hq[1] = H[1]; Hq[2] = H[2]; Hq[3] = H[1]; Hq[4] = H[3]; Hq[5] = H[4]; Hq[6] = H[2]; Hq[7] = H[4]; Hq[8] = H[5];
    d.bl[3 * q] = b[0]; d.bl[3 * q + 1] = b[1]; d.bl[3 * q + 2] = b[2];
}

// Hpp, bp of one free pose per workgroup
__device__ __forceinline__ void ba_pose_body(const BaDev &d, const BaEdgeSet &es, int s) {
    __shared__ double s_red[(BA_BUILD_T / 64) * 32], s_out[27];
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    for (int64_t i = d.ps_off[s] + threadIdx.x; i < d.ps_off[s + 1]; i += BA_BUILD_T) {
        const int e = d.ps_edges[i];
        if (d.level[e]) continue;
        const double wo = es.wo[e];
        const double *jp = es.Jp + 18 * (int64_t)e, *er = es.err + 3 * (int64_t)e;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c) acc[k++] += wo * (jp[a] * jp[c] + jp[6 + a] * jp[6 + c] + jp[12 + a] * jp[12 + c]);
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] -= wo * (jp[a] * er[0] + jp[6 + a] * er[1] + jp[12 + a] * er[2]);
    }
    block_sum_h<27, 32, BA_BUILD_T>(acc, s_red, s_out);
    if (threadIdx.x == 0) {
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) { d.Hpp[36 * s + 6 * a + c] = s_out[k]; d.Hpp[36 * s + 6 * c + a] = s_out[k]; ++k; }
        for (int a = 0; a < 6; ++a) d.bp[6 * s + a] = s_out[21 + a];
    }
}

// One launch behind the edge kernel for the three independent reductions of a linearisation: blocks [0, point_blocks) the
// points' Hll / bl, the next nF blocks the free poses' Hpp / bp, the last block chi2 of the linearisation point (scal[6]).
// (Three launches before: 9 + 21 + 25 us back to back on a 36 k-edge problem, each mostly latency.)
__global__ __launch_bounds__(BA_BUILD_T) void ba_build_kernel(BaDev d, int point_blocks) {
    if (ba_lm_skip_lin(d)) return;
    const BaEdgeSet es = ba_set(d, __builtin_amdgcn_readfirstlane(d.lm->cur));
    const int b = blockIdx.x;
    if (b < point_blocks) { ba_point_body(d, es, b); return; }
    if (b < point_blocks + d.nF) { ba_pose_body(d, es, b - point_blocks); return; }
    constexpr int T_ = BA_BUILD_T;
    __shared__ double s_red[T_ / 64], s_out[1];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 3 * T_ < d.nE; i += 4 * T_) { p0 += es.rchi[i]; p1 += es.rchi[i + T_]; p2 += es.rchi[i + 2 * T_]; p3 += es.rchi[i + 3 * T_]; }
    for (; i < d.nE; i += T_) p0 += es.rchi[i];
    double v[1] = {(p0 + p1) + (p2 + p3)};
    block_sum<1, T_>(v, s_red, s_out);
    if (threadIdx.x == 0) d.scal[6] = s_out[0];
}

// scal[2] = max |diag| over Hpp and Hll (computeLambdaInit)
__global__ __launch_bounds__(1024) void ba_maxdiag_kernel(BaDev d, int with_points) {
    if (ba_lm_idle(d) || !d.lm->first) return;
    __shared__ double s_m[1024];
    double m = 0;
    for (int i = threadIdx.x; i < 6 * d.nF; i += 1024) m = fmax(m, fabs(d.Hpp[36 * (i / 6) + 7 * (i % 6)]));
    if (with_points)
        for (int i = threadIdx.x; i < 3 * d.nX; i += 1024) m = fmax(m, fabs(d.Hll[9 * (int64_t)(i / 3) + 4 * (i % 3)]));
    s_m[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.scal[2] = s_m[0];
        d.lm->lambda = 1e-5 * s_m[0]; d.lm->ni = 2; d.lm->first = 0;          // computeLambdaInit
    }
}

// (Hll_q + lambda I)^-1 — formed where it is used (every edge of the point, and the point's update) instead of by a launch of
// its own: 9 loads and ~40 flops
__device__ __forceinline__ void ba_point_inverse(const BaDev &d, int q, double lambda, double (&Ai)[9]) {
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = d.Hll[9 * (int64_t)q + i];
    A[0] += lambda; A[4] += lambda; A[8] += lambda;
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c0 + A[1] * c1 + A[2] * c2, id = 1.0 / det;
    Ai[0] = c0 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c1 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c2 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// S block (i, j), i <= j:  [i == j] (Hpp_i + lambda I)  -  sum_q Y_{e(i,q)} W_{e(j,q)}' ; rhs_i = bp_i - sum_q Y_{e(i,q)} bl_q
// (1024 threads per block — 3 points per thread instead of 12 — was measured in round 3: the whole call went from 3.4 to 3.9 ms;
// the 42-value block reduction over 16 waves costs more than the shorter chains of dependent loads save)
constexpr int BA_SCHUR_T = 512;       // a pair of keyframes shares up to nX points and every thread walks a chain of dependent loads per point: 6 points per thread (42 f64 accumulators: 1024 threads would spill)
// (Round 6: Y_e = W_e (Hll_q + lambda I)^-1 is formed here, where it is used — 9 more loads that do not depend on the table chain, ~100 flops —
// instead of by a launch of its own over the edges that wrote 144 bytes per edge for this kernel to read back; the same expression, so the
// same doubles.)
__global__ __launch_bounds__(BA_SCHUR_T) void ba_schur_kernel(BaDev d) {
    if (ba_lm_idle(d)) return;
    const double *Wc = ba_set(d, __builtin_amdgcn_readfirstlane(d.lm->cur)).W;
    const double lambda = d.lm->lambda;
    __shared__ double s_red[(BA_SCHUR_T / 64) * 64], s_out[42];
    // block index -> (i, j) of the upper triangle
    int i = 0, rem = blockIdx.x;
    while (rem >= d.nF - i) { rem -= d.nF - i; ++i; }
    const int j = i + rem;
    const int32_t *ti = d.table + (int64_t)i * d.nX, *tj = d.table + (int64_t)j * d.nX;
    double acc[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) acc[k] = 0.0;
    // A point costs a chain of dependent loads (table entry -> level flag -> 18 + 18 doubles) and a thread has several: the table
    // entries and flags of SCH_U points are fetched together before anything depends on them (points in increasing order, as before).
    constexpr int SCH_U = 3;
    for (int q0 = threadIdx.x; q0 < d.nX; q0 += SCH_U * BA_SCHUR_T) {
        int e1[SCH_U], e2[SCH_U];
#pragma unroll
        for (int u = 0; u < SCH_U; ++u) {
            const int q = q0 + u * BA_SCHUR_T;
            e1[u] = q < d.nX ? ti[q] : -1;
            e2[u] = q < d.nX ? tj[q] : -1;
        }
        bool on1[SCH_U], on2[SCH_U];
#pragma unroll
        for (int u = 0; u < SCH_U; ++u) {
            const uint8_t l1 = d.level[e1[u] < 0 ? 0 : e1[u]], l2 = d.level[e2[u] < 0 ? 0 : e2[u]];
            on1[u] = e1[u] >= 0 && !l1;
            on2[u] = on1[u] && e2[u] >= 0 && !l2;
        }
#pragma unroll
        for (int u = 0; u < SCH_U; ++u) {
            if (!on1[u]) continue;
            const int q = q0 + u * BA_SCHUR_T;
            double Y[18];
            {
                const double *W1 = Wc + 18 * (int64_t)e1[u];
                double A9[9];
                ba_point_inverse(d, q, lambda, A9);
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) Y[3 * a + b] = W1[3 * a] * A9[b] + W1[3 * a + 1] * A9[3 + b] + W1[3 * a + 2] * A9[6 + b];
            }
            if (i == j) {
                const double *bl = d.bl + 3 * (int64_t)q;
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[36 + a] += Y[3 * a] * bl[0] + Y[3 * a + 1] * bl[1] + Y[3 * a + 2] * bl[2];
            }
            if (!on2[u]) continue;
            const double *W = Wc + 18 * (int64_t)e2[u];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[6 * a + b] += Y[3 * a] * W[3 * b] + Y[3 * a + 1] * W[3 * b + 1] + Y[3 * a + 2] * W[3 * b + 2];
        }
    }
    block_sum_h<42, 64, BA_SCHUR_T>(acc, s_red, s_out);
    const int n6 = 6 * d.nF;
    if (threadIdx.x < 36) {
        const int a = threadIdx.x / 6, b = threadIdx.x % 6;
        double v = -s_out[6 * a + b];
        if (i == j) { v += d.Hpp[36 * i + 6 * a + b]; if (a == b) v += lambda; }
        d.S[(int64_t)(6 * i + a) * n6 + 6 * j + b] = v;
        if (i != j) d.S[(int64_t)(6 * j + b) * n6 + 6 * i + a] = v;
    }
    if (i == j && threadIdx.x < 6) d.xs[6 * i + threadIdx.x] = d.bp[6 * i + threadIdx.x] - s_out[36 + threadIdx.x];
}

// dense Cholesky solve S x = xs (single workgroup); scal[3] = 1 on success.  The system is a grid of 6x6 keyframe blocks,
// and the factorisation is blocked accordingly (factor the diagonal block, solve the panel, update the trailing
// matrix: 3 barriers per keyframe instead of 3 per column), as are the two triangular solves.  This is the form for systems
// that do not fit in LDS (more than 21 free keyframes: a global bundle adjustment); it works in global memory.  A local BA
// (n <= 126) takes ba_dense_solve_lds_kernel below.
constexpr int BA_SOLVE_T = 512;        // (256 threads: 59.7 us, 512: 57.1, 1024: 89 — every wave repeats the diagonal block's Cholesky and the waves of a SIMD take turns at it)
__global__ __launch_bounds__(BA_SOLVE_T) void ba_dense_solve_kernel(BaDev d) {
    if (ba_lm_idle(d)) return;
    __shared__ int s_ok;
    __shared__ double s_x[6];
    const int nb = d.nF, n = 6 * nb, tid = threadIdx.x;
    double *S = d.S, *x = d.xs;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int jb = 0; jb < nb; ++jb) {
        const int j0 = 6 * jb;
        if (tid == 0) {                       // 6x6 Cholesky of the diagonal block, in place (lower)
            for (int j = 0; j < 6 && s_ok; ++j) {
                double dj = S[(int64_t)(j0 + j) * n + j0 + j];
                for (int k = 0; k < j; ++k) dj -= S[(int64_t)(j0 + j) * n + j0 + k] * S[(int64_t)(j0 + j) * n + j0 + k];
                if (!(dj > 0.0)) { s_ok = 0; break; }
                dj = sqrt(dj);
                S[(int64_t)(j0 + j) * n + j0 + j] = dj;
                for (int i = j + 1; i < 6; ++i) {
                    double v = S[(int64_t)(j0 + i) * n + j0 + j];
                    for (int k = 0; k < j; ++k) v -= S[(int64_t)(j0 + i) * n + j0 + k] * S[(int64_t)(j0 + j) * n + j0 + k];
                    S[(int64_t)(j0 + i) * n + j0 + j] = v / dj;
                }
            }
        }
        __syncthreads();
        if (!s_ok) break;
        // panel: rows below the block, L_ij = S_ij L_jj^-T (one row per thread, 6 unknowns)
        for (int i = j0 + 6 + tid; i < n; i += BA_SOLVE_T) {
            double r[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = S[(int64_t)i * n + j0 + c];
                for (int k = 0; k < c; ++k) v -= r[k] * S[(int64_t)(j0 + c) * n + j0 + k];
                r[c] = v / S[(int64_t)(j0 + c) * n + j0 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) S[(int64_t)i * n + j0 + c] = r[c];
        }
        __syncthreads();
        // trailing update of the lower triangle: rows i >= j0 + 6, columns j0 + 6 <= k <= i
        // (thread (tid >> 4, tid & 15) walks rows / columns with stride 16: no integer division in the loop; every element
        // is one 6-term sum in the same order as before)
        for (int i = j0 + 6 + (tid >> 4); i < n; i += BA_SOLVE_T / 16)
            for (int k = j0 + 6 + (tid & 15); k <= i; k += 16) {
                double v = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) v += S[(int64_t)i * n + j0 + c] * S[(int64_t)k * n + j0 + c];
                S[(int64_t)i * n + k] -= v;
            }
        __syncthreads();
    }
    if (s_ok) {
        for (int jb = 0; jb < nb; ++jb) {          // L y = b, one keyframe block per step
            const int j0 = 6 * jb;
            if (tid == 0) {
                for (int c = 0; c < 6; ++c) {
                    double v = x[j0 + c];
                    for (int k = 0; k < c; ++k) v -= S[(int64_t)(j0 + c) * n + j0 + k] * s_x[k];
                    s_x[c] = v / S[(int64_t)(j0 + c) * n + j0 + c];
                    x[j0 + c] = s_x[c];
                }
            }
            __syncthreads();
            for (int i = j0 + 6 + tid; i < n; i += BA_SOLVE_T) {
                double v = x[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= S[(int64_t)i * n + j0 + c] * s_x[c];
                x[i] = v;
            }
            __syncthreads();
        }
        for (int jb = nb - 1; jb >= 0; --jb) {     // L' x = y
            const int j0 = 6 * jb;
            if (tid == 0) {
                for (int c = 5; c >= 0; --c) {
                    double v = x[j0 + c];
                    for (int k = c + 1; k < 6; ++k) v -= S[(int64_t)(j0 + k) * n + j0 + c] * s_x[k];
                    s_x[c] = v / S[(int64_t)(j0 + c) * n + j0 + c];
                    x[j0 + c] = s_x[c];
                }
            }
            __syncthreads();
            for (int i = tid; i < j0; i += BA_SOLVE_T) {
                double v = x[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= S[(int64_t)(j0 + c) * n + i] * s_x[c];
                x[i] = v;
            }
            __syncthreads();
        }
    }
    if (tid == 0) d.scal[3] = s_ok ? 1.0 : 0.0;
}

// The reduced system of a local BA (n = 6 nF <= 126) solved entirely in LDS — round 3's second form (the first, the kernel above
// on an LDS copy of the matrix, took 151 of an LM iteration's 367 us: rocprofv3, profiles/r03_a_kernel_stats_ba.csv).  What that one spent its time
// on was hand-overs: thread 0 factorised the 6 x 6 diagonal block through ~100 dependent LDS round trips while 255 threads
// waited, 3 barriers per keyframe, then 2 x nF more steps of 2 barriers for the triangular solves.  Here
//   * EVERY thread factorises the diagonal block in registers (21 broadcast reads, ~100 FMAs, 6 sqrt, 6 divisions) and inverts
//     the factor: no hand-over and no barrier in front of the panel, whose rows become one 6-term product each (no divisions);
//   * the right-hand side is row n of the matrix: its "panel" and "trailing update" ARE the forward substitution;
//   * the backward substitution takes one barrier per keyframe (every thread forms x_j = L_jj^-T y_j itself from the stored
//     inverse).
// 3 barriers per keyframe in all instead of 7.  LDS: (n + 1)^2 + 21 nF doubles <= 133 KB.
__global__ __launch_bounds__(BA_SOLVE_T) void ba_dense_solve_lds_kernel(BaDev d) {
    // (Round 6, tried and removed: the trailing matrix in REGISTERS, one 6 x 6 block per thread for the whole factorisation — 72 LDS
    // reads and no write per block and step instead of 8 LDS operations per element.  72 us instead of 61: a thread then updates its
    // whole block in EVERY step, 20 x 36 elements in series, where this form spreads the shrinking trailing matrix evenly over the
    // 256 threads — 195 elements per thread in all.  profiles/r06_h_kernel_stats_ba.csv.)
    if (ba_lm_idle(d)) return;
    extern __shared__ double s_mat[];          // rows 0 .. n - 1: the matrix (lower triangle); row n: the right-hand side
    __shared__ double s_li[21 * 21];           // inverse of each diagonal block's Cholesky factor, packed lower triangle
    const int nb = d.nF, n = 6 * nb, tid = threadIdx.x;
    // row stride n + 1 doubles (odd): the trailing update reads element (k, j0 + c) of 16 consecutive rows k at once, and with a
    // stride of n = 6 nF doubles those fall on 4 bank pairs (48 k mod 64), a 4-way conflict on the kernel's dominant access
    const int ld = n + 1;
    for (int i = tid >> 4; i < n; i += BA_SOLVE_T / 16)
        for (int k = tid & 15; k <= i; k += 16) s_mat[i * ld + k] = d.S[(int64_t)i * n + k];
    for (int i = tid; i < n; i += BA_SOLVE_T) s_mat[n * ld + i] = d.xs[i];
    __syncthreads();
    bool ok = true;
    for (int jb = 0; jb < nb && ok; ++jb) {
        const int j0 = 6 * jb;
        double l[6][6], li[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = 0; k <= i; ++k) l[i][k] = s_mat[(j0 + i) * ld + j0 + k];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double dj = l[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) dj -= l[j][k] * l[j][k];
            if (!(dj > 0.0)) ok = false;                 // (the same value in every thread: uniform)
            double sd, inv;
            sqrt_and_rsqrt(dj, sd, inv);
            l[j][j] = sd;
            li[j][j] = inv;
#pragma unroll
            for (int i = j + 1; i < 6; ++i) {
                double v = l[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) v -= l[i][k] * l[j][k];
                l[i][j] = v * inv;
            }
        }
        if (!ok) break;
        // inverse of the lower-triangular factor: Li L = I
#pragma unroll
        for (int i = 1; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                double sum = 0;
#pragma unroll
                for (int k = j; k < i; ++k) sum += l[i][k] * li[k][j];
                li[i][j] = -sum * li[i][i];
            }
        if (tid == 0) {
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) s_li[21 * jb + k++] = li[i][j];
        }
        // panel: L_ij = S_ij L_jj^-T for the rows below the block, and for the right-hand side (row n): y_j = L_jj^-1 b_j
        for (int i = j0 + 6 + tid; i <= n; i += BA_SOLVE_T) {
            double *row = s_mat + i * ld + j0;
            double sv[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) sv[c] = row[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double r = 0;
#pragma unroll
                for (int k = 0; k <= c; ++k) r += sv[k] * li[c][k];
                row[c] = r;
            }
        }
        __syncthreads();
        // trailing update, rows j0 + 6 .. n (the right-hand side included), columns j0 + 6 .. min(row, n - 1)
        for (int i = j0 + 6 + (tid >> 4); i <= n; i += BA_SOLVE_T / 16) {
            double ri[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) ri[c] = s_mat[i * ld + j0 + c];
            const int kend = i < n ? i : n - 1;
            // (Round 6, tried and removed: four elements per pass with every operand read before anything is written — the panel columns (read) and
            // the trailing columns (written) never overlap, which hipcc cannot know.  65.8 us instead of 59.8: with one wave per SIMD the 24 + 4
            // reads in flight cost more registers and address arithmetic than the serialisation they remove; profiles/r06_k_kernel_stats_ba_unrolled_trailing.csv.)
            for (int k = j0 + 6 + (tid & 15); k <= kend; k += 16) {
                double v = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) v += ri[c] * s_mat[k * ld + j0 + c];
                s_mat[i * ld + k] -= v;
            }
        }
        __syncthreads();
    }
    if (ok) {
        double *y = s_mat + n * ld;
        for (int jb = nb - 1; jb >= 0; --jb) {         // L' x = y, one keyframe per step
            const int j0 = 6 * jb;
            double yj[6], x6[6], li[6][6];
#pragma unroll
            for (int c = 0; c < 6; ++c) yj[c] = y[j0 + c];
            {
                int k = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) li[i][j] = s_li[21 * jb + k++];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = 0;
#pragma unroll
                for (int k = c; k < 6; ++k) v += li[k][c] * yj[k];
                x6[c] = v;
            }
            for (int i = tid; i < j0; i += BA_SOLVE_T) {
                double v = y[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= s_mat[(j0 + c) * ld + i] * x6[c];
                y[i] = v;
            }
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) d.xs[j0 + c] = x6[c];
            }
            __syncthreads();
        }
    }
    if (tid == 0) d.scal[3] = ok ? 1.0 : 0.0;
}

// dx_l = Hll^-1 (bl - sum_e W_e' dx_p), points_trial = points + dx_l, per-point share of computeScale
__device__ __forceinline__ void ba_point_update_body(const BaDev &d, const double *Wc, int block, double lambda, const double *points, double *points_trial) {
    const int t = block * BA_T + threadIdx.x, sub = t & (BA_PL - 1);
    const int q = t / BA_PL < d.nX ? t / BA_PL : d.nX - 1;
    const bool mine = t / BA_PL < d.nX && sub == 0;
    const bool ok = d.scal[3] != 0.0;
    const double *bl = d.bl + 3 * (int64_t)q;
    double v[3] = {0, 0, 0}, x[3] = {0, 0, 0};
    if (ok) {
        for (int64_t i = d.pt_off[q] + sub; i < d.pt_off[q + 1]; i += BA_PL) {
            const int e = d.pt_edges[i];
            if (d.level[e]) continue;
            const int s = d.slot[d.edges[e].pose];
            if (s < 0) continue;
            const double *W = Wc + 18 * (int64_t)e, *xp = d.xs + 6 * s;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int a = 0; a < 6; ++a) v[b] -= W[3 * a + b] * xp[a];
        }
    }
    ba_lanes_sum<3>(v);
    if (!mine) return;
#pragma unroll
    for (int b = 0; b < 3; ++b) v[b] += bl[b];
    if (ok) {
        double Ai[9];
        ba_point_inverse(d, q, lambda, Ai);
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = Ai[3 * a] * v[0] + Ai[3 * a + 1] * v[1] + Ai[3 * a + 2] * v[2];
    }
    double sc = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        points_trial[3 * (int64_t)q + a] = points[3 * (int64_t)q + a] + x[a];
        sc += x[a] * (lambda * x[a] + bl[a]);
    }
    d.sc_pt[q] = sc;
}

// poses_trial = exp(dx) * poses for the free poses (copy for the fixed ones); scal[1] = pose share of computeScale
__device__ __forceinline__ void ba_pose_update_body(const BaDev &d, double lambda, const double *poses, double *poses_trial) {
    __shared__ double s_red[BA_T / 64], s_out[1];
    const bool ok = d.scal[3] != 0.0;
    double sc[1] = {0.0};
    for (int p = threadIdx.x; p < d.nP; p += BA_T) {
        const int s = d.slot[p];
        double T[12];
        if (s >= 0 && ok) {
            double u[6];
            for (int a = 0; a < 6; ++a) { u[a] = d.xs[6 * s + a]; sc[0] += u[a] * (lambda * u[a] + d.bp[6 * s + a]); }
            se3_oplus(poses + 12 * (int64_t)p, u, T);
        } else {
            for (int a = 0; a < 12; ++a) T[a] = poses[12 * (int64_t)p + a];
        }
        for (int a = 0; a < 12; ++a) poses_trial[12 * (int64_t)p + a] = T[a];
    }
    block_sum<1, BA_T>(sc, s_red, s_out);
    if (threadIdx.x == 0) d.scal[1] = s_out[0];
}

// the trial estimates in one launch: block 0 the poses, the others the points
__global__ __launch_bounds__(BA_T) void ba_update_kernel(BaDev d) {
    if (ba_lm_idle(d)) return;
    const int cur = __builtin_amdgcn_readfirstlane(d.lm->cur);
    const double lambda = d.lm->lambda;
    if (blockIdx.x == 0) ba_pose_update_body(d, lambda, ba_poses(d, cur), ba_poses(d, cur ^ 1));
    else ba_point_update_body(d, ba_set(d, cur).W, (int)blockIdx.x - 1, lambda, ba_points(d, cur), ba_points(d, cur ^ 1));
}

// pose-only variant of the reduced system (no landmarks in the state): S = blockdiag(Hpp + lambda I), xs = bp
__global__ void ba_pose_only_system_kernel(BaDev d) {
    if (ba_lm_idle(d)) return;
    const double lambda = d.lm->lambda;
    const int n6 = 6 * d.nF;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n6 * n6; idx += gridDim.x * blockDim.x) {
        const int r = idx / n6, c = idx % n6;
        double v = 0;
        if (r / 6 == c / 6) { v = d.Hpp[36 * (r / 6) + 6 * (r % 6) + c % 6]; if (r == c) v += lambda; }
        d.S[idx] = v;
        if (c == 0) d.xs[r] = d.bp[r];
    }
}

// final chi2 / depth test of LocalBundleAdjustment (:774-821, :826-858)
__global__ __launch_bounds__(BA_T) void ba_classify_kernel(BaDev d, const double *poses, const double *points, uint8_t *out) {
    const int64_t e = (int64_t)blockIdx.x * BA_T + threadIdx.x;
    if (e >= d.nE) return;
    const SivoEdge ed = d.edges[e];
    const double *er = (d.lm->last_set ? d.sets[1].err : d.sets[0].err) + 3 * e;          // the errors of the last evaluation, as g2o's edges hold them
    const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * ed.inv_sigma2;
    const double *R = poses + 12 * (int64_t)ed.pose, *X = points + 3 * (int64_t)ed.point;
    const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + R[11];
    out[e] = (c2 > (ed.stereo ? 7.815 : 5.991)) || !(z > 0.0);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Per-thread, grow-only device arena: a solver call makes ~30 allocations; hipMalloc / hipFree each cost tens of
// microseconds and hipFree synchronises the device, which is most of a 10 ms LocalBundleAdjustment.  Chunks are kept
// between calls (PoseOptimization runs every frame on the tracking thread, LocalBundleAdjustment on the mapping
// thread: one arena each) and released when the thread exits.
struct BaArena {
    struct Chunk { char *p; size_t cap, used; int device; };
    std::vector<Chunk> chunks;
    ~BaArena() { for (Chunk &c : chunks) (void)hipFree(c.p); }
    void reset() { for (Chunk &c : chunks) c.used = 0; }
    void *take(size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        int dev = 0;
        SIVO_HIP(hipGetDevice(&dev));
        for (Chunk &c : chunks)
            if (c.device == dev && c.cap - c.used >= bytes) { void *r = c.p + c.used; c.used += bytes; return r; }
        Chunk c{nullptr, std::max(bytes, (size_t)64 << 20), 0, dev};
        SIVO_HIP(hipMalloc((void **)&c.p, c.cap));
        c.used = bytes;
        chunks.push_back(c);
        return c.p;
    }
};
static thread_local BaArena t_arena;
// Per-thread, grow-only pinned staging buffer for a solver's one upload (a pinned allocation costs ~0.2 ms: never per call).  Its content
// is in flight until the call's first blocking read (the LM state after the first batch of steps), which every call makes before it returns.
struct BaStage {
    char *p = nullptr;
    size_t cap = 0;
    ~BaStage() { if (p) (void)hipHostFree(p); }
    char *reserve(size_t bytes) {
        if (bytes > cap) {
            if (p) SIVO_HIP(hipHostFree(p));
            p = nullptr; cap = 0;
            const size_t want = std::max(bytes * 2, (size_t)8 << 20);
            SIVO_HIP(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
            cap = want;
        }
        return p;
    }
};
static thread_local BaStage t_stage;

struct Buf {
    void *p = nullptr;
    void alloc(size_t bytes) { p = t_arena.take(bytes ? bytes : 8); }
    void upload(const void *src, size_t bytes) { alloc(bytes); if (bytes) SIVO_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice)); }
    void zero(size_t bytes) { alloc(bytes); SIVO_HIP(hipMemset(p, 0, bytes ? bytes : 8)); }
    template <class T> T *as() const { return (T *)p; }
};

// The LDS-resident Cholesky needs up to 127 KB of dynamic LDS: the opt-in is per device and per process, so it is made once
// per device under a lock and its result is kept; a device that refuses it uses the global-memory kernel.
static bool dense_solve_lds_ok() {
    static std::mutex mu;
    static int state[64] = {0};                 // per device: 0 unknown, 1 granted, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0)
        state[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(ba_dense_solve_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         127 * 127 * 8) == hipSuccess ? 1 : -1;
    return state[dev] == 1;
}

// What the stop flag of a solver call needs, once per calling thread (a pinned allocation costs 0.2 ms: not per call): a pinned word the
// device reads where the host loop read *stop, and an event to wait on while the host polls the caller's flag.
struct BaStopCtx {
    int *word = nullptr;
    hipEvent_t ev = nullptr;
    ~BaStopCtx() {
        if (word) (void)hipHostFree(word);
        if (ev) (void)hipEventDestroy(ev);
    }
    void ensure() {
        if (!word) { SIVO_HIP(hipHostMalloc((void **)&word, 64, hipHostMallocCoherent | hipHostMallocMapped)); *word = 0; }      // (fine-grained: the device must not cache it)
        if (!ev) SIVO_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
};
static thread_local BaStopCtx t_stop;

class BaSolver {
 public:
    BaSolver(const double *poses, const uint8_t *fixed, int nP, const double *points, int nX, bool points_fixed,
             const SivoEdge *edges, int64_t nE, const double intr[5], double dm, double ds)
        : nP_(nP), nX_(nX), nE_(nE), points_fixed_(points_fixed) {
        if (nP < 0 || nX < 0 || nE < 0) throw std::invalid_argument("negative size");
        if (nE > INT32_MAX) throw std::invalid_argument("too many edges");
        std::vector<int32_t> slot((size_t)std::max(nP, 1), -1);
        nF_ = 0;
        for (int i = 0; i < nP; ++i) slot[i] = fixed && fixed[i] ? -1 : nF_++;
        for (int64_t e = 0; e < nE; ++e)
            if (edges[e].pose < 0 || edges[e].pose >= nP || edges[e].point < 0 || edges[e].point >= nX)
                throw std::invalid_argument("edge refers to a pose/point outside the arrays");
        if ((int64_t)nF_ * nX > ((int64_t)1 << 28)) throw std::invalid_argument("problem too large for the dense (free pose x point) edge table");
        // CSR by point and by free pose; (free pose, point) -> edge table
        std::vector<int64_t> pt_off((size_t)nX + 1, 0), ps_off((size_t)nF_ + 1, 0);
        for (int64_t e = 0; e < nE; ++e) {
            pt_off[edges[e].point + 1]++;
            if (slot[edges[e].pose] >= 0) ps_off[slot[edges[e].pose] + 1]++;
        }
        for (int q = 0; q < nX; ++q) pt_off[q + 1] += pt_off[q];
        for (int s = 0; s < nF_; ++s) ps_off[s + 1] += ps_off[s];
        std::vector<int32_t> pt_edges((size_t)std::max<int64_t>(nE, 1)), ps_edges((size_t)std::max<int64_t>(ps_off[nF_], 1));
        std::vector<int32_t> table(points_fixed ? 1 : (size_t)std::max<int64_t>((int64_t)nF_ * nX, 1), -1);
        {
            std::vector<int64_t> f1(pt_off.begin(), pt_off.end() - 1), f2(ps_off.begin(), ps_off.end() - 1);
            for (int64_t e = 0; e < nE; ++e) {
                pt_edges[f1[edges[e].point]++] = (int32_t)e;
                const int s = slot[edges[e].pose];
                if (s >= 0) {
                    ps_edges[f2[s]++] = (int32_t)e;
                    if (!points_fixed) {
                        int32_t &t = table[(size_t)s * nX + edges[e].point];
                        if (t >= 0) throw std::invalid_argument("two edges join the same (keyframe, map point) pair");
                        t = (int32_t)e;
                    }
                }
            }
        }
        // ONE staged copy and ONE memset for the whole problem (round 6): the eleven uploads and thirteen memsets this constructor made until
        // round 5 were 24 blocking calls of ~15 us each — a tenth of the LocalBundleAdjustment call.  The host arrays are packed into the
        // calling thread's pinned staging buffer and go over in one asynchronous copy; everything that starts at zero is one region.
        {
            struct Seg { Buf *b; const void *src; size_t bytes; };
            const Seg up[] = {{&slot_, slot.data(), slot.size() * 4}, {&pt_off_, pt_off.data(), pt_off.size() * 8}, {&pt_edges_, pt_edges.data(), pt_edges.size() * 4},
                              {&ps_off_, ps_off.data(), ps_off.size() * 8}, {&ps_edges_, ps_edges.data(), ps_edges.size() * 4}, {&table_, table.data(), table.size() * 4},
                              {&edges_, edges, (size_t)nE * sizeof(SivoEdge)}, {&poses_[0], poses, (size_t)nP * 96}, {&poses_[1], poses, (size_t)nP * 96},
                              {&points_[0], points, (size_t)nX * 24}, {&points_[1], points, (size_t)nX * 24}};
            auto pad = [](size_t n) { return (std::max<size_t>(n, 8) + 255) / 256 * 256; };
            size_t total = 0;
            for (const Seg &g : up) total += pad(g.bytes);
            char *dev = (char *)t_arena.take(total), *host = t_stage.reserve(total);
            size_t off = 0;
            for (const Seg &g : up) {
                if (g.bytes) std::memcpy(host + off, g.src, g.bytes);
                g.b->p = dev + off;
                off += pad(g.bytes);
            }
            SIVO_HIP(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, 0));
            struct Z { Buf *b; size_t bytes; };
            const Z zs[] = {{&level_, (size_t)nE}, {&err_[0], (size_t)nE * 24}, {&err_[1], (size_t)nE * 24}, {&rchi_[0], (size_t)nE * 8}, {&rchi_[1], (size_t)nE * 8},
                            {&sc_pt_, (size_t)nX * 8}, {&Hpp_, (size_t)nF_ * 288}, {&bp_, (size_t)nF_ * 48}, {&xs_, (size_t)nF_ * 48}, {&scal_, 8 * 8},
                            {&partial_, (size_t)cdiv64(std::max<int64_t>(nE, 1), BA_T) * 8}, {&counter_, 8}, {&lm_, sizeof(BaLm)}};
            size_t ztotal = 0;
            for (const Z &z : zs) ztotal += pad(z.bytes);
            char *zdev = (char *)t_arena.take(ztotal);
            off = 0;
            for (const Z &z : zs) { z.b->p = zdev + off; off += pad(z.bytes); }
            SIVO_HIP(hipMemsetAsync(zdev, 0, ztotal, 0));
        }
        robust_.alloc((size_t)nE);
        SIVO_HIP(hipMemsetAsync(robust_.p, 1, (size_t)std::max<int64_t>(nE, 1), 0));
        for (int k = 0; k < 2; ++k) { Jp_[k].alloc((size_t)nE * 144); Jx_[k].alloc((size_t)nE * 72); wo_[k].alloc((size_t)nE * 8); W_[k].alloc((size_t)nE * 144); }
        Hll_.alloc((size_t)nX * 72); bl_.alloc((size_t)nX * 24); xl_.alloc((size_t)nX * 24);
        S_.alloc((size_t)36 * nF_ * nF_ * 8);
        hpp_last_.assign((size_t)std::max(nF_, 1) * 36, 0.0);
        d_.edges = edges_.as<SivoEdge>(); d_.nE = nE; d_.slot = slot_.as<int32_t>();
        d_.level = level_.as<uint8_t>(); d_.robust = robust_.as<uint8_t>();
        d_.nP = nP; d_.nF = nF_; d_.nX = nX;
        d_.K = Intr{intr[0], intr[1], intr[2], intr[3], intr[4]}; d_.delta_mono = dm; d_.delta_stereo = ds;
        d_.pt_off = pt_off_.as<int64_t>(); d_.pt_edges = pt_edges_.as<int32_t>();
        d_.ps_off = ps_off_.as<int64_t>(); d_.ps_edges = ps_edges_.as<int32_t>(); d_.table = table_.as<int32_t>();
        for (int k = 0; k < 2; ++k) {
            d_.sets[k].err = err_[k].as<double>(); d_.sets[k].Jp = Jp_[k].as<double>(); d_.sets[k].Jx = Jx_[k].as<double>();
            d_.sets[k].wo = wo_[k].as<double>(); d_.sets[k].rchi = rchi_[k].as<double>(); d_.sets[k].W = W_[k].as<double>();
        }
        d_.Hll = Hll_.as<double>(); d_.bl = bl_.as<double>(); d_.xl = xl_.as<double>();
        d_.sc_pt = sc_pt_.as<double>(); d_.Hpp = Hpp_.as<double>(); d_.bp = bp_.as<double>();
        d_.S = S_.as<double>(); d_.xs = xs_.as<double>(); d_.scal = scal_.as<double>();
        d_.lm = lm_.as<BaLm>();
        for (int k = 0; k < 2; ++k) { d_.poses[k] = poses_[k].as<double>(); d_.points[k] = points_[k].as<double>(); }
        d_.abort = nullptr;
        slot_host_ = slot;
    }

    void set_flags(const uint8_t *level, const uint8_t *robust) {
        if (level && nE_) SIVO_HIP(hipMemcpy(level_.p, level, (size_t)nE_, hipMemcpyHostToDevice));
        if (robust && nE_) SIVO_HIP(hipMemcpy(robust_.p, robust, (size_t)nE_, hipMemcpyHostToDevice));
    }
    void get_level(uint8_t *level) const { if (nE_) SIVO_HIP(hipMemcpy(level, level_.p, (size_t)nE_, hipMemcpyDeviceToHost)); }

    // g2o::SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg.  The loop itself runs on the device (BaLm): the
    // host enqueues one step per remaining iteration, reads the state once, and enqueues more only when a trial was rejected.
    int optimize(int iterations, const volatile uint8_t *stop, int *trials) {
        if (iterations <= 0 || (stop && *stop)) return 0;
        const unsigned gE = (unsigned)cdiv64(std::max<int64_t>(nE_, 1), BA_T), gX4 = (unsigned)cdiv(std::max(nX_, 1) * BA_PL, BA_T);
        const bool landmarks = !points_fixed_ && nX_ > 0;
        const unsigned gXb = (unsigned)cdiv(std::max(nX_, 1) * BA_PL, BA_BUILD_T);
        if (stop) {
            t_stop.ensure();
            abort_host_ = t_stop.word;
            *abort_host_ = 0;
            d_.abort = abort_host_;
        } else {
            d_.abort = nullptr;
        }
        BaLm lm{};
        lm.ni = 2; lm.cur = cur_; lm.need_lin = 1; lm.first = 1; lm.it_limit = iterations; lm.last_set = last_set_;
        SIVO_HIP(hipMemcpyAsync(d_.lm, &lm, sizeof lm, hipMemcpyHostToDevice, 0));          // (pageable source: the copy is staged before the call returns)
        if (!nF_) { const double one = 1.0; SIVO_HIP(hipMemcpyAsync(d_.scal + 3, &one, 8, hipMemcpyHostToDevice, 0)); }      // nothing to solve: "ok"
        const bool lds_solve = 6 * nF_ <= 126 && dense_solve_lds_ok();
        auto step = [&](bool first_of_call) {
            // linearise at the current estimate: errors, Jacobians, W (first step of a call only: afterwards the accepted trial left them);
            // Hll / bl of the points, Hpp / bp of the free poses, chi2 of the linearisation point (scal[6]) — skipped by the kernel itself
            // while trials of the same iteration go on
            if (first_of_call) hipLaunchKernelGGL(ba_edge_kernel<true>, dim3(gE), dim3(BA_T), 0, 0, d_, (double *)nullptr, (unsigned *)nullptr, 0);
            hipLaunchKernelGGL(ba_build_kernel, dim3((landmarks ? gXb : 0u) + (unsigned)nF_ + 1u), dim3(BA_BUILD_T), 0, 0, d_, landmarks ? (int)gXb : 0);
            if (first_of_call) hipLaunchKernelGGL(ba_maxdiag_kernel, dim3(1), dim3(1024), 0, 0, d_, landmarks ? 1 : 0);
            // one trial
            if (nF_) {
                if (landmarks) {
                    hipLaunchKernelGGL(ba_schur_kernel, dim3((unsigned)(nF_ * (nF_ + 1) / 2)), dim3(BA_SCHUR_T), 0, 0, d_);
                } else {
                    hipLaunchKernelGGL(ba_pose_only_system_kernel, dim3(64), dim3(256), 0, 0, d_);
                }
                if (lds_solve) hipLaunchKernelGGL(ba_dense_solve_lds_kernel, dim3(1), dim3(BA_SOLVE_T), (size_t)(6 * nF_ + 1) * (6 * nF_ + 1) * 8, 0, d_);
                else hipLaunchKernelGGL(ba_dense_solve_kernel, dim3(1), dim3(BA_SOLVE_T), 0, 0, d_);
            }
            // (without landmarks in the state the points never move: both point buffers hold the caller's values from the upload on)
            hipLaunchKernelGGL(ba_update_kernel, dim3(1u + (landmarks ? gX4 : 0u)), dim3(BA_T), 0, 0, d_);
            // the trial's errors and chi2; its last workgroup forms the sums and takes the decision (ba_lm_decide)
            hipLaunchKernelGGL(ba_edge_kernel<false>, dim3(gE), dim3(BA_T), 0, 0, d_, partial_.as<double>(), counter_.as<unsigned>(), landmarks ? 1 : 0);
        };
        bool first = true;
        int steps_total = 0;
        for (;;) {
            const int batch = std::max(1, iterations - lm.it);
            for (int b = 0; b < batch; ++b) { step(first); first = false; }
            steps_total += batch;
            SIVO_HIP(hipGetLastError());
            if (stop) {
                // the caller's flag is host memory the device cannot see: poll it while the batch runs and pass it on through the pinned word
                hipEvent_t ev = t_stop.ev;
                SIVO_HIP(hipEventRecord(ev, 0));
                while (hipEventQuery(ev) == hipErrorNotReady)
                    if (*stop) *abort_host_ = 1;
            }
            SIVO_HIP(hipMemcpy(&lm, d_.lm, sizeof lm, hipMemcpyDeviceToHost));
            if (lm.done || steps_total > 10 * iterations + 1) break;
        }
        cur_ = lm.cur; last_set_ = lm.last_set;
        if (trials) *trials += lm.trials;
        return lm.it;
    }

    // chi2 / depth classification at the current estimates using the error vectors g2o would hold
    void classify(uint8_t *outlier_host, bool to_level, bool drop_kernels) {
        if (!nE_) return;
        Buf out;
        if (to_level && !outlier_host) out.p = level_.p;          // (the kernel does not read the flags it replaces)
        else out.alloc((size_t)nE_);
        hipLaunchKernelGGL(ba_classify_kernel, dim3((unsigned)cdiv64(nE_, BA_T)), dim3(BA_T), 0, 0, d_,
                           (const double *)poses_[cur_].p, (const double *)points_[cur_].p, out.as<uint8_t>());
        SIVO_HIP(hipGetLastError());
        if (to_level && out.p != level_.p) SIVO_HIP(hipMemcpyAsync(level_.p, out.p, (size_t)nE_, hipMemcpyDeviceToDevice, 0));
        if (drop_kernels) SIVO_HIP(hipMemsetAsync(robust_.p, 0, (size_t)nE_, 0));
        if (outlier_host) SIVO_HIP(hipMemcpy(outlier_host, out.p, (size_t)nE_, hipMemcpyDeviceToHost));
    }

    void download(double *poses, double *points, double *err) const {
        if (poses && nP_) SIVO_HIP(hipMemcpy(poses, poses_[cur_].p, (size_t)nP_ * 96, hipMemcpyDeviceToHost));
        if (points && nX_ && !points_fixed_) SIVO_HIP(hipMemcpy(points, points_[cur_].p, (size_t)nX_ * 24, hipMemcpyDeviceToHost));
        if (err && nE_) SIVO_HIP(hipMemcpy(err, err_[last_set_].p, (size_t)nE_ * 24, hipMemcpyDeviceToHost));
    }
    // H_pp of the last linearisation (computeMarginals inverts its block, Optimizer.cc:482-487, 900-907); fetched when asked for
    const std::vector<double> &hpp_last() {
        if (nF_) SIVO_HIP(hipMemcpy(hpp_last_.data(), d_.Hpp, (size_t)nF_ * 288, hipMemcpyDeviceToHost));
        return hpp_last_;
    }
    int free_slot(int pose) const { return pose >= 0 && pose < nP_ ? slot_host_[pose] : -1; }
    int n_free() const { return nF_; }

 private:
    int nP_, nX_, nF_ = 0;
    int64_t nE_;
    bool points_fixed_;
    int cur_ = 0;
    Buf slot_, pt_off_, pt_edges_, ps_off_, ps_edges_, table_, edges_, level_, robust_, poses_[2], points_[2];
    Buf err_[2], Jp_[2], Jx_[2], wo_[2], rchi_[2], W_[2], Hll_, bl_, xl_, sc_pt_, Hpp_, bp_, S_, xs_, scal_, partial_, counter_, lm_;
    int last_set_ = 0;
    int *abort_host_ = nullptr;            // (the calling thread's pinned word, BaStopCtx)
    std::vector<double> hpp_last_;
    std::vector<int32_t> slot_host_;
    BaDev d_{};
};

}  // namespace sivo

using namespace sivo;

static void need_gpu() {
    if (sivo_device_count() < 1) throw std::runtime_error("no HIP device: libsivo_hip has no CPU fallback");
}

extern "C" int sivo_ba_optimize(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points,
                                const SivoEdge *edges, int64_t n_edges, const double intr[5], double delta_mono,
                                double delta_stereo, const uint8_t *level, const uint8_t *robust, int iterations,
                                const volatile uint8_t *stop_flag, double *err_out, double *hpp_last_out,
                                int *iterations_run, int *trials) {
    return guarded([&] {
        if (!poses || !intr || (n_edges && !edges) || (n_points && !points)) throw std::invalid_argument("null argument");
        if (iterations < 0) throw std::invalid_argument("negative iteration count");
        need_gpu();
        t_arena.reset();
        BaSolver s(poses, pose_fixed, n_poses, points, n_points, false, edges, n_edges, intr, delta_mono, delta_stereo);
        s.set_flags(level, robust);
        int tr = 0;
        const int n = s.optimize(iterations, stop_flag, &tr);
        s.download(poses, points, err_out);
        if (hpp_last_out && s.n_free()) std::memcpy(hpp_last_out, s.hpp_last().data(), (size_t)s.n_free() * 288);
        if (iterations_run) *iterations_run = n;
        if (trials) *trials = tr;
        return SIVO_OK;
    });
}

extern "C" int sivo_local_ba(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points,
                             const SivoEdge *edges, int64_t n_edges, const double intr[5], const volatile uint8_t *stop_flag,
                             uint8_t *outlier, int cov_pose, double *cov, int *cov_ok, int *iterations, int *trials) {
    return guarded([&] {
        if (!poses || !intr || (n_edges && !edges) || (n_points && !points)) throw std::invalid_argument("null argument");
        need_gpu();
        if (iterations) *iterations = 0;
        if (trials) *trials = 0;
        if (cov_ok) *cov_ok = 0;
        if (outlier && n_edges) std::memset(outlier, 0, (size_t)n_edges);
        if (stop_flag && *stop_flag) return SIVO_OK;                         // :757-761
        t_arena.reset();
        BaSolver s(poses, pose_fixed, n_poses, points, n_points, false, edges, n_edges, intr,
                   (double)std::sqrt(5.991f), (double)std::sqrt(7.815f));      // const float thHuber = sqrt(5.991f) (:646-647)
        int tr = 0;
        int n = s.optimize(5, stop_flag, &tr);                                // :763-764
        if (!(stop_flag && *stop_flag)) {                                     // bDoMore (:766-772)
            s.classify(nullptr, true, true);                                  // :774-817
            n += s.optimize(10, stop_flag, &tr);                              // :820-821
        }
        s.classify(outlier, false, false);                                    // :824-858
        s.download(poses, points, nullptr);
        if (cov && s.free_slot(cov_pose) >= 0) {
            const bool ok = inv6_spd(s.hpp_last().data() + 36 * s.free_slot(cov_pose), cov);
            if (cov_ok) *cov_ok = ok;
        }
        if (iterations) *iterations = n;
        if (trials) *trials = tr;
        return SIVO_OK;
    });
}

// Per-thread state of sivo_pose_optimize (PoseOptimization runs once or more per frame on the tracking thread): pinned host
// buffers the kernel reads its input from and writes its results to, and a stream of its own (the null stream is also
// PyTorch's default stream: a per-frame solve must not queue behind whatever the host application runs there).  Grow-only,
// released when the thread exits; nothing is allocated in a call once the buffers fit.
struct PoseCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void *h_in = nullptr, *h_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    void release() {
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        if (stream) (void)hipStreamDestroy(stream);
        h_in = h_out = nullptr; stream = nullptr; cap_in = cap_out = 0;
    }
    ~PoseCtx() { release(); }
    void reserve(size_t in_bytes, size_t out_bytes) {
        if (in_bytes > cap_in) {
            if (h_in) SIVO_HIP(hipHostFree(h_in));
            h_in = nullptr; cap_in = 0;
            const size_t cap = std::max(in_bytes * 2, (size_t)256 << 10);
            SIVO_HIP(hipHostMalloc(&h_in, cap, hipHostMallocDefault));
            cap_in = cap;
        }
        if (out_bytes > cap_out) {
            if (h_out) SIVO_HIP(hipHostFree(h_out));
            h_out = nullptr; cap_out = 0;
            const size_t cap = std::max(out_bytes * 2, (size_t)64 << 10);
            SIVO_HIP(hipHostMalloc(&h_out, cap, hipHostMallocDefault));
            cap_out = cap;
        }
    }
};
static PoseCtx &pose_ctx() {
    static thread_local PoseCtx c;
    int dev = 0;
    SIVO_HIP(hipGetDevice(&dev));
    if (c.device != dev) {
        c.release();
        int lo = 0, hi = 0;
        SIVO_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SIVO_HIP(hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi));
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pose_optimize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS_BYTES));
        c.device = dev;
    }
    return c;
}

extern "C" int sivo_pose_optimize(const double pose0[12], const double *points, int n_points, const SivoEdge *edges,
                                  int64_t n_edges, const double intr[5], uint8_t *outlier, double pose_out[12],
                                  double cov[36], int *cov_ok, double *chi2, int *n_inliers, int *iterations,
                                  int *trials) {
    return guarded([&] {
        if (!pose0 || !pose_out || !intr || (n_edges && (!edges || !points || !outlier))) throw std::invalid_argument("null argument");
        if (n_edges < 0 || n_edges > (1 << 24)) throw std::invalid_argument("edge count out of range");
        for (int64_t e = 0; e < n_edges; ++e)
            if (edges[e].point < 0 || edges[e].point >= n_points) throw std::invalid_argument("edge refers to a point outside the array");
        std::memcpy(pose_out, pose0, 96);
        if (cov_ok) *cov_ok = 0;
        if (n_inliers) *n_inliers = 0;
        if (iterations) *iterations = 0;
        if (trials) *trials = 0;
        if (n_edges < 3) {                                                    // nInitialCorrespondences < 3 (:409-411)
            if (n_edges) std::memset(outlier, 0, (size_t)n_edges);
            return SIVO_OK;
        }
        need_gpu();
        PoseCtx &c = pose_ctx();
        const size_t in_doubles = 12 + 8 * (size_t)n_edges, out_doubles = 52 + (chi2 ? (size_t)n_edges : 0);
        c.reserve(in_doubles * 8, out_doubles * 8 + (size_t)n_edges);
        double *in = (double *)c.h_in;
        std::memcpy(in, pose0, 96);
        for (int64_t e = 0; e < n_edges; ++e) {
            double *q = in + 12 + 8 * e;
            const double *X = points + 3 * (int64_t)edges[e].point;
            q[0] = edges[e].obs[0]; q[1] = edges[e].obs[1]; q[2] = edges[e].obs[2]; q[3] = edges[e].inv_sigma2;
            q[4] = X[0]; q[5] = X[1]; q[6] = X[2]; q[7] = edges[e].stereo ? 1.0 : 0.0;
        }
        PoseOptArgs a;
        a.n = (int)n_edges;
        a.K = Intr{intr[0], intr[1], intr[2], intr[3], intr[4]};
        a.delta_mono = (double)std::sqrt(5.991f); a.delta_stereo = (double)std::sqrt(7.815f);   // :307-308
        a.want_chi2 = chi2 != nullptr;
        double *h_out = (double *)c.h_out;
        uint8_t *h_flag = (uint8_t *)(h_out + out_doubles);
        if (n_edges <= PO_CAP) {
            // the kernel reads the staged edges once (into LDS) and writes a few hundred bytes: both straight through the pinned
            // host buffers — one launch and one synchronisation, no copy calls
            a.in = in; a.out = h_out; a.outlier = h_flag;
        } else {
            // edges beyond the LDS copy are read in every pass: they, and their flags, live in device memory
            t_arena.reset();
            Buf dIn, dOut;
            dIn.alloc(in_doubles * 8); dOut.alloc(out_doubles * 8 + (size_t)n_edges);
            SIVO_HIP(hipMemcpyAsync(dIn.p, in, in_doubles * 8, hipMemcpyHostToDevice, c.stream));
            a.in = dIn.as<double>(); a.out = dOut.as<double>(); a.outlier = (uint8_t *)(dOut.as<double>() + out_doubles);
        }
        hipLaunchKernelGGL(pose_optimize_kernel, dim3(1), dim3(PO_THREADS), (size_t)PO_LDS_BYTES, c.stream, a);
        SIVO_HIP(hipGetLastError());
        if (n_edges > PO_CAP) SIVO_HIP(hipMemcpyAsync(h_out, a.out, out_doubles * 8 + (size_t)n_edges, hipMemcpyDeviceToHost, c.stream));
        SIVO_HIP(hipStreamSynchronize(c.stream));
        std::memcpy(pose_out, h_out, 96);
        std::memcpy(outlier, h_flag, (size_t)n_edges);
        if (cov) std::memcpy(cov, h_out + 12, 288);
        if (chi2) std::memcpy(chi2, h_out + 52, (size_t)n_edges * 8);
        const int info[4] = {(int)h_out[48], (int)h_out[49], (int)h_out[50], (int)h_out[51]};
        if (cov_ok) *cov_ok = info[0];
        if (n_inliers) *n_inliers = (int)n_edges - info[1];
        if (iterations) *iterations = info[2];
        if (trials) *trials = info[3];
        return SIVO_OK;
    });
}
