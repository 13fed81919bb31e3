// ba.hip — batched per-edge reprojection residual / Jacobian for bundle adjustment.
//
// Stands behind the g2o edges SIVO::Optimizer builds (reference
// src/orbslam/Optimizer.cc:318-409 PoseOptimization, :651-755 LocalBundleAdjustment):
// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ computeError + linearizeOplus, chi2,
// RobustKernelHuber.  g2o itself is an un-vendored submodule (.gitmodules:4-6); the
// formulas are those of ORB-SLAM2's types_six_dof_expmap.
//
// fp64, one thread per edge, HBM-bound (~0.37 KB of traffic per edge).  The 30 output
// doubles of an edge are staged through LDS so a workgroup writes err / Jx / Jp as
// contiguous, fully coalesced slabs instead of 144-byte-strided scalars.
// Floating-point contraction is off so results are bit-identical to a plain IEEE
// evaluation of the same expressions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace sivo {

constexpr int BA_THREADS = 128;

__global__ __launch_bounds__(BA_THREADS) void ba_linearize_kernel(
    const double *__restrict__ poses, const double *__restrict__ points, const SivoEdge *__restrict__ edges,
    int64_t nE, double fx, double fy, double cx, double cy, double bf, double delta_mono, double delta_stereo,
    double *err, double *Jx, double *Jp, double *chi2, double *rho, double *w, uint8_t *depth_ok) {
    __shared__ double s_out[BA_THREADS * 30];
    const int64_t e0 = (int64_t)blockIdx.x * BA_THREADS;
    const int64_t e = e0 + threadIdx.x;
    double *mine = s_out + threadIdx.x * 30;  // [0..3) err, [3..12) Jx, [12..30) Jp
    if (e < nE) {
        const SivoEdge ed = edges[e];
        const double *R = poses + 12 * (int64_t)ed.pose, *t = R + 9;
        const double *X = points + 3 * (int64_t)ed.point;
        const double X0 = X[0], X1 = X[1], X2 = X[2];
        double Rm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rm[i] = R[i];
        const double x = Rm[0] * X0 + Rm[1] * X1 + Rm[2] * X2 + t[0];
        const double y = Rm[3] * X0 + Rm[4] * X1 + Rm[5] * X2 + t[1];
        const double z = Rm[6] * X0 + Rm[7] * X1 + Rm[8] * X2 + t[2];
        const double invz = 1.0 / z, z_2 = z * z;
        const bool stereo = ed.stereo != 0;

        const double e0v = ed.obs[0] - (x * invz * fx + cx);
        const double e1v = ed.obs[1] - (y * invz * fy + cy);
        const double e2v = stereo ? ed.obs[2] - (x * invz * fx + cx - bf * invz) : 0.0;
        mine[0] = e0v; mine[1] = e1v; mine[2] = e2v;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double j0 = -fx * Rm[j] / z + fx * x * Rm[6 + j] / z_2;
            mine[3 + j] = j0;
            mine[6 + j] = -fy * Rm[3 + j] / z + fy * y * Rm[6 + j] / z_2;
            mine[9 + j] = stereo ? j0 - bf * Rm[6 + j] / z_2 : 0.0;
        }
        double *jp = mine + 12;
        jp[0] = x * y / z_2 * fx;
        jp[1] = -(1 + (x * x / z_2)) * fx;
        jp[2] = y / z * fx;
        jp[3] = -1. / z * fx;
        jp[4] = 0;
        jp[5] = x / z_2 * fx;
        jp[6] = (1 + y * y / z_2) * fy;
        jp[7] = -x * y / z_2 * fy;
        jp[8] = -x / z * fy;
        jp[9] = 0;
        jp[10] = -1. / z * fy;
        jp[11] = y / z_2 * fy;
        if (stereo) {
            jp[12] = jp[0] - bf * y / z_2;
            jp[13] = jp[1] + bf * x / z_2;
            jp[14] = jp[2];
            jp[15] = jp[3];
            jp[16] = 0;
            jp[17] = jp[5] - bf / z_2;
        } else {
#pragma unroll
            for (int j = 12; j < 18; ++j) jp[j] = 0.0;
        }
        const double c2 = (e0v * e0v + e1v * e1v + e2v * e2v) * ed.inv_sigma2;
        const double delta = stereo ? delta_stereo : delta_mono;
        const double dsqr = delta * delta;
        double r, wt;
        if (c2 <= dsqr) { r = c2; wt = 1.0; }
        else { const double s = sqrt(c2); r = 2 * s * delta - dsqr; wt = delta / s; }
        if (chi2) chi2[e] = c2;
        if (rho) rho[e] = r;
        if (w) w[e] = wt;
        if (depth_ok) depth_ok[e] = z > 0.0;
    }
    __syncthreads();
    const int64_t nblk = nE - e0 < BA_THREADS ? nE - e0 : BA_THREADS;
    // coalesced slab writes: element k of the block's slab comes from edge k/len, slot k%len
    if (err)
        for (int k = threadIdx.x; k < nblk * 3; k += BA_THREADS) err[e0 * 3 + k] = s_out[(k / 3) * 30 + k % 3];
    if (Jx)
        for (int k = threadIdx.x; k < nblk * 9; k += BA_THREADS) Jx[e0 * 9 + k] = s_out[(k / 9) * 30 + 3 + k % 9];
    if (Jp)
        for (int k = threadIdx.x; k < nblk * 18; k += BA_THREADS) Jp[e0 * 18 + k] = s_out[(k / 18) * 30 + 12 + k % 18];
}

}  // namespace sivo

using namespace sivo;

extern "C" int sivo_ba_linearize_dev(const double *d_poses, const double *d_points, const SivoEdge *d_edges,
                                     int64_t n_edges, const double intr[5], double delta_mono, double delta_stereo,
                                     double *d_err, double *d_jx, double *d_jp, double *d_chi2, double *d_rho,
                                     double *d_w, uint8_t *d_depth_ok, void *stream) {
    return guarded([&] {
        if (n_edges < 0) throw std::invalid_argument("negative edge count");
        if (n_edges == 0) return SIVO_OK;
        if (!d_poses || !d_points || !d_edges || !intr) throw std::invalid_argument("null argument");
        hipLaunchKernelGGL(ba_linearize_kernel, dim3((unsigned)cdiv64(n_edges, BA_THREADS)), dim3(BA_THREADS), 0,
                           (hipStream_t)stream, d_poses, d_points, d_edges, n_edges, intr[0], intr[1], intr[2], intr[3],
                           intr[4], delta_mono, delta_stereo, d_err, d_jx, d_jp, d_chi2, d_rho, d_w, d_depth_ok);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_ba_linearize(const double *poses, int n_poses, const double *points, int n_points,
                                 const SivoEdge *edges, int64_t n_edges, const double intr[5], double delta_mono,
                                 double delta_stereo, double *err, double *jx, double *jp, double *chi2, double *rho,
                                 double *w, uint8_t *depth_ok) {
    return guarded([&] {
        if (n_edges < 0 || n_poses < 0 || n_points < 0) throw std::invalid_argument("negative size");
        if (n_edges == 0) return SIVO_OK;
        if (!poses || !points || !edges || !intr) throw std::invalid_argument("null argument");
        for (int64_t e = 0; e < n_edges; ++e)
            if (edges[e].pose < 0 || edges[e].pose >= n_poses || edges[e].point < 0 || edges[e].point >= n_points)
                throw std::invalid_argument("edge refers to a pose/point outside the arrays");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        struct Buf { void *p = nullptr; ~Buf() { (void)hipFree(p); } };
        auto up = [](Buf &b, const void *src, size_t bytes) {
            SIVO_HIP(hipMalloc(&b.p, bytes ? bytes : 1));
            if (src) SIVO_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        Buf dP, dX, dE, dErr, dJx, dJp, dC, dR, dW, dOk;
        up(dP, poses, (size_t)n_poses * 12 * 8); up(dX, points, (size_t)n_points * 3 * 8);
        up(dE, edges, (size_t)n_edges * sizeof(SivoEdge));
        up(dErr, nullptr, n_edges * 3 * 8); up(dJx, nullptr, n_edges * 9 * 8); up(dJp, nullptr, n_edges * 18 * 8);
        up(dC, nullptr, n_edges * 8); up(dR, nullptr, n_edges * 8); up(dW, nullptr, n_edges * 8); up(dOk, nullptr, n_edges);
        int rc = sivo_ba_linearize_dev((const double *)dP.p, (const double *)dX.p, (const SivoEdge *)dE.p, n_edges, intr,
                                       delta_mono, delta_stereo, (double *)dErr.p, (double *)dJx.p, (double *)dJp.p,
                                       (double *)dC.p, (double *)dR.p, (double *)dW.p, (uint8_t *)dOk.p, nullptr);
        if (rc) return rc;
        auto down = [](void *dst, const Buf &b, size_t bytes) { if (dst) SIVO_HIP(hipMemcpy(dst, b.p, bytes, hipMemcpyDeviceToHost)); };
        down(err, dErr, n_edges * 3 * 8); down(jx, dJx, n_edges * 9 * 8); down(jp, dJp, n_edges * 18 * 8);
        down(chi2, dC, n_edges * 8); down(rho, dR, n_edges * 8); down(w, dW, n_edges * 8); down(depth_ok, dOk, n_edges);
        return SIVO_OK;
    });
}
