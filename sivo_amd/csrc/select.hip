// select.hip — SIVO's information-theoretic map-point selection gate, batched over the semantic
// keypoints of a frame.  Stands behind
//   SIVO::computeStereoJacobianPose / computeStereoCovariance / computeStereoMutualInformation
//   (reference src/sivo_helpers/sivo_helpers.cpp:64-88, 160-180, 201-219)
// as they are applied in
//   Tracking::CreateNewKeyFrame (reference src/orbslam/Tracking.cc:934-1023): entropy lookup at the truncated keypoint
//     position in the entropy map the SegNet path left in HBM, depth > 0, accept iff MI - entropy > ThEntropyReduction
//     (sivo_entropy_gate);
//   LocalMapping::CheckSemantics (reference src/orbslam/LocalMapping.cc:474-538, compute_information = true): also a
//     static class (<= TERRAIN) and confidence >= ThConfidence, and the point is rejected only when
//     MI - entropy < ThEntropyReduction, i.e. it passes at equality (sivo_check_semantics).
// One thread per keypoint, fp64, determinants as Eigen takes them (3x3 cofactors, 6x6 / 9x9 partial-
// pivot LU); a few thousand independent 9x9 factorizations: latency bound, microseconds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include "common.hpp"

namespace sivo {

// diagonal().prod() in the order of Eigen's unrolled reduction (halves, recursively), written out for the two sizes used
__device__ double diag_product(const double *d, int n) {
    if (n == 6) return (d[0] * (d[1] * d[2])) * (d[3] * (d[4] * d[5]));
    return ((d[0] * d[1]) * (d[2] * d[3])) * ((d[4] * d[5]) * (d[6] * (d[7] * d[8])));     // n == 9
}

// Eigen::PartialPivLU::determinant (what Matrix<double, 6, 6> / <9, 9>::determinant() evaluates: sivo_helpers.cpp:207-216)
__device__ double det_lu(double *a, int n) {
    double diag[9];
    double sign = 1.0;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); piv = i; }
        if (best == 0.0) return 0.0;
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = a[k * n + j]; a[k * n + j] = a[piv * n + j]; a[piv * n + j] = t; }
            sign = -sign;
        }
        diag[k] = a[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = a[i * n + k] / a[k * n + k];
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
        }
    }
    return sign * diag_product(diag, n);
}

struct GateArgs {
    const SivoKeyPoint *kps;
    const float *depth;
    const double *xyz;
    const double *entropy;
    int rows, cols, n;
    double Sx[36];
    double fx, fy, bl, th;
    float level_sigma2[16];
    double *mi, *reduction;
    uint8_t *accept;
    // CheckSemantics form (classes != nullptr): accept[] receives the detected class, or VOID (255)
    const double *confidence;
    const uint8_t *classes;
    double th_conf;
};

__global__ void entropy_gate_kernel(GateArgs g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    double m = 0.0, red = 0.0;
    uint8_t acc = 0;
    const SivoKeyPoint kp = g.kps[i];
    const int col = (int)kp.x, row = (int)kp.y;
    bool ok = g.depth[i] > 0 && row >= 0 && row < g.rows && col >= 0 && col < g.cols;
    int cls = 255;
    if (g.classes) {
        acc = 255;                                                   // Classes::VOID
        if (ok) {
            cls = g.classes[(int64_t)row * g.cols + col];
            ok = cls <= 8 && g.confidence[(int64_t)row * g.cols + col] >= g.th_conf;     // <= Classes::TERRAIN, >= mThConfidence
        }
    }
    if (ok) {
        const double X = g.xyz[3 * i], Y = g.xyz[3 * i + 1], Z = g.xyz[3 * i + 2];
        const double fx = g.fx, fy = g.fy, bl = g.bl;
        double J[18];
        for (int k = 0; k < 18; ++k) J[k] = 0.0;
        if (Z != 0) {
            J[0] = fx / Z; J[1] = 0.0; J[2] = -fx * X / (Z * Z);
            J[3] = -fx * X * Y / (Z * Z); J[4] = fx * (1.0 + (X * X) / (Z * Z)); J[5] = -fx * Y / Z;
            J[6] = 0.0; J[7] = fy / Z; J[8] = -fy * Y / (Z * Z);
            J[9] = -fy * (1 + (Y * Y) / (Z * Z)); J[10] = fy * X * Y / (Z * Z); J[11] = fy * X / Z;
            J[12] = fx / Z; J[13] = 0.0; J[14] = -fx * (X - bl) / (Z * Z);
            J[15] = -fx * (X - bl) * Y / (Z * Z); J[16] = fx * (1.0 + (X * (X - bl)) / (Z * Z)); J[17] = -fx * Y / Z;
        }
        const double sigma2 = g.level_sigma2[kp.octave];
        double S9[81], JS[18], Sz[9], Sxc[36];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 6; ++b) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += J[a * 6 + k] * g.Sx[k * 6 + b];
                JS[a * 6 + b] = s;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += JS[a * 6 + k] * J[b * 6 + k];
                Sz[a * 3 + b] = s + (a == b ? sigma2 : 0.0);
            }
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) { S9[a * 9 + b] = g.Sx[a * 6 + b]; Sxc[a * 6 + b] = g.Sx[a * 6 + b]; }
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += g.Sx[a * 6 + k] * J[b * 6 + k];
                S9[a * 9 + 6 + b] = s;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 6; ++b) S9[(6 + a) * 9 + b] = JS[a * 6 + b];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S9[(6 + a) * 9 + 6 + b] = Sz[a * 3 + b];
        const double state_det = det_lu(Sxc, 6);
        const double meas_det = Sz[0] * (Sz[4] * Sz[8] - Sz[5] * Sz[7]) - Sz[1] * (Sz[3] * Sz[8] - Sz[5] * Sz[6]) +
                                Sz[2] * (Sz[3] * Sz[7] - Sz[4] * Sz[6]);
        const double cov_det = det_lu(S9, 9);
        m = 0.5 * log2(state_det * meas_det / cov_det);
        red = m - g.entropy[(int64_t)row * g.cols + col];
        if (g.classes) acc = red < g.th ? 255 : (uint8_t)cls;         // LocalMapping.cc:529-532: rejected only below the threshold
        else acc = red > g.th;
    }
    if (g.mi) g.mi[i] = m;
    if (g.reduction) g.reduction[i] = red;
    if (g.accept) g.accept[i] = acc;
}

}  // namespace sivo

using namespace sivo;

extern "C" int sivo_entropy_gate_dev(int n, const SivoKeyPoint *d_kps, const float *d_depth, const double *d_xyz,
                                     const double *d_entropy, int rows, int cols, const double state_cov[36], double fx,
                                     double fy, double bl, const float *level_sigma2, int nlevels, double th,
                                     double *d_mi, double *d_reduction, uint8_t *d_accept, void *stream) {
    return guarded([&] {
        if (n < 0 || nlevels < 1 || nlevels > 16) throw std::invalid_argument("bad sizes (nlevels <= 16)");
        if (n == 0) return SIVO_OK;
        if (!d_kps || !d_depth || !d_xyz || !d_entropy || !state_cov || !level_sigma2) throw std::invalid_argument("null argument");
        GateArgs g{};
        g.kps = d_kps; g.depth = d_depth; g.xyz = d_xyz; g.entropy = d_entropy; g.rows = rows; g.cols = cols; g.n = n;
        for (int i = 0; i < 36; ++i) g.Sx[i] = state_cov[i];
        g.fx = fx; g.fy = fy; g.bl = bl; g.th = th;
        for (int i = 0; i < nlevels; ++i) g.level_sigma2[i] = level_sigma2[i];
        g.mi = d_mi; g.reduction = d_reduction; g.accept = d_accept;
        hipLaunchKernelGGL(entropy_gate_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, g);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

// Host keypoints against the entropy map the network left in HBM (the per-frame form: the keys come out of the semantic filter on
// the host, the 2.9 MB f64 map never has to leave the device for this).  The key arrays are staged into a pinned buffer of the
// calling thread which the kernel reads directly, and the three outputs are written straight into pinned memory: one launch, one
// synchronisation, no allocation once the buffers fit.  The caller has synchronised with whatever produced d_entropy (the frame
// has: it reads the class map back before it filters the keys).
namespace {
struct GateCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    char *h = nullptr;
    size_t cap = 0;
    void release() {
        if (h) (void)hipHostFree(h);
        if (stream) (void)hipStreamDestroy(stream);
        h = nullptr; stream = nullptr; cap = 0;
    }
    ~GateCtx() { release(); }
};
GateCtx &gate_ctx(size_t bytes) {
    static thread_local GateCtx c;
    int dev = 0;
    SIVO_HIP(hipGetDevice(&dev));
    if (c.device != dev) {
        c.release();
        int lo = 0, hi = 0;          // (a dozen workgroups that a host thread waits for: ahead of whatever else the device is running)
        SIVO_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SIVO_HIP(hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi));
        c.device = dev;
    }
    if (bytes > c.cap) {
        if (c.h) SIVO_HIP(hipHostFree(c.h));
        c.h = nullptr; c.cap = 0;
        const size_t cap = std::max(bytes * 2, (size_t)256 << 10);
        SIVO_HIP(hipHostMalloc((void **)&c.h, cap, hipHostMallocDefault));
        c.cap = cap;
    }
    return c;
}
}  // namespace

extern "C" int sivo_entropy_gate_map_dev(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz,
                                         const double *d_entropy, int rows, int cols, const double state_cov[36], double fx,
                                         double fy, double bl, const float *level_sigma2, int nlevels, double th, double *mi,
                                         double *reduction, uint8_t *accept) {
    return guarded([&] {
        if (n < 0) throw std::invalid_argument("negative size");
        if (n == 0) return SIVO_OK;
        if (!kps || !depth || !xyz || !d_entropy) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        const size_t N = (size_t)n, o_xyz = 0, o_mi = o_xyz + N * 24, o_red = o_mi + N * 8, o_kps = o_red + N * 8,
                     o_depth = o_kps + N * sizeof(SivoKeyPoint), o_acc = o_depth + N * 4, total = o_acc + N;
        GateCtx &c = gate_ctx(total);
        std::memcpy(c.h + o_xyz, xyz, N * 24);
        std::memcpy(c.h + o_kps, kps, N * sizeof(SivoKeyPoint));
        std::memcpy(c.h + o_depth, depth, N * 4);
        const int rc = sivo_entropy_gate_dev(n, (const SivoKeyPoint *)(c.h + o_kps), (const float *)(c.h + o_depth), (const double *)(c.h + o_xyz),
                                             d_entropy, rows, cols, state_cov, fx, fy, bl, level_sigma2, nlevels, th,
                                             (double *)(c.h + o_mi), (double *)(c.h + o_red), (uint8_t *)(c.h + o_acc), c.stream);
        if (rc) return rc;
        SIVO_HIP(hipStreamSynchronize(c.stream));
        if (mi) std::memcpy(mi, c.h + o_mi, N * 8);
        if (reduction) std::memcpy(reduction, c.h + o_red, N * 8);
        if (accept) std::memcpy(accept, c.h + o_acc, N);
        return SIVO_OK;
    });
}

extern "C" int sivo_check_semantics_dev(int n, const SivoKeyPoint *d_kps, const float *d_depth, const double *d_xyz,
                                        const double *d_entropy, const double *d_confidence, const uint8_t *d_classes, int rows,
                                        int cols, const double state_cov[36], double fx, double fy, double bl,
                                        const float *level_sigma2, int nlevels, double th_entropy, double th_confidence,
                                        double *d_mi, double *d_reduction, uint8_t *d_detected_class, void *stream) {
    return guarded([&] {
        if (n < 0 || nlevels < 1 || nlevels > 16) throw std::invalid_argument("bad sizes (nlevels <= 16)");
        if (n == 0) return SIVO_OK;
        if (!d_kps || !d_depth || !d_xyz || !d_entropy || !d_confidence || !d_classes || !state_cov || !level_sigma2 || !d_detected_class)
            throw std::invalid_argument("null argument");
        GateArgs g{};
        g.kps = d_kps; g.depth = d_depth; g.xyz = d_xyz; g.entropy = d_entropy; g.rows = rows; g.cols = cols; g.n = n;
        for (int i = 0; i < 36; ++i) g.Sx[i] = state_cov[i];
        g.fx = fx; g.fy = fy; g.bl = bl; g.th = th_entropy;
        for (int i = 0; i < nlevels; ++i) g.level_sigma2[i] = level_sigma2[i];
        g.mi = d_mi; g.reduction = d_reduction; g.accept = d_detected_class;
        g.confidence = d_confidence; g.classes = d_classes; g.th_conf = th_confidence;
        hipLaunchKernelGGL(entropy_gate_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, g);
        SIVO_HIP(hipGetLastError());
        return SIVO_OK;
    });
}

extern "C" int sivo_check_semantics(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz, const double *entropy,
                                    const double *confidence, const uint8_t *classes, int rows, int cols, const double state_cov[36],
                                    double fx, double fy, double bl, const float *level_sigma2, int nlevels, double th_entropy,
                                    double th_confidence, double *mi, double *reduction, uint8_t *detected_class) {
    return guarded([&] {
        if (n < 0) throw std::invalid_argument("negative size");
        if (n == 0) return SIVO_OK;
        if (!kps || !depth || !xyz || !entropy || !confidence || !classes || !detected_class) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        struct Buf { void *p = nullptr; ~Buf() { (void)hipFree(p); } };
        auto up = [](Buf &b, const void *src, size_t bytes) {
            SIVO_HIP(hipMalloc(&b.p, bytes ? bytes : 1));
            if (src) SIVO_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        Buf dk, dd, dx, de, dc, dl, dm, dr, da;
        const size_t px = (size_t)rows * cols;
        up(dk, kps, (size_t)n * sizeof(SivoKeyPoint)); up(dd, depth, (size_t)n * 4); up(dx, xyz, (size_t)n * 24);
        up(de, entropy, px * 8); up(dc, confidence, px * 8); up(dl, classes, px);
        up(dm, nullptr, (size_t)n * 8); up(dr, nullptr, (size_t)n * 8); up(da, nullptr, (size_t)n);
        const int rc = sivo_check_semantics_dev(n, (const SivoKeyPoint *)dk.p, (const float *)dd.p, (const double *)dx.p, (const double *)de.p,
                                                (const double *)dc.p, (const uint8_t *)dl.p, rows, cols, state_cov, fx, fy, bl, level_sigma2,
                                                nlevels, th_entropy, th_confidence, (double *)dm.p, (double *)dr.p, (uint8_t *)da.p, nullptr);
        if (rc) return rc;
        if (mi) SIVO_HIP(hipMemcpy(mi, dm.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (reduction) SIVO_HIP(hipMemcpy(reduction, dr.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        SIVO_HIP(hipMemcpy(detected_class, da.p, (size_t)n, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}

extern "C" int sivo_entropy_gate(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz,
                                 const double *entropy, int rows, int cols, const double state_cov[36], double fx,
                                 double fy, double bl, const float *level_sigma2, int nlevels, double th, double *mi,
                                 double *reduction, uint8_t *accept) {
    return guarded([&] {
        if (n < 0) throw std::invalid_argument("negative size");
        if (n == 0) return SIVO_OK;
        if (!kps || !depth || !xyz || !entropy) throw std::invalid_argument("null argument");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        struct Buf { void *p = nullptr; ~Buf() { (void)hipFree(p); } };
        auto up = [](Buf &b, const void *src, size_t bytes) {
            SIVO_HIP(hipMalloc(&b.p, bytes ? bytes : 1));
            if (src) SIVO_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        Buf dk, dd, dx, de, dm, dr, da;
        up(dk, kps, (size_t)n * sizeof(SivoKeyPoint)); up(dd, depth, (size_t)n * 4); up(dx, xyz, (size_t)n * 24);
        up(de, entropy, (size_t)rows * cols * 8);
        up(dm, nullptr, (size_t)n * 8); up(dr, nullptr, (size_t)n * 8); up(da, nullptr, (size_t)n);
        const int rc = sivo_entropy_gate_dev(n, (const SivoKeyPoint *)dk.p, (const float *)dd.p, (const double *)dx.p,
                                             (const double *)de.p, rows, cols, state_cov, fx, fy, bl, level_sigma2, nlevels, th,
                                             (double *)dm.p, (double *)dr.p, (uint8_t *)da.p, nullptr);
        if (rc) return rc;
        if (mi) SIVO_HIP(hipMemcpy(mi, dm.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (reduction) SIVO_HIP(hipMemcpy(reduction, dr.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (accept) SIVO_HIP(hipMemcpy(accept, da.p, (size_t)n, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}
