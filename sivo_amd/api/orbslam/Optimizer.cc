// SIVO::Optimizer edge batch over libsivo_hip (reference src/orbslam/Optimizer.cc:318-409, 651-755, 774-821).
#include "Optimizer.h"

#include <cmath>
#include <stdexcept>
#include <string>

namespace SIVO {

void Optimizer::LinearizeEdges(const std::vector<double> &poses, const std::vector<double> &points,
                               const std::vector<SivoEdge> &edges, const double intr[5], EdgeBatchResult &out) {
    const size_t n = edges.size();
    out.err.resize(3 * n); out.Jpoint.resize(9 * n); out.Jpose.resize(18 * n);
    out.chi2.resize(n); out.rho.resize(n); out.weight.resize(n); out.depthPositive.resize(n);
    if (n == 0) return;
    const int rc = sivo_ba_linearize(poses.data(), (int)(poses.size() / 12), points.data(), (int)(points.size() / 3), edges.data(),
                                     (int64_t)n, intr, std::sqrt(CHI2_MONO), std::sqrt(CHI2_STEREO), out.err.data(),
                                     out.Jpoint.data(), out.Jpose.data(), out.chi2.data(), out.rho.data(), out.weight.data(),
                                     out.depthPositive.data());
    if (rc == SIVO_ERR_INVALID_ARGUMENT) throw std::invalid_argument(sivo_last_error());
    if (rc != SIVO_OK) throw std::runtime_error(std::string("Optimizer: ") + sivo_last_error());
}

int Optimizer::ClassifyOutliers(const std::vector<SivoEdge> &edges, const EdgeBatchResult &lin, std::vector<uint8_t> &outlier) {
    outlier.assign(edges.size(), 0);
    int n = 0;
    for (size_t e = 0; e < edges.size(); ++e) {
        const double th = edges[e].stereo ? CHI2_STEREO : CHI2_MONO;
        if (lin.chi2[e] > th || !lin.depthPositive[e]) { outlier[e] = 1; ++n; }
    }
    return n;
}

namespace {
void raise(int rc) {
    if (rc == SIVO_ERR_INVALID_ARGUMENT) throw std::invalid_argument(sivo_last_error());
    if (rc != SIVO_OK) throw std::runtime_error(std::string("Optimizer: ") + sivo_last_error());
}
// g2o polls the caller's `bool *` (setForceStopFlag, Optimizer.cc:573-575); the C ABI polls the same byte
static_assert(sizeof(bool) == 1, "pbStopFlag is handed to the C ABI as a byte");
const volatile uint8_t *stop_byte(const bool *p) { return reinterpret_cast<const volatile uint8_t *>(p); }
}  // namespace

int Optimizer::PoseOptimization(double pose[12], const std::vector<double> &mapPoints, const std::vector<SivoEdge> &edges,
                                const double intr[5], std::vector<uint8_t> &outlier, double covariance[36],
                                bool *covarianceValid) {
    outlier.assign(edges.size(), 0);
    double out[12], cov[36];
    int ok = 0, inliers = 0;
    raise(sivo_pose_optimize(pose, mapPoints.data(), (int)(mapPoints.size() / 3), edges.data(), (int64_t)edges.size(), intr,
                             outlier.data(), out, cov, &ok, nullptr, &inliers, nullptr, nullptr));
    for (int i = 0; i < 12; ++i) pose[i] = out[i];
    if (ok && covariance)
        for (int i = 0; i < 36; ++i) covariance[i] = cov[i];
    if (covarianceValid) *covarianceValid = ok != 0;
    return inliers;
}

void Optimizer::LocalBundleAdjustment(std::vector<double> &poses, const std::vector<uint8_t> &fixedPose,
                                      std::vector<double> &points, const std::vector<SivoEdge> &edges, const double intr[5],
                                      const bool *pbStopFlag, std::vector<uint8_t> &erase, int covariancePose,
                                      double *covariance, bool *covarianceValid) {
    if (fixedPose.size() * 12 != poses.size()) throw std::invalid_argument("fixedPose must hold one flag per keyframe");
    erase.assign(edges.size(), 0);
    double cov[36];
    int ok = 0;
    raise(sivo_local_ba(poses.data(), fixedPose.data(), (int)fixedPose.size(), points.data(), (int)(points.size() / 3),
                        edges.data(), (int64_t)edges.size(), intr, stop_byte(pbStopFlag), erase.data(),
                        covariancePose, cov, &ok, nullptr, nullptr));
    if (ok && covariance)
        for (int i = 0; i < 36; ++i) covariance[i] = cov[i];
    if (covarianceValid) *covarianceValid = ok != 0;
}

void Optimizer::BundleAdjustment(std::vector<double> &poses, const std::vector<uint8_t> &fixedPose, std::vector<double> &points,
                                 const std::vector<SivoEdge> &edges, const double intr[5], int nIterations,
                                 const bool *pbStopFlag, bool bRobust) {
    if (fixedPose.size() * 12 != poses.size()) throw std::invalid_argument("fixedPose must hold one flag per keyframe");
    const std::vector<uint8_t> robust(edges.size(), bRobust ? 1 : 0);
    raise(sivo_ba_optimize(poses.data(), fixedPose.data(), (int)fixedPose.size(), points.data(), (int)(points.size() / 3),
                           edges.data(), (int64_t)edges.size(), intr, (double)std::sqrt(5.991f), (double)std::sqrt(7.815f),   // :133-134
                           nullptr, robust.data(), nIterations, stop_byte(pbStopFlag), nullptr, nullptr, nullptr,
                           nullptr));
}

}  // namespace SIVO
