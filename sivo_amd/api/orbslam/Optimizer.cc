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

}  // namespace SIVO
