// SIVO::Optimizer — the per-edge arithmetic of the reference class (reference include/orbslam/Optimizer.h:43-79).
//
// The reference's static members build g2o graphs from KeyFrame / MapPoint objects and hand the solve to
// g2o + CHOLMOD; that control plane and the sparse solve stay on the host and outside this library
// (SURVEY.md 8f-3 ranks the device-side Hessian assembly as a later step).  What runs once per LM
// iteration over ALL edges — computeError + linearizeOplus + chi2 + Huber — is provided here on arrays,
// plus the chi2 inlier classification the reference applies between optimisation rounds
// (Optimizer.cc:423-471, 774-821).
#ifndef OPTIMIZER_H
#define OPTIMIZER_H

#include <cstdint>
#include <vector>

#include "../../../include/sivo_hip.h"

namespace SIVO {

struct EdgeBatchResult {
    std::vector<double> err;      // 3 per edge
    std::vector<double> Jpoint;   // 3x3 row-major per edge (d err / d XYZ)
    std::vector<double> Jpose;    // 3x6 row-major per edge (d err / d [omega, upsilon])
    std::vector<double> chi2, rho, weight;
    std::vector<uint8_t> depthPositive;
};

class Optimizer {
 public:
    // chi2 thresholds for 95 % confidence (Optimizer.cc:296-297, 647-648)
    static constexpr double CHI2_MONO = 5.991, CHI2_STEREO = 7.815;

    // poses: 12 doubles per keyframe (Rcw row-major, tcw); points: 3 doubles per map point;
    // intr = {fx, fy, cx, cy, bf}.  One GPU launch for the whole batch.
    static void LinearizeEdges(const std::vector<double> &poses, const std::vector<double> &points,
                               const std::vector<SivoEdge> &edges, const double intr[5], EdgeBatchResult &out);

    // Outlier test of LocalBundleAdjustment (Optimizer.cc:774-821): an edge is an outlier iff
    // chi2 > 5.991 (mono) / 7.815 (stereo) or the point is behind the camera.
    static int ClassifyOutliers(const std::vector<SivoEdge> &edges, const EdgeBatchResult &lin, std::vector<uint8_t> &outlier);
};

}  // namespace SIVO
#endif
