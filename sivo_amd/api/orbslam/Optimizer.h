// SIVO::Optimizer — the per-edge arithmetic of the reference class (reference include/orbslam/Optimizer.h:43-79).
//
// The reference's static members build g2o graphs from KeyFrame / MapPoint objects and hand the solve to
// g2o + CHOLMOD.  The graph walk (which keyframes / map points take part, writing the results back under the
// map mutex) is SLAM control plane and stays with the caller; everything from "the graph is built" to "the
// estimates are recovered" is provided here on arrays and runs on the GPU: per-edge computeError +
// linearizeOplus + chi2 + Huber, the Levenberg-Marquardt / Schur-complement loops with the reference's
// iteration and re-classification schedules, and the marginal pose covariance (Optimizer.cc:409-491, 757-926).
// (guard: NOT the reference's OPTIMIZER_H — a translation unit may include the reference's include/orbslam/Optimizer.h beside this one)
#ifndef SIVO_AMD_API_OPTIMIZER_H
#define SIVO_AMD_API_OPTIMIZER_H

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

    // int Optimizer::PoseOptimization(Frame *pFrame) (Optimizer.cc:273-491) from the built edges on: `pose` (12:
    // Rcw row-major, tcw = pFrame->mTcw) is optimised in place against the fixed map points; outlier = mvbOutlier;
    // covariance = the 6x6 block SetCovariance receives (left untouched and covarianceValid = false when
    // computeMarginals would fail).  Returns nInitialCorrespondences - nBad.
    static int PoseOptimization(double pose[12], const std::vector<double> &mapPoints, const std::vector<SivoEdge> &edges,
                                const double intr[5], std::vector<uint8_t> &outlier, double covariance[36],
                                bool *covarianceValid = nullptr);

    // void Optimizer::LocalBundleAdjustment(KeyFrame*, bool *pbStopFlag, Map*, bool) (Optimizer.cc:493-926) from the
    // built graph on: poses (fixedPose[i] != 0 for lFixedCameras and keyframe 0) and points are optimised in place;
    // erase[e] = 1 for the observations the caller removes (:824-878); covariance = marginal block of keyframe
    // covariancePose (:900-907).
    static void LocalBundleAdjustment(std::vector<double> &poses, const std::vector<uint8_t> &fixedPose,
                                      std::vector<double> &points, const std::vector<SivoEdge> &edges, const double intr[5],
                                      const bool *pbStopFlag, std::vector<uint8_t> &erase, int covariancePose = -1,
                                      double *covariance = nullptr, bool *covarianceValid = nullptr);

    // void Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) (Optimizer.cc:49-271)
    // from the built graph on: one optimize(nIterations) with (bRobust) or without Huber kernels.
    static void BundleAdjustment(std::vector<double> &poses, const std::vector<uint8_t> &fixedPose, std::vector<double> &points,
                                 const std::vector<SivoEdge> &edges, const double intr[5], int nIterations = 5,
                                 const bool *pbStopFlag = nullptr, bool bRobust = true);

    // ---- the reference's own static members over the SLAM object graph (reference include/orbslam/Optimizer.h:46-60), so
    // that its callers compile against this class as they are: Tracking.cc:617,753,792,1329 `Optimizer::PoseOptimization(
    // &mCurrentFrame)`, LocalMapping.cc:83 `Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap)`,
    // LoopClosing.cc:667 `Optimizer::GlobalBundleAdjustment(mpMap, 10, &mbStopGBA, nLoopKF, false)`, Tracking.cc
    // `Optimizer::GlobalBundleAdjustment(mpMap, 20)`.  Member templates over the SLAM types (the data model is outside
    // this library): any Frame / KeyFrame / MapPoint / Map exposing the members the reference's Optimizer.cc
    // uses binds, the reference's own classes included.  Defined in OptimizerAdapter.h (included below): the graph walk
    // of Optimizer.cc fills arrays, the array forms above run on the GPU, the results are written back as the reference does.
    template <class FrameT>
    static int PoseOptimization(FrameT *pFrame);
    template <class KeyFrameT, class MapT>
    static void LocalBundleAdjustment(KeyFrameT *pKF, bool *pbStopFlag, MapT *pMap);
    template <class KeyFrameT, class MapPointT>
    static void BundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP, int nIterations = 5,
                                 bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul, const bool bRobust = true);
    template <class MapT>
    static void GlobalBundleAdjustment(MapT *pMap, int nIterations = 5, bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul,
                                       const bool bRobust = true);

    // ---- loop closing (reference include/orbslam/Optimizer.h:64-79; called at LoopClosing.cc:333 `Optimizer::OptimizeSim3(mpCurrentKF,
    // pKF, vpMapPointMatches, gScm, 10, mbFixScale)` and :582 `Optimizer::OptimizeEssentialGraph(mpMap, mpMatchedKF, mpCurrentKF,
    // NonCorrectedSim3, CorrectedSim3, LoopConnections, mbFixScale)`).  Sim3 pose-graph optimisation: sequential sparse algebra over a
    // pointer graph, a handful of calls per loop closure — SLAM back end, outside the per-frame path this library accelerates
    // (out of scope).  The members exist so that LoopClosing.cc compiles against this class unchanged: with
    // -DSIVO_HAVE_G2O they forward to SIVO_G2O_BACKEND (a class with these two static members: the reference's own Optimizer.cc
    // compiled under another name is one — tests/cpp/pin_optimizer.cpp links exactly that); without it, instantiating them is a
    // compile-time error that says so.
    template <class MapT, class KeyFrameT, class KFPoseMapT, class ConnectionsT>
    static void OptimizeEssentialGraph(MapT *pMap, KeyFrameT *pLoopKF, KeyFrameT *pCurKF, const KFPoseMapT &NonCorrectedSim3,
                                       const KFPoseMapT &CorrectedSim3, const ConnectionsT &LoopConnections, const bool &bFixScale);
    template <class KeyFrameT, class MapPointT, class Sim3T>
    static int OptimizeSim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches1, Sim3T &g2oS12, const float th2,
                            const bool bFixScale);
};

}  // namespace SIVO

#include "OptimizerAdapter.h"

#endif
