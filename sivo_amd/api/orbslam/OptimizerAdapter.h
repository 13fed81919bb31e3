// The reference signatures of SIVO::Optimizer over the SLAM object graph (reference include/orbslam/Optimizer.h:46-60):
//     int  Optimizer::PoseOptimization(Frame *pFrame)                                  Optimizer.cc:273-491
//     void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *)    Optimizer.cc:493-926
//     void Optimizer::BundleAdjustment(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust)      Optimizer.cc:49-271
//     void Optimizer::GlobalBundleAdjustment(Map *, nIterations, pbStopFlag, nLoopKF, bRobust)     Optimizer.cc:37-47
// as header-only templates (free functions, and the static members of class Optimizer that forward to them): the graph walk the reference does in front of g2o (which observations become edges, mono
// or stereo by mvRight, information = mvInvLevelSigma2[octave], which keyframes are fixed) fills SivoEdge arrays, the
// array-level Optimizer entry points (GPU: Levenberg-Marquardt + Schur, chi2 schedules, marginal covariance) do what
// g2o + CHOLMOD do, and the results are written back the way the reference does (SetPose, mvbOutlier, SetCovariance,
// EraseMapPointMatch / EraseObservation under the map mutex, SetWorldPos + UpdateNormalAndDepth).
//
// Frame / KeyFrame / MapPoint / Map are template parameters (SLAM data model, outside this library — SURVEY.md 8): any
// types exposing the members used below under the reference's names compile, the reference's own classes included.
// tests/cpp/test_api.cpp instantiates both with minimal stand-ins.
#ifndef OPTIMIZER_ADAPTER_H
#define OPTIMIZER_ADAPTER_H

#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <vector>

#include "Optimizer.h"

#ifdef SIVO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "../compat/cv_min.hpp"
#endif
#ifdef SIVO_HAVE_EIGEN
#include <Eigen/Core>      // Frame / KeyFrame::SetCovariance(Eigen::MatrixXd) (reference Frame.cc:254-260)
#endif

namespace SIVO {
namespace optimizer_detail {

// Converter::toSE3Quat (reference src/orbslam/Converter.cc:33-42) turns the float pose into an Eigen quaternion (unit
// norm) + translation, and g2o works with the rotation matrix of THAT quaternion: the float matrix is re-orthogonalised
// on the way in.  Same here: matrix -> quaternion (the trace-branch construction Eigen uses) -> normalise -> matrix.
inline void se3_from_cv(const cv::Mat &Tcw, double pose[12]) {
    double R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = Tcw.at<float>(r, c);
    double q[4];     // x y z w
    const double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0.0) {
        double s = std::sqrt(t + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[2][1] - R[1][2]) * s; q[1] = (R[0][2] - R[2][0]) * s; q[2] = (R[1][0] - R[0][1]) * s;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R[k][j] - R[j][k]) * s; q[j] = (R[j][i] + R[i][j]) * s; q[k] = (R[k][i] + R[i][k]) * s;
    }
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double &v : q) v /= nrm;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
                 tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double M[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int i = 0; i < 9; ++i) pose[i] = M[i];
    for (int r = 0; r < 3; ++r) pose[9 + r] = Tcw.at<float>(r, 3);
}

// Converter::toCvMat(g2o::SE3Quat) (Converter.cc:44-48): 4 x 4 CV_32F
inline cv::Mat cv_from_se3(const double pose[12]) {
    cv::Mat T = cv::Mat::zeros(4, 4, CV_32F);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T.at<float>(r, c) = (float)pose[3 * r + c];
        T.at<float>(r, 3) = (float)pose[9 + r];
    }
    T.at<float>(3, 3) = 1.f;
    return T;
}

// one observation as the reference sets it up (Optimizer.cc:318-409 resp. 668-755)
template <class FrameLike>
SivoEdge observation(const FrameLike &F, size_t idx, int pose, int point) {
    SivoEdge e{};
    const cv::KeyPoint &kp = F.mvKeysSemantic[idx];
    e.pose = pose; e.point = point;
    e.obs[0] = kp.pt.x; e.obs[1] = kp.pt.y;
    e.stereo = F.mvRight[idx] < 0 ? 0 : 1;
    if (e.stereo) e.obs[2] = F.mvRight[idx];
    e.inv_sigma2 = F.mvInvLevelSigma2[kp.octave];
    return e;
}

template <class T>
void set_covariance(T *obj, const double cov[36]) {
#ifdef SIVO_HAVE_EIGEN
    Eigen::MatrixXd S(6, 6);
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) S(r, c) = cov[6 * r + c];
    obj->SetCovariance(S);
#else
    obj->SetCovariance(cov);      // 6 x 6 row-major
#endif
}

}  // namespace optimizer_detail

// int Optimizer::PoseOptimization(Frame *pFrame)
template <class FrameT>
int PoseOptimization(FrameT *pFrame) {
    using namespace optimizer_detail;
    const int N = pFrame->numSemanticKeys;
    std::vector<SivoEdge> edges;
    std::vector<size_t> index;
    std::vector<double> points;
    edges.reserve((size_t)N); index.reserve((size_t)N); points.reserve(3 * (size_t)N);
    {
        // (the reference holds MapPoint::mGlobalMutex here, :312; the stand-in types decide what that means)
        for (int i = 0; i < N; ++i) {
            auto *pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            pFrame->mvbOutlier[i] = false;
            const cv::Mat Xw = pMP->GetWorldPos();
            edges.push_back(observation(*pFrame, (size_t)i, 0, (int)index.size()));
            index.push_back((size_t)i);
            for (int r = 0; r < 3; ++r) points.push_back(Xw.at<float>(r, 0));
        }
    }
    const int nInitialCorrespondences = (int)edges.size();
    if (nInitialCorrespondences < 3) return 0;                                   // :409-411
    double pose[12], cov[36];
    se3_from_cv(pFrame->mTcw, pose);
    const double intr[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};
    std::vector<uint8_t> outlier;
    bool covOk = false;
    const int inliers = Optimizer::PoseOptimization(pose, points, edges, intr, outlier, cov, &covOk);
    for (size_t e = 0; e < edges.size(); ++e) pFrame->mvbOutlier[index[e]] = outlier[e] != 0;
    pFrame->SetPose(cv_from_se3(pose));                                          // :470-475
    if (covOk) set_covariance(pFrame, cov);                                      // :477-485
    return inliers;
}

// void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)
template <class KeyFrameT, class MapT>
void LocalBundleAdjustment(KeyFrameT *pKF, bool *pbStopFlag, MapT *pMap) {
    using namespace optimizer_detail;
    typedef typename std::remove_pointer<typename decltype(pKF->GetMapPointMatches())::value_type>::type MapPointT;
    // local keyframes: the current one and its covisible neighbours (:496-512)
    std::list<KeyFrameT *> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    for (KeyFrameT *pKFi : pKF->GetVectorCovisibleKeyFrames()) {
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    // local map points: everything the local keyframes see (:514-535)
    std::list<MapPointT *> lLocalMapPoints;
    for (KeyFrameT *pKFi : lLocalKeyFrames)
        for (MapPointT *pMP : pKFi->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
    // fixed keyframes: see local points without being local (:537-562)
    std::list<KeyFrameT *> lFixedKFs;
    for (MapPointT *pMP : lLocalMapPoints)
        for (const auto &ob : pMP->GetObservations()) {
            KeyFrameT *pKFi = ob.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedKFs.push_back(pKFi);
            }
        }
    // vertices -> arrays (:577-618): local keyframes first (keyframe 0 of the map is held fixed), then the fixed ones
    std::map<KeyFrameT *, int> poseIndex;
    std::vector<double> poses;
    std::vector<uint8_t> fixedPose;
    auto add_pose = [&](KeyFrameT *k, bool fixed) {
        double p[12];
        se3_from_cv(k->GetPose(), p);
        poseIndex[k] = (int)fixedPose.size();
        poses.insert(poses.end(), p, p + 12);
        fixedPose.push_back(fixed ? 1 : 0);
    };
    for (KeyFrameT *k : lLocalKeyFrames) add_pose(k, k->mnId == 0);
    for (KeyFrameT *k : lFixedKFs) add_pose(k, true);
    // points and edges (:646-755)
    std::vector<double> points;
    std::vector<SivoEdge> edges;
    std::vector<KeyFrameT *> edgeKF;
    std::vector<MapPointT *> edgeMP;
    int pointIndex = 0;
    for (MapPointT *pMP : lLocalMapPoints) {
        const cv::Mat Xw = pMP->GetWorldPos();
        for (int r = 0; r < 3; ++r) points.push_back(Xw.at<float>(r, 0));
        for (const auto &ob : pMP->GetObservations()) {
            KeyFrameT *pKFi = ob.first;
            if (pKFi->isBad()) continue;
            const auto it = poseIndex.find(pKFi);
            if (it == poseIndex.end()) continue;          // (bad keyframes never became vertices)
            edges.push_back(observation(*pKFi, ob.second, it->second, pointIndex));
            edgeKF.push_back(pKFi); edgeMP.push_back(pMP);
        }
        ++pointIndex;
    }
    if (pbStopFlag && *pbStopFlag) return;                                       // :757-761
    const double intr[5] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf};
    std::vector<uint8_t> erase;
    double cov[36];
    bool covOk = false;
    Optimizer::LocalBundleAdjustment(poses, fixedPose, points, edges, intr, pbStopFlag, erase, /*covariancePose=*/0, cov, &covOk);

    // vToErase (:824-858): the monocular edges in edge order, THEN the stereo ones; a map point's isBad() is read while the list
    // is built, i.e. before anything is erased (an erasure can turn a point bad: its later entries are still erased).  Both
    // pinned against the reference's own Optimizer.cc by tests/cpp/pin_optimizer.cpp.
    std::vector<size_t> vToErase;
    for (int stereo = 0; stereo < 2; ++stereo)
        for (size_t e = 0; e < edges.size(); ++e)
            if ((edges[e].stereo != 0) == (stereo != 0) && erase[e] && !edgeMP[e]->isBad()) vToErase.push_back(e);
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                    // :860-861
    for (size_t e : vToErase) {                                                  // :863-871
        edgeKF[e]->EraseMapPointMatch(edgeMP[e]);
        edgeMP[e]->EraseObservation(edgeKF[e]);
    }
    for (KeyFrameT *k : lLocalKeyFrames) {                                       // :884-910
        k->SetPose(cv_from_se3(poses.data() + 12 * (size_t)poseIndex[k]));
        if (k->mnId == pKF->mnId && covOk) set_covariance(pKF, cov);
    }
    pointIndex = 0;
    for (MapPointT *pMP : lLocalMapPoints) {                                     // :912-925
        cv::Mat X(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) X.at<float>(r, 0) = (float)points[3 * (size_t)pointIndex + r];
        pMP->SetWorldPos(X);
        pMP->UpdateNormalAndDepth();
        ++pointIndex;
    }
}

// void Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) (Optimizer.cc:49-271)
template <class KeyFrameT, class MapPointT>
void BundleAdjustment(const std::vector<KeyFrameT *> &vpKFs, const std::vector<MapPointT *> &vpMP, int nIterations = 5,
                      bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul, const bool bRobust = true) {
    using namespace optimizer_detail;
    // keyframe vertices (:77-97): every keyframe that is not bad, the map's first one held fixed
    std::map<KeyFrameT *, int> poseIndex;
    std::vector<double> poses;
    std::vector<uint8_t> fixedPose;
    unsigned long maxKFid = 0;
    for (KeyFrameT *pKF : vpKFs) {
        if (pKF->isBad()) continue;
        double p[12];
        se3_from_cv(pKF->GetPose(), p);
        poseIndex[pKF] = (int)fixedPose.size();
        poses.insert(poses.end(), p, p + 12);
        fixedPose.push_back(pKF->mnId == 0 ? 1 : 0);
        if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
    }
    if (poseIndex.empty()) return;
    // map point vertices and their observations (:102-211); a point without any edge is removed again (:204-210)
    std::vector<int> pointIndex(vpMP.size(), -1);          // vbNotIncludedMP[i] <=> pointIndex[i] < 0
    std::vector<double> points;
    std::vector<SivoEdge> edges;
    int nPoints = 0;
    for (size_t i = 0; i < vpMP.size(); ++i) {
        MapPointT *pMP = vpMP[i];
        if (pMP->isBad()) continue;
        const size_t first = edges.size();
        for (const auto &ob : pMP->GetObservations()) {
            KeyFrameT *pKF = ob.first;
            if (pKF->isBad() || pKF->mnId > maxKFid) continue;
            const auto it = poseIndex.find(pKF);
            if (it == poseIndex.end()) continue;            // (no vertex of that id: g2o refuses the edge)
            edges.push_back(observation(*pKF, ob.second, it->second, nPoints));
        }
        if (edges.size() == first) continue;
        const cv::Mat Xw = pMP->GetWorldPos();
        for (int r = 0; r < 3; ++r) points.push_back(Xw.at<float>(r, 0));
        pointIndex[i] = nPoints++;
    }
    KeyFrameT *any = poseIndex.begin()->first;
    const double intr[5] = {any->fx, any->fy, any->cx, any->cy, any->mbf};
    if (!edges.empty()) Optimizer::BundleAdjustment(poses, fixedPose, points, edges, intr, nIterations, pbStopFlag, bRobust);   // :214-217
    // keyframes (:219-235)
    for (KeyFrameT *pKF : vpKFs) {
        if (pKF->isBad()) continue;
        const cv::Mat T = cv_from_se3(poses.data() + 12 * (size_t)poseIndex[pKF]);
        if (nLoopKF == 0) {
            pKF->SetPose(T);
        } else {
            pKF->mTcwGBA = T.clone();
            pKF->mnBAGlobalForKF = nLoopKF;
        }
    }
    // points (:237-260)
    for (size_t i = 0; i < vpMP.size(); ++i) {
        if (pointIndex[i] < 0) continue;
        MapPointT *pMP = vpMP[i];
        if (pMP->isBad()) continue;
        cv::Mat X(3, 1, CV_32F);
        for (int r = 0; r < 3; ++r) X.at<float>(r, 0) = (float)points[3 * (size_t)pointIndex[i] + r];
        if (nLoopKF == 0) {
            pMP->SetWorldPos(X);
            pMP->UpdateNormalAndDepth();
        } else {
            pMP->mPosGBA = X.clone();
            pMP->mnBAGlobalForKF = nLoopKF;
        }
    }
}

// void Optimizer::GlobalBundleAdjustment(Map *pMap, nIterations, pbStopFlag, nLoopKF, bRobust) (Optimizer.cc:37-47)
template <class MapT>
void GlobalBundleAdjustment(MapT *pMap, int nIterations = 5, bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul,
                            const bool bRobust = true) {
    const auto vpKFs = pMap->GetAllKeyFrames();
    const auto vpMP = pMap->GetAllMapPoints();
    BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

// the static members of class Optimizer (declared in Optimizer.h)
template <class FrameT>
int Optimizer::PoseOptimization(FrameT *pFrame) { return SIVO::PoseOptimization(pFrame); }
template <class KeyFrameT, class MapT>
void Optimizer::LocalBundleAdjustment(KeyFrameT *pKF, bool *pbStopFlag, MapT *pMap) { SIVO::LocalBundleAdjustment(pKF, pbStopFlag, pMap); }
template <class KeyFrameT, class MapPointT>
void Optimizer::BundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP, int nIterations, bool *pbStopFlag,
                                 const unsigned long nLoopKF, const bool bRobust) {
    SIVO::BundleAdjustment(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}
template <class MapT>
void Optimizer::GlobalBundleAdjustment(MapT *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    SIVO::GlobalBundleAdjustment(pMap, nIterations, pbStopFlag, nLoopKF, bRobust);
}

// Loop closing: Optimizer::OptimizeEssentialGraph / OptimizeSim3 (declared in Optimizer.h)
#ifdef SIVO_HAVE_G2O
template <class MapT, class KeyFrameT, class KFPoseMapT, class ConnectionsT>
void Optimizer::OptimizeEssentialGraph(MapT *pMap, KeyFrameT *pLoopKF, KeyFrameT *pCurKF, const KFPoseMapT &NonCorrectedSim3, const KFPoseMapT &CorrectedSim3,
                                       const ConnectionsT &LoopConnections, const bool &bFixScale) {
    SIVO_G2O_BACKEND::OptimizeEssentialGraph(pMap, pLoopKF, pCurKF, NonCorrectedSim3, CorrectedSim3, LoopConnections, bFixScale);
}
template <class KeyFrameT, class MapPointT, class Sim3T>
int Optimizer::OptimizeSim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches1, Sim3T &g2oS12, const float th2, const bool bFixScale) {
    return SIVO_G2O_BACKEND::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale);
}
#else
template <class MapT, class KeyFrameT, class KFPoseMapT, class ConnectionsT>
void Optimizer::OptimizeEssentialGraph(MapT *, KeyFrameT *, KeyFrameT *, const KFPoseMapT &, const KFPoseMapT &, const ConnectionsT &, const bool &) {
    static_assert(sizeof(MapT) == 0, "Optimizer::OptimizeEssentialGraph (Sim3 pose graph, g2o) is outside this library: build with -DSIVO_HAVE_G2O and "
                                     "-DSIVO_G2O_BACKEND=<a class providing it, e.g. the reference's Optimizer.cc compiled under another name>");
}
template <class KeyFrameT, class MapPointT, class Sim3T>
int Optimizer::OptimizeSim3(KeyFrameT *, KeyFrameT *, std::vector<MapPointT *> &, Sim3T &, const float, const bool) {
    static_assert(sizeof(KeyFrameT) == 0, "Optimizer::OptimizeSim3 (Sim3 alignment, g2o) is outside this library: build with -DSIVO_HAVE_G2O and "
                                          "-DSIVO_G2O_BACKEND=<a class providing it>");
    return 0;
}
#endif

}  // namespace SIVO
#endif
