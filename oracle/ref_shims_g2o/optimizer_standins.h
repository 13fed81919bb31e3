// Stand-ins for the SLAM data model as src/orbslam/Optimizer.cc sees it (Frame, KeyFrame, MapPoint, Map, LoopClosing), with the
// reference's member names — TEST INFRASTRUCTURE ONLY.  The reference's own Optimizer.cc is compiled against them (oracle/Makefile
// `ref`), the SIVO::Optimizer member templates of sivo_amd/api/orbslam/OptimizerAdapter.h are instantiated on the very same types,
// and tests/cpp/pin_optimizer.cpp compares the two on identical scenes.  Every mutation the optimizer performs on the object graph
// is logged in order (opt_log()), which is what the comparison reads besides the final state.
//   restated here (the reference's classes pull in OpenCV / DBoW2 / Pangolin):
//     KeyFrame::EraseMapPointMatch(MapPoint*)   KeyFrame.cc:238-245  (GetIndexInKeyFrame, then the slot is cleared)
//     MapPoint::EraseObservation                MapPoint.cc:164-193  (observation count by mono / stereo; bad below three)
#pragma once
#include <opencv2/core/core.hpp>

#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "g2o_standin.hpp"

namespace SIVO {

class KeyFrame;
class MapPoint;

struct OptEvent {
    int kind;          // 0 Frame::SetPose, 1 Frame::SetCovariance, 2 KeyFrame::SetPose, 3 KeyFrame::SetCovariance, 4 EraseMapPointMatch(kf a, point b),
                       // 5 EraseObservation(point a, kf b), 6 SetWorldPos(point a), 7 UpdateNormalAndDepth(point a)
    long a, b;
    bool operator==(const OptEvent &o) const { return kind == o.kind && a == o.a && b == o.b; }
};
inline std::vector<OptEvent> &opt_log() { static std::vector<OptEvent> log; return log; }

class Frame {
 public:
    std::vector<cv::KeyPoint> mvKeysSemantic;
    std::vector<float> mvRight, mvInvLevelSigma2;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int numSemanticKeys = 0;
    long mnId = 0;
    double mSigmacw[36] = {0};
    bool covarianceSet = false;
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); opt_log().push_back(OptEvent{0, mnId, 0}); }
    void SetCovariance(const Eigen::MatrixXd &S) {               // the reference's signature (Frame.cc:254-260)
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) mSigmacw[6 * r + c] = S(r, c);
        covarianceSet = true;
        opt_log().push_back(OptEvent{1, mnId, 0});
    }
    void SetCovariance(const double *c) {                        // sivo_amd/api without Eigen: 6 x 6 row-major
        std::memcpy(mSigmacw, c, sizeof mSigmacw);
        covarianceSet = true;
        opt_log().push_back(OptEvent{1, mnId, 0});
    }
};

class MapPoint {
 public:
    static std::mutex mGlobalMutex;
    unsigned long mnId = 0;
    cv::Mat mWorldPos, mPosGBA;
    bool mbBad = false;
    int nObs = 0;
    std::map<KeyFrame *, size_t> mObservations;
    unsigned long mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0, mnCorrectedByKF = 0, mnCorrectedReference = 0;
    KeyFrame *mpRefKF = nullptr;
    int normalUpdates = 0;

    bool isBad() const { return mbBad; }
    cv::Mat GetWorldPos() const { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat &Pos) { Pos.copyTo(mWorldPos); opt_log().push_back(OptEvent{6, (long)mnId, 0}); }
    void UpdateNormalAndDepth() { ++normalUpdates; opt_log().push_back(OptEvent{7, (long)mnId, 0}); }
    std::map<KeyFrame *, size_t> GetObservations() const { return mObservations; }
    int GetIndexInKeyFrame(KeyFrame *pKF) const {
        const auto it = mObservations.find(pKF);
        return it == mObservations.end() ? -1 : (int)it->second;
    }
    KeyFrame *GetReferenceKeyFrame() const { return mpRefKF; }
    inline void EraseObservation(KeyFrame *pKF);
};

class KeyFrame : public Frame {
 public:
    unsigned long mnId = 0;                  // (shadows Frame::mnId: the reference's KeyFrame has its own counter)
    bool mbBad = false;
    unsigned long mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    cv::Mat mTcwGBA, mK;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    std::map<KeyFrame *, int> mConnectedKeyFrameWeights;
    KeyFrame *mpParent = nullptr;
    std::set<KeyFrame *> mspChildrens, mspLoopEdges;

    bool isBad() const { return mbBad; }
    cv::Mat GetPose() const { return mTcw.clone(); }
    void SetPose(const cv::Mat &Tcw) { Tcw.copyTo(mTcw); opt_log().push_back(OptEvent{2, (long)mnId, 0}); }
    void SetCovariance(const Eigen::MatrixXd &S) {
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) mSigmacw[6 * r + c] = S(r, c);
        covarianceSet = true;
        opt_log().push_back(OptEvent{3, (long)mnId, 0});
    }
    void SetCovariance(const double *c) {
        std::memcpy(mSigmacw, c, sizeof mSigmacw);
        covarianceSet = true;
        opt_log().push_back(OptEvent{3, (long)mnId, 0});
    }
    cv::Mat GetRotation() const { return mTcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() const { return mTcw.rowRange(0, 3).col(3).clone(); }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() const { return mvpOrderedConnectedKeyFrames; }
    std::vector<KeyFrame *> GetCovisiblesByWeight(const int &w) const {
        std::vector<KeyFrame *> out;
        for (KeyFrame *k : mvpOrderedConnectedKeyFrames)
            if (GetWeight(k) >= w) out.push_back(k);
        return out;
    }
    int GetWeight(KeyFrame *pKF) const {
        const auto it = mConnectedKeyFrameWeights.find(pKF);
        return it == mConnectedKeyFrameWeights.end() ? 0 : it->second;
    }
    KeyFrame *GetParent() const { return mpParent; }
    bool hasChild(KeyFrame *pKF) const { return mspChildrens.count(pKF) != 0; }
    std::set<KeyFrame *> GetLoopEdges() const { return mspLoopEdges; }
    std::vector<MapPoint *> GetMapPointMatches() const { return mvpMapPoints; }
    void EraseMapPointMatch(MapPoint *pMP) {
        opt_log().push_back(OptEvent{4, (long)mnId, (long)pMP->mnId});
        const int idx = pMP->GetIndexInKeyFrame(this);
        if (idx >= 0) mvpMapPoints[(size_t)idx] = nullptr;
    }
};

inline void MapPoint::EraseObservation(KeyFrame *pKF) {
    opt_log().push_back(OptEvent{5, (long)mnId, (long)pKF->mnId});
    const auto it = mObservations.find(pKF);
    if (it == mObservations.end()) return;
    if (pKF->mvRight[it->second] >= 0) nObs -= 2;
    else nObs--;
    mObservations.erase(it);
    if (mpRefKF == pKF) mpRefKF = mObservations.empty() ? nullptr : mObservations.begin()->first;
    if (nObs <= 2) mbBad = true;                 // (SetBadFlag: the full clean-up is Map / KeyFrame bookkeeping outside the optimizer)
}

class Map {
 public:
    std::mutex mMutexMapUpdate;
    std::vector<KeyFrame *> keyframes;
    std::vector<MapPoint *> points;
    std::vector<KeyFrame *> GetAllKeyFrames() const { return keyframes; }
    std::vector<MapPoint *> GetAllMapPoints() const { return points; }
    long unsigned int GetMaxKFid() const {
        unsigned long m = 0;
        for (KeyFrame *k : keyframes) m = k->mnId > m ? k->mnId : m;
        return m;
    }
};

class LoopClosing {
 public:
    typedef std::map<KeyFrame *, g2o::Sim3, std::less<KeyFrame *>, Eigen::aligned_allocator<std::pair<KeyFrame *const, g2o::Sim3>>> KeyFrameAndPose;
};

}  // namespace SIVO
