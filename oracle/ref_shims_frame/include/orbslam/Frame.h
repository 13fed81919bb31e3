// Declaration under which the reference's src/orbslam/Frame.cc is compiled UNTOUCHED for oracle/_ref/libref_frame.so:
// the members that file defines and reads (reference include/orbslam/Frame.h:46-275), with stand-ins for everything its
// own header would drag in and that cannot be built here — Caffe's BayesianSegNet, DBoW2's vocabulary, Eigen, the
// KeyFrame / MapPoint / Map graph.  ORBextractor is the reference's real class (ORBextractor.cc compiled beside it).
// Test infrastructure only (tests/test_pin_frame.py).
#ifndef PIN_REFERENCE_FRAME_DECL_H
#define PIN_REFERENCE_FRAME_DECL_H

#include <opencv2/opencv.hpp>

#include <climits>   // the real headers bring INT_MAX in transitively
#include <cmath>
#include <cstddef>
#include <map>
#include <vector>

#include "../../../../sivo_amd/api/compat/eigen_min.hpp"
#include "dependencies/DBoW2/DBoW2/BowVector.h"
#include "dependencies/DBoW2/DBoW2/FeatureVector.h"
#include "include/orbslam/ORBextractor.h"

// the reference's headers leak this, and Frame.cc relies on it (min / max / vector unqualified)
using namespace std;

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace SIVO {

typedef sivo_compat::RowMatrix<double> MatXd;
typedef sivo_compat::RowMatrix<unsigned char> MatXu;
struct StateCovarianceType {                                     // Eigen::Matrix6d in the reference; Frame.cc copies it and asks isZero()
    double v[36] = {0};
    bool isZero(double prec = 0) const { for (double x : v) if (std::fabs(x) > prec) return false; return true; }
};

// bayesian_segnet.hpp:67-83
enum Classes { ROAD, SIDEWALK, BUILDING, WALL, POLE, TRAFFIC_LIGHT, TRAFFIC_SIGN, VEGETATION, TERRAIN, SKY, PERSON, CAR, COMMERCIAL_VEHICLE, BIKE, VOID = 255 };

// hands Frame::SegmentImage the maps the test prepared
class BayesianSegNet {
 public:
    MatXu classes;
    MatXd confidence, entropy;
    void segmentImage(const cv::Mat &, MatXu &c, MatXd &conf, MatXd &ent) { c = classes; conf = confidence; ent = entropy; }
    cv::Mat generateSegmentedImage(const MatXu &, const cv::Mat &) { return cv::Mat(); }
};

class ORBVocabulary {
 public:
    void transform(const std::vector<cv::Mat> &, DBoW2::BowVector &, DBoW2::FeatureVector &, int) {}
};

class Frame;
class KeyFrame;

class MapPoint {
 public:
    cv::Mat mWorldPos, mNormalVector;
    float mfMinDistance = 0, mfMaxDistance = 0;
    bool mbTrackInView = false;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    int mnTrackScaleLevel = 0;
    cv::Mat GetWorldPos() const { return mWorldPos.clone(); }
    cv::Mat GetNormal() const { return mNormalVector.clone(); }
    float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
    inline int PredictScale(const float &currentDist, Frame *pF);        // MapPoint.cc:439-453
};

class Frame {
 public:
    Frame();
    Frame(const Frame &frame);
    Frame(const cv::Mat &imLeftGrey, const cv::Mat &imLeftColour, const cv::Mat &imRight, const double &timeStamp,
          ORBextractor *pORBextractorLeft, ORBextractor *pORBextractorRight, ORBVocabulary *voc, BayesianSegNet *pBayesianSegNet,
          cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth, const float &thConfidence,
          const float &thEntropyReduction);

    void ExtractORB(int flag, const cv::Mat &im);
    void SegmentImage(const cv::Mat &im);
    cv::Mat getSegmentedImage();
    void ComputeBoW();
    void SetPose(cv::Mat Tcw);
    void SetCovariance(const StateCovarianceType &Sigmacw);
    void UpdatePoseMatrices();
    cv::Mat GetCameraCenter() { return mOw.clone(); }
    cv::Mat GetRotationInverse() { return mRwc.clone(); }
    bool isInFrustum(MapPoint *pMP, float viewingCosLimit);
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                          const int maxLevel = -1) const;
    void ComputeStereoMatches();
    cv::Mat UnprojectStereo(const unsigned long &i);

    ORBVocabulary *mpORBvocabulary;
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    BayesianSegNet *mpBayesianSegNet;
    double mTimeStamp;
    cv::Mat mK;
    static float fx, fy, cx, cy, invfx, invfy;
    cv::Mat mDistCoef, mImSemantic;
    float mbf, mb, mThDepth, mThConfidence, mThEntropyReduction;
    int numSemanticKeys;
    std::vector<cv::KeyPoint> mvKeysLeft, mvKeysSemantic, mvKeysRight;
    std::vector<float> mvRight, mvDepth;
    MatXu mClasses;
    MatXd mConfidence, mEntropy;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptorsLeft, mDescriptorsRight, mDescriptorsSemantic;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    StateCovarianceType mSigmacw;
    static long unsigned int nNextId;
    long unsigned int mnId;
    KeyFrame *mpReferenceKF;
    int mnScaleLevels;
    float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static bool mbInitialComputations;

 private:
    void SelectSemanticKeys();
    void ComputeImageBounds(const cv::Mat &imLeft);
    void AssignFeaturesToGrid();
    cv::Mat mRcw, mRwc, mtcw, mOw;
};

inline int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = ceil(log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}

}  // namespace SIVO
#endif
