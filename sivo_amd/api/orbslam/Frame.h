// SIVO::Frame — the perception part of the reference's stereo Frame (reference include/orbslam/Frame.h,
// src/orbslam/Frame.cc:85-260, 326-404, 444-629): segmentation, the two ORB extractions, SelectSemanticKeys,
// ComputeStereoMatches, the feature grid and GetFeaturesInArea.  Map points, poses, BoW and everything else the
// SLAM back end hangs on a Frame stay with the caller (out of scope).
//
// Same results as the reference's sequence (SegmentImage; then two extractor threads; SelectSemanticKeys;
// ComputeStereoMatches on the semantic keys), different schedule: the extractors and the stereo matching of ALL left
// keypoints run on their own threads / HIP streams while the network computes the class map; only the median cull of
// ComputeStereoMatches waits for SelectSemanticKeys (sivo_stereo_match_begin / _cull, bit-identical by construction).
#ifndef FRAME_H
#define FRAME_H

#include <vector>

#include "../bayesian_segnet/bayesian_segnet.hpp"
#include "ORBextractor.h"

namespace SIVO {

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

class Frame {
 public:
    // Frame.cc:85-181.  K is given as fx, fy, cx, cy (KITTI is rectified: distCoef = 0, image bounds = image).
    Frame(const cv::Mat &imLeftGrey, const cv::Mat &imLeftColour, const cv::Mat &imRight, const double &timeStamp,
          ORBextractor *pORBextractorLeft, ORBextractor *pORBextractorRight, BayesianSegNet *pBayesianSegNet, float fx,
          float fy, float cx, float cy, const float &bf, const float &thDepth);

    void ExtractORB(int flag, const cv::Mat &im);          // Frame.cc:215-220
    void SegmentImage(const cv::Mat &im);                  // Frame.cc:222-236
    void SelectSemanticKeys();                             // Frame.cc:177-203
    void ComputeStereoMatches();                           // Frame.cc:444-629 (one-call form, used when constructed stepwise)
    void AssignFeaturesToGrid();                           // Frame.cc:205-221
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);                                   // Frame.cc:392-404
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                          const int maxLevel = -1) const;                           // Frame.cc:326-390
    // Camera-frame back-projection of semantic key i (Frame.cc:631-645 without the pose): false when it has no depth.
    bool UnprojectStereoCamera(const unsigned long &i, float xyz[3]) const;

    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    BayesianSegNet *mpBayesianSegNet;
    double mTimeStamp;
    float fx, fy, cx, cy, invfx, invfy, mbf, mb, mThDepth;

    int numSemanticKeys = 0;
    std::vector<cv::KeyPoint> mvKeysLeft, mvKeysRight, mvKeysSemantic;
    cv::Mat mDescriptorsLeft, mDescriptorsRight, mDescriptorsSemantic;
    std::vector<float> mvRight, mvDepth;      // per semantic key: right u coordinate / depth, -1 when unmatched
    MatXu mClasses;
    MatXd mConfidence, mEntropy;
    cv::Mat mImSemantic;

    int mnScaleLevels;
    float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];

 private:
    std::vector<uint8_t> mKeepLeft;          // SelectSemanticKeys decision per left keypoint
};

}  // namespace SIVO
#endif
