// Declaration under which the reference's src/orbslam/ORBmatcher.cc is compiled for oracle/_ref/: the public and
// protected members that file defines (reference include/orbslam/ORBmatcher.h:36-142), over the stand-in SLAM types
// of slam_standins.h instead of the reference's Frame.h / KeyFrame.h / MapPoint.h (which need OpenCV, DBoW2's
// vocabulary, g2o and Caffe).  The build defines ORBmatcher=RefORBmatcher so that the class can live in one test
// binary with this repository's SIVO::ORBmatcher.  Test infrastructure only.
#ifndef PIN_REFERENCE_ORBMATCHER_DECL_H
#define PIN_REFERENCE_ORBMATCHER_DECL_H

#include <set>
#include <utility>
#include <vector>

#include "slam_standins.h"

namespace SIVO {

class ORBmatcher {
 public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               const bool bOnlyStereo);
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                     const cv::Mat &t12, const float th);
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

 protected:
    bool CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF);
    float RadiusByViewingCos(const float &viewCos);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace SIVO
#endif
