// SIVO::ORBmatcher — the data-parallel core of the reference class (reference include/orbslam/ORBmatcher.h:36-142).
//
// The reference's Search* / Fuse members walk the SLAM object graph (Frame, KeyFrame, MapPoint), which is
// outside this library's scope ("pointer chasing + Hamming").  What every one of
// them does with the candidates it gathered — brute-force Hamming argmin with best / second best, the
// TH_LOW / TH_HIGH thresholds, the nearest-neighbour ratio test and the 30-bin rotation-consistency
// histogram — is provided here on plain arrays, with the same constants and the same tie rules, so each
// reference routine becomes: gather candidates on the host -> MatchCandidates() -> apply the result.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#ifdef SIVO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "../compat/cv_min.hpp"
#endif

#include <cstdint>
#include <vector>

namespace SIVO {

class ORBmatcher {
 public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (1 x 32 CV_8U rows).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // For query i (row i of `queries`, N x 32 CV_8U) the candidates are rows
    // candIdx[candOff[i] .. candOff[i+1]) of `train`.  Returns per query the best row (or -1), best and
    // second-best distance (256 when absent).  Runs on the GPU (sivo_hamming_argmin2).
    void BestTwo(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> &candOff,
                 const std::vector<int32_t> &candIdx, std::vector<int> &bestIdx, std::vector<int> &bestDist,
                 std::vector<int> &secondDist) const;

    // The acceptance logic shared by SearchByProjection(Frame&, vector<MapPoint*>&) (ORBmatcher.cc:105-121):
    // accept query i iff bestDist <= thDist and (no second || bestDist <= mfNNratio * secondDist);
    // then, if mbCheckOrientation, keep only matches whose rotation bin (query angle - train angle)
    // is one of the three most populated of HISTO_LENGTH bins (ComputeThreeMaxima, :1545-1577).
    // matches[i] = train row or -1.  Returns the number of matches.
    int MatchCandidates(const cv::Mat &queries, const std::vector<float> &queryAngles, const cv::Mat &train,
                        const std::vector<float> &trainAngles, const std::vector<int32_t> &candOff,
                        const std::vector<int32_t> &candIdx, int thDist, bool useRatio, std::vector<int> &matches) const;

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) const;

 protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace SIVO
#endif
