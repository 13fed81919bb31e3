// SIVO::ORBextractor with the reference's public interface (reference include/orbslam/ORBextractor.h:46-123)
// over libsivo_hip.  Two instances may be used concurrently from two host threads (Frame.cc:126-129):
// every instance owns its HIP streams and buffers.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#ifdef SIVO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "../compat/cv_min.hpp"
#endif

#include <vector>

struct sivo_orb;

namespace SIVO {

class ORBextractor {
 public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored (as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    double inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Level images of the last extraction (interior views into padded host copies, 19-px reflect-101
    // border around each), refreshed by operator() because Frame::ComputeStereoMatches reads them
    // (Frame.cc:451,546,568).  Set mbDownloadPyramid = false to skip the read-back when the caller
    // matches on the device (sivo_stereo_match).
    std::vector<cv::Mat> mvImagePyramid;
    bool mbDownloadPyramid = true;

    sivo_orb *handle() { return mpHandle; }   // for the device-side stereo matcher

    // Which OpenCV's GaussianBlur 8U taps new extractors blur the descriptor image with (sivo_orb_set_gaussian: 0 = OpenCV 3.2 - 3.4.12 /
    // 4.0 - 4.5.0, 1 = OpenCV >= 3.4.13 / >= 4.5.1).  Initialised from the OpenCV version this file is compiled against.
    static int sGaussianVariant;

 protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<cv::Mat> mvPadded;
    sivo_orb *mpHandle = nullptr;
};

}  // namespace SIVO
#endif
