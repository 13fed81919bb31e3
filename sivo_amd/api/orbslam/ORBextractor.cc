// SIVO::ORBextractor over libsivo_hip (reference src/orbslam/ORBextractor.cc:412-475, 1019-1083).
#include "ORBextractor.h"

#include <cassert>
#include <stdexcept>
#include <string>

#include "../../../include/sivo_hip.h"

namespace SIVO {

static const int EDGE_THRESHOLD = 19;

// The reference blurs the descriptor image with cv::GaussianBlur (ORBextractor.cc:1060-1062), whose 8-bit taps depend on the OpenCV it is
// linked with (sivo_orb_set_gaussian): built against a real OpenCV this class picks the taps of THAT version, so that its descriptors
// are the ones the reference's own ORBextractor produces in the same build; against the compat shim (no OpenCV) the default (0).
#if defined(CV_VERSION_MAJOR) && defined(CV_VERSION_MINOR) && defined(CV_VERSION_REVISION)
#if (CV_VERSION_MAJOR > 4) || (CV_VERSION_MAJOR == 4 && (CV_VERSION_MINOR > 5 || (CV_VERSION_MINOR == 5 && CV_VERSION_REVISION >= 1))) || \
    (CV_VERSION_MAJOR == 3 && CV_VERSION_MINOR == 4 && CV_VERSION_REVISION >= 13)
int ORBextractor::sGaussianVariant = 1;      // getGaussianKernelFixedPoint_ED: 18 34 48 56 48 34 18
#else
int ORBextractor::sGaussianVariant = 0;      // round(g * 256): 18 34 49 55 49 34 18
#endif
#else
int ORBextractor::sGaussianVariant = 0;
#endif

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    if (sivo_orb_create(_nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, 0, &mpHandle) != SIVO_OK)
        throw std::runtime_error(std::string("ORBextractor: ") + sivo_last_error());
    if (sivo_orb_set_gaussian(mpHandle, sGaussianVariant) != SIVO_OK)
        throw std::runtime_error(std::string("ORBextractor: ") + sivo_last_error());
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    std::vector<int32_t> fpl(nlevels);
    sivo_orb_tables(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), fpl.data());
    mnFeaturesPerLevel.assign(fpl.begin(), fpl.end());
    mvImagePyramid.resize(nlevels);
    mvPadded.resize(nlevels);
}

ORBextractor::~ORBextractor() {
    if (mpHandle) sivo_orb_destroy(mpHandle);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;                       // :1023-1024
    const cv::Mat &image = _image;
    assert(image.type() == CV_8UC1);                  // :1027
    static_assert(sizeof(cv::KeyPoint) == sizeof(SivoKeyPoint), "cv::KeyPoint layout");
    const int cap = nfeatures * 2 + 64;
    std::vector<cv::KeyPoint> kps(cap);
    cv::Mat desc(cap, 32, CV_8UC1);
    int n = 0;
    const int rc = sivo_orb_extract(mpHandle, image.data, image.rows, image.cols, (int)image.step,
                                    reinterpret_cast<SivoKeyPoint *>(kps.data()), desc.data, cap, &n);
    if (rc != SIVO_OK) throw std::runtime_error(std::string("ORBextractor: ") + sivo_last_error());
    _keypoints.assign(kps.begin(), kps.begin() + n);
    if (n == 0) {
        _descriptors.release();                       // :1040-1041
    } else {
        _descriptors.create(n, 32, CV_8UC1);
        std::memcpy(_descriptors.data, desc.data, (size_t)n * 32);
    }
    if (mbDownloadPyramid) {
        for (int l = 0; l < nlevels; ++l) {
            int32_t r = 0, c = 0;
            sivo_orb_level(mpHandle, l, nullptr, 0, &r, &c);
            mvPadded[l].create(r + 2 * EDGE_THRESHOLD, c + 2 * EDGE_THRESHOLD, CV_8UC1);
            sivo_orb_level(mpHandle, l, mvPadded[l].data, mvPadded[l].step * (size_t)mvPadded[l].rows, &r, &c);
            mvImagePyramid[l] = cv::Mat(r, c, CV_8UC1, mvPadded[l].ptr(EDGE_THRESHOLD) + EDGE_THRESHOLD, mvPadded[l].step);
        }
    }
}

}  // namespace SIVO
