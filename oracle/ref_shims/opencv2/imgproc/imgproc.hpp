// Stand-in for <opencv2/imgproc/imgproc.hpp> + the features2d / core functions src/orbslam/ORBextractor.cc calls, for
// compiling that file into oracle/_ref.  OpenCV is a third-party dependency that is not in the reference tree
// (README.md asks for OpenCV > 3.2): each primitive forwards to the restatement of the published OpenCV 3.2-3.4
// algorithm in oracle/orb_oracle.c (header there).  So in oracle/_ref/libref_orb.so the reference's OWN code is real —
// pyramid construction and its in-place border trick, the 30 x 30 cell walk with the two FAST thresholds, the octree
// distribution, IC_Angle, the 256-pair pattern and the steered descriptor, key scaling — and only these primitives are
// restated.  Test infrastructure only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../core/core.hpp"

#ifndef CV_PI
#define CV_PI 3.1415926535897932384626433832795
#endif

extern "C" {
int orc_cvround(double v);
float orc_fast_atan2(float y, float x);
void orc_resize_linear_u8(const uint8_t *src, int sh, int sw, int sstep, uint8_t *dst, int dh, int dw, int dstep);
void orc_border101(uint8_t *img, int rows, int cols, int step, int b);
void orc_gaussian7_u8(const uint8_t *src, int rows, int cols, int sstep, uint8_t *dst, int dstep);
int orc_fast9_16(const uint8_t *img, int rows, int cols, int step, int threshold, int nonmax, int32_t *out_xy, uint8_t *out_score,
                 int max_out);
}

inline int cvRound(double v) { return orc_cvround(v); }
inline int cvRound(float v) { return orc_cvround((double)v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { const int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { const int i = (int)v; return i + (i < v); }

namespace cv {

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };

inline float fastAtan2(float y, float x) { return orc_fast_atan2(y, x); }

// cv::FAST(image, keypoints, threshold, nonmaxSuppression): TYPE_9_16, keys in raster order, size 7, response = score
inline void FAST(const Mat &image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true) {
    keypoints.clear();
    const int cap = image.rows * image.cols;
    if (cap <= 0) return;
    std::vector<int32_t> xy(2 * (size_t)cap);
    std::vector<uint8_t> score((size_t)cap);
    const int n = orc_fast9_16(image.data, image.rows, image.cols, (int)image.step, threshold, nonmaxSuppression ? 1 : 0, xy.data(),
                               score.data(), cap);
    for (int i = 0; i < n; ++i) keypoints.push_back(KeyPoint((float)xy[2 * i], (float)xy[2 * i + 1], 7.f, -1, (float)score[i]));
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) on 8UC1.  dst keeps its storage when it already has dsize (a view
// into a larger buffer stays one: ORBextractor::ComputePyramid resizes straight into the bordered buffer).
inline void resize(const Mat &src, Mat &dst, Size dsize, double, double, int interpolation) {
    if (interpolation != INTER_LINEAR || src.type() != CV_8UC1) std::abort();
    dst.create(dsize.height, dsize.width, src.type());
    orc_resize_linear_u8(src.data, src.rows, src.cols, (int)src.step, dst.data, dst.rows, dst.cols, (int)dst.step);
}

// cv::copyMakeBorder(src, dst, b, b, b, b, BORDER_REFLECT_101 [+ BORDER_ISOLATED]); src may be the interior of dst.
// (Without BORDER_ISOLATED OpenCV would read real pixels around a src that is a view; the only such call in the
// reference, level 0, passes the caller's whole image.)
inline void copyMakeBorder(const Mat &src, Mat &dst, int top, int bottom, int left, int right, int borderType) {
    if ((borderType & ~BORDER_ISOLATED) != BORDER_REFLECT_101 || top != bottom || top != left || top != right || src.type() != CV_8UC1)
        std::abort();
    const Mat keep = src;                                   // src may alias dst's storage: hold it across create()
    dst.create(keep.rows + 2 * top, keep.cols + 2 * top, keep.type());
    uint8_t *interior = dst.data + (size_t)top * dst.step + (size_t)top;
    if (interior != keep.data)
        for (int r = 0; r < keep.rows; ++r) std::memcpy(interior + (size_t)r * dst.step, keep.ptr(r), (size_t)keep.cols);
    orc_border101(interior, keep.rows, keep.cols, (int)dst.step, top);
}

// cv::GaussianBlur(src, dst, Size(7, 7), 2, 2, BORDER_REFLECT_101) on 8UC1, in place allowed
inline void GaussianBlur(const Mat &src, Mat &dst, Size ksize, double sigmaX, double sigmaY, int borderType) {
    if (ksize.width != 7 || ksize.height != 7 || sigmaX != 2 || sigmaY != 2 || borderType != BORDER_REFLECT_101 || src.type() != CV_8UC1)
        std::abort();
    const Mat in = src.clone();
    dst.create(in.rows, in.cols, in.type());
    orc_gaussian7_u8(in.data, in.rows, in.cols, (int)in.step, dst.data, (int)dst.step);
}

// only reached for cameras with lens distortion (Frame::ComputeImageBounds); KITTI's rectified images have none
inline void undistortPoints(const Mat &, Mat &, const Mat &, const Mat &, const Mat &, const Mat &) { std::abort(); }

// cv::KeyPointsFilter::retainBest — only ORBextractor::ComputeKeyPointsOld (never called) uses it
struct KeyPointsFilter {
    static void retainBest(std::vector<KeyPoint> &keypoints, int n_points) {
        if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
            if (n_points == 0) { keypoints.clear(); return; }
            std::nth_element(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(),
                             [](const KeyPoint &a, const KeyPoint &b) { return a.response > b.response; });
            const float ambiguous = keypoints[(size_t)n_points - 1].response;
            const auto end = std::partition(keypoints.begin() + n_points, keypoints.end(),
                                            [ambiguous](const KeyPoint &k) { return k.response >= ambiguous; });
            keypoints.resize((size_t)(end - keypoints.begin()));
        }
    }
};

}  // namespace cv
