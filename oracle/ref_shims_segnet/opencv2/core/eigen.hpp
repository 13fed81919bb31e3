// Stand-in for <opencv2/core/eigen.hpp> plus the few cv functions only bayesian_segnet.cpp uses (split, normalize, cvtColor,
// LUT, addWeighted), next to the cv::Mat subset of oracle/ref_shims.  Test infrastructure only.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../../ref_shims/opencv2/core/core.hpp"
#include "../../Eigen/Eigen"

#ifndef CV_GRAY2BGR
#define CV_GRAY2BGR 8
#endif

namespace cv {

enum { NORM_MINMAX = 32 };

// cv::split(m, mv): every mv[k] keeps its storage when it already has the plane's size and type — which is how
// BayesianSegNet::wrapInputLayer makes split() write straight into the network's input blob
inline void split(const Mat &m, std::vector<Mat> &mv) {
    const int cn = m.channels();
    mv.resize((size_t)cn);
    for (int k = 0; k < cn; ++k) {
        mv[(size_t)k].create(m.rows, m.cols, CV_MAKETYPE(m.depth(), 1));
        for (int r = 0; r < m.rows; ++r) {
            const float *src = m.ptr<float>(r);
            float *dst = mv[(size_t)k].ptr<float>(r);
            for (int c = 0; c < m.cols; ++c) dst[c] = src[c * cn + k];
        }
    }
}

template <class T, int R, int C, int O>
void eigen2cv(const Eigen::Matrix<T, R, C, O> &src, Mat &dst) {
    dst.create((int)src.rows(), (int)src.cols(), sizeof(T) == 8 ? CV_64FC1 : CV_8UC1);
    for (int r = 0; r < dst.rows; ++r)
        for (int c = 0; c < dst.cols; ++c) dst.ptr<T>(r)[c] = src(r, c);
}

// NORM_MINMAX to [a, b] on CV_64F
inline void normalize(const Mat &src, Mat &dst, double a, double b, int, int) {
    double lo = src.at<double>(0, 0), hi = lo;
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c) { lo = std::min(lo, src.at<double>(r, c)); hi = std::max(hi, src.at<double>(r, c)); }
    const double scale = (b - a) * (hi - lo > 2.220446049250313e-16 ? 1. / (hi - lo) : 0), shift = a - lo * scale;
    dst.create(src.rows, src.cols, CV_64FC1);
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c) dst.at<double>(r, c) = src.at<double>(r, c) * scale + shift;
}

inline void cvtColor(const Mat &src, Mat &dst, int) {           // CV_GRAY2BGR
    dst.create(src.rows, src.cols, CV_8UC3);
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c) { const uchar v = src.ptr<uchar>(r)[c]; uchar *d = dst.ptr<uchar>(r) + 3 * c; d[0] = d[1] = d[2] = v; }
}

inline void LUT(const Mat &src, const Mat &lut, Mat &dst) {      // 8UC3 through a 256 x 1 8UC3 table, channel-wise
    dst.create(src.rows, src.cols, CV_8UC3);
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < 3 * src.cols; ++c) dst.ptr<uchar>(r)[c] = lut.ptr<uchar>(src.ptr<uchar>(r)[c])[c % 3];
}

inline void addWeighted(const Mat &a, double alpha, const Mat &b, double beta, double gamma, Mat &dst) {   // 8U, saturate_cast<uchar>(float sum)
    Mat out(a.rows, a.cols, a.type());
    const int n = a.cols * a.channels();
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < n; ++c) {
            const float v = (float)a.ptr<uchar>(r)[c] * (float)alpha + (float)b.ptr<uchar>(r)[c] * (float)beta + (float)gamma;
            const long iv = std::lrint(v);
            out.ptr<uchar>(r)[c] = (uchar)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
        }
    dst = out;
}

}  // namespace cv
