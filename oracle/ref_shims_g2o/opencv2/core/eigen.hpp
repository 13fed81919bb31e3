// stand-in for <opencv2/core/eigen.hpp>: cv::eigen2cv as Converter.cc uses it (a double matrix becomes CV_64F)
#pragma once
#include "core.hpp"
#include "../../eigen_g2o_ext.hpp"
namespace cv {
template <class T, int R, int C>
void eigen2cv(const Eigen::Matrix<T, R, C> &src, Mat &dst) {
    dst.create((int)src.rows(), (int)src.cols(), CV_64FC1);
    for (int r = 0; r < dst.rows; ++r)
        for (int c = 0; c < dst.cols; ++c) dst.ptr<double>(r)[c] = (double)src(r, c);
}
}  // namespace cv
