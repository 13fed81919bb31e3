// Stand-in for <opencv2/core/core.hpp> when the reference's own sources are compiled for oracle/_ref (OpenCV is not
// installed here): the cv::Mat / KeyPoint / Point subset of sivo_amd/api/compat/cv_min.hpp, whose CV_32F algebra restates
// the rounding of OpenCV 3.x's small-matrix gemm, convertTo, dot and norm.  Test infrastructure only.
#pragma once
#include <cassert>   // the real header brings it in transitively; ORBmatcher.cc relies on that
#include "../../../../sivo_amd/api/compat/cv_min.hpp"
typedef unsigned char uchar;   // OpenCV declares it at global scope (cvdef.h)
