// Stand-in for include/orbslam/Converter.h (the real one needs Eigen and g2o): the one function Frame.cc calls.
#pragma once
#include <opencv2/core/core.hpp>

#include <vector>
namespace SIVO {
class Converter {
 public:
    static std::vector<cv::Mat> toDescriptorVector(const cv::Mat &Descriptors) {       // Converter.cc:31-38
        std::vector<cv::Mat> vDesc;
        vDesc.reserve((size_t)Descriptors.rows);
        for (int j = 0; j < Descriptors.rows; j++) vDesc.push_back(Descriptors.row(j));
        return vDesc;
    }
};
}  // namespace SIVO
