// What Frame.cc needs of ORBmatcher: the two thresholds (ORBmatcher.cc:37-38) and DescriptorDistance — here the oracle's
// orc_descriptor_distance, which tests/cpp/pin_matcher.cpp pins against the reference's own (ORBmatcher.cc:1579-1596).
#pragma once
#include <opencv2/core/core.hpp>
extern "C" int orc_descriptor_distance(const unsigned char *a, const unsigned char *b);
namespace SIVO {
class ORBmatcher {
 public:
    static const int TH_LOW = 50;
    static const int TH_HIGH = 100;
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orc_descriptor_distance(a.ptr(0), b.ptr(0)); }
};
}  // namespace SIVO
