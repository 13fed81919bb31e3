// prototxt.hpp — the subset of Caffe's text format the two SIVO nets use
// (reference config/bayesian_segnet/{basic,standard}/kitti/*.prototxt).
#pragma once
#include <string>
#include <vector>

namespace sivo {

struct ProtoLayer {
    std::string name, type;
    std::vector<std::string> bottom, top;
    // Convolution / Pooling
    int num_output = 0, pad = 0, kernel_size = 0, stride = 1;
    std::string pool = "MAX";
    // Upsample
    int scale = 2;
    // Dropout
    float dropout_ratio = 0.5f;
    bool sample_weights_test = false;
    // LRN
    int local_size = 5;
    float alpha = 1.f, beta = 0.75f;
    // BN
    std::string bn_mode = "LEARN";
};

struct ProtoNet {
    std::string name, input = "data";
    int shape[4] = {0, 0, 0, 0};  // T, C, H, W (T = 0 when the file leaves it blank)
    std::vector<ProtoLayer> layers;
};

// Throws std::invalid_argument on malformed text.
ProtoNet parse_prototxt(const std::string &text);

}  // namespace sivo
