// Stand-in for <caffe/caffe.hpp> under which the reference's src/bayesian_segnet/bayesian_segnet.cpp compiles untouched
// (caffe-segnet-cudnn7 is an empty submodule and cannot be built here).  The "network" has the two blobs the class
// touches, `data` (T, 3, H, W) and `prob` (T, classes, H, W); Forward() calls a hook the test installs (it fills `prob`).
// So what oracle/_ref/libref_segnet.so pins is everything of BayesianSegNet AROUND the forward pass: constructor checks,
// wrapInputLayer / resizeImage / preprocessImage, extractMeanConfidence, computeClasses, computeMaxConfidence,
// computeClassificationEntropy, computeVariance.  Test infrastructure only.
#pragma once
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}

namespace caffe {

enum Phase { TRAIN, TEST };

class Caffe {
 public:
    enum Brew { CPU, GPU };
    static void set_mode(Brew) {}
};

template <class T>
class Blob {
 public:
    std::vector<int> shape_;
    std::vector<T> data_;
    int shape(int i) const { return shape_[(size_t)i]; }
    T *mutable_cpu_data() { return data_.data(); }
};

struct StandIn {                    // configured by the wrapper before a BayesianSegNet is constructed
    int T = 0, C = 3, H = 0, W = 0, classes = 0;
    void (*forward)(const float *data, float *prob, void *user) = nullptr;
    void *user = nullptr;
};
inline StandIn &standin() { static StandIn s; return s; }

template <class T>
class Net {
 public:
    Net(const std::string &, Phase) {
        const StandIn &s = standin();
        data_.reset(new Blob<T>); prob_.reset(new Blob<T>);
        data_->shape_ = {s.T, s.C, s.H, s.W}; data_->data_.assign((size_t)s.T * s.C * s.H * s.W, T(0));
        prob_->shape_ = {s.T, s.classes, s.H, s.W}; prob_->data_.assign((size_t)s.T * s.classes * s.H * s.W, T(0));
    }
    void CopyTrainedLayersFrom(const std::string &) {}
    boost::shared_ptr<Blob<T> > blob_by_name(const std::string &name) {
        if (name == "data") return data_;
        if (name == "prob") return prob_;
        throw std::runtime_error("stand-in network has no blob " + name);
    }
    void Forward() {
        const StandIn &s = standin();
        if (s.forward) s.forward(data_->data_.data(), prob_->data_.data(), s.user);
    }

 private:
    boost::shared_ptr<Blob<T> > data_, prob_;
};

}  // namespace caffe
