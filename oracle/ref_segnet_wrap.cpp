// C entry points over the reference's OWN SIVO::BayesianSegNet (src/bayesian_segnet/bayesian_segnet.cpp compiled untouched
// into oracle/_ref/libref_segnet.so) with a stand-in for Caffe's network: Forward() copies in the softmax probabilities the
// caller supplies.  Everything around the forward pass is the reference's code: the constructor's checks, wrapInputLayer /
// resizeImage / preprocessImage (what the network would be fed), extractMeanConfidence, computeClasses,
// computeMaxConfidence, computeClassificationEntropy, and the never-called computeVariance.  Eigen's Tensor module and
// OpenCV are stand-ins (ref_shims_segnet/).  Test infrastructure: tests/test_pin_segnet_post.py.
#include <cstdint>
#include <cstring>
#include <stdexcept>

#define private public          // computeVariance & co. are private; the object code is the untouched reference file
#include "bayesian_segnet/bayesian_segnet.hpp"
#undef private

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
struct Ctx { const float *prob; float *data_out; size_t n_prob, n_data; };
void forward_hook(const float *data, float *prob, void *user) {
    Ctx *c = static_cast<Ctx *>(user);
    if (c->data_out) std::memcpy(c->data_out, data, c->n_data * sizeof(float));
    std::memcpy(prob, c->prob, c->n_prob * sizeof(float));
}
}  // namespace

// segmentImage on a (rows x cols) BGR image with the given network output.  data_blob (T x 3 x H x W, may be NULL)
// receives what the input layer held when Forward() was called; variance (H x W, may be NULL) = computeVariance(classes).
// Returns 0, or 1 when the image is smaller than the network (the reference then runs into undefined behaviour: not run).
REF_API int ref_segnet_segment(int T, int classes, int H, int W, const float *prob, const uint8_t *bgr, int rows, int cols,
                               float *data_blob, uint8_t *cls, double *conf, double *ent, double *variance) {
    if (rows < H || cols < W) return 1;
    caffe::StandIn &s = caffe::standin();
    s.T = T; s.C = 3; s.H = H; s.W = W; s.classes = classes;
    Ctx ctx{prob, data_blob, (size_t)T * classes * H * W, (size_t)T * 3 * H * W};
    s.forward = forward_hook; s.user = &ctx;
    SIVO::BayesianSegNetParams params("stand-in.prototxt", "stand-in.caffemodel");
    SIVO::BayesianSegNet net(params);
    cv::Mat image(rows, cols, CV_8UC3, const_cast<uint8_t *>(bgr));
    SIVO::MatXu c; SIVO::MatXd f, e;
    net.segmentImage(image, c, f, e);
    std::memcpy(cls, c.data(), (size_t)H * W);
    std::memcpy(conf, f.data(), (size_t)H * W * 8);
    std::memcpy(ent, e.data(), (size_t)H * W * 8);
    if (variance) {
        const SIVO::MatXd v = net.computeVariance(c);
        std::memcpy(variance, v.data(), (size_t)H * W * 8);
    }
    return 0;
}

// The constructor's error behaviour: 0 = constructed, 1 = std::invalid_argument (message copied out), 2 = anything else.
REF_API int ref_segnet_construct(const char *model, const char *weights, int T, int C, char *what, int what_len) {
    caffe::StandIn &s = caffe::standin();
    s.T = T; s.C = C; s.H = 8; s.W = 8; s.classes = 2; s.forward = nullptr;
    try {
        SIVO::BayesianSegNetParams params(model, weights);
        SIVO::BayesianSegNet net(params);
        return 0;
    } catch (const std::invalid_argument &e) {
        std::strncpy(what, e.what(), (size_t)what_len - 1); what[what_len - 1] = 0;
        return 1;
    } catch (...) {
        return 2;
    }
}

REF_API double ref_compute_entropy(double p) { return SIVO::computeEntropy(p); }
