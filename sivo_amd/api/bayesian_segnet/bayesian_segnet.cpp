// SIVO::BayesianSegNet over libsivo_hip (reference src/bayesian_segnet/bayesian_segnet.cpp).
#include "bayesian_segnet.hpp"

#include <cmath>
#include <stdexcept>

#include "../../../include/sivo_hip.h"

namespace SIVO {

// bayesian_segnet.cpp:38-44
double computeEntropy(const double probability) {
    if (probability == 0) return 0;
    return -1.0 * probability * std::log2(probability);
}

static void throw_status(int rc) {
    const std::string msg = sivo_last_error();
    if (rc == SIVO_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);   // the reference's exception type
    throw std::runtime_error(msg);
}

BayesianSegNet::BayesianSegNet(const BayesianSegNetParams &params) : params(params) {
    this->checkConfig();
    if (!this->params.use_gpu) throw std::runtime_error("use_gpu = false: libsivo_hip has no CPU path");
    const int rc =
        this->params.devices.size() > 1
            ? sivo_segnet_create_multi_from_files_opts(this->params.model_file.c_str(), this->params.weights_file.c_str(),
                                                       this->params.monte_carlo_samples, this->params.devices.data(),
                                                       (int)this->params.devices.size(), &this->params.options, &this->handle)
            : sivo_segnet_create_from_files_opts(this->params.model_file.c_str(), this->params.weights_file.c_str(),
                                                 this->params.monte_carlo_samples,
                                                 this->params.devices.empty() ? this->params.device : this->params.devices[0],
                                                 &this->params.options, &this->handle);
    if (rc != SIVO_OK) throw_status(rc);    // C != 3 / T <= 1 -> std::invalid_argument, as bayesian_segnet.cpp:64-70
    int32_t T, C, H, W, K;
    sivo_segnet_shape(this->handle, &T, &C, &H, &W, &K);
    this->input_geometry = cv::Size{W, H};
    this->generateSegmentationColours();
}

BayesianSegNet::~BayesianSegNet() {
    if (this->handle) sivo_segnet_destroy(this->handle);
}

// bayesian_segnet.cpp:80-89
void BayesianSegNet::checkConfig() {
    if (this->params.model_file.empty()) throw std::invalid_argument("model_file (.prototxt file) is empty!");
    if (this->params.weights_file.empty()) throw std::invalid_argument("weights_file (.caffemodel file) is empty!");
}

// bayesian_segnet.cpp:91-117 (BGR triplets of the Cityscapes-style palette)
void BayesianSegNet::generateSegmentationColours() {
    static const unsigned char lut[14][3] = {{128, 64, 128}, {232, 35, 244}, {69, 69, 69}, {156, 102, 102}, {153, 153, 153},
                                             {30, 170, 250}, {0, 220, 220}, {35, 142, 107}, {152, 251, 152}, {180, 130, 70},
                                             {60, 20, 220}, {142, 0, 0}, {70, 0, 0}, {32, 11, 119}};
    for (int c = 0; c < 14; ++c) this->class_colours.at<cv::Vec3b>(c, 0) = cv::Vec3b(lut[c][0], lut[c][1], lut[c][2]);
    this->class_colours.at<cv::Vec3b>(Classes::VOID, 0) = cv::Vec3b(0, 0, 0);
}

// bayesian_segnet.cpp:142-162: same size -> as is; larger -> centre crop (clone); smaller -> empty Mat
cv::Mat BayesianSegNet::resizeImage(const cv::Mat &image) {
    cv::Mat resized_image;
    if (image.size() == this->input_geometry) return image;
    if (image.rows >= this->input_geometry.height && image.cols >= this->input_geometry.width) {
        const int x_tl = image.cols / 2 - this->input_geometry.width / 2;
        const int y_tl = image.rows / 2 - this->input_geometry.height / 2;
        resized_image.create(this->input_geometry.height, this->input_geometry.width, image.type());
        const size_t es = image.elemSize();
        for (int r = 0; r < resized_image.rows; ++r)
            std::memcpy(resized_image.ptr(r), image.ptr(r + y_tl) + (size_t)x_tl * es, (size_t)resized_image.cols * es);
    }
    return resized_image;
}

// bayesian_segnet.cpp:299-318
void BayesianSegNet::segmentImage(const cv::Mat &image, MatXu &classes, MatXd &confidence, MatXd &entropy) {
    if (image.empty() || image.type() != CV_8UC3) throw std::invalid_argument("segmentImage expects a CV_8UC3 BGR image");
    const int H = this->input_geometry.height, W = this->input_geometry.width;
    classes.resize(H, W);
    confidence.resize(H, W);
    entropy.resize(H, W);
    cv::Mat contiguous = image.isContinuous() ? image : image.clone();
    // every frame draws fresh dropout masks (Caffe's RNG advances between Forward() calls)
    const uint64_t seed = this->params.seed + 0x9E3779B97F4A7C15ull * this->frame_counter++;
    const int rc = sivo_segnet_segment(this->handle, contiguous.data, contiguous.rows, contiguous.cols, seed, classes.data(),
                                       confidence.data(), entropy.data());
    if (rc != SIVO_OK) throw_status(rc);
}

// bayesian_segnet.cpp:320-330 (cv::eigen2cv)
cv::Mat BayesianSegNet::generateConfidenceImage(const MatXd &confidence) {
    cv::Mat img((int)confidence.rows(), (int)confidence.cols(), CV_64FC1);
    std::memcpy(img.data, confidence.data(), sizeof(double) * (size_t)confidence.size());
    return img;
}

static cv::Mat normalise_minmax(const MatXd &m) {   // cv::normalize(.., 0, 1, NORM_MINMAX)
    cv::Mat img((int)m.rows(), (int)m.cols(), CV_64FC1);
    const double lo = m.minCoeff(), hi = m.maxCoeff();
    const double scale = hi > lo ? 1.0 / (hi - lo) : 0.0;
    for (std::ptrdiff_t i = 0; i < m.size(); ++i) img.ptr<double>()[i] = (m.data()[i] - lo) * scale;
    return img;
}
cv::Mat BayesianSegNet::generateVarianceImage(MatXd &variance) { return normalise_minmax(variance); }
cv::Mat BayesianSegNet::generateEntropyImage(MatXd &entropy) { return normalise_minmax(entropy); }

// bayesian_segnet.cpp:362-389: LUT colouring + 0.5 / 0.5 blend with the (cropped) input
cv::Mat BayesianSegNet::generateSegmentedImage(const MatXu &classes, const cv::Mat &test_image) {
    cv::Mat out((int)classes.rows(), (int)classes.cols(), CV_8UC3);
    cv::Mat resized = this->resizeImage(test_image);
    for (int r = 0; r < out.rows; ++r)
        for (int c = 0; c < out.cols; ++c) {
            const cv::Vec3b col = this->class_colours.at<cv::Vec3b>(classes(r, c), 0);
            cv::Vec3b px;
            for (int k = 0; k < 3; ++k) {
                const double v = 0.5 * col[k] + (resized.empty() ? 0.0 : 0.5 * resized.at<cv::Vec3b>(r, c)[k]);
                px[k] = (unsigned char)std::lrint(v > 255 ? 255 : v);    // saturate_cast rounds half to even
            }
            out.at<cv::Vec3b>(r, c) = px;
        }
    return out;
}

}  // namespace SIVO
