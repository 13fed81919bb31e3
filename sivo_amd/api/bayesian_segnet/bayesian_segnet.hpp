// SIVO::BayesianSegNet with the reference's public interface
// (reference include/bayesian_segnet/bayesian_segnet.hpp:46-170), implemented over the C ABI of
// libsivo_hip.so instead of Caffe.  Drop-in for libbayesian_segnet behind src/sivo.cc / System.cc:94-95.
#ifndef BAYESIAN_SEGNET_BAYESIAN_SEGNET_HPP
#define BAYESIAN_SEGNET_BAYESIAN_SEGNET_HPP

#ifdef SIVO_HAVE_EIGEN
#include <Eigen/Eigen>
#else
#include "../compat/eigen_min.hpp"
#endif
#ifdef SIVO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "../compat/cv_min.hpp"
#endif

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/sivo_hip.h"      // SivoSegnetOptions (plain C: no other dependency)

struct sivo_segnet;

namespace SIVO {

#ifdef SIVO_HAVE_EIGEN
using MatXd = Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
using MatXu = Eigen::Matrix<uint8_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
#else
using MatXd = sivo_compat::RowMatrix<double>;
using MatXu = sivo_compat::RowMatrix<uint8_t>;
#endif

double computeEntropy(const double probability);

/// Classes that can be detected (reference bayesian_segnet.hpp:67-83).
enum Classes {
    ROAD, SIDEWALK, BUILDING, WALL, POLE, TRAFFIC_LIGHT, TRAFFIC_SIGN, VEGETATION, TERRAIN, SKY, PERSON, CAR,
    COMMERCIAL_VEHICLE, BIKE, VOID = 255
};

struct BayesianSegNetParams {
    BayesianSegNetParams(const std::string model_filepath, const std::string weights_filepath)
        : model_file(model_filepath), weights_file(weights_filepath) {}
    /// The reference selects Caffe's CPU or GPU mode with this; this library has no CPU path and
    /// throws std::runtime_error when it is false.
    bool use_gpu = true;
    std::string model_file;    ///< .prototxt
    std::string weights_file;  ///< the trained .caffemodel (read directly) or a .sivow parameter container (sivo_amd/weights.py)
    /// Additions (defaults keep the reference behaviour): MC sample count when the prototxt leaves it
    /// blank, dropout seed (Caffe's RNG is unseeded in the reference), HIP device.
    int monte_carlo_samples = 0;
    uint64_t seed = 0;
    int device = 0;
    /// More than one entry: the T samples of every segmentImage call are spread over these HIP devices inside this one
    /// object (sivo_segnet_create_multi: per-device stream + RCCL communicator, reduce-scatter / all-gather over xGMI).
    /// Empty: the single `device` above.
    std::vector<int> devices;
    /// How the handle runs (SivoSegnetOptions of include/sivo_hip.h; zero = the library's defaults: two lanes, f16x3 arithmetic on the
    /// matrix-core layers, packed activations, 16 GiB F(4x4) workspace).  The library reads no environment variable: what used to be
    /// SIVO_LANES / SIVO_GEMM / ... is set here, next to the file names.
    SivoSegnetOptions options = {sizeof(SivoSegnetOptions), 0, 0, 0, 0, 0, 0, 0};
};

class BayesianSegNet {
 public:
    explicit BayesianSegNet(const BayesianSegNetParams &params);
    ~BayesianSegNet();
    BayesianSegNet(const BayesianSegNet &) = delete;
    BayesianSegNet &operator=(const BayesianSegNet &) = delete;

    void segmentImage(const cv::Mat &image, MatXu &classes, MatXd &confidence, MatXd &entropy);
    cv::Mat generateConfidenceImage(const MatXd &confidence);
    cv::Mat generateVarianceImage(MatXd &variance);
    cv::Mat generateEntropyImage(MatXd &entropy);
    cv::Mat generateSegmentedImage(const MatXu &classes, const cv::Mat &test_image);
    cv::Size getInputGeometry() { return this->input_geometry; }

 private:
    void checkConfig();
    void generateSegmentationColours();
    cv::Mat resizeImage(const cv::Mat &image);

    sivo_segnet *handle = nullptr;
    cv::Size input_geometry;
    cv::Mat class_colours = cv::Mat::zeros(256, 1, CV_8UC3);
    BayesianSegNetParams params;
    uint64_t frame_counter = 0;
};

}  // namespace SIVO
#endif
