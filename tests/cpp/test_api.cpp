// test_api.cpp — exercises the SIVO:: classes (sivo_amd/api) the way the reference's own
// tests/test_bayesian_segnet.cpp does: InitializationTest (:138-150, exception convention),
// SegmentationTest (:152-168, output sizes), plus the ORB / matcher / optimizer classes.
//   test_api cpu                                 host-only checks (no GPU needed)
//   test_api gpu <prototxt> <weights.sivow> <frame.bin> <outdir>
//        frame.bin = int32 rows, int32 cols, then rows*cols*3 BGR bytes; writes the outputs for
//        tests/test_gpu_cpp_api.py to compare with the Python binding (same library, bit-identical).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bayesian_segnet/bayesian_segnet.hpp"
#include "orbslam/ORBextractor.h"
#include "orbslam/ORBmatcher.h"
#include "orbslam/Optimizer.h"
#include "orbslam/Frame.h"
#include "kitti_io.hpp"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

template <class F>
static bool throws_invalid_argument(F &&f) {
    try { f(); } catch (const std::invalid_argument &) { return true; } catch (...) { return false; }
    return false;
}

static void write_file(const std::string &path, const void *p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char *>(p), (std::streamsize)n);
}

static int run_cpu() {
    // InitializationTest: empty model / weights path -> std::invalid_argument
    CHECK(throws_invalid_argument([] { SIVO::BayesianSegNet s(SIVO::BayesianSegNetParams("", "weights")); }));
    CHECK(throws_invalid_argument([] { SIVO::BayesianSegNet s(SIVO::BayesianSegNetParams("model", "")); }));
    CHECK(SIVO::computeEntropy(0.0) == 0.0 && SIVO::computeEntropy(0.5) == 0.5);
    CHECK(SIVO::ORBmatcher::TH_LOW == 50 && SIVO::ORBmatcher::TH_HIGH == 100 && SIVO::ORBmatcher::HISTO_LENGTH == 30);
    cv::Mat a = cv::Mat::zeros(1, 32, CV_8UC1), b = cv::Mat::zeros(1, 32, CV_8UC1);
    CHECK(SIVO::ORBmatcher::DescriptorDistance(a, b) == 0);
    for (int i = 0; i < 32; ++i) b.data[i] = 0xff;
    CHECK(SIVO::ORBmatcher::DescriptorDistance(a, b) == 256);
    b.data[0] = 0x0f;
    CHECK(SIVO::ORBmatcher::DescriptorDistance(a, b) == 252);
    SIVO::ORBmatcher m;
    std::vector<int> hist[30];
    hist[3].assign(10, 0); hist[7].assign(5, 0); hist[9].assign(20, 0); hist[11].assign(1, 0);
    int i1 = -1, i2 = -1, i3 = -1;
    m.ComputeThreeMaxima(hist, 30, i1, i2, i3);
    CHECK(i1 == 9 && i2 == 3 && i3 == 7);
    hist[3].clear(); hist[7].clear();
    i1 = i2 = i3 = -1;
    m.ComputeThreeMaxima(hist, 30, i1, i2, i3);
    CHECK(i1 == 9 && i2 == -1 && i3 == -1);                 // max2 < 0.1 * max1
    std::printf(failures ? "cpu checks FAILED\n" : "cpu checks ok\n");
    return failures;
}

static int run_gpu(int argc, char **argv) {
    if (argc < 6) { std::printf("usage: test_api gpu prototxt weights frame.bin outdir\n"); return 2; }
    const std::string proto = argv[2], weights = argv[3], frame = argv[4], out = argv[5];
    std::ifstream f(frame, std::ios::binary);
    int32_t rows = 0, cols = 0;
    f.read(reinterpret_cast<char *>(&rows), 4); f.read(reinterpret_cast<char *>(&cols), 4);
    cv::Mat bgr(rows, cols, CV_8UC3);
    f.read(reinterpret_cast<char *>(bgr.data), (std::streamsize)rows * cols * 3);

    // SegmentationTest (tests/test_bayesian_segnet.cpp:152-168)
    SIVO::BayesianSegNetParams params(proto, weights);
    params.seed = 7;
    SIVO::BayesianSegNet segnet(params);
    const cv::Size g = segnet.getInputGeometry();
    SIVO::MatXu classes; SIVO::MatXd confidence, entropy;
    segnet.segmentImage(bgr, classes, confidence, entropy);
    const int output_size = g.height * g.width;
    CHECK(classes.size() == output_size && confidence.size() == output_size && entropy.size() == output_size);
    cv::Mat conf_img = segnet.generateConfidenceImage(confidence);
    cv::Mat ent_img = segnet.generateEntropyImage(entropy);
    cv::Mat seg_img = segnet.generateSegmentedImage(classes, bgr);
    CHECK(conf_img.rows == g.height && ent_img.cols == g.width && seg_img.type() == CV_8UC3);
    CHECK(entropy.minCoeff() >= 0.0 && entropy.maxCoeff() <= 3.91);
    write_file(out + "/classes.bin", classes.data(), (size_t)output_size);
    write_file(out + "/confidence.bin", confidence.data(), (size_t)output_size * 8);
    write_file(out + "/entropy.bin", entropy.data(), (size_t)output_size * 8);
    // smaller than the network: the reference yields an empty Mat; here a std::runtime_error
    bool threw = false;
    try { cv::Mat tiny(8, 8, CV_8UC3); segnet.segmentImage(tiny, classes, confidence, entropy); } catch (const std::exception &) { threw = true; }
    CHECK(threw);

    // ORB: two extractors on two threads (Frame.cc:126-129), gray = the blue plane (any 8UC1 image will do)
    cv::Mat gray(rows, cols, CV_8UC1);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) gray.at<unsigned char>(r, c) = bgr.ptr(r)[3 * c];
    SIVO::ORBextractor left(2000, 1.2f, 8, 20, 7), right(2000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kl, kr; cv::Mat dl, dr, nomask;
    std::thread t1([&] { left(gray, nomask, kl, dl); });
    std::thread t2([&] { right(gray, nomask, kr, dr); });
    t1.join(); t2.join();
    CHECK(kl.size() == kr.size() && kl.size() > 100 && dl.rows == (int)kl.size() && dl.cols == 32);
    CHECK(std::memcmp(dl.data, dr.data, (size_t)dl.rows * 32) == 0);
    CHECK(left.GetLevels() == 8 && left.mvImagePyramid.size() == 8 && left.mvImagePyramid[0].rows == rows);
    CHECK(std::memcmp(left.mvImagePyramid[0].ptr(5), gray.ptr(5), (size_t)cols) == 0);
    CHECK(left.GetScaleFactors()[1] == 1.2f);
    write_file(out + "/kps.bin", kl.data(), kl.size() * sizeof(cv::KeyPoint));
    write_file(out + "/desc.bin", dl.data, (size_t)dl.rows * 32);
    std::vector<cv::KeyPoint> none; cv::Mat nd, empty;
    left(empty, nomask, none, nd);                         // empty image: returns silently
    CHECK(none.empty());

    // Frame (Frame.cc:85-181): grey left, right = left shifted by 8 px (disparity 8), the network on the colour frame.
    // Results go to files; the Python test rebuilds the same frame through the Python binding and compares bit for bit.
    {
        cv::Mat grayR(rows, cols, CV_8UC1);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) grayR.at<unsigned char>(r, c) = gray.at<unsigned char>(r, c + 8 < cols ? c + 8 : cols - 1);
        // the network of this test is smaller than the frame: segmentImage centre-crops, so the Frame is built on the crop
        const int y0 = (rows - g.height) / 2, x0 = (cols - g.width) / 2;
        cv::Mat cl(g.height, g.width, CV_8UC1), cr(g.height, g.width, CV_8UC1), cc(g.height, g.width, CV_8UC3);
        for (int r = 0; r < g.height; ++r) {
            std::memcpy(cl.ptr(r), gray.ptr(r + y0) + x0, (size_t)g.width);
            std::memcpy(cr.ptr(r), grayR.ptr(r + y0) + x0, (size_t)g.width);
            std::memcpy(cc.ptr(r), bgr.ptr(r + y0) + 3 * x0, (size_t)g.width * 3);
        }
        SIVO::ORBextractor fl(500, 1.2f, 1, 20, 7), fr(500, 1.2f, 1, 20, 7);     // one level: the 64-row crop holds no second one
        {   // errors inside the worker threads come back as exceptions of the constructor
            SIVO::ORBextractor bl(500, 1.2f, 4, 20, 7), br(500, 1.2f, 4, 20, 7);
            SIVO::BayesianSegNetParams p3(proto, weights);
            SIVO::BayesianSegNet seg3(p3);
            bool threw2 = false;
            try { SIVO::Frame bad(cl, cc, cr, 0.0, &bl, &br, &seg3, 718.856f, 718.856f, 64.f, 32.f, 386.1448f, 40.f); } catch (const std::runtime_error &) { threw2 = true; }
            CHECK(threw2);
        }
        SIVO::BayesianSegNetParams p2(proto, weights);
        p2.seed = 7;
        SIVO::BayesianSegNet seg2(p2);
        SIVO::Frame F(cl, cc, cr, 0.0, &fl, &fr, &seg2, 718.856f, 718.856f, 0.5f * g.width, 0.5f * g.height, 386.1448f, 40.f);
        CHECK(F.numSemanticKeys == (int)F.mvKeysSemantic.size() && F.mvRight.size() == F.mvKeysSemantic.size());
        CHECK(F.mDescriptorsSemantic.rows == F.numSemanticKeys && F.mvDepth.size() == F.mvRight.size());
        size_t in_grid = 0;
        for (int i = 0; i < FRAME_GRID_COLS; ++i) for (int j = 0; j < FRAME_GRID_ROWS; ++j) in_grid += F.mGrid[i][j].size();
        CHECK((int)in_grid == F.numSemanticKeys);
        if (F.numSemanticKeys > 0) {
            const cv::KeyPoint &k0 = F.mvKeysSemantic[0];
            const std::vector<size_t> near = F.GetFeaturesInArea(k0.pt.x, k0.pt.y, 5.f);
            bool found = false;
            for (size_t i : near) found |= (i == 0);
            CHECK(found);
            // one-call ComputeStereoMatches on the semantic keys == what the constructor produced
            std::vector<float> r0 = F.mvRight, d0 = F.mvDepth;
            F.ComputeStereoMatches();
            CHECK(r0.size() == F.mvRight.size() && std::memcmp(r0.data(), F.mvRight.data(), r0.size() * 4) == 0);
            CHECK(std::memcmp(d0.data(), F.mvDepth.data(), d0.size() * 4) == 0);
            float xyz[3];
            for (size_t i = 0; i < F.mvDepth.size(); ++i)
                if (F.mvDepth[i] > 0) { CHECK(F.UnprojectStereoCamera(i, xyz) && xyz[2] == F.mvDepth[i]); break; }
        }
        write_file(out + "/frame_keys.bin", F.mvKeysSemantic.data(), F.mvKeysSemantic.size() * sizeof(cv::KeyPoint));
        write_file(out + "/frame_right.bin", F.mvRight.data(), F.mvRight.size() * 4);
        write_file(out + "/frame_depth.bin", F.mvDepth.data(), F.mvDepth.size() * 4);
        write_file(out + "/frame_classes.bin", F.mClasses.data(), (size_t)g.height * g.width);
    }

    // matcher: every descriptor against itself + neighbours -> best = itself at distance 0
    SIVO::ORBmatcher matcher(0.9f, true);
    const int n = dl.rows;
    std::vector<int32_t> off(n + 1), idx;
    std::vector<float> ang(n);
    for (int i = 0; i < n; ++i) {
        off[i] = (int32_t)idx.size();
        for (int d = -2; d <= 2; ++d) if (i + d >= 0 && i + d < n) idx.push_back(i + d);
        ang[i] = kl[i].angle;
    }
    off[n] = (int32_t)idx.size();
    std::vector<int> matches;
    const int nm = matcher.MatchCandidates(dl, ang, dr, ang, off, idx, SIVO::ORBmatcher::TH_LOW, true, matches);
    int self = 0;
    for (int i = 0; i < n; ++i) self += matches[i] == i;
    CHECK(nm > n * 8 / 10 && self == nm);

    // optimizer: one pose at identity, points in front, perfect observations -> zero error
    std::vector<double> poses = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, points;
    std::vector<SivoEdge> edges;
    const double intr[5] = {718.856, 718.856, 498.692, 173.215, 386.1448};
    for (int i = 0; i < 100; ++i) {
        const double X = -5 + 0.1 * i, Y = 0.5 + 0.3 * ((i * 7) % 11 - 5), Z = 10 + i;   // not collinear
        points.insert(points.end(), {X, Y, Z});
        SivoEdge e{};
        e.pose = 0; e.point = i; e.stereo = i & 1; e.inv_sigma2 = 1.0;
        e.obs[0] = intr[0] * X / Z + intr[2]; e.obs[1] = intr[1] * Y / Z + intr[3]; e.obs[2] = e.obs[0] - intr[4] / Z;
        if (i == 50) e.obs[0] += 10;                       // one outlier
        edges.push_back(e);
    }
    SIVO::EdgeBatchResult lin;
    SIVO::Optimizer::LinearizeEdges(poses, points, edges, intr, lin);
    std::vector<uint8_t> outlier;
    CHECK(SIVO::Optimizer::ClassifyOutliers(edges, lin, outlier) == 1 && outlier[50] == 1);
    CHECK(lin.chi2[0] < 1e-20 && lin.weight[50] < 1.0 && lin.Jpose[3] == -1.0 / 10 * intr[0]);

    // PoseOptimization: start 5 cm / 0.3 deg off, all-stereo edges -> back at the identity, the planted outlier flagged
    for (auto &e : edges) e.stereo = 1;
    double pose[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.05, -0.02, 0.03}, cov[36];
    {
        const double a = 0.005;
        pose[0] = std::cos(a); pose[2] = std::sin(a); pose[6] = -std::sin(a); pose[8] = std::cos(a);
    }
    bool covValid = false;
    const int inliers = SIVO::Optimizer::PoseOptimization(pose, points, edges, intr, outlier, cov, &covValid);
    CHECK(inliers == 99 && outlier[50] == 1 && covValid && cov[0] > 0 && cov[35] > 0);
    CHECK(std::fabs(pose[9]) < 2e-3 && std::fabs(pose[10]) < 2e-3 && std::fabs(pose[11]) < 2e-3 && std::fabs(pose[2]) < 1e-4);

    // LocalBundleAdjustment: two keyframes (first fixed), perturbed points, noise-free observations
    std::vector<double> kfs = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, -0.5, 0, -1.0};
    std::vector<SivoEdge> ba;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 100; ++i) {
            const double X = points[3 * i] + kfs[12 * k + 9], Y = points[3 * i + 1] + kfs[12 * k + 10], Z = points[3 * i + 2] + kfs[12 * k + 11];
            SivoEdge e{};
            e.pose = k; e.point = i; e.stereo = 1; e.inv_sigma2 = 1.0;
            e.obs[0] = intr[0] * X / Z + intr[2]; e.obs[1] = intr[1] * Y / Z + intr[3]; e.obs[2] = e.obs[0] - intr[4] / Z;
            ba.push_back(e);
        }
    std::vector<double> kfs0 = kfs, pts0 = points;
    kfs[12 + 9] += 0.03; kfs[12 + 11] -= 0.02;
    for (int i = 0; i < 100; ++i) pts0[3 * i + 2] += (i % 2 ? 0.05 : -0.05);
    std::vector<uint8_t> erase;
    bool stop = false;
    SIVO::Optimizer::LocalBundleAdjustment(kfs, {1, 0}, pts0, ba, intr, &stop, erase, 1, cov, &covValid);
    int nErase = 0;
    for (uint8_t b : erase) nErase += b;
    CHECK(nErase == 0 && covValid && std::fabs(kfs[12 + 9] - kfs0[12 + 9]) < 5e-3 && std::fabs(kfs[12 + 11] - kfs0[12 + 11]) < 5e-3);
    CHECK(kfs[9] == 0.0 && kfs[0] == 1.0);                                          // fixed keyframe untouched
    stop = true;
    std::vector<double> kfs1 = kfs;
    SIVO::Optimizer::LocalBundleAdjustment(kfs1, {1, 0}, pts0, ba, intr, &stop, erase);
    CHECK(kfs1 == kfs);                                                              // pbStopFlag honoured
    SIVO::Optimizer::BundleAdjustment(kfs1, {1, 0}, pts0, ba, intr, 3, nullptr, false);
    std::printf(failures ? "gpu checks FAILED\n" : "gpu checks ok\n");
    return failures;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "cpu") return run_cpu();
    if (argc >= 2 && std::string(argv[1]) == "gpu") return run_gpu(argc, argv);
    if (argc >= 5 && std::string(argv[1]) == "kitti") {          // kitti <sequence dir> <Tcw.bin> <out.txt>
        std::vector<std::string> l, r;
        std::vector<double> t;
        SIVO::loadImages(argv[2], l, r, t);
        std::printf("%zu %s %s %g\n", t.size(), l.back().c_str(), r.back().c_str(), t.back());
        std::vector<float> T;
        if (FILE *f = std::fopen(argv[3], "rb")) {
            float v;
            while (std::fread(&v, 4, 1, f) == 1) T.push_back(v);
            std::fclose(f);
        }
        return SIVO::saveTrajectoryKITTI(argv[4], T) ? 0 : 1;
    }
    std::printf("usage: test_api cpu | gpu ...\n");
    return 2;
}
