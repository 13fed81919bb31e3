// SIVO::Frame perception part over libsivo_hip (reference src/orbslam/Frame.cc:85-260, 326-404, 444-629).
#include "Frame.h"

#include <cmath>
#include <stdexcept>
#include <string>
#include <thread>

#include "../../../include/sivo_hip.h"

namespace SIVO {

static_assert(sizeof(cv::KeyPoint) == sizeof(SivoKeyPoint), "cv::KeyPoint and SivoKeyPoint must share a layout");

Frame::Frame(const cv::Mat &imLeftGrey, const cv::Mat &imLeftColour, const cv::Mat &imRight, const double &timeStamp,
             ORBextractor *pORBextractorLeft, ORBextractor *pORBextractorRight, BayesianSegNet *pBayesianSegNet, float fx_,
             float fy_, float cx_, float cy_, const float &bf, const float &thDepth)
    : mpORBextractorLeft(pORBextractorLeft), mpORBextractorRight(pORBextractorRight), mpBayesianSegNet(pBayesianSegNet),
      mTimeStamp(timeStamp), fx(fx_), fy(fy_), cx(cx_), cy(cy_), invfx(1.0f / fx_), invfy(1.0f / fy_), mbf(bf), mb(bf / fx_),
      mThDepth(thDepth) {
    // Scale level info (:117-124)
    mnScaleLevels = mpORBextractorLeft->GetLevels();
    mfScaleFactor = static_cast<float>(mpORBextractorLeft->GetScaleFactor());
    mfLogScaleFactor = std::log(mfScaleFactor);
    mvScaleFactors = mpORBextractorLeft->GetScaleFactors();
    mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();
    mvLevelSigma2 = mpORBextractorLeft->GetScaleSigmaSquares();
    mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
    // ComputeImageBounds with distCoef = 0 (:430-435)
    mnMinX = 0.0f; mnMaxX = (float)imLeftGrey.cols; mnMinY = 0.0f; mnMaxY = (float)imLeftGrey.rows;
    mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (mnMaxX - mnMinX);
    mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (mnMaxY - mnMinY);

    // The reference runs SegmentImage, then the two extractor threads (:126-131).  Here the extractors start first and a
    // third thread matches every left keypoint as soon as both are done, all beside the network.
    std::vector<float> uR, depth;
    std::vector<int32_t> sad;
    int match_rc = SIVO_OK;
    std::string match_err;
    // (no exception may leave a worker thread: errors are carried back and rethrown here)
    std::string orb_err[2];
    auto extract = [&](int flag, const cv::Mat &im) {
        try { ExtractORB(flag, im); } catch (const std::exception &e) { orb_err[flag] = e.what(); if (orb_err[flag].empty()) orb_err[flag] = "error"; }
    };
    std::thread threadLeft(extract, 0, std::cref(imLeftGrey));
    std::thread threadRight(extract, 1, std::cref(imRight));
    std::thread threadMatch([&] {
        threadLeft.join();
        threadRight.join();
        if (!orb_err[0].empty() || !orb_err[1].empty()) return;
        const int nL = (int)mvKeysLeft.size(), nR = (int)mvKeysRight.size();
        uR.assign((size_t)nL, -1.f); depth.assign((size_t)nL, -1.f); sad.assign((size_t)nL, -1);
        if (nL == 0) return;
        match_rc = sivo_stereo_match_begin(mpORBextractorLeft->handle(), mpORBextractorRight->handle(),
                                           reinterpret_cast<const SivoKeyPoint *>(mvKeysLeft.data()), mDescriptorsLeft.data, nL,
                                           reinterpret_cast<const SivoKeyPoint *>(mvKeysRight.data()), mDescriptorsRight.data, nR, mbf,
                                           mb, uR.data(), depth.data(), nullptr, sad.data());
        if (match_rc != SIVO_OK) match_err = sivo_last_error();
    });
    try {
        SegmentImage(imLeftColour);
    } catch (...) {
        threadMatch.join();
        throw;
    }
    threadMatch.join();
    for (int k = 0; k < 2; ++k)
        if (!orb_err[k].empty()) throw std::runtime_error(std::string("Frame: ORB extraction (") + (k ? "right" : "left") + ") failed: " + orb_err[k]);
    if (match_rc != SIVO_OK) throw std::runtime_error("Frame: stereo matching failed: " + match_err);

    if (mvKeysLeft.empty()) return;                      // :133-141
    SelectSemanticKeys();
    if (mvKeysSemantic.empty()) return;
    numSemanticKeys = static_cast<int>(mvKeysSemantic.size());

    // ComputeStereoMatches on the semantic keys = median cull over the kept keypoints of the all-keypoint matching
    const int rc = sivo_stereo_match_cull((int)mKeepLeft.size(), mKeepLeft.data(), sad.data(), uR.data(), depth.data());
    if (rc != SIVO_OK) throw std::runtime_error(std::string("Frame: ") + sivo_last_error());
    mvRight.clear(); mvDepth.clear();
    mvRight.reserve((size_t)numSemanticKeys); mvDepth.reserve((size_t)numSemanticKeys);
    for (size_t i = 0; i < mKeepLeft.size(); ++i)
        if (mKeepLeft[i]) { mvRight.push_back(uR[i]); mvDepth.push_back(depth[i]); }

    AssignFeaturesToGrid();
}

void Frame::ExtractORB(int flag, const cv::Mat &im) {
    if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeysLeft, mDescriptorsLeft);
    else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
}

void Frame::SegmentImage(const cv::Mat &im) {
    mpBayesianSegNet->segmentImage(im, mClasses, mConfidence, mEntropy);
    mImSemantic = mpBayesianSegNet->generateSegmentedImage(mClasses, im);
}

void Frame::SelectSemanticKeys() {
    mKeepLeft.assign(mvKeysLeft.size(), 0);
    mvKeysSemantic.clear();
    size_t n = 0;
    for (size_t i = 0; i < mvKeysLeft.size(); ++i) {
        const int col = static_cast<int>(mvKeysLeft[i].pt.x), row = static_cast<int>(mvKeysLeft[i].pt.y);
        if (static_cast<Classes>(mClasses(row, col)) <= Classes::TERRAIN) { mKeepLeft[i] = 1; ++n; }   // static classes (bayesian_segnet.hpp Classes)
    }
    mvKeysSemantic.reserve(n);
    mDescriptorsSemantic.create((int)n, 32, CV_8UC1);
    size_t k = 0;
    for (size_t i = 0; i < mvKeysLeft.size(); ++i)
        if (mKeepLeft[i]) {
            mvKeysSemantic.push_back(mvKeysLeft[i]);
            std::memcpy(mDescriptorsSemantic.ptr((int)k++), mDescriptorsLeft.ptr((int)i), 32);
        }
}

void Frame::ComputeStereoMatches() {
    const int nL = (int)mvKeysSemantic.size();
    mvRight.assign((size_t)nL, -1.f); mvDepth.assign((size_t)nL, -1.f);
    if (nL == 0) return;
    const int rc = sivo_stereo_match(mpORBextractorLeft->handle(), mpORBextractorRight->handle(),
                                     reinterpret_cast<const SivoKeyPoint *>(mvKeysSemantic.data()), mDescriptorsSemantic.data, nL,
                                     reinterpret_cast<const SivoKeyPoint *>(mvKeysRight.data()), mDescriptorsRight.data,
                                     (int)mvKeysRight.size(), mbf, mb, mvRight.data(), mvDepth.data(), nullptr);
    if (rc != SIVO_OK) throw std::runtime_error(std::string("Frame::ComputeStereoMatches: ") + sivo_last_error());
}

void Frame::AssignFeaturesToGrid() {
    const int nReserve = (int)(0.5f * numSemanticKeys / (FRAME_GRID_COLS * FRAME_GRID_ROWS));
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) { mGrid[i][j].clear(); mGrid[i][j].reserve((size_t)nReserve); }
    for (int i = 0; i < numSemanticKeys; i++) {
        int gx, gy;
        if (PosInGrid(mvKeysSemantic[i], gx, gy)) mGrid[gx][gy].push_back((size_t)i);
    }
}

bool Frame::PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY) {
    posX = static_cast<int>(std::round((kp.pt.x - mnMinX) * mfGridElementWidthInv));
    posY = static_cast<int>(std::round((kp.pt.y - mnMinY) * mfGridElementHeightInv));
    return !(posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS);
}

std::vector<size_t> Frame::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel,
                                             const int maxLevel) const {
    std::vector<size_t> vIndices;
    vIndices.reserve((size_t)numSemanticKeys);
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - mnMinX + r) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - mnMinY + r) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (size_t idx : mGrid[ix][iy]) {
                const cv::KeyPoint &kp = mvKeysSemantic[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (std::fabs(kp.pt.x - x) < r && std::fabs(kp.pt.y - y) < r) vIndices.push_back(idx);
            }
    return vIndices;
}

bool Frame::UnprojectStereoCamera(const unsigned long &i, float xyz[3]) const {
    const float z = mvDepth.at(i);
    if (!(z > 0)) return false;
    xyz[0] = (mvKeysSemantic.at(i).pt.x - cx) * z * invfx;
    xyz[1] = (mvKeysSemantic.at(i).pt.y - cy) * z * invfy;
    xyz[2] = z;
    return true;
}

}  // namespace SIVO
