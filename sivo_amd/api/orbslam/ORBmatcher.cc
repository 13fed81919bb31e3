// SIVO::ORBmatcher, non-template members (reference src/orbslam/ORBmatcher.cc:37-39, 78-121, 1545-1596); the Search* / Fuse
// members are templates in ORBmatcher.h.
#include "ORBmatcher.h"

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../../include/sivo_hip.h"

namespace SIVO {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// One pair on the host, exactly the reference's scalar routine's result (32-bit words, popcount); the
// batched forms below are the GPU path.
int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
    const unsigned char *pa = a.ptr(), *pb = b.ptr();
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y;
        std::memcpy(&x, pa + 4 * i, 4);
        std::memcpy(&y, pb + 4 * i, 4);
        dist += __builtin_popcount(x ^ y);
    }
    return dist;
}

void ORBmatcher::BestTwo(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> &candOff,
                         const std::vector<int32_t> &candIdx, std::vector<int> &bestIdx, std::vector<int> &bestDist,
                         std::vector<int> &secondDist, std::vector<int> *secondIdx) const {
    const int n = queries.rows;
    if ((int)candOff.size() != n + 1) throw std::invalid_argument("candOff must hold rows + 1 offsets");
    bestIdx.assign(n, -1); bestDist.assign(n, 256); secondDist.assign(n, 256);
    if (secondIdx) secondIdx->assign(n, -1);
    if (n == 0) return;
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
    const int rc = sivo_hamming_argmin2(queries.data, n, train.data, train.rows, candOff.data(), candIdx.data(),
                                        bestIdx.data(), bestDist.data(), secondDist.data(), secondIdx ? secondIdx->data() : nullptr);
    if (rc != SIVO_OK) throw std::runtime_error(std::string("ORBmatcher: ") + sivo_last_error());
}

int ORBmatcher::MatchCandidates(const cv::Mat &queries, const std::vector<float> &queryAngles, const cv::Mat &train,
                                const std::vector<float> &trainAngles, const std::vector<int32_t> &candOff,
                                const std::vector<int32_t> &candIdx, int thDist, RatioRule rule,
                                const std::vector<int> &trainOctaves, std::vector<int> &matches) const {
    std::vector<int> bi, bd, sd, si;
    BestTwo(queries, train, candOff, candIdx, bi, bd, sd, &si);
    const int n = queries.rows;
    if (rule == RATIO_SAME_LEVEL && (int)trainOctaves.size() != train.rows)
        throw std::invalid_argument("RATIO_SAME_LEVEL needs one octave per train row");
    matches.assign(n, -1);
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n; ++i) {
        if (bi[i] < 0 || bd[i] > thDist) continue;
        if (rule == RATIO_SAME_LEVEL) {
            // ORBmatcher.cc:117-119: only when best and second lie on the same pyramid level (no second: levels differ)
            const int bestLevel = trainOctaves[bi[i]], bestLevel2 = si[i] >= 0 ? trainOctaves[si[i]] : -1;
            if (bestLevel == bestLevel2 && bd[i] > mfNNratio * sd[i]) continue;
        } else if (rule == RATIO_ALWAYS) {
            // ORBmatcher.cc:230, :582: static_cast<float>(best) < mfNNratio * static_cast<float>(second)
            if (!(static_cast<float>(bd[i]) < mfNNratio * static_cast<float>(sd[i]))) continue;
        }
        matches[i] = bi[i];
        ++nmatches;
        if (mbCheckOrientation) {
            float rot = queryAngles[i] - trainAngles[bi[i]];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(i);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int q : rotHist[b]) { matches[q] = -1; --nmatches; }
    }
    return nmatches;
}

// ORBmatcher.cc:1545-1577
void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) const {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; ++i) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace SIVO
