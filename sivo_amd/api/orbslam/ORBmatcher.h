// SIVO::ORBmatcher over libsivo_hip (reference include/orbslam/ORBmatcher.h:36-142, src/orbslam/ORBmatcher.cc).
//
// Same member names, argument meaning and results as the reference class.  Every Search* / Fuse member is split the
// same way: a GATHER pass over the SLAM objects (what the reference does per point in front of the window query:
// isBad(), projection, distance / viewing-angle tests, PredictScale ...) fills plain arrays, ONE call into the C ABI
// does the window queries, gates, Hamming scans, the sequential-consistency repair and the rotation histogram on the
// GPU (sivo_amd/csrc/search.hip), and a SCATTER pass stores the matches back into the objects.
//
// Frame, KeyFrame and MapPoint themselves are SLAM data model and not part of this library (out of
// scope), so the members that take them are templates: they compile against any types that expose the members the
// reference routines read, under the reference's own names — the reference's Frame / KeyFrame / MapPoint satisfy them
// as they are.  tests/cpp/test_api.cpp instantiates every one of them with minimal stand-ins.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#ifdef SIVO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "../compat/cv_min.hpp"
#endif

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeinfo>
#include <utility>
#include <vector>

#include "../../../include/sivo_hip.h"

namespace SIVO {

namespace matcher_detail {

static_assert(sizeof(cv::KeyPoint) == sizeof(SivoKeyPoint), "cv::KeyPoint and SivoKeyPoint must share a layout");

inline void check(int rc, const char *what) {
    if (rc != SIVO_OK) throw std::runtime_error(std::string("ORBmatcher::") + what + ": " + sivo_last_error());
}

// The HIP device the matcher's frame views live on: -1 (default) = the calling thread's current device.  Set once, before the
// first Search* call, by a host that drives several GPUs (ORBmatcher::SetDevice).
inline int &matcher_device() {
    static int device = -1;
    return device;
}

// The matcher's view of a Frame / KeyFrame on the device (keys, mvRight, descriptors, 64 x 48 grid).
class DeviceFrame {
 public:
    template <class FrameT>
    DeviceFrame(const FrameT &F, int device) {
        const int n = static_cast<int>(F.mvKeysSemantic.size());
        if (n && (!F.mDescriptorsSemantic.isContinuous() || F.mDescriptorsSemantic.rows != n || F.mDescriptorsSemantic.cols != 32))
            throw std::invalid_argument("ORBmatcher: mDescriptorsSemantic must be a continuous N x 32 CV_8U matrix");
        check(sivo_mframe_create(reinterpret_cast<const SivoKeyPoint *>(F.mvKeysSemantic.data()), n,
                                 F.mvRight.empty() ? nullptr : F.mvRight.data(), F.mDescriptorsSemantic.data, (float)F.mnMinX,
                                 (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY, F.mvScaleFactors.data(), F.mvLevelSigma2.data(),
                                 F.mvInvLevelSigma2.data(), (int)F.mvScaleFactors.size(), device, &h_),
              "frame upload");
        n_ = n;
    }
    ~DeviceFrame() { sivo_mframe_destroy(h_); }
    DeviceFrame(const DeviceFrame &) = delete;
    DeviceFrame &operator=(const DeviceFrame &) = delete;
    sivo_mframe_t get() const { return h_; }
    int size() const { return n_; }
    std::mutex &mutex() { return mu_; }

 private:
    sivo_mframe_t h_ = nullptr;
    int n_ = 0;
    std::mutex mu_;        // a handle serves one search at a time (its stream and scratch are its own)
};

// Views are kept per Frame / KeyFrame: Tracking searches the current frame two or three times per image (motion model, local
// map, relocalisation candidates), LocalMapping and LoopClosing search and fuse into the same keyframes again and again.  The
// key is what identifies the data the view was built from — the object's type and mnId, the number of semantic keys, the
// address of the descriptor matrix's pixels (a copied Frame clones its descriptors, a recycled address has another mnId) and
// a fingerprint of three of its entries — so that a view never outlives the data it mirrors in any way the matcher could observe.  The least recently used views
// are dropped beyond CAPACITY (their device memory goes back to the library's pool).  Types without an mnId get a view per call.
class FrameCache {
 public:
    enum : size_t { CAPACITY = 128 };
    struct Key {
        size_t type;
        unsigned long long id, print;
        int n, device;
        const void *desc;
        bool operator==(const Key &o) const {
            return type == o.type && id == o.id && print == o.print && n == o.n && device == o.device && desc == o.desc;
        }
    };
    // FNV-1a over the first, middle and last key / descriptor / mvRight entry: stand-in types in tests re-use ids and addresses
    template <class FrameT>
    static unsigned long long fingerprint(const FrameT &F) {
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&h](const void *p, size_t bytes) {
            const unsigned char *c = static_cast<const unsigned char *>(p);
            for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
        };
        const size_t n = F.mvKeysSemantic.size();
        if (!n) return h;
        for (size_t i : {(size_t)0, n / 2, n - 1}) {
            mix(&F.mvKeysSemantic[i], sizeof(cv::KeyPoint));
            mix(F.mDescriptorsSemantic.data + 32 * i, 32);
            if (!F.mvRight.empty()) mix(&F.mvRight[i], sizeof(float));
        }
        return h;
    }
    static FrameCache &instance() {
        static FrameCache c;
        return c;
    }
    template <class FrameT>
    std::shared_ptr<DeviceFrame> get(const FrameT &F) {
        return get_impl(F, matcher_device(), 0);
    }
    void clear() {
        std::lock_guard<std::mutex> lock(mu_);
        lru_.clear();
    }
    size_t size() {
        std::lock_guard<std::mutex> lock(mu_);
        return lru_.size();
    }

 private:
    // chosen when FrameT has an mnId ...
    template <class FrameT>
    auto get_impl(const FrameT &F, int device, int) -> decltype((void)F.mnId, std::shared_ptr<DeviceFrame>()) {
        const Key k{typeid(FrameT).hash_code(), (unsigned long long)F.mnId, fingerprint(F), (int)F.mvKeysSemantic.size(), device,
                    (const void *)F.mDescriptorsSemantic.data};
        std::lock_guard<std::mutex> lock(mu_);
        for (auto it = lru_.begin(); it != lru_.end(); ++it)
            if (it->first == k) {
                lru_.splice(lru_.begin(), lru_, it);
                return lru_.front().second;
            }
        lru_.emplace_front(k, std::make_shared<DeviceFrame>(F, device));
        if (lru_.size() > CAPACITY) lru_.pop_back();
        return lru_.front().second;
    }
    // ... a view per call otherwise
    template <class FrameT>
    std::shared_ptr<DeviceFrame> get_impl(const FrameT &F, int device, long) {
        return std::make_shared<DeviceFrame>(F, device);
    }
    std::mutex mu_;
    std::list<std::pair<Key, std::shared_ptr<DeviceFrame>>> lru_;
};

// A frame's view for the duration of one matcher call.
class FrameLease {
 public:
    template <class FrameT>
    explicit FrameLease(const FrameT &F) : frame_(FrameCache::instance().get(F)), lock_(frame_->mutex()) {}
    sivo_mframe_t get() const { return frame_->get(); }
    int size() const { return frame_->size(); }

 private:
    std::shared_ptr<DeviceFrame> frame_;
    std::unique_lock<std::mutex> lock_;
};

// The float arithmetic of the cv::Mat expressions the reference routines evaluate per point (OpenCV 3.x, matmul.cpp /
// convert.cpp / stat.cpp; tests/cpp/pin_matcher.cpp compares against the reference's own ORBmatcher.cc):
//   `R * x + t` (3 x 3 by 3 x 1, nothing transposed) is gemm's small-matrix path: each dot product summed left to right in
//   float, then + t;   `-R.t() * t` goes through the general kernel: products and sum in double, rounded once;
//   `M / s` is convertTo with the factor narrowed to float: m * (float)(1.0 / s);   Mat::dot and cv::norm sum in double.
struct Pose {            // [R | t] of a 4 x 4 (or 3 x 4) CV_32F matrix, optionally with a similarity's scale divided out
    float R[9], t[3];
    explicit Pose(const cv::Mat &T, float scale = 1.0f) {
        const float inv = (float)(1.0 / (double)scale);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R[3 * r + c] = scale == 1.0f ? T.at<float>(r, c) : T.at<float>(r, c) * inv;
            t[r] = scale == 1.0f ? T.at<float>(r, 3) : T.at<float>(r, 3) * inv;
        }
    }
    Pose(const cv::Mat &Rm, const cv::Mat &tm) {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R[3 * r + c] = Rm.at<float>(r, c);
            t[r] = tm.at<float>(r, 0);
        }
    }
    // Xc = R * X + t
    void apply(const float X[3], float Xc[3]) const {
        for (int r = 0; r < 3; ++r) {
            float s = R[3 * r] * X[0];
            s = s + R[3 * r + 1] * X[1];
            s = s + R[3 * r + 2] * X[2];
            Xc[r] = s + t[r];
        }
    }
    void apply(const cv::Mat &Xw, float Xc[3]) const {
        const float X[3] = {Xw.at<float>(0, 0), Xw.at<float>(1, 0), Xw.at<float>(2, 0)};
        apply(X, Xc);
    }
    // O = -R.t() * t
    void centre(float O[3]) const {
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int r = 0; r < 3; ++r) s += (double)R[3 * r + c] * (double)t[r];
            O[c] = (float)(-s);
        }
    }
};

inline float norm3(const float a[3]) { return (float)std::sqrt((double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2]); }

inline void descriptor_row(const cv::Mat &d, std::vector<uint8_t> &dst, size_t i) { std::memcpy(dst.data() + 32 * i, d.ptr(0), 32); }

// The vocabulary nodes two DBoW2::FeatureVector hold in common (both are std::map<NodeId, std::vector<unsigned>>), as
// the two CSR lists the C ABI takes — the lock-step walk of ORBmatcher.cc:181-260.
template <class FeatVec>
void common_nodes(const FeatVec &v1, const FeatVec &v2, std::vector<int32_t> &off1, std::vector<int32_t> &idx1,
                  std::vector<int32_t> &off2, std::vector<int32_t> &idx2) {
    off1.assign(1, 0); off2.assign(1, 0); idx1.clear(); idx2.clear();
    auto it1 = v1.begin(), it2 = v2.begin();
    while (it1 != v1.end() && it2 != v2.end()) {
        if (it1->first == it2->first) {
            for (auto i : it1->second) idx1.push_back((int32_t)i);
            for (auto i : it2->second) idx2.push_back((int32_t)i);
            off1.push_back((int32_t)idx1.size()); off2.push_back((int32_t)idx2.size());
            ++it1; ++it2;
        } else if (it1->first < it2->first) {
            it1 = v1.lower_bound(it2->first);
        } else {
            it2 = v2.lower_bound(it1->first);
        }
    }
}

}  // namespace matcher_detail

class ORBmatcher {
 public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (1 x 32 CV_8U rows).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // (not in the reference) the HIP device the frames' views are kept on — default -1: the calling thread's current device — and
    // the release of every cached view (e.g. at System::Reset).
    static void SetDevice(int device) { matcher_detail::matcher_device() = device; matcher_detail::FrameCache::instance().clear(); }
    static void ReleaseDeviceFrames() { matcher_detail::FrameCache::instance().clear(); }

    // Search matches between Frame keypoints and projected MapPoints.  Returns number of matches.
    // Used to track the local map (Tracking).                                              ORBmatcher.cc:44-127
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3);

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking).                                         ORBmatcher.cc:1278-1418
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking).                                                    ORBmatcher.cc:1420-1543
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound, const float th,
                           const int ORBdist);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing).                                                ORBmatcher.cc:286-399
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints,
                           std::vector<MapPointT *> &vpMatched, int th);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame, by vocabulary node.
    // Used in Relocalisation and Loop Detection.                                            ORBmatcher.cc:161-284, 508-629
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches);
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12);

    // Matching for the Map Initialization (only used in the monocular case).              ORBmatcher.cc:401-506
    // Sequential by construction (a later key may take over an earlier key's match): host loop over the frame grid.
    template <class FrameT>
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10);

    // Matching to triangulate new MapPoints.  Check Epipolar Constraint.                   ORBmatcher.cc:631-785
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, cv::Mat F12,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12].
    // In the stereo and RGB-D case, s12 = 1.                                                ORBmatcher.cc:1055-1276
    template <class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12,
                     const cv::Mat &R12, const cv::Mat &t12, const float th);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints.                 ORBmatcher.cc:787-929
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th = 3.0);
    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.   ORBmatcher.cc:931-1053
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, float th,
             std::vector<MapPointT *> &vpReplacePoint);

    // ---- array-level entry points (no SLAM types) ------------------------------------------------------------------
    // For query i (row i of `queries`, N x 32 CV_8U) the candidates are rows candIdx[candOff[i] .. candOff[i+1]) of
    // `train`.  Per query: best row (or -1), best and second-best distance (256 when absent) and the row holding the
    // second-best distance (-1 when absent).  Runs on the GPU (sivo_hamming_argmin2).
    void BestTwo(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> &candOff,
                 const std::vector<int32_t> &candIdx, std::vector<int> &bestIdx, std::vector<int> &bestDist,
                 std::vector<int> &secondDist, std::vector<int> *secondIdx = nullptr) const;

    // How the nearest-neighbour ratio enters the acceptance of a match; the reference routines differ:
    enum RatioRule {
        RATIO_NONE,        // frame-to-frame / relocalisation projection search, Fuse, Sim3 (ORBmatcher.cc:1372, 1504)
        RATIO_SAME_LEVEL,  // SearchByProjection(Frame&, MapPoints): only when best and second share an octave (:117-119)
        RATIO_ALWAYS       // SearchByBoW, SearchForInitialization: best < mfNNratio * second (:230, :455, :582)
    };
    // Acceptance + rotation consistency over candidate lists: accept query i iff bestDist <= thDist and the ratio rule
    // holds; then, if mbCheckOrientation, keep only the matches whose rotation bin (query angle - train angle) is one of
    // the three most populated of HISTO_LENGTH bins.  trainOctaves is read for RATIO_SAME_LEVEL only.
    int MatchCandidates(const cv::Mat &queries, const std::vector<float> &queryAngles, const cv::Mat &train,
                        const std::vector<float> &trainAngles, const std::vector<int32_t> &candOff,
                        const std::vector<int32_t> &candIdx, int thDist, RatioRule rule, const std::vector<int> &trainOctaves,
                        std::vector<int> &matches) const;

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) const;
    float RadiusByViewingCos(const float &viewCos) const { return viewCos > 0.998 ? 2.5f : 4.0f; }      // ORBmatcher.cc:129-134

 protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

// =====================================================================================================================
// template members
// =====================================================================================================================

template <class FrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th) {
    using namespace matcher_detail;
    const size_t n = vpMapPoints.size();
    std::vector<uint8_t> inView(n, 0), desc(32 * n, 0);
    std::vector<float> px(n, 0.f), py(n, 0.f), pxr(n, 0.f), viewCos(n, 0.f);
    std::vector<int32_t> level(n, 0), obs(n, 0);
    for (size_t i = 0; i < n; ++i) {
        MapPointT *pMP = vpMapPoints[i];
        if (!pMP->mbTrackInView || pMP->isBad()) continue;
        inView[i] = 1;
        px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR;
        level[i] = pMP->mnTrackScaleLevel; viewCos[i] = pMP->mTrackViewCos;
        obs[i] = pMP->Observations();
        descriptor_row(pMP->GetDescriptor(), desc, i);
    }
    FrameLease dF(F);
    std::vector<int32_t> occ(dF.size()), match(dF.size());
    for (int k = 0; k < dF.size(); ++k) occ[k] = F.mvpMapPoints[k] ? (int32_t)F.mvpMapPoints[k]->Observations() : -1;
    int nmatches = 0;
    check(sivo_search_by_projection_mappoints(dF.get(), (int)n, inView.data(), px.data(), py.data(), pxr.data(), level.data(),
                                              viewCos.data(), desc.data(), obs.data(), th, mfNNratio, occ.data(), match.data(),
                                              &nmatches),
          "SearchByProjection");
    for (int k = 0; k < dF.size(); ++k)
        if (match[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[match[k]];
    return nmatches;
}

template <class FrameT>
int ORBmatcher::SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono) {
    using namespace matcher_detail;
    // relative motion along the optical axis decides the octave range searched (:1284-1300)
    const Pose Tcw(CurrentFrame.mTcw), Tlw(LastFrame.mTcw);
    float twc[3], tlc[3];
    Tcw.centre(twc);
    Tlw.apply(twc, tlc);
    const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;

    const int n = LastFrame.numSemanticKeys;
    std::vector<uint8_t> valid((size_t)n, 0), desc(32 * (size_t)n, 0);
    std::vector<float> u((size_t)n, 0.f), v((size_t)n, 0.f), invz((size_t)n, 0.f), angle((size_t)n, 0.f);
    std::vector<int32_t> octave((size_t)n, 0), obs((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        auto *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        float x3Dc[3];
        Tcw.apply(pMP->GetWorldPos(), x3Dc);
        const float invzc = 1.0 / x3Dc[2];
        valid[i] = 1;
        invz[i] = invzc;
        u[i] = CurrentFrame.fx * x3Dc[0] * invzc + CurrentFrame.cx;
        v[i] = CurrentFrame.fy * x3Dc[1] * invzc + CurrentFrame.cy;
        octave[i] = LastFrame.mvKeysSemantic[i].octave;
        angle[i] = LastFrame.mvKeysSemantic[i].angle;
        obs[i] = pMP->Observations();
        descriptor_row(pMP->GetDescriptor(), desc, (size_t)i);
    }
    FrameLease dF(CurrentFrame);
    std::vector<int32_t> occ(dF.size()), match(dF.size());
    for (int k = 0; k < dF.size(); ++k) occ[k] = CurrentFrame.mvpMapPoints[k] ? (int32_t)CurrentFrame.mvpMapPoints[k]->Observations() : -1;
    int nmatches = 0;
    check(sivo_search_by_projection_frame(dF.get(), n, valid.data(), u.data(), v.data(), invz.data(), octave.data(), angle.data(),
                                          desc.data(), obs.data(), th, bForward, bBackward, CurrentFrame.mbf, mbCheckOrientation,
                                          occ.data(), match.data(), &nmatches),
          "SearchByProjection");
    for (int k = 0; k < dF.size(); ++k) {
        if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[match[k]];
        else if (match[k] == -2) CurrentFrame.mvpMapPoints[k] = nullptr;
    }
    return nmatches;
}

template <class FrameT, class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound,
                                   const float th, const int ORBdist) {
    using namespace matcher_detail;
    const Pose Tcw(CurrentFrame.mTcw);
    float Ow[3];
    Tcw.centre(Ow);
    const std::vector<MapPointT *> vpMPs = pKF->GetMapPointMatches();
    const size_t n = vpMPs.size();
    std::vector<uint8_t> valid(n, 0), desc(32 * n, 0);
    std::vector<float> u(n, 0.f), v(n, 0.f), angle(n, 0.f);
    std::vector<int32_t> level(n, 0);
    for (size_t i = 0; i < n; ++i) {
        MapPointT *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        float x3Dc[3];
        Tcw.apply(x3Dw, x3Dc);
        const float invzc = 1.0 / x3Dc[2];
        u[i] = CurrentFrame.fx * x3Dc[0] * invzc + CurrentFrame.cx;
        v[i] = CurrentFrame.fy * x3Dc[1] * invzc + CurrentFrame.cy;
        if (u[i] < CurrentFrame.mnMinX || u[i] > CurrentFrame.mnMaxX || v[i] < CurrentFrame.mnMinY || v[i] > CurrentFrame.mnMaxY) continue;
        const float PO[3] = {x3Dw.at<float>(0, 0) - Ow[0], x3Dw.at<float>(1, 0) - Ow[1], x3Dw.at<float>(2, 0) - Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
        valid[i] = 1;
        level[i] = pMP->PredictScale(dist3D, &CurrentFrame);
        angle[i] = pKF->mvKeysSemantic[i].angle;
        descriptor_row(pMP->GetDescriptor(), desc, i);
    }
    FrameLease dF(CurrentFrame);
    std::vector<uint8_t> occupied((size_t)dF.size());
    std::vector<int32_t> match((size_t)dF.size());
    for (int k = 0; k < dF.size(); ++k) occupied[k] = CurrentFrame.mvpMapPoints[k] != nullptr;
    int nmatches = 0;
    check(sivo_search_by_projection_reloc(dF.get(), (int)n, valid.data(), u.data(), v.data(), level.data(), angle.data(), desc.data(), th,
                                          ORBdist, mbCheckOrientation, occupied.data(), match.data(), &nmatches),
          "SearchByProjection");
    for (int k = 0; k < dF.size(); ++k) {
        if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = vpMPs[match[k]];
        else if (match[k] == -2) CurrentFrame.mvpMapPoints[k] = nullptr;
    }
    return nmatches;
}

namespace matcher_detail {
// The per-point tests shared by SearchByProjection(KF, Scw), the two Fuse and SearchBySim3 (ORBmatcher.cc:313-353, 806-848,
// 968-1006): positive depth, inside the image, distance inside the scale-invariance range, viewing angle below 60 deg.
template <class KeyFrameT, class MapPointT>
bool project_for_fusion(KeyFrameT *pKF, MapPointT *pMP, const Pose &Tcw, const float Ow[3], bool checkNormal, float &u, float &v,
                        float &invz, int &level) {
    const cv::Mat p3Dw = pMP->GetWorldPos();
    float p3Dc[3];
    Tcw.apply(p3Dw, p3Dc);
    if (p3Dc[2] < 0.0f) return false;
    invz = 1.0 / p3Dc[2];
    u = pKF->fx * (p3Dc[0] * invz) + pKF->cx;
    v = pKF->fy * (p3Dc[1] * invz) + pKF->cy;
    if (!pKF->IsInImage(u, v)) return false;
    const float PO[3] = {p3Dw.at<float>(0, 0) - Ow[0], p3Dw.at<float>(1, 0) - Ow[1], p3Dw.at<float>(2, 0) - Ow[2]};
    const float dist3D = norm3(PO);
    if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) return false;
    if (checkNormal) {
        const cv::Mat Pn = pMP->GetNormal();
        const double dot = (double)PO[0] * Pn.at<float>(0, 0) + (double)PO[1] * Pn.at<float>(1, 0) + (double)PO[2] * Pn.at<float>(2, 0);
        if (dot < 0.5 * dist3D) return false;
    }
    level = pMP->PredictScale(dist3D, pKF);
    return true;
}
inline float sim3_scale(const cv::Mat &Scw) {      // sqrt(sRcw.row(0).dot(sRcw.row(0)))
    const double a = Scw.at<float>(0, 0), b = Scw.at<float>(0, 1), c = Scw.at<float>(0, 2);
    return (float)std::sqrt(a * a + b * b + c * c);
}
}  // namespace matcher_detail

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByProjection(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints,
                                   std::vector<MapPointT *> &vpMatched, int th) {
    using namespace matcher_detail;
    const Pose Tcw(Scw, sim3_scale(Scw));
    float Ow[3];
    Tcw.centre(Ow);
    std::set<MapPointT *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPointT *>(nullptr));
    const size_t n = vpPoints.size();
    std::vector<uint8_t> valid(n, 0), desc(32 * n, 0);
    std::vector<float> u(n, 0.f), v(n, 0.f);
    std::vector<int32_t> level(n, 0);
    for (size_t i = 0; i < n; ++i) {
        MapPointT *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        float invz;
        int lvl;
        if (!project_for_fusion(pKF, pMP, Tcw, Ow, true, u[i], v[i], invz, lvl)) continue;
        valid[i] = 1; level[i] = lvl;
        descriptor_row(pMP->GetDescriptor(), desc, i);
    }
    FrameLease dKF(*pKF);
    std::vector<uint8_t> matched((size_t)dKF.size());
    std::vector<int32_t> match((size_t)dKF.size());
    for (int k = 0; k < dKF.size(); ++k) matched[k] = vpMatched[k] != nullptr;
    int nmatches = 0;
    check(sivo_search_by_projection_kf(dKF.get(), (int)n, valid.data(), u.data(), v.data(), level.data(), desc.data(), th,
                                       matched.data(), match.data(), &nmatches),
          "SearchByProjection");
    for (int k = 0; k < dKF.size(); ++k)
        if (match[k] >= 0) vpMatched[k] = vpPoints[match[k]];
    return nmatches;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th) {
    using namespace matcher_detail;
    const Pose Tcw(pKF->GetRotation(), pKF->GetTranslation());
    const cv::Mat OwM = pKF->GetCameraCenter();
    const float Ow[3] = {OwM.at<float>(0, 0), OwM.at<float>(1, 0), OwM.at<float>(2, 0)};
    const size_t n = vpMapPoints.size();
    std::vector<uint8_t> valid(n, 0), desc(32 * n, 0);
    std::vector<float> u(n, 0.f), v(n, 0.f), ur(n, 0.f);
    std::vector<int32_t> level(n, 0), best(n, -1);
    for (size_t i = 0; i < n; ++i) {
        MapPointT *pMP = vpMapPoints[i];
        if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        float invz;
        int lvl;
        if (!project_for_fusion(pKF, pMP, Tcw, Ow, true, u[i], v[i], invz, lvl)) continue;
        ur[i] = u[i] - pKF->mbf * invz;
        valid[i] = 1; level[i] = lvl;
        descriptor_row(pMP->GetDescriptor(), desc, i);
    }
    FrameLease dKF(*pKF);
    int nFused = 0;
    check(sivo_fuse(dKF.get(), (int)n, valid.data(), u.data(), v.data(), ur.data(), level.data(), desc.data(), th, 0, best.data(),
                    nullptr, &nFused),
          "Fuse");
    // If there is already a MapPoint replace otherwise add new measurement (:909-923), in the reference's order.  The
    // reference tests isBad() / IsInKeyFrame() at the top of every iteration, i.e. AFTER the surgery of the earlier ones
    // (a Replace can retire a point that comes later in the list); which key a point would fuse with does not depend on it.
    nFused = 0;
    for (size_t i = 0; i < n; ++i) {
        if (best[i] < 0) continue;
        MapPointT *pMP = vpMapPoints[i];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        ++nFused;
        MapPointT *pMPinKF = pKF->GetMapPoint(best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, best[i]);
            pKF->AddMapPoint(pMP, best[i]);
        }
    }
    return nFused;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT *pKF, cv::Mat Scw, const std::vector<MapPointT *> &vpPoints, float th,
                     std::vector<MapPointT *> &vpReplacePoint) {
    using namespace matcher_detail;
    const Pose Tcw(Scw, sim3_scale(Scw));
    float Ow[3];
    Tcw.centre(Ow);
    const std::set<MapPointT *> spAlreadyFound = pKF->GetMapPoints();
    const size_t n = vpPoints.size();
    std::vector<uint8_t> valid(n, 0), desc(32 * n, 0);
    std::vector<float> u(n, 0.f), v(n, 0.f);
    std::vector<int32_t> level(n, 0), best(n, -1);
    for (size_t i = 0; i < n; ++i) {
        MapPointT *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        float invz;
        int lvl;
        if (!project_for_fusion(pKF, pMP, Tcw, Ow, true, u[i], v[i], invz, lvl)) continue;
        valid[i] = 1; level[i] = lvl;
        descriptor_row(pMP->GetDescriptor(), desc, i);
    }
    FrameLease dKF(*pKF);
    int nFused = 0;
    check(sivo_fuse(dKF.get(), (int)n, valid.data(), u.data(), v.data(), nullptr, level.data(), desc.data(), th, 1, best.data(), nullptr,
                    &nFused),
          "Fuse");
    for (size_t i = 0; i < n; ++i) {
        if (best[i] < 0) continue;
        MapPointT *pMP = vpPoints[i];
        MapPointT *pMPinKF = pKF->GetMapPoint(best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, best[i]);
            pKF->AddMapPoint(pMP, best[i]);
        }
    }
    return nFused;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12,
                             const cv::Mat &R12, const cv::Mat &t12, const float th) {
    using namespace matcher_detail;
    const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    // [sR12 | t12] and [sR21 | t21] (:1074-1077): sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t(), t21 = -sR21 * t12
    Pose T12(R12, t12), T21(R12, t12);
    {
        const float inv = (float)(1.0 / (double)s12);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                T12.R[3 * r + c] = s12 * R12.at<float>(r, c);
                T21.R[3 * r + c] = R12.at<float>(c, r) * inv;
            }
        for (int r = 0; r < 3; ++r) {
            float acc = T21.R[3 * r] * t12.at<float>(0, 0);
            acc = acc + T21.R[3 * r + 1] * t12.at<float>(1, 0);
            acc = acc + T21.R[3 * r + 2] * t12.at<float>(2, 0);
            T21.t[r] = -acc;
        }
    }
    std::vector<uint8_t> already1((size_t)N1, 0), already2((size_t)N2, 0);
    for (int i = 0; i < N1; ++i) {
        MapPointT *pMP = vpMatches12[i];
        if (!pMP) continue;
        already1[i] = 1;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if (idx2 >= 0 && idx2 < N2) already2[idx2] = 1;
    }
    // one direction: the points of `from`, through its pose and the relative similarity, into `into`
    auto direction = [&](KeyFrameT *from, KeyFrameT *into, const std::vector<MapPointT *> &pts, const std::vector<uint8_t> &already,
                         const Pose &Trel, std::vector<int32_t> &vnMatch) {
        const size_t n = pts.size();
        const Pose Tfw(from->GetRotation(), from->GetTranslation());
        std::vector<uint8_t> valid(n, 0), desc(32 * n, 0);
        std::vector<float> u(n, 0.f), v(n, 0.f);
        std::vector<int32_t> level(n, 0);
        for (size_t i = 0; i < n; ++i) {
            MapPointT *pMP = pts[i];
            if (!pMP || already[i] || pMP->isBad()) continue;
            float pa[3], pb[3];
            Tfw.apply(pMP->GetWorldPos(), pa);
            Trel.apply(pa, pb);
            if (pb[2] < 0.0) continue;
            const float invz = 1.0 / pb[2];
            u[i] = into->fx * (pb[0] * invz) + into->cx;
            v[i] = into->fy * (pb[1] * invz) + into->cy;
            if (!into->IsInImage(u[i], v[i])) continue;
            const float dist3D = norm3(pb);
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            valid[i] = 1;
            level[i] = pMP->PredictScale(dist3D, into);
            descriptor_row(pMP->GetDescriptor(), desc, i);
        }
        FrameLease d(*into);
        vnMatch.assign(n, -1);
        check(sivo_search_by_sim3_dir(d.get(), (int)n, valid.data(), u.data(), v.data(), level.data(), desc.data(), th, vnMatch.data()),
              "SearchBySim3");
    };
    std::vector<int32_t> vnMatch1, vnMatch2;
    direction(pKF1, pKF2, vpMapPoints1, already1, T21, vnMatch1);
    direction(pKF2, pKF1, vpMapPoints2, already2, T12, vnMatch2);
    int nFound = 0;                                                            // check agreement (:1254-1273)
    for (int i1 = 0; i1 < N1; ++i1) {
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; ++nFound; }
    }
    return nFound;
}

template <class KeyFrameT, class FrameT, class MapPointT>
int ORBmatcher::SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches) {
    using namespace matcher_detail;
    const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches.assign((size_t)F.numSemanticKeys, nullptr);
    std::vector<int32_t> off1, idx1, off2, idx2;
    common_nodes(pKF->mFeatVec, F.mFeatVec, off1, idx1, off2, idx2);
    const size_t nKF = vpMapPointsKF.size();
    std::vector<uint8_t> valid(nKF, 0);
    for (size_t i = 0; i < nKF; ++i) valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
    FrameLease dF(F);
    std::vector<int32_t> match((size_t)dF.size(), -1);
    int nmatches = 0;
    check(sivo_search_by_bow_kf_frame((int)off1.size() - 1, off1.data(), idx1.data(), off2.data(), idx2.data(), valid.data(),
                                      reinterpret_cast<const SivoKeyPoint *>(pKF->mvKeysSemantic.data()), pKF->mDescriptorsSemantic.data,
                                      (int)nKF, dF.get(), mfNNratio, mbCheckOrientation, match.data(), &nmatches),
          "SearchByBoW");
    for (int k = 0; k < dF.size(); ++k)
        if (match[k] >= 0) vpMapPointMatches[k] = vpMapPointsKF[match[k]];
    return nmatches;
}

template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12) {
    using namespace matcher_detail;
    const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12.assign(vpMapPoints1.size(), nullptr);
    std::vector<int32_t> off1, idx1, off2, idx2;
    common_nodes(pKF1->mFeatVec, pKF2->mFeatVec, off1, idx1, off2, idx2);
    std::vector<uint8_t> valid1(vpMapPoints1.size()), valid2(vpMapPoints2.size());
    for (size_t i = 0; i < valid1.size(); ++i) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
    for (size_t i = 0; i < valid2.size(); ++i) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
    FrameLease d2(*pKF2);
    std::vector<int32_t> m12(vpMapPoints1.size(), -1);
    int nmatches = 0;
    check(sivo_search_by_bow_kf_kf((int)off1.size() - 1, off1.data(), idx1.data(), off2.data(), idx2.data(), valid1.data(),
                                   reinterpret_cast<const SivoKeyPoint *>(pKF1->mvKeysSemantic.data()), pKF1->mDescriptorsSemantic.data,
                                   (int)vpMapPoints1.size(), valid2.data(), d2.get(), mfNNratio, mbCheckOrientation, m12.data(),
                                   &nmatches),
          "SearchByBoW");
    for (size_t i = 0; i < m12.size(); ++i)
        if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
    return nmatches;
}

template <class KeyFrameT>
int ORBmatcher::SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, cv::Mat F12,
                                       std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo) {
    using namespace matcher_detail;
    // epipole in the second image (:639-647)
    float C2[3];
    Pose(pKF2->GetRotation(), pKF2->GetTranslation()).apply(pKF1->GetCameraCenter(), C2);
    const float invz = 1.0f / C2[2];
    const float ex = pKF2->fx * C2[0] * invz + pKF2->cx, ey = pKF2->fy * C2[1] * invz + pKF2->cy;
    std::vector<int32_t> off1, idx1, off2, idx2;
    common_nodes(pKF1->mFeatVec, pKF2->mFeatVec, off1, idx1, off2, idx2);
    const int n1 = pKF1->numSemanticKeys, n2 = pKF2->numSemanticKeys;
    std::vector<uint8_t> has1((size_t)n1), has2((size_t)n2);
    for (int i = 0; i < n1; ++i) has1[i] = pKF1->GetMapPoint(i) != nullptr;
    for (int i = 0; i < n2; ++i) has2[i] = pKF2->GetMapPoint(i) != nullptr;
    float F[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
    FrameLease d2(*pKF2);
    std::vector<int32_t> m12((size_t)n1, -1);
    int nmatches = 0;
    check(sivo_search_for_triangulation((int)off1.size() - 1, off1.data(), idx1.data(), off2.data(), idx2.data(),
                                        reinterpret_cast<const SivoKeyPoint *>(pKF1->mvKeysSemantic.data()),
                                        pKF1->mvRight.empty() ? nullptr : pKF1->mvRight.data(), has1.data(), pKF1->mDescriptorsSemantic.data,
                                        n1, d2.get(), has2.data(), F, ex, ey, bOnlyStereo, mbCheckOrientation, m12.data(), &nmatches),
          "SearchForTriangulation");
    vMatchedPairs.clear();
    vMatchedPairs.reserve((size_t)(nmatches > 0 ? nmatches : 0));
    for (int i = 0; i < n1; ++i)
        if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return nmatches;
}

template <class FrameT>
int ORBmatcher::SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                        std::vector<int> &vnMatches12, int windowSize) {
    int nmatches = 0;
    const size_t n1 = F1.mvKeysSemantic.size(), n2 = F2.mvKeysSemantic.size();
    vnMatches12.assign(n1, -1);
    std::vector<std::vector<int> > rotHist((size_t)HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> matchedDistance(n2, INT_MAX), matches21(n2, -1);
    for (size_t i1 = 0; i1 < n1; ++i1) {
        const int level1 = F1.mvKeysSemantic[i1].octave;
        if (level1 > 0) continue;
        const std::vector<size_t> window = F2.GetFeaturesInArea(vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, level1, level1);
        if (window.empty()) continue;
        const cv::Mat d1 = F1.mDescriptorsSemantic.row((int)i1);
        int best = INT_MAX, second = INT_MAX, bestIdx2 = -1;
        for (size_t i2 : window) {
            const int dist = DescriptorDistance(d1, F2.mDescriptorsSemantic.row((int)i2));
            if (matchedDistance[i2] <= dist) continue;
            if (dist < best) { second = best; best = dist; bestIdx2 = (int)i2; }
            else if (dist < second) second = dist;
        }
        if (best > TH_LOW || !(best < (float)second * mfNNratio)) continue;
        if (matches21[bestIdx2] >= 0) { vnMatches12[matches21[bestIdx2]] = -1; --nmatches; }
        vnMatches12[i1] = bestIdx2;
        matches21[bestIdx2] = (int)i1;
        matchedDistance[bestIdx2] = best;
        ++nmatches;
        if (mbCheckOrientation) {
            float rot = F1.mvKeysSemantic[i1].angle - F2.mvKeysSemantic[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back((int)i1);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist.data(), HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; ++b) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int idx1 : rotHist[b])
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; --nmatches; }
        }
    }
    for (size_t i1 = 0; i1 < n1; ++i1)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysSemantic[vnMatches12[i1]].pt;
    return nmatches;
}

}  // namespace SIVO
#endif
