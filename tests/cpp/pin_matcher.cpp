// Pins the guided-matching path against the reference's OWN code.
//
// oracle/Makefile compiles /root/reference/src/orbslam/ORBmatcher.cc, untouched, against the stand-in SLAM types of
// oracle/ref_shims/slam_standins.h (class renamed SIVO::RefORBmatcher by -DORBmatcher=RefORBmatcher).  This program builds
// deterministic stereo scenes out of those stand-in types, runs every Search* / Fuse member of
//     (a) the reference's ORBmatcher.cc                       ("ref"), and
//     (b) this repository's SIVO::ORBmatcher templates        ("mine": gather -> C ABI -> scatter)
// on identical copies of each scene and requires identical results: return value, every output vector, and the ordered
// log of object-graph mutations (Replace / AddObservation / AddMapPoint).
//
// Two link variants (sivo_amd/api/Makefile):
//   oracle/_ref/pin_matcher_cpu   the C ABI is tests/cpp/abi_on_oracle.cpp -> oracle/search_oracle.c: pins the CPU oracle
//                                 (and the templates' gather / scatter) to the reference, no GPU needed;
//   oracle/_ref/pin_matcher_gpu   the C ABI is libsivo_hip.so: pins the device path to the reference.
// Built with -DPIN_NO_REFERENCE (tests/cpp/golden_matcher_{cpu,gpu}) the reference is not linked and (b) is compared
// with tests/golden/matcher_reference.txt, which `pin_matcher_cpu --write-golden` wrote from (a).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "slam_standins.h"
#ifndef PIN_NO_REFERENCE
#define ORBmatcher RefORBmatcher
#include "include/orbslam/ORBmatcher.h"      // oracle/ref_shims: the declaration the reference's .cc was compiled under
#undef ORBmatcher
#endif
#include "../../sivo_amd/api/orbslam/ORBmatcher.h"

using SIVO::Frame;
using SIVO::KeyFrame;
using SIVO::MapPoint;

// ---------------------------------------------------------------------------------------------------------------------
// deterministic scene
// ---------------------------------------------------------------------------------------------------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) {}
    uint32_t u32() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
    float uni() { return (float)(u32() & 0xFFFFFF) / 16777216.0f; }
    float range(float a, float b) { return a + (b - a) * uni(); }
    int below(int n) { return (int)(u32() % (uint32_t)n); }
};

struct SceneSpec {
    uint64_t seed = 1;
    int n_points = 700, n_clutter = 220, n_proto = 160;
    float yaw_deg = 1.0f, tx = 0.12f, ty = -0.02f, tz = 0.75f;    // pose of B relative to A (tz > mb: "forward")
    float angle_shift = 10.0f;
    bool level0_only = false;                                        // SearchForInitialization works on octave 0
    float noise_px = 1.5f;                                           // key position noise in B
    int desc_noise = 14;                                             // up to this many flipped descriptor bits per key
    int cluster = 1;                                                 // > 1: groups of this many look-alike points (same prototype and level) a few pixels apart
};

struct Scene {
    KeyFrame A, B;
    std::vector<std::unique_ptr<MapPoint> > points;   // [0, n_points): world points; then the temporal points of B
    std::vector<MapPoint *> world;                    // the first n_points
    std::vector<int> keyA_of, keyB_of;                // world point -> key index in A / B
};

static void flip_bits(uint8_t *d, int n, Rng &r) { for (int i = 0; i < n; ++i) { const int b = r.below(256); d[b >> 3] ^= (uint8_t)(1u << (b & 7)); } }

static cv::Mat pose(float yaw_deg, float tx, float ty, float tz) {
    cv::Mat T = cv::Mat::zeros(4, 4, CV_32F);
    const float a = yaw_deg * 3.14159265f / 180.0f, c = std::cos(a), s = std::sin(a);
    T.at<float>(0, 0) = c; T.at<float>(0, 2) = s; T.at<float>(1, 1) = 1.f; T.at<float>(2, 0) = -s; T.at<float>(2, 2) = c; T.at<float>(3, 3) = 1.f;
    T.at<float>(0, 3) = tx; T.at<float>(1, 3) = ty; T.at<float>(2, 3) = tz;
    return T;
}

static void camera(Frame &F, long id, const cv::Mat &Tcw) {
    F.mnId = id;
    F.fx = F.fy = 718.856f; F.cx = 607.1928f; F.cy = 185.2157f; F.mbf = 386.1448f; F.mb = F.mbf / F.fx;
    F.mnMinX = 0.f; F.mnMaxX = 1241.f; F.mnMinY = 0.f; F.mnMaxY = 376.f;
    F.mnScaleLevels = 8; F.mfLogScaleFactor = std::log(1.2f);
    F.mvScaleFactors.assign(8, 1.0f); F.mvLevelSigma2.assign(8, 1.0f); F.mvInvLevelSigma2.assign(8, 1.0f);
    for (int i = 1; i < 8; ++i) {
        F.mvScaleFactors[i] = F.mvScaleFactors[i - 1] * 1.2f;
        F.mvLevelSigma2[i] = F.mvScaleFactors[i] * F.mvScaleFactors[i];
        F.mvInvLevelSigma2[i] = 1.0f / F.mvLevelSigma2[i];
    }
    F.mTcw = Tcw.clone();
}

static void set_centre(KeyFrame &K) {
    const cv::Mat R = K.mTcw.rowRange(0, 3).colRange(0, 3), t = K.mTcw.rowRange(0, 3).col(3);
    K.mOw = -R.t() * t;
}

static bool project(const Frame &F, const cv::Mat &Xw, float &u, float &v, float &z) {
    const cv::Mat R = F.mTcw.rowRange(0, 3).colRange(0, 3), t = F.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Xc = R * Xw + t;
    z = Xc.at<float>(2);
    if (z <= 0.1f) return false;
    u = F.fx * Xc.at<float>(0) / z + F.cx; v = F.fy * Xc.at<float>(1) / z + F.cy;
    return u >= F.mnMinX + 2 && u < F.mnMaxX - 2 && v >= F.mnMinY + 2 && v < F.mnMaxY - 2;
}

struct KeyDraft { cv::KeyPoint kp; float right; uint8_t desc[32]; int point; unsigned node; };

static void commit_keys(KeyFrame &K, std::vector<KeyDraft> &d, Rng &r, std::vector<int> &key_of, int n_points) {
    for (size_t i = d.size(); i > 1; --i) std::swap(d[i - 1], d[(size_t)r.below((int)i)]);      // shuffled key order
    const int n = (int)d.size();
    K.numSemanticKeys = n;
    K.mvKeysSemantic.resize((size_t)n); K.mvRight.resize((size_t)n); K.mDescriptorsSemantic = cv::Mat(n, 32, CV_8UC1);
    K.mvpMapPoints.assign((size_t)n, nullptr); K.mvbOutlier.assign((size_t)n, false);
    key_of.assign((size_t)n_points, -1);
    for (int i = 0; i < n; ++i) {
        K.mvKeysSemantic[i] = d[i].kp; K.mvRight[i] = d[i].right;
        std::memcpy(K.mDescriptorsSemantic.ptr(i), d[i].desc, 32);
        if (d[i].point >= 0) key_of[(size_t)d[i].point] = i;
        K.mFeatVec[d[i].node].push_back((unsigned)i);
    }
}

static std::unique_ptr<Scene> build_scene(const SceneSpec &spec) {
    std::unique_ptr<Scene> S(new Scene);
    Rng r(spec.seed);
    camera(S->A, 1, pose(0.f, 0.f, 0.f, 0.f));
    camera(S->B, 2, pose(spec.yaw_deg, spec.tx, spec.ty, spec.tz));
    set_centre(S->A); set_centre(S->B);
    std::vector<std::vector<uint8_t> > proto((size_t)spec.n_proto, std::vector<uint8_t>(32));
    for (auto &p : proto) for (auto &b : p) b = (uint8_t)r.u32();
    std::vector<KeyDraft> da, db;
    for (int i = 0; i < spec.n_points; ++i) {
        std::unique_ptr<MapPoint> P(new MapPoint);
        P->mnId = i;
        float z = r.range(4.f, 55.f), u = r.range(20.f, 1220.f), v = r.range(15.f, 360.f);
        static float cz, cu, cv_; static int cl;
        if (spec.cluster > 1) {
            if (i % spec.cluster == 0) { cz = z; cu = u; cv_ = v; cl = r.below(7); }
            z = cz * r.range(0.985f, 1.015f); u = std::min(1225.f, std::max(15.f, cu + r.range(-11.f, 11.f))); v = std::min(365.f, std::max(10.f, cv_ + r.range(-11.f, 11.f)));
        }
        P->mWorldPos = cv::Mat(3, 1, CV_32F);
        P->mWorldPos.at<float>(0) = (u - S->A.cx) * z / S->A.fx; P->mWorldPos.at<float>(1) = (v - S->A.cy) * z / S->A.fy; P->mWorldPos.at<float>(2) = z;
        const cv::Mat PO = P->mWorldPos - S->A.mOw;
        const float dist = (float)cv::norm(PO);
        P->mNormalVector = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; ++k) P->mNormalVector.at<float>(k) = PO.at<float>(k) / dist;
        if (r.below(12) == 0) P->mNormalVector.at<float>(2) = -P->mNormalVector.at<float>(2);       // seen from behind: fails the 60 deg test
        const int lA = spec.level0_only ? 0 : spec.cluster > 1 ? cl : r.below(7);
        P->mfMaxDistance = dist * S->A.mvScaleFactors[lA];
        P->mfMinDistance = P->mfMaxDistance / S->A.mvScaleFactors[7];
        P->mDescriptor = cv::Mat(1, 32, CV_8UC1);
        const int pr = spec.cluster > 1 ? (i / spec.cluster) % spec.n_proto : i % spec.n_proto;
        std::memcpy(P->mDescriptor.data, proto[(size_t)pr].data(), 32);
        flip_bits(P->mDescriptor.data, 6, r);
        P->mbBad = r.below(33) == 0;
        const float theta = r.range(0.f, 360.f);
        // key in A
        {
            KeyDraft k; k.point = i; k.node = (unsigned)(pr % 37);
            k.kp.pt.x = u + r.range(-1.2f, 1.2f); k.kp.pt.y = v + r.range(-1.2f, 1.2f); k.kp.octave = lA; k.kp.size = 31.f * S->A.mvScaleFactors[lA];
            k.kp.angle = std::fmod(theta + r.range(-3.f, 3.f) + 360.f, 360.f); k.kp.response = r.range(20.f, 120.f);
            k.right = r.below(7) == 0 ? -1.f : k.kp.pt.x - S->A.mbf / z + r.range(-0.3f, 0.3f);
            std::memcpy(k.desc, P->mDescriptor.data, 32); flip_bits(k.desc, r.below(12), r);
            da.push_back(k);
        }
        // key in B (if the point projects into it)
        float ub, vb, zb;
        if (project(S->B, P->mWorldPos, ub, vb, zb) && r.below(10) != 0) {
            const cv::Mat POb = P->mWorldPos - S->B.mOw;
            const int pred = P->PredictScale((float)cv::norm(POb), &S->B);
            int lB = spec.level0_only ? 0 : pred + (r.below(5) == 0 ? -1 : 0) + (r.below(9) == 0 ? 1 : 0);
            lB = lB < 0 ? 0 : lB > 7 ? 7 : lB;
            KeyDraft k; k.point = i; k.node = (unsigned)(pr % 37);
            k.kp.pt.x = ub + r.range(-spec.noise_px, spec.noise_px); k.kp.pt.y = vb + r.range(-spec.noise_px, spec.noise_px); k.kp.octave = lB; k.kp.size = 31.f * S->B.mvScaleFactors[lB];
            const float a = r.below(9) == 0 ? r.range(0.f, 360.f) : theta + spec.angle_shift + r.range(-4.f, 4.f);
            k.kp.angle = std::fmod(a + 720.f, 360.f); k.kp.response = r.range(20.f, 120.f);
            k.right = r.below(7) == 0 ? -1.f : k.kp.pt.x - S->B.mbf / zb + r.range(-0.3f, 0.3f);
            std::memcpy(k.desc, P->mDescriptor.data, 32); flip_bits(k.desc, r.below(spec.desc_noise), r);
            db.push_back(k);
        }
        S->world.push_back(P.get());
        S->points.push_back(std::move(P));
    }
    for (int f = 0; f < 2; ++f)
        for (int i = 0; i < spec.n_clutter; ++i) {
            KeyDraft k; k.point = -1; k.node = (unsigned)r.below(37);
            k.kp.pt.x = r.range(3.f, 1238.f); k.kp.pt.y = r.range(3.f, 373.f); k.kp.octave = spec.level0_only ? 0 : r.below(8);
            k.kp.size = 31.f; k.kp.angle = r.range(0.f, 360.f); k.kp.response = r.range(20.f, 120.f);
            k.right = r.below(3) == 0 ? -1.f : k.kp.pt.x - r.range(5.f, 90.f);
            if (r.below(3) == 0) { std::memcpy(k.desc, proto[(size_t)r.below(spec.n_proto)].data(), 32); flip_bits(k.desc, 10 + r.below(25), r); }
            else for (auto &b : k.desc) b = (uint8_t)r.u32();
            (f ? db : da).push_back(k);
        }
    commit_keys(S->A, da, r, S->keyA_of, spec.n_points);
    commit_keys(S->B, db, r, S->keyB_of, spec.n_points);
    // associations: A holds most of its points; B holds some (occupied slots), a few through temporal points (no observations)
    for (int i = 0; i < spec.n_points; ++i) {
        MapPoint *P = S->world[(size_t)i];
        const int ka = S->keyA_of[(size_t)i], kb = S->keyB_of[(size_t)i];
        if (ka >= 0 && r.below(5) != 0) {
            S->A.mvpMapPoints[(size_t)ka] = P;
            P->mObservations[&S->A] = (size_t)ka; P->nObs += S->A.mvRight[(size_t)ka] >= 0 ? 2 : 1;
            S->A.mvbOutlier[(size_t)ka] = r.below(20) == 0;
        }
        if (kb >= 0) {
            const int what = r.below(10);
            if (what < 3) {
                S->B.mvpMapPoints[(size_t)kb] = P;
                P->mObservations[&S->B] = (size_t)kb; P->nObs += S->B.mvRight[(size_t)kb] >= 0 ? 2 : 1;
            } else if (what == 3) {                                    // a temporal point of the stereo odometry: Observations() == 0
                std::unique_ptr<MapPoint> Tp(new MapPoint(*P));
                Tp->mnId = (long)S->points.size(); Tp->nObs = 0; Tp->mObservations.clear(); Tp->mbBad = false;
                S->B.mvpMapPoints[(size_t)kb] = Tp.get();
                S->points.push_back(std::move(Tp));
            }
        }
        // what Frame::isInFrustum(B) would leave
        float ub, vb, zb;
        P->mbTrackInView = project(S->B, P->mWorldPos, ub, vb, zb) && r.below(8) != 0;
        if (P->mbTrackInView) {
            P->mTrackProjX = ub; P->mTrackProjY = vb; P->mTrackProjXR = ub - S->B.mbf / zb;
            const cv::Mat POb = P->mWorldPos - S->B.mOw;
            P->mnTrackScaleLevel = P->PredictScale((float)cv::norm(POb), &S->B);
            P->mTrackViewCos = r.below(2) ? r.range(0.9985f, 1.0f) : r.range(0.6f, 0.998f);
        }
    }
    return S;
}

// ---------------------------------------------------------------------------------------------------------------------
// one routine = one function over a scene and a matcher type, returning everything observable as integers
// ---------------------------------------------------------------------------------------------------------------------
typedef std::vector<long> Result;
static long id_of(const MapPoint *p) { return p ? p->mnId : -1; }
static void put_points(Result &o, const std::vector<MapPoint *> &v) { o.push_back((long)v.size()); for (auto *p : v) o.push_back(id_of(p)); }
static void put_log(Result &o) {
    o.push_back((long)SIVO::pin_log().size());
    for (const auto &e : SIVO::pin_log()) { o.push_back(e.kind); o.push_back(e.a); o.push_back(e.b); o.push_back(e.c); }
}
static void put_state(Result &o, const Scene &S) {
    put_points(o, S.A.mvpMapPoints); put_points(o, S.B.mvpMapPoints);
    for (const auto &p : S.points) { o.push_back(p->mbBad); o.push_back(p->nObs); o.push_back(id_of(p->mpReplaced)); }
}
static long bits(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }

static cv::Mat sim3(const cv::Mat &Tcw, float s) {       // Converter::toCvMat(g2o::Sim3) = [sR | t]
    cv::Mat S = Tcw.clone();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S.at<float>(r, c) = s * Tcw.at<float>(r, c);
    for (int r = 0; r < 3; ++r) S.at<float>(r, 3) = s * Tcw.at<float>(r, 3);
    return S;
}
static cv::Mat skew(const cv::Mat &v) {
    cv::Mat m = cv::Mat::zeros(3, 3, CV_32F);
    m.at<float>(0, 1) = -v.at<float>(2); m.at<float>(0, 2) = v.at<float>(1); m.at<float>(1, 0) = v.at<float>(2);
    m.at<float>(1, 2) = -v.at<float>(0); m.at<float>(2, 0) = -v.at<float>(1); m.at<float>(2, 1) = v.at<float>(0);
    return m;
}
static cv::Mat fundamental(const KeyFrame &K1, const KeyFrame &K2) {    // LocalMapping::ComputeF12 (LocalMapping.cc:640-660), K1 == K2 here
    const cv::Mat R1w = K1.GetRotation(), t1w = K1.GetTranslation(), R2w = K2.GetRotation(), t2w = K2.GetTranslation();
    const cv::Mat R12 = R1w * cv::Mat(R2w.t());
    const cv::Mat t12 = cv::Mat(-R12 * t2w) + t1w;
    cv::Mat Kinv = cv::Mat::zeros(3, 3, CV_32F);
    Kinv.at<float>(0, 0) = 1.f / K1.fx; Kinv.at<float>(1, 1) = 1.f / K1.fy; Kinv.at<float>(0, 2) = -K1.cx / K1.fx; Kinv.at<float>(1, 2) = -K1.cy / K1.fy;
    Kinv.at<float>(2, 2) = 1.f;
    const cv::Mat a = cv::Mat(Kinv.t()) * skew(t12);
    const cv::Mat b = a * R12;
    return cv::Mat(b * Kinv);
}

template <class M> Result r_local_map(Scene &S, float th, float ratio) {
    M m(ratio, true);
    Result o; o.push_back(m.SearchByProjection(static_cast<Frame &>(S.B), S.world, th)); put_points(o, S.B.mvpMapPoints); return o;
}
template <class M> Result r_frame(Scene &S, float th, bool mono, bool ori) {
    M m(0.9f, ori);
    Result o; o.push_back(m.SearchByProjection(static_cast<Frame &>(S.B), static_cast<const Frame &>(S.A), th, mono)); put_points(o, S.B.mvpMapPoints);
    return o;
}
template <class M> Result r_reloc(Scene &S, float th, int orbdist) {
    M m(0.9f, true);
    std::set<MapPoint *> found;
    for (size_t i = 0; i < S.world.size(); i += 7) found.insert(S.world[i]);
    Result o; o.push_back(m.SearchByProjection(static_cast<Frame &>(S.B), &S.A, found, th, orbdist)); put_points(o, S.B.mvpMapPoints); return o;
}
template <class M> Result r_loop(Scene &S, float s, int th) {
    M m(0.75f, true);
    std::vector<MapPoint *> matched = S.B.mvpMapPoints;
    Result o; o.push_back(m.SearchByProjection(&S.B, sim3(S.B.mTcw, s), S.world, matched, th)); put_points(o, matched); return o;
}
template <class M> Result r_bow_frame(Scene &S, float ratio, bool ori) {
    M m(ratio, ori);
    std::vector<MapPoint *> out;
    Result o; o.push_back(m.SearchByBoW(&S.A, static_cast<Frame &>(S.B), out)); put_points(o, out); return o;
}
template <class M> Result r_bow_kf(Scene &S, float ratio, bool ori) {
    M m(ratio, ori);
    std::vector<MapPoint *> out;
    Result o; o.push_back(m.SearchByBoW(&S.A, &S.B, out)); put_points(o, out); return o;
}
template <class M> Result r_init(Scene &S, int window, float ratio) {
    M m(ratio, true);
    std::vector<cv::Point2f> prev;
    for (const auto &k : S.A.mvKeysSemantic) prev.push_back(k.pt);
    std::vector<int> m12;
    Result o; o.push_back(m.SearchForInitialization(static_cast<Frame &>(S.A), static_cast<Frame &>(S.B), prev, m12, window));
    o.push_back((long)m12.size()); for (int v : m12) o.push_back(v);
    for (const auto &p : prev) { o.push_back(bits(p.x)); o.push_back(bits(p.y)); }
    return o;
}
template <class M> Result r_triangulation(Scene &S, bool only_stereo, bool ori) {
    M m(0.6f, ori);
    std::vector<std::pair<size_t, size_t> > pairs;
    Result o; o.push_back(m.SearchForTriangulation(&S.A, &S.B, fundamental(S.A, S.B), pairs, only_stereo));
    o.push_back((long)pairs.size()); for (const auto &p : pairs) { o.push_back((long)p.first); o.push_back((long)p.second); }
    return o;
}
template <class M> Result r_sim3(Scene &S, float s12, float th) {
    M m(0.75f, true);
    std::vector<MapPoint *> m12(S.A.mvpMapPoints.size(), nullptr);
    for (size_t i = 0; i < m12.size(); i += 9)                                  // a few matches known beforehand (from SearchByBoW)
        if (S.A.mvpMapPoints[i] && S.keyB_of[(size_t)S.A.mvpMapPoints[i]->mnId] >= 0) m12[i] = S.B.mvpMapPoints[(size_t)S.keyB_of[(size_t)S.A.mvpMapPoints[i]->mnId]];
    // [s12 R12 | t12]: camera 2 -> camera 1
    const cv::Mat R1w = S.A.GetRotation(), t1w = S.A.GetTranslation(), R2w = S.B.GetRotation(), t2w = S.B.GetTranslation();
    const cv::Mat R12 = R1w * cv::Mat(R2w.t());
    const cv::Mat t12 = cv::Mat(-R12 * t2w) + t1w;
    Result o; o.push_back(m.SearchBySim3(&S.A, &S.B, m12, s12, R12, t12, th)); put_points(o, m12); return o;
}
template <class M> Result r_fuse(Scene &S, float th) {
    M m(0.6f, true);
    SIVO::pin_log().clear();
    Result o; o.push_back(m.Fuse(&S.B, S.world, th)); put_log(o); put_state(o, S); return o;
}
template <class M> Result r_fuse_sim3(Scene &S, float s, float th) {
    M m(0.6f, true);
    SIVO::pin_log().clear();
    std::vector<MapPoint *> replace(S.world.size(), nullptr);
    Result o; o.push_back(m.Fuse(&S.B, sim3(S.B.mTcw, s), S.world, th, replace)); put_points(o, replace); put_log(o); put_state(o, S); return o;
}

struct Case { std::string name; SceneSpec spec; Result (*ref)(Scene &); Result (*mine)(Scene &); };

// the table: every routine on several scenes / parameter sets
#ifndef PIN_NO_REFERENCE
#define BOTH(expr_ref, expr_mine) [](Scene &S) { return expr_ref; }, [](Scene &S) { return expr_mine; }
typedef SIVO::RefORBmatcher R;
#else
#define BOTH(expr_ref, expr_mine) nullptr, [](Scene &S) { return expr_mine; }
#endif
typedef SIVO::ORBmatcher Mine;

static SceneSpec spec(uint64_t seed, float tz = 0.75f, int np = 700, int nproto = 160, bool l0 = false) {
    SceneSpec s; s.seed = seed; s.tz = tz; s.n_points = np; s.n_proto = nproto; s.level0_only = l0; return s;
}
// crowded and ambiguous: 1100 points that share 24 descriptor prototypes, keys up to 6 px off their projection, noisy
// descriptors -> windows hold several plausible candidates, the ratio tests reject, queries collide on one key
static SceneSpec hard(uint64_t seed, float tz = 0.75f, bool l0 = false) {
    SceneSpec s; s.seed = seed; s.tz = tz; s.n_points = 1100; s.n_clutter = 500; s.n_proto = 24; s.noise_px = 6.f; s.desc_noise = 40; s.level0_only = l0;
    return s;
}

static std::vector<Case> cases() {
    std::vector<Case> c;
    for (uint64_t seed : {11ull, 12ull}) {
        const std::string t = " seed " + std::to_string(seed);
        c.push_back({"local map th=1" + t, spec(seed), BOTH(r_local_map<R>(S, 1.f, 0.8f), r_local_map<Mine>(S, 1.f, 0.8f))});
        c.push_back({"local map th=5" + t, spec(seed), BOTH(r_local_map<R>(S, 5.f, 0.8f), r_local_map<Mine>(S, 5.f, 0.8f))});
        c.push_back({"frame forward th=7" + t, spec(seed), BOTH(r_frame<R>(S, 7.f, false, true), r_frame<Mine>(S, 7.f, false, true))});
        c.push_back({"frame backward th=15" + t, spec(seed, -0.8f), BOTH(r_frame<R>(S, 15.f, false, true), r_frame<Mine>(S, 15.f, false, true))});
        c.push_back({"frame mono th=15 no-ori" + t, spec(seed), BOTH(r_frame<R>(S, 15.f, true, false), r_frame<Mine>(S, 15.f, true, false))});
        c.push_back({"frame still th=7" + t, spec(seed, 0.2f), BOTH(r_frame<R>(S, 7.f, false, true), r_frame<Mine>(S, 7.f, false, true))});
        c.push_back({"reloc th=10 d=100" + t, spec(seed), BOTH(r_reloc<R>(S, 10.f, 100), r_reloc<Mine>(S, 10.f, 100))});
        c.push_back({"reloc th=3 d=64" + t, spec(seed), BOTH(r_reloc<R>(S, 3.f, 64), r_reloc<Mine>(S, 3.f, 64))});
        c.push_back({"loop s=1 th=10" + t, spec(seed), BOTH(r_loop<R>(S, 1.0f, 10), r_loop<Mine>(S, 1.0f, 10))});
        c.push_back({"loop s=1.03 th=4" + t, spec(seed), BOTH(r_loop<R>(S, 1.03f, 4), r_loop<Mine>(S, 1.03f, 4))});
        c.push_back({"bow kf-frame" + t, spec(seed), BOTH(r_bow_frame<R>(S, 0.75f, true), r_bow_frame<Mine>(S, 0.75f, true))});
        c.push_back({"bow kf-frame loose no-ori" + t, spec(seed, 0.75f, 700, 60), BOTH(r_bow_frame<R>(S, 0.95f, false), r_bow_frame<Mine>(S, 0.95f, false))});
        c.push_back({"bow kf-kf" + t, spec(seed), BOTH(r_bow_kf<R>(S, 0.75f, true), r_bow_kf<Mine>(S, 0.75f, true))});
        c.push_back({"bow kf-kf loose" + t, spec(seed, 0.75f, 700, 60), BOTH(r_bow_kf<R>(S, 0.95f, true), r_bow_kf<Mine>(S, 0.95f, true))});
        c.push_back({"initialization w=100" + t, spec(seed, 0.3f, 500, 120, true), BOTH(r_init<R>(S, 100, 0.9f), r_init<Mine>(S, 100, 0.9f))});
        c.push_back({"initialization w=30" + t, spec(seed, 0.3f, 500, 40, true), BOTH(r_init<R>(S, 30, 0.97f), r_init<Mine>(S, 30, 0.97f))});
        c.push_back({"triangulation" + t, spec(seed), BOTH(r_triangulation<R>(S, false, true), r_triangulation<Mine>(S, false, true))});
        c.push_back({"triangulation stereo-only no-ori" + t, spec(seed, 0.75f, 700, 60), BOTH(r_triangulation<R>(S, true, false), r_triangulation<Mine>(S, true, false))});
        c.push_back({"sim3 s=1 th=7.5" + t, spec(seed), BOTH(r_sim3<R>(S, 1.0f, 7.5f), r_sim3<Mine>(S, 1.0f, 7.5f))});
        c.push_back({"sim3 s=0.97 th=10" + t, spec(seed), BOTH(r_sim3<R>(S, 0.97f, 10.f), r_sim3<Mine>(S, 0.97f, 10.f))});
        c.push_back({"fuse th=3" + t, spec(seed), BOTH(r_fuse<R>(S, 3.f), r_fuse<Mine>(S, 3.f))});
        c.push_back({"fuse th=8 few prototypes" + t, spec(seed, 0.75f, 700, 40), BOTH(r_fuse<R>(S, 8.f), r_fuse<Mine>(S, 8.f))});
        c.push_back({"fuse sim3 s=1 th=4" + t, spec(seed), BOTH(r_fuse_sim3<R>(S, 1.0f, 4.f), r_fuse_sim3<Mine>(S, 1.0f, 4.f))});
        c.push_back({"fuse sim3 s=1.02 th=10" + t, spec(seed, 0.75f, 700, 40), BOTH(r_fuse_sim3<R>(S, 1.02f, 10.f), r_fuse_sim3<Mine>(S, 1.02f, 10.f))});
    }
    for (uint64_t seed : {21ull, 22ull, 23ull, 31ull, 32ull, 33ull}) {
        const bool clustered = seed > 30;
        const std::string t = (clustered ? " look-alikes seed " : " hard seed ") + std::to_string(seed);
        auto hard = [clustered](uint64_t sd, float tz = 0.75f, bool l0 = false) { SceneSpec s = ::hard(sd, tz, l0); if (clustered) { s.cluster = 5; s.n_proto = 200; s.noise_px = 3.f; s.desc_noise = 30; } return s; };
        c.push_back({"local map th=3" + t, hard(seed), BOTH(r_local_map<R>(S, 3.f, 0.8f), r_local_map<Mine>(S, 3.f, 0.8f))});
        c.push_back({"local map th=1 strict" + t, hard(seed), BOTH(r_local_map<R>(S, 1.f, 0.6f), r_local_map<Mine>(S, 1.f, 0.6f))});
        c.push_back({"frame th=7" + t, hard(seed), BOTH(r_frame<R>(S, 7.f, false, true), r_frame<Mine>(S, 7.f, false, true))});
        c.push_back({"frame th=15 backward" + t, hard(seed, -0.8f), BOTH(r_frame<R>(S, 15.f, false, true), r_frame<Mine>(S, 15.f, false, true))});
        c.push_back({"reloc th=10 d=100" + t, hard(seed), BOTH(r_reloc<R>(S, 10.f, 100), r_reloc<Mine>(S, 10.f, 100))});
        c.push_back({"reloc th=3 d=64" + t, hard(seed), BOTH(r_reloc<R>(S, 3.f, 64), r_reloc<Mine>(S, 3.f, 64))});
        c.push_back({"loop th=10" + t, hard(seed), BOTH(r_loop<R>(S, 1.0f, 10), r_loop<Mine>(S, 1.0f, 10))});
        c.push_back({"loop th=3" + t, hard(seed), BOTH(r_loop<R>(S, 1.01f, 3), r_loop<Mine>(S, 1.01f, 3))});
        c.push_back({"bow kf-frame" + t, hard(seed), BOTH(r_bow_frame<R>(S, 0.9f, true), r_bow_frame<Mine>(S, 0.9f, true))});
        c.push_back({"bow kf-kf" + t, hard(seed), BOTH(r_bow_kf<R>(S, 0.9f, true), r_bow_kf<Mine>(S, 0.9f, true))});
        c.push_back({"initialization w=60" + t, hard(seed, 0.3f, true), BOTH(r_init<R>(S, 60, 0.95f), r_init<Mine>(S, 60, 0.95f))});
        c.push_back({"triangulation" + t, hard(seed), BOTH(r_triangulation<R>(S, false, true), r_triangulation<Mine>(S, false, true))});
        c.push_back({"sim3 th=7.5" + t, hard(seed), BOTH(r_sim3<R>(S, 1.0f, 7.5f), r_sim3<Mine>(S, 1.0f, 7.5f))});
        c.push_back({"fuse th=3" + t, hard(seed), BOTH(r_fuse<R>(S, 3.f), r_fuse<Mine>(S, 3.f))});
        c.push_back({"fuse th=10" + t, hard(seed), BOTH(r_fuse<R>(S, 10.f), r_fuse<Mine>(S, 10.f))});
        c.push_back({"fuse sim3 th=6" + t, hard(seed), BOTH(r_fuse_sim3<R>(S, 1.0f, 6.f), r_fuse_sim3<Mine>(S, 1.0f, 6.f))});
    }
    return c;
}

// what a golden line keeps of a result: return value, number of values, FNV-1a over all of them
static Result digest(const Result &r) {
    uint64_t h = 1469598103934665603ull;
    for (long v : r)
        for (int b = 0; b < 8; ++b) { h ^= (uint64_t)((v >> (8 * b)) & 0xFF); h *= 1099511628211ull; }
    Result d; d.push_back(r.empty() ? -1 : r[0]); d.push_back((long)r.size()); d.push_back((long)(h >> 1));
    return d;
}

static void describe_difference(const Result &a, const Result &b) {
    std::printf("    sizes %zu / %zu, return values %ld / %ld\n", a.size(), b.size(), a.empty() ? -99 : a[0], b.empty() ? -99 : b[0]);
    int shown = 0;
    for (size_t i = 0; i < a.size() && i < b.size() && shown < 8; ++i)
        if (a[i] != b[i]) { std::printf("    [%zu] %ld / %ld\n", i, a[i], b[i]); ++shown; }
}

int main(int argc, char **argv) {
    std::string golden_in, golden_out;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--golden") && i + 1 < argc) golden_in = argv[++i];
        else if (!std::strcmp(argv[i], "--write-golden") && i + 1 < argc) golden_out = argv[++i];
    }
    std::map<std::string, Result> golden;
    if (!golden_in.empty()) {
        std::ifstream f(golden_in);
        if (!f) { std::printf("cannot read %s\n", golden_in.c_str()); return 2; }
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            const size_t bar = line.find('|');
            std::istringstream is(line.substr(bar + 1));
            Result r; long v;
            while (is >> v) r.push_back(v);
            golden[line.substr(0, bar)] = r;
        }
    }
#ifdef PIN_NO_REFERENCE
    if (golden_in.empty()) { std::printf("built without the reference: --golden FILE is required\n"); return 2; }
#endif
    std::ofstream out;
    if (!golden_out.empty()) {
        out.open(golden_out);
        out << "# Results of the reference's own src/orbslam/ORBmatcher.cc (compiled by oracle/Makefile against oracle/ref_shims) on the\n"
               "# scenes of tests/cpp/pin_matcher.cpp; written by `oracle/_ref/pin_matcher_cpu --write-golden`.\n"
               "# name|return value, number of result values, FNV-1a hash of all of them (return value, output vectors, mutation log)\n";
    }
    int failures = 0, total = 0, nonzero = 0;
#ifndef PIN_NO_REFERENCE
    {   // ORBmatcher::DescriptorDistance (ORBmatcher.cc:1579-1596) against SIVO::ORBmatcher::DescriptorDistance
        Rng r(5);
        cv::Mat a(1, 32, CV_8UC1), b(1, 32, CV_8UC1);
        int bad = 0;
        for (int t = 0; t < 4000; ++t) {
            for (int k = 0; k < 32; ++k) { a.data[k] = t == 0 ? 0 : t == 1 ? 255 : (uint8_t)r.u32(); b.data[k] = t < 2 ? 255 : t % 3 ? (uint8_t)r.u32() : a.data[k]; }
            if (t % 3 == 0) flip_bits(b.data, r.below(40), r);
            bad += SIVO::RefORBmatcher::DescriptorDistance(a, b) != SIVO::ORBmatcher::DescriptorDistance(a, b);
        }
        std::printf("%s DescriptorDistance on 4000 pairs\n", bad ? "FAIL" : "ok  ");
        failures += bad != 0;
    }
#endif
    for (const Case &c : cases()) {
        ++total;
        Result expect;
        bool have = false;
        if (c.ref) {
            std::unique_ptr<Scene> S = build_scene(c.spec);
            expect = c.ref(*S);
            have = true;
            if (out.is_open()) { out << c.name << "|"; for (long v : digest(expect)) out << v << ' '; out << "\n"; }
            if (!golden.empty()) {                         // the committed fixture must still be what the reference computes
                const auto it = golden.find(c.name);
                if (it == golden.end() || it->second != digest(expect)) { std::printf("STALE GOLDEN %s\n", c.name.c_str()); ++failures; }
            }
        } else {
            const auto it = golden.find(c.name);
            if (it != golden.end()) { expect = it->second; have = true; }
        }
        if (!have) { std::printf("NO EXPECTATION %s\n", c.name.c_str()); ++failures; continue; }
        std::unique_ptr<Scene> S = build_scene(c.spec);
        const Result got = c.ref ? c.mine(*S) : digest(c.mine(*S));
        nonzero += expect[0] > 0;
        if (got == expect) std::printf("ok   %-44s matches %4ld, %ld values\n", c.name.c_str(), expect[0], c.ref ? (long)expect.size() : expect[1]);
        else { std::printf("FAIL %s\n", c.name.c_str()); describe_difference(expect, got); ++failures; }
    }
    std::printf("%d cases, %d failures, %d with matches\n", total, failures, nonzero);
    if (!failures) std::printf("pin ok\n");
    return failures ? 1 : 0;
}
