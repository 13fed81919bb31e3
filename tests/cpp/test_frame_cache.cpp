// The matcher's frame views (sivo_amd/api/orbslam/ORBmatcher.h: FrameCache / FrameLease) on the CPU: the C ABI is the one over the
// oracle (abi_on_oracle.cpp), which counts the views it builds.  What is checked: a Frame / KeyFrame gets ONE view however often it is
// searched; a copy with cloned descriptors, another mnId, other keys or other descriptor bytes gets its own; the least recently used
// views are dropped beyond CAPACITY and are rebuilt when needed again; ReleaseDeviceFrames drops everything.
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "slam_standins.h"
#include "../../sivo_amd/api/orbslam/ORBmatcher.h"

extern "C" int abi_on_oracle_mframes_created(void);
extern "C" int abi_on_oracle_mframes_alive(void);

using namespace SIVO;

static int failures = 0;
#define CHECK(cond, ...)                                                      \
    do {                                                                      \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d %s: ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } \
    } while (0)

static void fill(KeyFrame &K, long id, int n, unsigned seed) {
    K.mnId = id;
    K.mnMinX = 0.f; K.mnMaxX = 1241.f; K.mnMinY = 0.f; K.mnMaxY = 376.f;
    K.mvScaleFactors.assign(8, 1.0f); K.mvLevelSigma2.assign(8, 1.0f); K.mvInvLevelSigma2.assign(8, 1.0f);
    K.mvKeysSemantic.resize((size_t)n); K.mvRight.assign((size_t)n, -1.f); K.mDescriptorsSemantic = cv::Mat(n, 32, CV_8UC1);
    unsigned s = seed * 2654435761u + 12345u;
    auto rnd = [&s] { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int i = 0; i < n; ++i) {
        cv::KeyPoint &kp = K.mvKeysSemantic[(size_t)i];
        kp.pt.x = (float)(rnd() % 1200u) + 10.f; kp.pt.y = (float)(rnd() % 340u) + 10.f; kp.octave = (int)(rnd() % 8u); kp.angle = (float)(rnd() % 360u);
        for (int b = 0; b < 32; ++b) K.mDescriptorsSemantic.ptr(i)[b] = (uint8_t)rnd();
    }
}

int main() {
    using matcher_detail::FrameCache;
    using matcher_detail::FrameLease;
    const int base = abi_on_oracle_mframes_created();
    KeyFrame A, B;
    fill(A, 1, 300, 1); fill(B, 2, 300, 2);
    for (int k = 0; k < 5; ++k) { FrameLease a(A); FrameLease b(B); CHECK(a.size() == 300 && b.size() == 300, "sizes"); }
    CHECK(abi_on_oracle_mframes_created() - base == 2, "two frames, five searches each: %d views", abi_on_oracle_mframes_created() - base);
    CHECK(FrameCache::instance().size() == 2, "cache holds %zu", FrameCache::instance().size());
    // the same content under another id, the same id with cloned (= other) descriptor storage, other descriptor bytes, other keys
    KeyFrame C; fill(C, 3, 300, 1);
    { FrameLease c(C); }
    KeyFrame A2 = A; A2.mDescriptorsSemantic = A.mDescriptorsSemantic.clone();
    { FrameLease a2(A2); }
    CHECK(abi_on_oracle_mframes_created() - base == 4, "new id / cloned descriptors: %d views", abi_on_oracle_mframes_created() - base);
    A.mDescriptorsSemantic.ptr(150)[7] ^= 0x10;           // a key in the middle of the fingerprint's sample
    { FrameLease a(A); }
    A.mvKeysSemantic[299].pt.x += 1.f;
    { FrameLease a(A); }
    CHECK(abi_on_oracle_mframes_created() - base == 6, "changed descriptor byte / changed key: %d views", abi_on_oracle_mframes_created() - base);
    { FrameLease a(A); FrameLease b(B); }
    CHECK(abi_on_oracle_mframes_created() - base == 6, "unchanged since: %d views", abi_on_oracle_mframes_created() - base);
    // least recently used views go beyond CAPACITY; one that is needed again is rebuilt
    const int cap = (int)FrameCache::CAPACITY;
    std::vector<std::unique_ptr<KeyFrame> > many;
    for (int i = 0; i < cap + 10; ++i) {
        many.emplace_back(new KeyFrame);
        fill(*many.back(), 100 + i, 40, 100u + (unsigned)i);
        FrameLease m(*many.back());
    }
    CHECK((int)FrameCache::instance().size() == cap, "cache holds %zu of %d", FrameCache::instance().size(), cap);
    CHECK(abi_on_oracle_mframes_alive() == cap, "%d views alive", abi_on_oracle_mframes_alive());
    const int before = abi_on_oracle_mframes_created();
    { FrameLease b(B); }                                   // evicted by now
    { FrameLease m(*many.back()); }                        // most recent: still there
    CHECK(abi_on_oracle_mframes_created() - before == 1, "one rebuilt, one found: %d", abi_on_oracle_mframes_created() - before);
    ORBmatcher::ReleaseDeviceFrames();
    CHECK(FrameCache::instance().size() == 0 && abi_on_oracle_mframes_alive() == 0, "released: %zu cached, %d alive", FrameCache::instance().size(), abi_on_oracle_mframes_alive());
    // a view in use survives its eviction
    {
        FrameLease a(A);
        ORBmatcher::ReleaseDeviceFrames();
        CHECK(abi_on_oracle_mframes_alive() == 1 && a.size() == 300, "in-use view: %d alive", abi_on_oracle_mframes_alive());
    }
    CHECK(abi_on_oracle_mframes_alive() == 0, "after the lease: %d alive", abi_on_oracle_mframes_alive());
    std::printf(failures ? "%d FAILURES\n" : "frame cache: ok\n", failures);
    return failures ? 1 : 0;
}
