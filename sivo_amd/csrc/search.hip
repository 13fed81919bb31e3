// search.hip — guided matching on CDNA4: the Search* / Fuse members of SIVO::ORBmatcher from the projected point on
// (reference src/orbslam/ORBmatcher.cc:44-127, 161-284, 286-399, 508-629, 631-785, 787-1053, 1055-1276, 1278-1543) and
// the frame grid they query (reference src/orbslam/Frame.cc:205-221, 326-390).
//
// Integer / latency-bound work: a frame holds <= a few thousand 32-byte descriptors (64 KB), a call runs <= a few
// thousand queries with 10-100 candidates each.  One wave per query: the grid cells a window covers are contiguous per
// grid column in a CSR laid out column-major like the reference's mGrid[ix][iy] walk, so a window is <= 2r/16+2 spans
// that the 64 lanes stride over (coalesced index loads, 2 x 16-byte descriptor loads per candidate, v_bcnt for the
// distance, wave-butterfly for best / second best on a (distance, visiting order) key, which reproduces the
// reference's first-minimum-wins scans bit for bit).
//
// Sequential semantics.  In the reference, iteration i of a routine skips keypoints matched by iterations < i.  Here all
// queries run at once against the initial state ("round 0"); accepted picks are registered with atomics, and only if
// two accepted queries picked the same keypoint do repair rounds follow, in which query i excludes the keypoints
// picked in the previous round by BLOCKING queries < i.  By induction query i is final after round i; the loop ends at
// the first round that changes nothing, whose state satisfies the sequential recurrence and is therefore the
// reference's result.  Typical calls end after round 0 (one launch + one 8-byte read-back).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>
#include <stdexcept>
#include <vector>

#include "common.hpp"

#define SIVO_GRID_COLS 64   // FRAME_GRID_COLS (reference include/orbslam/Frame.h:38-39)
#define SIVO_GRID_ROWS 48
#define SIVO_HISTO 30       // ORBmatcher::HISTO_LENGTH
#define SIVO_TH_HIGH 100
#define SIVO_TH_LOW 50

// Device side of a frame: ONE device allocation (keys, descriptors, grid, scale tables, then the engine's scratch), one pinned
// host buffer of the same layout to stage uploads and read results back through, one stream.  Slabs outlive the frames that
// use them: sivo_mframe_destroy hands the slab to a per-process pool and the next sivo_mframe_create on that device takes it
// back, so a tracking loop that builds a frame view per image allocates nothing after its first frames (hipMalloc +
// hipHostMalloc + hipStreamCreate are ~0.3 ms together, more than the searches they would serve).
struct MframeSlab {
    int device = 0;
    char *d = nullptr, *h = nullptr;           // device memory / pinned host mirror
    size_t cap = 0;
    hipStream_t stream = nullptr;
    ~MframeSlab() {
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct sivo_mframe {
    int device = 0, n = 0, nlevels = 0;
    float min_x = 0, max_x = 0, min_y = 0, max_y = 0, inv_w = 0, inv_h = 0;
    // host copies (GetFeaturesInArea on the host, wrappers)
    std::vector<SivoKeyPoint> keys;
    std::vector<float> u_right, scale, sigma2, inv_sigma2;
    std::vector<int32_t> cell_off, cell_idx;
    // device (offsets into the slab)
    std::unique_ptr<MframeSlab> slab;
    size_t frame_bytes = 0;                    // the frame's own arrays end here; the engine's scratch follows
    float *d_x = nullptr, *d_y = nullptr, *d_angle = nullptr, *d_ur = nullptr, *d_scale = nullptr, *d_sigma2 = nullptr,
          *d_inv_sigma2 = nullptr;
    int32_t *d_oct = nullptr, *d_cell_off = nullptr, *d_cell_idx = nullptr;
    uint4 *d_desc = nullptr;
    hipStream_t stream = nullptr;
    ~sivo_mframe();
};

namespace sivo {
namespace {

struct SearchArgs {
    // train frame
    int n;
    const float *x, *y, *angle, *ur, *scale, *sigma2, *inv_sigma2;
    const int32_t *oct, *cell_off, *cell_idx;
    const uint4 *desc;
    float min_x, min_y, inv_w, inv_h;
    // queries
    int nq;
    const SivoSearchQuery *q;
    const uint4 *qdesc;
    const int32_t *cbeg, *cend, *cidx;
    SivoSearchRule rule;
    const uint8_t *blocked;
    const int32_t *owner_prev;   // min blocking accepted query that picked k in the previous round (INT_MAX none); null in round 0
    int32_t *owner_next;
    int32_t *npick;              // accepted picks per keypoint this round
    int32_t *pick, *bdist, *sdist;   // per query: accepted keypoint or -1, best / second distance
    int32_t *flags;              // [0] collision (two accepted picks of one keypoint), [1] a pick changed
};

__device__ __forceinline__ int ham256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// top-2 of a (key, index) multiset; key = dist << 22 | visiting order (or its complement when the last candidate wins)
struct Top2 { uint32_t k1, k2; int i1, i2; };
constexpr uint32_t KEY_NONE = 0xffffffffu;
__device__ __forceinline__ void top2_push(Top2 &t, uint32_t k, int i) {
    if (k < t.k1) { t.k2 = t.k1; t.i2 = t.i1; t.k1 = k; t.i1 = i; }
    else if (k < t.k2) { t.k2 = k; t.i2 = i; }
}

__global__ __launch_bounds__(256) void search_round_kernel(SearchArgs a) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= a.nq) return;
    const SivoSearchQuery Q = a.q[qi];
    Top2 t{KEY_NONE, KEY_NONE, -1, -1};
    if (Q.flags & SIVO_Q_VALID) {
        const uint4 q0 = a.qdesc[(int64_t)qi * 2], q1 = a.qdesc[(int64_t)qi * 2 + 1];
        const int gate = a.rule.gate_mode;
        // one candidate (lane-local): static gates, dynamic exclusion, distance
        auto visit = [&](int idx, uint32_t ord) {
            if (a.blocked && a.blocked[idx]) return;
            if (a.owner_prev && a.owner_prev[idx] < qi) return;
            const int lvl = a.oct[idx];
            if (lvl < Q.lvl_lo || lvl > Q.lvl_hi) return;
            const float kx = a.x[idx], ky = a.y[idx];
            if (!a.cidx) {
                // GetFeaturesInArea (Frame.cc:379-384)
                const float distx = kx - Q.u, disty = ky - Q.v;
                if (!(fabsf(distx) < Q.radius && fabsf(disty) < Q.radius)) return;
            }
            const float kur = a.ur ? a.ur[idx] : -1.0f;
            if (gate == 1) {
                if (kur > 0) {
                    const float er = fabsf(Q.ur - kur);
                    if (er > Q.gate) return;
                }
            } else if (gate == 2) {
                // Fuse (ORBmatcher.cc:880-902)
                const float ex = Q.u - kx, ey = Q.v - ky;
                if (kur >= 0) {
                    const float er = Q.ur - kur;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if ((double)(e2 * a.inv_sigma2[lvl]) > 7.8) return;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if ((double)(e2 * a.inv_sigma2[lvl]) > 5.99) return;
                }
            }
            const int d = ham256(q0, q1, a.desc[(int64_t)idx * 2], a.desc[(int64_t)idx * 2 + 1]);
            if (gate == 3) {
                // SearchForTriangulation (ORBmatcher.cc:703-719)
                if (d > a.rule.th_dist) return;
                const bool st1 = (Q.flags & SIVO_Q_STEREO) != 0, st2 = kur >= 0;
                if (!st1 && !st2) {
                    const float distex = a.rule.ex - kx, distey = a.rule.ey - ky;
                    if (distex * distex + distey * distey < 100 * a.scale[lvl]) return;
                }
                const float *F = a.rule.F12;
                const float ea = Q.u * F[0] + Q.v * F[3] + F[6];
                const float eb = Q.u * F[1] + Q.v * F[4] + F[7];
                const float ec = Q.u * F[2] + Q.v * F[5] + F[8];
                const float num = ea * kx + eb * ky + ec;
                const float den = ea * ea + eb * eb;
                if (den == 0) return;
                const float dsqr = num * num / den;
                if (!((double)dsqr < 3.84 * (double)a.sigma2[lvl])) return;
            }
            if (d >= 256) return;                        // bestDist starts at 256 and the test is `dist < bestDist`
            const uint32_t o = a.rule.tie_last ? (0x3fffffu - ord) : ord;
            top2_push(t, ((uint32_t)d << 22) | o, idx);
        };
        if (a.cidx) {
            const int s = a.cbeg[qi], e = a.cend[qi];
            for (int j = s + lane; j < e; j += 64) visit(a.cidx[j], (uint32_t)(j - s));
        } else {
            // window -> cell ranges, exactly as Frame::GetFeaturesInArea (Frame.cc:334-357)
            int cx0 = max(0, (int)floorf((Q.u - a.min_x - Q.radius) * a.inv_w));
            int cx1 = min(SIVO_GRID_COLS - 1, (int)ceilf((Q.u - a.min_x + Q.radius) * a.inv_w));
            int cy0 = max(0, (int)floorf((Q.v - a.min_y - Q.radius) * a.inv_h));
            int cy1 = min(SIVO_GRID_ROWS - 1, (int)ceilf((Q.v - a.min_y + Q.radius) * a.inv_h));
            if (cx0 < SIVO_GRID_COLS && cx1 >= 0 && cy0 < SIVO_GRID_ROWS && cy1 >= 0) {
                uint32_t base = 0;
                for (int ix = cx0; ix <= cx1; ++ix) {
                    const int s = a.cell_off[ix * SIVO_GRID_ROWS + cy0], e = a.cell_off[ix * SIVO_GRID_ROWS + cy1 + 1];
                    for (int j = s + lane; j < e; j += 64) visit(a.cell_idx[j], base + (uint32_t)(j - s));
                    base += (uint32_t)(e - s);
                }
            }
        }
    }
    // butterfly: merge the lanes' top-2 sets
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ok1 = __shfl_xor(t.k1, off), ok2 = __shfl_xor(t.k2, off);
        const int oi1 = __shfl_xor(t.i1, off), oi2 = __shfl_xor(t.i2, off);
        top2_push(t, ok1, oi1);
        top2_push(t, ok2, oi2);
    }
    if (lane != 0) return;
    const int best = t.i1 >= 0 ? (int)(t.k1 >> 22) : 256, second = t.i2 >= 0 ? (int)(t.k2 >> 22) : 256;
    bool accept = t.i1 >= 0 && (a.rule.accept_lt ? best < a.rule.th_dist : best <= a.rule.th_dist);
    if (accept && a.rule.ratio_mode == 1) {
        const int bestLevel = a.oct[t.i1], bestLevel2 = t.i2 >= 0 ? a.oct[t.i2] : -1;
        if (bestLevel == bestLevel2 && (float)best > a.rule.nn_ratio * (float)second) accept = false;
    } else if (accept && a.rule.ratio_mode == 2) {
        if (!((float)best < a.rule.nn_ratio * (float)second)) accept = false;
    }
    const int p = accept ? t.i1 : -1;
    if (a.pick[qi] != p) a.flags[1] = 1;
    a.pick[qi] = p;
    a.bdist[qi] = best;
    a.sdist[qi] = second;
    if (p >= 0 && a.rule.dynamic) {
        if (atomicAdd(&a.npick[p], 1) > 0) a.flags[0] = 1;
        if (Q.flags & SIVO_Q_BLOCKS) atomicMin(&a.owner_next[p], qi);
    }
}

struct FinalArgs {
    int n, nq;
    const SivoSearchQuery *q;
    const float *angle;
    const int32_t *pick;
    int check_orientation;
    int32_t *match_query, *match_train;   // device outputs (nq / n)
    int32_t *counters;                    // [0] accepted, [1] culled by the rotation check
};

// One workgroup: rotation histogram (ORBmatcher.cc:1376-1412), three maxima (:1545-1577), final slot owners.
__global__ __launch_bounds__(1024) void search_finalize_kernel(FinalArgs a) {
    __shared__ int hist[SIVO_HISTO];
    __shared__ int keep[SIVO_HISTO];
    const int tid = threadIdx.x;
    if (tid < SIVO_HISTO) { hist[tid] = 0; keep[tid] = 1; }
    if (tid < 2) a.counters[tid] = 0;
    for (int k = tid; k < a.n; k += 1024) a.match_train[k] = -1;
    __syncthreads();
    const float factor = 1.0f / SIVO_HISTO;
    int accepted = 0;
    for (int qi = tid; qi < a.nq; qi += 1024) {
        const int p = a.pick[qi];
        if (p < 0) continue;
        ++accepted;
        if (a.check_orientation) {
            float rot = a.q[qi].angle - a.angle[p];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == SIVO_HISTO) bin = 0;
            atomicAdd(&hist[bin], 1);
        }
    }
    if (accepted) atomicAdd(&a.counters[0], accepted);
    __syncthreads();
    if (tid == 0 && a.check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < SIVO_HISTO; ++i) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < SIVO_HISTO; ++i) keep[i] = (i == ind1 || i == ind2 || i == ind3);
    }
    __syncthreads();
    // slot owners: the blocking picker if there is one (no later query can take the slot), else the last non-blocking one
    for (int qi = tid; qi < a.nq; qi += 1024) {
        const int p = a.pick[qi];
        if (p >= 0 && !(a.q[qi].flags & SIVO_Q_BLOCKS)) atomicMax(&a.match_train[p], qi);
    }
    __syncthreads();
    for (int qi = tid; qi < a.nq; qi += 1024) {
        const int p = a.pick[qi];
        if (p >= 0 && (a.q[qi].flags & SIVO_Q_BLOCKS)) a.match_train[p] = qi;
    }
    __syncthreads();
    int culled = 0;
    for (int qi = tid; qi < a.nq; qi += 1024) {
        const int p = a.pick[qi];
        int m = p;
        if (p >= 0 && a.check_orientation) {
            float rot = a.q[qi].angle - a.angle[p];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == SIVO_HISTO) bin = 0;
            if (!keep[bin]) { m = -1; ++culled; a.match_train[p] = -2; }
        }
        a.match_query[qi] = m;
    }
    if (culled) atomicAdd(&a.counters[1], culled);
}

template <class T>
T *carve(char *&p, size_t count) {
    T *r = reinterpret_cast<T *>(p);
    p += (count * sizeof(T) + 255) / 256 * 256;
    return r;
}

// ---- slab pool ------------------------------------------------------------------------------------------------------
struct SlabPool {
    std::mutex mu;
    std::vector<std::unique_ptr<MframeSlab>> idle;
    static constexpr size_t KEEP = 64;          // a local map's keyframes + the tracking frames; beyond that slabs are freed
    std::unique_ptr<MframeSlab> take(int device, size_t bytes) {
        {
            std::lock_guard<std::mutex> lock(mu);
            int best = -1;
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i]->device == device && idle[i]->cap >= bytes && (best < 0 || idle[i]->cap < idle[(size_t)best]->cap)) best = (int)i;
            if (best >= 0) {
                std::unique_ptr<MframeSlab> r = std::move(idle[(size_t)best]);
                idle.erase(idle.begin() + best);
                return r;
            }
        }
        std::unique_ptr<MframeSlab> r(new MframeSlab);
        r->device = device;
        r->cap = std::max(bytes, (size_t)1 << 20);
        SIVO_HIP(hipMalloc((void **)&r->d, r->cap));
        SIVO_HIP(hipHostMalloc((void **)&r->h, r->cap, hipHostMallocDefault));
        SIVO_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        return r;
    }
    void give(std::unique_ptr<MframeSlab> sl) {
        if (!sl) return;
        (void)hipStreamSynchronize(sl->stream);      // (an upload nobody searched against may still be in flight)
        std::lock_guard<std::mutex> lock(mu);
        if (idle.size() < KEEP) idle.push_back(std::move(sl));
    }
};
SlabPool &slab_pool() {
    static SlabPool *p = new SlabPool;           // never destroyed: frames may be released after main() returns
    return *p;
}

// Lays the frame's arrays out in the slab (device pointers into F, the same offsets in the pinned mirror); returns the bytes used.
size_t bind_frame(sivo_mframe &F, char *base) {
    char *p = base;
    const size_t n = (size_t)F.n;
    F.d_x = carve<float>(p, n); F.d_y = carve<float>(p, n); F.d_angle = carve<float>(p, n); F.d_oct = carve<int32_t>(p, n);
    F.d_ur = carve<float>(p, F.u_right.size());
    F.d_desc = carve<uint4>(p, n * 2);
    F.d_cell_off = carve<int32_t>(p, F.cell_off.size()); F.d_cell_idx = carve<int32_t>(p, F.cell_idx.size());
    F.d_scale = carve<float>(p, (size_t)F.nlevels); F.d_sigma2 = carve<float>(p, (size_t)F.nlevels); F.d_inv_sigma2 = carve<float>(p, (size_t)F.nlevels);
    return (size_t)(p - base);
}

// The engine's scratch behind the frame's arrays; a call that needs more than the slab holds moves the frame to a bigger slab
// (the pinned mirror still holds the frame's arrays: one copy host to host, one upload).
char *scratch(sivo_mframe &F, size_t bytes) {
    if (F.frame_bytes + bytes > F.slab->cap) {
        std::unique_ptr<MframeSlab> big = slab_pool().take(F.device, (F.frame_bytes + bytes) * 2);
        SIVO_HIP(hipStreamSynchronize(F.slab->stream));
        std::memcpy(big->h, F.slab->h, F.frame_bytes);
        SIVO_HIP(hipMemcpyAsync(big->d, big->h, F.frame_bytes, hipMemcpyHostToDevice, big->stream));
        slab_pool().give(std::move(F.slab));
        F.slab = std::move(big);
        F.stream = F.slab->stream;
        bind_frame(F, F.slab->d);
    }
    return F.slab->d + F.frame_bytes;
}

// The engine on host arrays: upload queries, rounds, finalize, download.
void run_search(sivo_mframe &F, const SivoSearchQuery *queries, const uint8_t *query_desc, int nq, const int32_t *cand_begin,
                const int32_t *cand_end, const int32_t *cand_idx, int n_cand, const SivoSearchRule &rule, const uint8_t *blocked,
                int32_t *match_query, int32_t *match_train, int32_t *best_dist, int32_t *second_dist, int *n_matches,
                int *rounds_out) {
    if (rounds_out) *rounds_out = 0;
    if (n_matches) *n_matches = 0;
    if (match_train) for (int k = 0; k < F.n; ++k) match_train[k] = -1;
    if (nq <= 0) return;
    if (nq >= (1 << 22) || F.n >= (1 << 22) || n_cand >= (1 << 22)) throw std::invalid_argument("sivo_search: at most 2^22 - 1 queries / keypoints / list entries");
    // explicit candidate lists index the frame's keypoints: a stale or foreign FeatureVector entry would be an out-of-bounds
    // read of descriptors / keys and, through the pick, an out-of-bounds atomic on the scratch buffer
    if (cand_idx)
        for (int j = 0; j < n_cand; ++j)
            if ((uint32_t)cand_idx[j] >= (uint32_t)F.n) throw std::invalid_argument("sivo_search: candidate index outside the frame's keypoints");
    DeviceGuard dg(F.device);
    const int n = F.n;
    const size_t need = 256 * 20 + (size_t)nq * (sizeof(SivoSearchQuery) + 32 + 8 + 4 * 4) + (size_t)(cand_idx ? n_cand : 0) * 4 +
                        (size_t)(n + 64) * (1 + 4 * 4) + 1024;
    // Scratch = [ upload block | device-only | download block ], the pinned mirror has the same layout: ONE copy up (queries,
    // their descriptors, candidate lists, blocked flags and the initial values of pick / owner / npick / flags), one copy down.
    char *const d0 = scratch(F, need);
    char *const h0 = F.slab->h + (d0 - F.slab->d);
    char *p = d0;
    SivoSearchQuery *d_q = carve<SivoSearchQuery>(p, nq);
    uint4 *d_qdesc = carve<uint4>(p, (size_t)nq * 2);
    int32_t *d_cbeg = carve<int32_t>(p, nq), *d_cend = carve<int32_t>(p, nq);
    int32_t *d_cidx = carve<int32_t>(p, cand_idx ? n_cand : 0);
    uint8_t *d_blocked = carve<uint8_t>(p, n);
    int32_t *d_pick = carve<int32_t>(p, nq);
    int32_t *d_owner0 = carve<int32_t>(p, rule.dynamic ? n : 0), *d_npick = carve<int32_t>(p, rule.dynamic ? n : 0);
    int32_t *d_flags = carve<int32_t>(p, 4);     // collision, changed, accepted, culled
    const size_t up_bytes = (size_t)(p - d0);
    int32_t *d_owner1 = carve<int32_t>(p, rule.dynamic ? n : 0);
    if (!rule.dynamic) d_npick = carve<int32_t>(p, n);
    char *const down0 = p;
    int32_t *d_counters = carve<int32_t>(p, 4);
    int32_t *d_mq = carve<int32_t>(p, nq), *d_mtrain = carve<int32_t>(p, n), *d_bd = carve<int32_t>(p, nq), *d_sd = carve<int32_t>(p, nq);
    const size_t down_bytes = (size_t)(p - down0);
    int32_t *d_owner[2] = {d_owner0, d_owner1};
    auto host = [&](auto *dev) { return reinterpret_cast<std::remove_reference_t<decltype(*dev)> *>(h0 + ((char *)dev - d0)); };
    hipStream_t st = F.stream;
    SIVO_HIP(hipStreamSynchronize(st));          // (the mirror may still feed the frame's own upload)
    std::memcpy(host(d_q), queries, (size_t)nq * sizeof(SivoSearchQuery));
    std::memcpy(host(d_qdesc), query_desc, (size_t)nq * 32);
    if (cand_idx) {
        std::memcpy(host(d_cbeg), cand_begin, (size_t)nq * 4);
        std::memcpy(host(d_cend), cand_end, (size_t)nq * 4);
        if (n_cand) std::memcpy(host(d_cidx), cand_idx, (size_t)n_cand * 4);
    }
    if (blocked && n) std::memcpy(host(d_blocked), blocked, (size_t)n);
    std::memset(host(d_pick), 0xff, (size_t)nq * 4);        // -1
    if (rule.dynamic && n) {
        std::memset(host(d_owner0), 0x7f, (size_t)n * 4);    // 0x7f7f7f7f > any query index
        std::memset(host(d_npick), 0, (size_t)n * 4);
    }
    std::memset(host(d_flags), 0, 16);
    SIVO_HIP(hipMemcpyAsync(d0, h0, up_bytes, hipMemcpyHostToDevice, st));

    SearchArgs a{};
    a.n = n; a.x = F.d_x; a.y = F.d_y; a.angle = F.d_angle; a.ur = F.u_right.empty() ? nullptr : F.d_ur;
    a.scale = F.d_scale; a.sigma2 = F.d_sigma2; a.inv_sigma2 = F.d_inv_sigma2;
    a.oct = F.d_oct; a.cell_off = F.d_cell_off; a.cell_idx = F.d_cell_idx; a.desc = F.d_desc;
    a.min_x = F.min_x; a.min_y = F.min_y; a.inv_w = F.inv_w; a.inv_h = F.inv_h;
    a.nq = nq; a.q = d_q; a.qdesc = d_qdesc;
    a.cbeg = d_cbeg; a.cend = d_cend; a.cidx = cand_idx ? d_cidx : nullptr;
    a.rule = rule; a.blocked = blocked ? d_blocked : nullptr;
    a.npick = d_npick; a.pick = d_pick; a.bdist = d_bd; a.sdist = d_sd; a.flags = d_flags;
    int rounds = 0;
    for (;; ++rounds) {
        a.owner_prev = rounds == 0 ? nullptr : d_owner[(rounds + 1) & 1];
        a.owner_next = d_owner[rounds & 1];
        if (rounds && rule.dynamic && n) {                                         // (round 0's values came with the upload)
            SIVO_HIP(hipMemsetAsync(a.owner_next, 0x7f, (size_t)n * 4, st));
            SIVO_HIP(hipMemsetAsync(d_npick, 0, (size_t)n * 4, st));
        }
        if (rounds) SIVO_HIP(hipMemsetAsync(d_flags, 0, 8, st));
        hipLaunchKernelGGL(search_round_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, st, a);
        if (!rule.dynamic) { ++rounds; break; }
        int32_t *fl = host(d_flags);
        SIVO_HIP(hipMemcpyAsync(fl, d_flags, 8, hipMemcpyDeviceToHost, st));
        SIVO_HIP(hipStreamSynchronize(st));
        // round 0 is final when no keypoint was picked twice — unless a ratio test is on: a keypoint an earlier query took
        // may have been this query's SECOND best, so a confirming round always runs then
        if (rounds == 0 ? (!fl[0] && rule.ratio_mode == 0) : !fl[1]) { ++rounds; break; }
        if (rounds > nq + 1) throw std::runtime_error("sivo_search: the repair rounds did not converge");
    }
    FinalArgs f{};
    f.n = n; f.nq = nq; f.q = d_q; f.angle = F.d_angle; f.pick = d_pick; f.check_orientation = rule.check_orientation;
    f.match_query = d_mq; f.match_train = d_mtrain; f.counters = d_counters;
    hipLaunchKernelGGL(search_finalize_kernel, dim3(1), dim3(1024), 0, st, f);
    SIVO_HIP(hipMemcpyAsync(h0 + (down0 - d0), down0, down_bytes, hipMemcpyDeviceToHost, st));
    SIVO_HIP(hipStreamSynchronize(st));
    SIVO_HIP(hipGetLastError());
    const int32_t *cnt = host(d_counters);
    if (match_query) std::memcpy(match_query, host(d_mq), (size_t)nq * 4);
    if (match_train && n) std::memcpy(match_train, host(d_mtrain), (size_t)n * 4);
    if (best_dist) std::memcpy(best_dist, host(d_bd), (size_t)nq * 4);
    if (second_dist) std::memcpy(second_dist, host(d_sd), (size_t)nq * 4);
    if (n_matches) *n_matches = cnt[0] - cnt[1];
    if (rounds_out) *rounds_out = rounds;
}

// Frame::GetFeaturesInArea (Frame.cc:326-390) on the host copy of the grid.
void features_in_area(const sivo_mframe &F, float x, float y, float r, int minLevel, int maxLevel, std::vector<int32_t> &out) {
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - F.min_x - r) * F.inv_w));
    if (nMinCellX >= SIVO_GRID_COLS) return;
    const int nMaxCellX = std::min(SIVO_GRID_COLS - 1, (int)std::ceil((x - F.min_x + r) * F.inv_w));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - F.min_y - r) * F.inv_h));
    if (nMinCellY >= SIVO_GRID_ROWS) return;
    const int nMaxCellY = std::min(SIVO_GRID_ROWS - 1, (int)std::ceil((y - F.min_y + r) * F.inv_h));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int j = F.cell_off[ix * SIVO_GRID_ROWS + iy]; j < F.cell_off[ix * SIVO_GRID_ROWS + iy + 1]; ++j) {
                const SivoKeyPoint &kp = F.keys[F.cell_idx[j]];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) out.push_back(F.cell_idx[j]);
            }
}

// GetFeaturesInArea(min, max) tests `octave < min` only when (min > 0 || max >= 0) and `octave > max` only when
// max >= 0; with octave >= 0 that is the plain range [min, max >= 0 ? max : inf).
void level_range(int minLevel, int maxLevel, int32_t &lo, int32_t &hi) {
    lo = minLevel;
    hi = maxLevel >= 0 ? maxLevel : INT_MAX;
}

SivoSearchRule make_rule(int th, int gate_mode, int ratio_mode, float nn, bool ori, bool dynamic) {
    SivoSearchRule r;
    std::memset(&r, 0, sizeof r);
    r.th_dist = th; r.gate_mode = gate_mode; r.ratio_mode = ratio_mode; r.nn_ratio = nn;
    r.check_orientation = ori ? 1 : 0; r.dynamic = dynamic ? 1 : 0;
    return r;
}

void require(bool ok, const char *what) {
    if (!ok) throw std::invalid_argument(what);
}

}  // namespace
}  // namespace sivo

using namespace sivo;

sivo_mframe::~sivo_mframe() { slab_pool().give(std::move(slab)); }

extern "C" int sivo_mframe_create(const SivoKeyPoint *keys, int n, const float *u_right, const uint8_t *descriptors,
                                  float min_x, float max_x, float min_y, float max_y, const float *scale_factors,
                                  const float *level_sigma2, const float *inv_level_sigma2, int nlevels, int device,
                                  sivo_mframe_t *out) {
    return guarded([&] {
        require(out != nullptr, "out is NULL");
        *out = nullptr;
        require(n >= 0 && (n == 0 || (keys && descriptors)), "keys / descriptors are NULL");
        require(nlevels > 0 && scale_factors && level_sigma2 && inv_level_sigma2, "scale tables are NULL");
        require(max_x > min_x && max_y > min_y, "empty image bounds");
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device: libsivo_hip has no CPU fallback");
        if (device < 0) SIVO_HIP(hipGetDevice(&device));                    // -1: the calling thread's current device
        if (sivo_device_count() <= device)
            return fail(SIVO_ERR_RUNTIME, "HIP device %d is not available (%d visible): libsivo_hip has no CPU fallback", device,
                        sivo_device_count());
        std::unique_ptr<sivo_mframe> F(new sivo_mframe);
        F->device = device; F->n = n; F->nlevels = nlevels;
        F->min_x = min_x; F->max_x = max_x; F->min_y = min_y; F->max_y = max_y;
        F->inv_w = (float)SIVO_GRID_COLS / (max_x - min_x);
        F->inv_h = (float)SIVO_GRID_ROWS / (max_y - min_y);
        F->keys.assign(keys, keys + n);
        if (u_right) F->u_right.assign(u_right, u_right + n);
        F->scale.assign(scale_factors, scale_factors + nlevels);
        F->sigma2.assign(level_sigma2, level_sigma2 + nlevels);
        F->inv_sigma2.assign(inv_level_sigma2, inv_level_sigma2 + nlevels);
        // AssignFeaturesToGrid (Frame.cc:205-221) + PosInGrid (:392-404) -> CSR in mGrid[ix][iy] order
        const int cells = SIVO_GRID_COLS * SIVO_GRID_ROWS;
        std::vector<int32_t> cell_of((size_t)n, -1);
        F->cell_off.assign((size_t)cells + 1, 0);
        for (int i = 0; i < n; ++i) {
            require(keys[i].octave >= 0 && keys[i].octave < nlevels, "keypoint octave outside the scale tables");
            const int px = (int)std::round((keys[i].x - min_x) * F->inv_w), py = (int)std::round((keys[i].y - min_y) * F->inv_h);
            if (px < 0 || px >= SIVO_GRID_COLS || py < 0 || py >= SIVO_GRID_ROWS) continue;
            cell_of[i] = px * SIVO_GRID_ROWS + py;
            ++F->cell_off[cell_of[i] + 1];
        }
        for (int c = 0; c < cells; ++c) F->cell_off[c + 1] += F->cell_off[c];
        F->cell_idx.assign((size_t)F->cell_off[cells], 0);
        std::vector<int32_t> fill(F->cell_off.begin(), F->cell_off.end() - 1);
        for (int i = 0; i < n; ++i)
            if (cell_of[i] >= 0) F->cell_idx[fill[cell_of[i]]++] = i;

        DeviceGuard dg(device);
        // one slab: the frame's arrays + scratch for a call with as many queries as the frame has keys (a bigger call grows it)
        const size_t frame_bytes = bind_frame(*F, reinterpret_cast<char *>((uintptr_t)4096));
        const size_t scratch_guess = 256 * 20 + (size_t)(n + 64) * (sizeof(SivoSearchQuery) + 32 + 8 + 16 + 17) + 1024;
        F->slab = slab_pool().take(device, frame_bytes + scratch_guess);
        F->stream = F->slab->stream;
        F->frame_bytes = frame_bytes;
        bind_frame(*F, F->slab->d);
        char *const h = F->slab->h, *const d = F->slab->d;
        auto host = [&](auto *dev) { return reinterpret_cast<std::remove_reference_t<decltype(*dev)> *>(h + ((char *)dev - d)); };
        float *hx = host(F->d_x), *hy = host(F->d_y), *ha = host(F->d_angle);
        int32_t *ho = host(F->d_oct);
        for (int i = 0; i < n; ++i) { hx[i] = keys[i].x; hy[i] = keys[i].y; ha[i] = keys[i].angle; ho[i] = keys[i].octave; }
        if (!F->u_right.empty()) std::memcpy(host(F->d_ur), F->u_right.data(), F->u_right.size() * 4);
        if (n) std::memcpy(host(F->d_desc), descriptors, (size_t)n * 32);
        std::memcpy(host(F->d_cell_off), F->cell_off.data(), F->cell_off.size() * 4);
        if (!F->cell_idx.empty()) std::memcpy(host(F->d_cell_idx), F->cell_idx.data(), F->cell_idx.size() * 4);
        std::memcpy(host(F->d_scale), F->scale.data(), (size_t)nlevels * 4);
        std::memcpy(host(F->d_sigma2), F->sigma2.data(), (size_t)nlevels * 4);
        std::memcpy(host(F->d_inv_sigma2), F->inv_sigma2.data(), (size_t)nlevels * 4);
        // (asynchronous: every search of this frame runs on the same stream, behind the upload)
        SIVO_HIP(hipMemcpyAsync(d, h, frame_bytes, hipMemcpyHostToDevice, F->stream));
        *out = F.release();
        return SIVO_OK;
    });
}

extern "C" int sivo_mframe_destroy(sivo_mframe_t h) {
    return guarded([&] {
        if (h) {
            DeviceGuard dg(h->device);
            delete h;
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_mframe_features_in_area(sivo_mframe_t h, float x, float y, float r, int min_level, int max_level,
                                            int32_t *out, int capacity, int *n_out) {
    return guarded([&] {
        require(h && n_out, "null argument");
        std::vector<int32_t> v;
        features_in_area(*h, x, y, r, min_level, max_level, v);
        *n_out = (int)v.size();
        if (!out) return SIVO_OK;
        if (capacity < (int)v.size()) return fail(SIVO_ERR_CAPACITY, "%zu indices, capacity %d", v.size(), capacity);
        std::memcpy(out, v.data(), v.size() * 4);
        return SIVO_OK;
    });
}

extern "C" int sivo_search(sivo_mframe_t train, const SivoSearchQuery *queries, const uint8_t *query_desc, int n_queries,
                           const int32_t *cand_begin, const int32_t *cand_end, const int32_t *cand_idx, int n_cand,
                           const SivoSearchRule *rule, const uint8_t *blocked, int32_t *match_query, int32_t *match_train,
                           int32_t *best_dist, int32_t *second_dist, int *n_matches, int *rounds) {
    return guarded([&] {
        require(train && rule && n_queries >= 0 && (n_queries == 0 || (queries && query_desc)), "null argument");
        require(!cand_idx || (cand_begin && cand_end), "explicit candidate lists need cand_begin and cand_end");
        if (cand_idx)
            for (int q = 0; q < n_queries; ++q)
                require(cand_begin[q] >= 0 && cand_begin[q] <= cand_end[q] && cand_end[q] <= n_cand, "candidate range outside the list");
        run_search(*train, queries, query_desc, n_queries, cand_begin, cand_end, cand_idx, n_cand, *rule, blocked, match_query,
                   match_train, best_dist, second_dist, n_matches, rounds);
        return SIVO_OK;
    });
}

// ---- the reference routines on arrays -------------------------------------------------------------------------------

extern "C" int sivo_search_by_projection_mappoints(sivo_mframe_t F, int n_mp, const uint8_t *track_in_view, const float *proj_x,
                                                   const float *proj_y, const float *proj_xr, const int32_t *level,
                                                   const float *view_cos, const uint8_t *mp_desc, const int32_t *mp_obs,
                                                   float th, float nn_ratio, int32_t *occ_obs, int32_t *match, int *n_matches) {
    return guarded([&] {
        require(F && n_mp >= 0 && occ_obs && match, "null argument");
        require(n_mp == 0 || (track_in_view && proj_x && proj_y && proj_xr && level && view_cos && mp_desc && mp_obs), "null argument");
        const bool bFactor = th != 1.0;
        std::vector<SivoSearchQuery> q((size_t)n_mp);
        for (int i = 0; i < n_mp; ++i) {
            SivoSearchQuery &s = q[i];
            std::memset(&s, 0, sizeof s);
            if (!track_in_view[i]) continue;
            require(level[i] >= 0 && level[i] < F->nlevels, "predicted level outside the scale tables");
            float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;                    // RadiusByViewingCos (ORBmatcher.cc:129-134)
            if (bFactor) r *= th;
            s.u = proj_x[i]; s.v = proj_y[i];
            s.radius = r * F->scale[level[i]];
            level_range(level[i] - 1, level[i], s.lvl_lo, s.lvl_hi);
            s.ur = proj_xr[i]; s.gate = r * F->scale[level[i]];
            s.flags = SIVO_Q_VALID | (mp_obs[i] > 0 ? SIVO_Q_BLOCKS : 0);
        }
        std::vector<uint8_t> blocked((size_t)F->n);
        for (int k = 0; k < F->n; ++k) blocked[k] = occ_obs[k] > 0;
        const SivoSearchRule rule = make_rule(SIVO_TH_HIGH, 1, 1, nn_ratio, false, true);
        run_search(*F, q.data(), mp_desc, n_mp, nullptr, nullptr, nullptr, 0, rule, blocked.data(), nullptr, match, nullptr, nullptr,
                   n_matches, nullptr);
        for (int k = 0; k < F->n; ++k)
            if (match[k] >= 0) occ_obs[k] = mp_obs[match[k]];
        return SIVO_OK;
    });
}

extern "C" int sivo_search_by_projection_frame(sivo_mframe_t C, int n_last, const uint8_t *valid, const float *u, const float *v,
                                               const float *inv_z, const int32_t *last_octave, const float *last_angle,
                                               const uint8_t *mp_desc, const int32_t *mp_obs, float th, int forward, int backward,
                                               float bf, int check_orientation, int32_t *occ_obs, int32_t *match,
                                               int *n_matches) {
    return guarded([&] {
        require(C && n_last >= 0 && occ_obs && match, "null argument");
        require(n_last == 0 || (valid && u && v && inv_z && last_octave && last_angle && mp_desc && mp_obs), "null argument");
        std::vector<SivoSearchQuery> q((size_t)n_last);
        for (int i = 0; i < n_last; ++i) {
            SivoSearchQuery &s = q[i];
            std::memset(&s, 0, sizeof s);
            if (!valid[i]) continue;
            if (inv_z[i] < 0) continue;                                                  // :1318-1319
            if (u[i] < C->min_x || u[i] > C->max_x) continue;                            // :1324-1327
            if (v[i] < C->min_y || v[i] > C->max_y) continue;
            const int oct = last_octave[i];
            require(oct >= 0 && oct < C->nlevels, "octave outside the scale tables");
            s.u = u[i]; s.v = v[i];
            s.radius = th * C->scale[oct];
            if (forward) level_range(oct, -1, s.lvl_lo, s.lvl_hi);                       // :1336-1343
            else if (backward) level_range(0, oct, s.lvl_lo, s.lvl_hi);
            else level_range(oct - 1, oct + 1, s.lvl_lo, s.lvl_hi);
            s.ur = u[i] - bf * inv_z[i];                                                 // :1354
            s.gate = s.radius;
            s.angle = last_angle[i];
            s.flags = SIVO_Q_VALID | (mp_obs[i] > 0 ? SIVO_Q_BLOCKS : 0);
        }
        std::vector<uint8_t> blocked((size_t)C->n);
        for (int k = 0; k < C->n; ++k) blocked[k] = occ_obs[k] > 0;
        const SivoSearchRule rule = make_rule(SIVO_TH_HIGH, 1, 0, 0.f, check_orientation != 0, true);
        run_search(*C, q.data(), mp_desc, n_last, nullptr, nullptr, nullptr, 0, rule, blocked.data(), nullptr, match, nullptr, nullptr,
                   n_matches, nullptr);
        for (int k = 0; k < C->n; ++k) {
            if (match[k] >= 0) occ_obs[k] = mp_obs[match[k]];
            else if (match[k] == -2) occ_obs[k] = -1;
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_search_by_projection_reloc(sivo_mframe_t C, int n_kf, const uint8_t *valid, const float *u, const float *v,
                                               const int32_t *pred_level, const float *kf_angle, const uint8_t *mp_desc, float th,
                                               int orb_dist, int check_orientation, uint8_t *occupied, int32_t *match,
                                               int *n_matches) {
    return guarded([&] {
        require(C && n_kf >= 0 && occupied && match, "null argument");
        require(n_kf == 0 || (valid && u && v && pred_level && kf_angle && mp_desc), "null argument");
        // (with ORBdist >= 256 the reference's `bestDist <= ORBdist` holds for a point without any candidate and it indexes
        // mvpMapPoints[-1]; its callers pass 100 and 64, Tracking.cc:1073-1083)
        require(orb_dist >= 0 && orb_dist < 256, "ORBdist must be below 256");
        std::vector<SivoSearchQuery> q((size_t)n_kf);
        for (int i = 0; i < n_kf; ++i) {
            SivoSearchQuery &s = q[i];
            std::memset(&s, 0, sizeof s);
            if (!valid[i]) continue;
            if (u[i] < C->min_x || u[i] > C->max_x) continue;
            if (v[i] < C->min_y || v[i] > C->max_y) continue;
            require(pred_level[i] >= 0 && pred_level[i] < C->nlevels, "predicted level outside the scale tables");
            s.u = u[i]; s.v = v[i];
            s.radius = th * C->scale[pred_level[i]];
            level_range(pred_level[i] - 1, pred_level[i] + 1, s.lvl_lo, s.lvl_hi);
            s.angle = kf_angle[i];
            s.flags = SIVO_Q_VALID | SIVO_Q_BLOCKS;
        }
        const SivoSearchRule rule = make_rule(orb_dist, 0, 0, 0.f, check_orientation != 0, true);
        run_search(*C, q.data(), mp_desc, n_kf, nullptr, nullptr, nullptr, 0, rule, occupied, nullptr, match, nullptr, nullptr, n_matches,
                   nullptr);
        for (int k = 0; k < C->n; ++k) {
            if (match[k] >= 0) occupied[k] = 1;
            else if (match[k] == -2) occupied[k] = 0;
        }
        return SIVO_OK;
    });
}

namespace {
// the shared shape of SearchByProjection(KF, Scw), Fuse x2 and SearchBySim3: window without level arguments, then an
// explicit octave test [pred - 1, pred]
void window_queries(const sivo_mframe &F, int n, const uint8_t *valid, const float *u, const float *v, const float *ur,
                    const int32_t *pred_level, float th, int flags, std::vector<SivoSearchQuery> &q) {
    q.assign((size_t)n, SivoSearchQuery{});
    for (int i = 0; i < n; ++i) {
        SivoSearchQuery &s = q[i];
        std::memset(&s, 0, sizeof s);
        if (!valid[i]) continue;
        require(pred_level[i] >= 0 && pred_level[i] < F.nlevels, "predicted level outside the scale tables");
        s.u = u[i]; s.v = v[i];
        s.radius = th * F.scale[pred_level[i]];
        s.lvl_lo = pred_level[i] - 1; s.lvl_hi = pred_level[i];
        s.ur = ur ? ur[i] : 0.f;
        s.flags = SIVO_Q_VALID | flags;
    }
}
}  // namespace

extern "C" int sivo_search_by_projection_kf(sivo_mframe_t KF, int n_mp, const uint8_t *valid, const float *u, const float *v,
                                            const int32_t *pred_level, const uint8_t *mp_desc, int th, uint8_t *matched,
                                            int32_t *match, int *n_matches) {
    return guarded([&] {
        require(KF && n_mp >= 0 && matched && match, "null argument");
        require(n_mp == 0 || (valid && u && v && pred_level && mp_desc), "null argument");
        std::vector<SivoSearchQuery> q;
        window_queries(*KF, n_mp, valid, u, v, nullptr, pred_level, (float)th, SIVO_Q_BLOCKS, q);
        const SivoSearchRule rule = make_rule(SIVO_TH_LOW, 0, 0, 0.f, false, true);
        run_search(*KF, q.data(), mp_desc, n_mp, nullptr, nullptr, nullptr, 0, rule, matched, nullptr, match, nullptr, nullptr, n_matches,
                   nullptr);
        for (int k = 0; k < KF->n; ++k)
            if (match[k] >= 0) matched[k] = 1;
        return SIVO_OK;
    });
}

extern "C" int sivo_fuse(sivo_mframe_t KF, int n_mp, const uint8_t *valid, const float *u, const float *v, const float *ur,
                         const int32_t *pred_level, const uint8_t *mp_desc, float th, int scw_variant, int32_t *best_idx,
                         int32_t *best_dist, int *n_fused) {
    return guarded([&] {
        require(KF && n_mp >= 0 && best_idx, "null argument");
        require(n_mp == 0 || (valid && u && v && pred_level && mp_desc && (scw_variant || ur)), "null argument");
        std::vector<SivoSearchQuery> q;
        window_queries(*KF, n_mp, valid, u, v, ur, pred_level, th, 0, q);
        const SivoSearchRule rule = make_rule(SIVO_TH_LOW, scw_variant ? 0 : 2, 0, 0.f, false, false);
        std::vector<int32_t> bd((size_t)n_mp);
        run_search(*KF, q.data(), mp_desc, n_mp, nullptr, nullptr, nullptr, 0, rule, nullptr, best_idx, nullptr, bd.data(), nullptr,
                   n_fused, nullptr);
        if (best_dist)      // bestDist starts at 256 (:865) resp. INT_MAX (:1019) and is returned as the loop left it
            for (int i = 0; i < n_mp; ++i) best_dist[i] = (bd[i] == 256 && scw_variant) ? INT_MAX : bd[i];
        return SIVO_OK;
    });
}

extern "C" int sivo_search_by_sim3_dir(sivo_mframe_t KF, int n, const uint8_t *valid, const float *u, const float *v,
                                       const int32_t *pred_level, const uint8_t *mp_desc, float th, int32_t *match_out) {
    return guarded([&] {
        require(KF && n >= 0 && match_out, "null argument");
        require(n == 0 || (valid && u && v && pred_level && mp_desc), "null argument");
        std::vector<SivoSearchQuery> q;
        window_queries(*KF, n, valid, u, v, nullptr, pred_level, th, 0, q);
        const SivoSearchRule rule = make_rule(SIVO_TH_HIGH, 0, 0, 0.f, false, false);
        run_search(*KF, q.data(), mp_desc, n, nullptr, nullptr, nullptr, 0, rule, nullptr, match_out, nullptr, nullptr, nullptr, nullptr,
                   nullptr);
        return SIVO_OK;
    });
}

namespace {
// queries of a BoW-guided routine in the reference's visiting order: node by node, the first operand's keys of the node
struct NodeQueries {
    std::vector<SivoSearchQuery> q;
    std::vector<uint8_t> desc;
    std::vector<int32_t> cbeg, cend, key;      // key[q] = index of the query's keypoint in operand 1
};
template <class ValidFn, class FlagFn>
void node_queries(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const SivoKeyPoint *keys1,
                  const uint8_t *desc1, int n1, int n2_list, ValidFn valid, FlagFn flags, NodeQueries &out) {
    require(n_nodes >= 0 && (n_nodes == 0 || (off1 && idx1 && off2)), "null node lists");
    for (int k = 0; k < n_nodes; ++k) {
        require(off1[k] <= off1[k + 1] && off2[k] <= off2[k + 1] && off2[k + 1] <= n2_list, "node offsets are not ascending");
        for (int j = off1[k]; j < off1[k + 1]; ++j) {
            const int i1 = idx1[j];
            require(i1 >= 0 && i1 < n1, "node list index outside the first operand");
            SivoSearchQuery s;
            std::memset(&s, 0, sizeof s);
            s.lvl_lo = INT_MIN; s.lvl_hi = INT_MAX;
            s.u = keys1[i1].x; s.v = keys1[i1].y; s.angle = keys1[i1].angle;
            s.flags = valid(i1) ? (SIVO_Q_VALID | flags(i1)) : 0;
            out.q.push_back(s);
            out.desc.insert(out.desc.end(), desc1 + 32 * (size_t)i1, desc1 + 32 * (size_t)i1 + 32);
            out.cbeg.push_back(off2[k]); out.cend.push_back(off2[k + 1]); out.key.push_back(i1);
        }
    }
}
}  // namespace

extern "C" int sivo_search_by_bow_kf_frame(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                                           const int32_t *idx2, const uint8_t *kf_valid, const SivoKeyPoint *keys_kf,
                                           const uint8_t *desc_kf, int n_kf, sivo_mframe_t F, float nn_ratio,
                                           int check_orientation, int32_t *match_f, int *n_matches) {
    return guarded([&] {
        require(F && match_f && kf_valid && keys_kf && desc_kf, "null argument");
        const int n2 = n_nodes > 0 ? off2[n_nodes] : 0;
        NodeQueries nq;
        node_queries(n_nodes, off1, idx1, off2, keys_kf, desc_kf, n_kf, n2, [&](int i) { return kf_valid[i] != 0; },
                     [](int) { return SIVO_Q_BLOCKS; }, nq);
        const SivoSearchRule rule = make_rule(SIVO_TH_LOW, 0, 2, nn_ratio, check_orientation != 0, true);
        run_search(*F, nq.q.data(), nq.desc.data(), (int)nq.q.size(), nq.cbeg.data(), nq.cend.data(), idx2, n2, rule, nullptr, nullptr,
                   match_f, nullptr, nullptr, n_matches, nullptr);
        for (int k = 0; k < F->n; ++k) match_f[k] = match_f[k] >= 0 ? nq.key[match_f[k]] : -1;      // :272-275 stores NULL
        return SIVO_OK;
    });
}

extern "C" int sivo_search_by_bow_kf_kf(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                                        const int32_t *idx2, const uint8_t *valid1, const SivoKeyPoint *keys1,
                                        const uint8_t *desc1, int n1, const uint8_t *valid2, sivo_mframe_t KF2, float nn_ratio,
                                        int check_orientation, int32_t *matches12, int *n_matches) {
    return guarded([&] {
        require(KF2 && matches12 && valid1 && valid2 && keys1 && desc1 && n1 >= 0, "null argument");
        const int n2 = n_nodes > 0 ? off2[n_nodes] : 0;
        NodeQueries nq;
        node_queries(n_nodes, off1, idx1, off2, keys1, desc1, n1, n2, [&](int i) { return valid1[i] != 0; },
                     [](int) { return SIVO_Q_BLOCKS; }, nq);
        std::vector<uint8_t> blocked((size_t)KF2->n);
        for (int k = 0; k < KF2->n; ++k) blocked[k] = !valid2[k];
        SivoSearchRule rule = make_rule(SIVO_TH_LOW, 0, 2, nn_ratio, check_orientation != 0, true);
        rule.accept_lt = 1;                                                                          // :580 `bestDist1 < TH_LOW`
        std::vector<int32_t> mq(nq.q.size());
        run_search(*KF2, nq.q.data(), nq.desc.data(), (int)nq.q.size(), nq.cbeg.data(), nq.cend.data(), idx2, n2, rule, blocked.data(),
                   mq.data(), nullptr, nullptr, nullptr, n_matches, nullptr);
        for (int i = 0; i < n1; ++i) matches12[i] = -1;
        for (size_t q = 0; q < mq.size(); ++q)
            if (mq[q] >= 0) matches12[nq.key[q]] = mq[q];
        return SIVO_OK;
    });
}

extern "C" int sivo_search_for_triangulation(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                                             const int32_t *idx2, const SivoKeyPoint *keys1, const float *u_right1,
                                             const uint8_t *has_mp1, const uint8_t *desc1, int n1, sivo_mframe_t KF2,
                                             const uint8_t *has_mp2, const float F12[9], float ex, float ey, int only_stereo,
                                             int check_orientation, int32_t *matches12, int *n_matches) {
    return guarded([&] {
        require(KF2 && matches12 && keys1 && has_mp1 && has_mp2 && desc1 && F12 && n1 >= 0, "null argument");
        const int n2 = n_nodes > 0 ? off2[n_nodes] : 0;
        NodeQueries nq;
        auto stereo1 = [&](int i) { return u_right1 && u_right1[i] >= 0; };
        node_queries(n_nodes, off1, idx1, off2, keys1, desc1, n1, n2,
                     [&](int i) { return !has_mp1[i] && (!only_stereo || stereo1(i)); },                 // :672-684
                     [&](int i) { return stereo1(i) ? SIVO_Q_STEREO : 0; }, nq);
        std::vector<uint8_t> blocked((size_t)KF2->n);
        for (int k = 0; k < KF2->n; ++k)                                                                  // :694-703 (vbMatched2 is never set)
            blocked[k] = has_mp2[k] || (only_stereo && !(!KF2->u_right.empty() && KF2->u_right[k] >= 0));
        SivoSearchRule rule = make_rule(SIVO_TH_LOW, 3, 0, 0.f, check_orientation != 0, false);
        rule.tie_last = 1;
        std::memcpy(rule.F12, F12, sizeof rule.F12);
        rule.ex = ex; rule.ey = ey;
        std::vector<int32_t> mq(nq.q.size());
        run_search(*KF2, nq.q.data(), nq.desc.data(), (int)nq.q.size(), nq.cbeg.data(), nq.cend.data(), idx2, n2, rule, blocked.data(),
                   mq.data(), nullptr, nullptr, nullptr, n_matches, nullptr);
        for (int i = 0; i < n1; ++i) matches12[i] = -1;
        for (size_t q = 0; q < mq.size(); ++q)
            if (mq[q] >= 0) matches12[nq.key[q]] = mq[q];
        return SIVO_OK;
    });
}
