// orb.hip — ORB extraction pipeline on CDNA4; stands behind SIVO::ORBextractor
// (reference src/orbslam/ORBextractor.cc: ComputePyramid :1085-1122,
// ComputeKeyPointsOctTree :752-847, IC_Angle :75-100, GaussianBlur + computeOrbDescriptor
// :1060-1066,104-150, operator() :1019-1083) and behind the pixel work of
// Frame::ComputeStereoMatches (reference src/orbslam/Frame.cc:444-629).
//
// All kernels are integer / byte kernels bound by HBM latency and launch overhead
// (~6.6 MB touched per image): one launch covers all 8 pyramid levels wherever the
// data dependence allows (border, FAST cells, blur), coalesced row-major byte
// accesses, wave ballot + popcount for the ordered compaction of FAST corners.
// The OpenCV primitives are restated bit-exactly (fixed-point resize, reflect-101,
// FAST-9/16 score ladder in closed form, 8-bit fixed-point Gaussian, fastAtan2
// polynomial with explicitly unfused float ops) — see oracle/orb_oracle.c for the
// same definitions on the CPU.
//
// Per image (round 6: FOUR launches and three copies; fifteen launches and five copies before):
//   pyramid_kernel       every level in one launch: a workgroup that owns a tile of level l recomputes the footprint of that tile on
//                        levels 1 .. l - 1 in LDS (level l needs l - 1: the chain is a few thousand pixels deep per tile, the launches it
//                        replaces were eight dependent dispatches of 3 us of work each)
//   blur_border_kernel   Gaussian blur of all levels + the reflect-101 borders (second stream, overlaps FAST and the host quadtree)
//   fast_cells_kernel    FAST per 30-px cell incl. the iniTh / minTh fallback and per-cell NMS, the exclusive scan of the cell counts
//                        (every workgroup adds up the published counts of the cells before it) and the ordered emission
//   -> ONE D2H (cell offsets + candidates) -> host quadtree (orb_host.cpp) -> H2D kept keypoints ->
//   orient_describe_kernel  IC angle + rBRIEF of a keypoint in one wave -> ONE D2H (angle | descriptor records).
// Geometries whose pyramid footprint does not fit the staging buffers (more levels / larger scale steps than ORB-SLAM uses) take the
// level-by-level launches (copy_level0_kernel + resize_kernel).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "orb.hpp"

namespace sivo {

constexpr int EDGE_THRESHOLD = 19, HALF_PATCH = 15, PATCH_SIZE = 31;
constexpr int MAX_LEVELS = 16;

static const int8_t h_pattern[1024] = {
#include "orb_pattern.inc"
};
__constant__ int8_t c_pattern[1024];
__constant__ int c_umax[16];

struct LevelInfo {
    int rows, cols, step;    // interior size; step = cols + 2*EDGE (bytes)
    int pad_;
    int64_t off;             // byte offset of the padded buffer inside the pyramid arena
    int64_t blur_off;        // byte offset of the (rows x cols, tight) blurred image
};
struct LevelTable { LevelInfo lv[MAX_LEVELS]; int n; };

struct CellInfo { int level, x0, y0, w, h; };   // tested region (absolute level coordinates)

struct XTab { int sx; short a0, a1; };
struct YTab { int sy0, sy1; short b0, b1; };

// ---------------------------------------------------------------- pyramid
__global__ void copy_level0_kernel(const uint8_t *src, int sstep, uint8_t *dst, int dstep, int rows, int cols) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < cols) dst[(int64_t)y * dstep + x] = src[(int64_t)y * sstep + x];
}

// cv::resize INTER_LINEAR on 8UC1: horizontal 11-bit fixed point into int, vertical
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Coefficient tables come from the host.
__global__ void resize_kernel(const uint8_t *src, int sstep, int sw, uint8_t *dst, int dstep, int dh, int dw,
                              const XTab *xt, const YTab *yt) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const XTab X = xt[dx];
    const YTab Y = yt[dy];
    const int sx1 = X.sx + 1 < sw ? X.sx + 1 : X.sx;
    const uint8_t *S0 = src + (int64_t)Y.sy0 * sstep, *S1 = src + (int64_t)Y.sy1 * sstep;
    const int r0 = S0[X.sx] * X.a0 + S0[sx1] * X.a1;
    const int r1 = S1[X.sx] * X.a0 + S1[sx1] * X.a1;
    int v = (((Y.b0 * (r0 >> 4)) >> 16) + ((Y.b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(int64_t)dy * dstep + dx] = (uint8_t)v;
}

// Every level in one launch.  Tile t of level l: the host has walked the resize tables down from the tile to level 1 (the source
// footprint of a pixel range is the range of its first and last pixels' taps: the maps are monotone) and left the footprint of every
// level in the tile's record; the workgroup brings the table slices of those footprints into LDS, then produces the footprint on level 1
// from the source image, on level 2 from that, ... and finally its tile from the footprint on level l - 1 — each pixel by
// resize_kernel's arithmetic from the same inputs, so every level is bit-identical to the level-by-level launches.  Only the first
// stage reads pixels from memory; tiles of the deepest levels (the longest chains) come first in the list.
struct PyrTile { short level, pad_; short r[MAX_LEVELS][4]; };      // r[j] = x0, x1 (exclusive), y0, y1 of the footprint on level j, j = 1 .. level
struct PyrTables { const XTab *xt[MAX_LEVELS]; const YTab *yt[MAX_LEVELS]; };
constexpr int PYR_TW = 64, PYR_TH = 16, PYR_LDS = 16 * 1024;      // output tile; bytes of each of the two staging buffers
constexpr int PYR_XT = 2048, PYR_YT = 768;                         // table entries a workgroup holds (all its levels together)
constexpr int PYR_T = 1024;
__global__ __launch_bounds__(PYR_T) void pyramid_kernel(const uint8_t *src, int sstep, uint8_t *pyr, LevelTable T, PyrTables tabs,
                                                       const PyrTile *tiles) {
    __shared__ uint8_t s_buf[2][PYR_LDS];
    __shared__ XTab s_xt[PYR_XT];
    __shared__ YTab s_yt[PYR_YT];
    __shared__ PyrTile s_t;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    for (int i = tid; i < (int)(sizeof(PyrTile) / 4); i += PYR_T) reinterpret_cast<int *>(&s_t)[i] = reinterpret_cast<const int *>(tiles + blockIdx.x)[i];
    __syncthreads();
    const int l = s_t.level;
    const LevelInfo L = T.lv[l];
    uint8_t *dst = pyr + L.off + (int64_t)EDGE_THRESHOLD * L.step + EDGE_THRESHOLD;
    if (l == 0) {
        const int x0 = s_t.r[0][0], x1 = s_t.r[0][1], y0 = s_t.r[0][2], y1 = s_t.r[0][3];
        for (int y = y0 + ty; y < y1; y += PYR_T / 64)
            for (int x = x0 + tx; x < x1; x += 64) dst[(int64_t)y * L.step + x] = src[(int64_t)y * sstep + x];
        return;
    }
    // table slices of every stage (one round trip to memory for all of them)
    {
        int xo = 0, yo = 0;
        for (int j = 1; j <= l; ++j) {
            const int rx0 = s_t.r[j][0], rw = s_t.r[j][1] - rx0, ry0 = s_t.r[j][2], rh = s_t.r[j][3] - ry0;
            for (int i = tid; i < rw; i += PYR_T) s_xt[xo + i] = tabs.xt[j][rx0 + i];
            for (int i = tid; i < rh; i += PYR_T) s_yt[yo + i] = tabs.yt[j][ry0 + i];
            xo += rw; yo += rh;
        }
    }
    __syncthreads();
    int xo = 0, yo = 0;
    for (int j = 1; j <= l; ++j) {
        const int rx0 = s_t.r[j][0], rw = s_t.r[j][1] - rx0, ry0 = s_t.r[j][2], rh = s_t.r[j][3] - ry0;
        const int px0 = j > 1 ? s_t.r[j - 1][0] : 0, pw = j > 1 ? s_t.r[j - 1][1] - px0 : 0, py0 = j > 1 ? s_t.r[j - 1][2] : 0;
        const uint8_t *prev = s_buf[(j - 1) & 1];
        uint8_t *out = s_buf[j & 1];
        const int sw = T.lv[j - 1].cols;
        for (int iy = ty; iy < rh; iy += PYR_T / 64) {
            const YTab Y = s_yt[yo + iy];
#pragma unroll 4
            for (int ix = tx; ix < rw; ix += 64) {
                const XTab X = s_xt[xo + ix];
                const int sx1 = X.sx + 1 < sw ? X.sx + 1 : X.sx;
                int p00, p01, p10, p11;
                if (j == 1) {
                    const uint8_t *S0 = src + (int64_t)Y.sy0 * sstep, *S1 = src + (int64_t)Y.sy1 * sstep;
                    p00 = S0[X.sx]; p01 = S0[sx1]; p10 = S1[X.sx]; p11 = S1[sx1];
                } else {
                    const uint8_t *S0 = prev + (Y.sy0 - py0) * pw - px0, *S1 = prev + (Y.sy1 - py0) * pw - px0;
                    p00 = S0[X.sx]; p01 = S0[sx1]; p10 = S1[X.sx]; p11 = S1[sx1];
                }
                const int r0 = p00 * X.a0 + p01 * X.a1;
                const int r1 = p10 * X.a0 + p11 * X.a1;
                int v = (((Y.b0 * (r0 >> 4)) >> 16) + ((Y.b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                if (j == l) dst[(int64_t)(ry0 + iy) * L.step + rx0 + ix] = (uint8_t)v;
                else out[iy * rw + ix] = (uint8_t)v;
            }
        }
        xo += rw; yo += rh;
        __syncthreads();
    }
}

__device__ __forceinline__ int reflect101(int p, int len) {
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// copyMakeBorder(BORDER_REFLECT_101) of one level, `nblocks` workgroups sharing it (part of blur_border_kernel).
constexpr int BORDER_BLOCKS = 64;
__device__ __forceinline__ void border_body(uint8_t *pyr, const LevelInfo &L, int block, int nblocks) {
    const int PW = L.cols + 2 * EDGE_THRESHOLD, PH = L.rows + 2 * EDGE_THRESHOLD;
    uint8_t *base = pyr + L.off;
    for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < (int64_t)PW * PH;
         i += (int64_t)nblocks * blockDim.x) {
        const int px = (int)(i % PW), py = (int)(i / PW);
        const int x = px - EDGE_THRESHOLD, y = py - EDGE_THRESHOLD;
        if (x >= 0 && x < L.cols && y >= 0 && y < L.rows) continue;
        const int sx = reflect101(x, L.cols), sy = reflect101(y, L.rows);
        base[(int64_t)py * L.step + px] = base[(int64_t)(sy + EDGE_THRESHOLD) * L.step + sx + EDGE_THRESHOLD];
    }
}

// ---------------------------------------------------------------- FAST-9/16
// Closed form of cv::FAST's cornerScore<16>: with d[k] = v - ring[k],
//   A = max over the 16 arcs of 9 contiguous ring pixels of min(d),  B = the same on -d,
//   pixel is a corner at threshold t  <=>  max(A,B) > t,   score = max(A,B) - 1.
__device__ __forceinline__ int fast_score(const uint8_t *p, int step) {
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * step];      d[1] = v - p[3 * step + 1];   d[2] = v - p[2 * step + 2];   d[3] = v - p[step + 3];
    d[4] = v - p[3];             d[5] = v - p[-step + 3];      d[6] = v - p[-2 * step + 2];  d[7] = v - p[-3 * step + 1];
    d[8] = v - p[-3 * step];     d[9] = v - p[-3 * step - 1];  d[10] = v - p[-2 * step - 2]; d[11] = v - p[-step - 3];
    d[12] = v - p[-3];           d[13] = v - p[step - 3];      d[14] = v - p[2 * step - 2];  d[15] = v - p[3 * step - 1];
    int lo2[16], hi2[16], lo4[16], hi4[16], lo8[16], hi8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo8[k] = min(lo4[k], lo4[(k + 4) & 15]); hi8[k] = max(hi4[k], hi4[(k + 4) & 15]); }
    int A = -256, Bn = 256;   // A = max arc-min(d); Bn = min arc-max(d)  (B = -Bn)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        A = max(A, min(lo8[k], d[(k + 8) & 15]));
        Bn = min(Bn, max(hi8[k], d[(k + 8) & 15]));
    }
    return max(A, -Bn) - 1;
}

constexpr int CELL_MAX = 64;                       // tested region is at most 64 x 64
__device__ __forceinline__ bool nms_keep(const uint8_t *s, int sw, int th) {
    const int c = s[0];
    if (c < th) return false;
#define NB(o) ((s[o] >= th ? s[o] : 0) < c)
    return NB(-1) && NB(1) && NB(-sw - 1) && NB(-sw) && NB(-sw + 1) && NB(sw - 1) && NB(sw) && NB(sw + 1);
#undef NB
}

// One wave per 30-px cell (ORBextractor.cc:775-819): scores of the cell's tested region in
// LDS, per-cell 3x3 NMS at iniTh, fall back to minTh when that leaves nothing, raster-order
// emission through ballot/popcount.  Word = x | y << 12 | score << 24 with (x, y)
// relative to the FAST border (level coordinate - 16), as vToDistributeKeys holds them.
// Round 6: the exclusive scan over the cells and the ordered compaction happen HERE (they were two more launches and a
// ncells x cap slot array): a cell publishes ONE word, status[c] = epoch of this extraction << 12 | its count, waits until every cell
// before it has published — workgroups are dispatched in index order, so those are resident or finished — adds their counts
// (integers: any order) and emits its corners straight to dense[offset ...].  offsets[c] goes to the host with the candidates.
// One word per cell, read and written with relaxed device-scope atomics: nothing else has to be ordered, no fence, no cache
// invalidation per poll (the first form — count and flag in two arrays, an acquire load per predecessor — took 118 us instead of 42).
// A wait is bounded: if it runs out (it never should), the cell raises *err and the host fails the extraction loudly.
constexpr int FAST_SPIN_LIMIT = 1 << 22;
constexpr int FAST_EPOCH_MASK = (1 << 20) - 1;
__global__ __launch_bounds__(64) void fast_cells_kernel(const uint8_t *pyr, LevelTable T, const CellInfo *cells, int ncells,
                                                       int ini_th, int min_th, int cap, uint32_t *status, uint32_t epoch,
                                                       int *offsets, uint32_t *dense, int *err) {
    __shared__ uint8_t s_score[(CELL_MAX + 2) * (CELL_MAX + 2)];
    const int cell = blockIdx.x;
    const CellInfo C = cells[cell];
    const LevelInfo L = T.lv[C.level];
    const int lane = threadIdx.x;
    const uint8_t *img = pyr + L.off + (int64_t)EDGE_THRESHOLD * L.step + EDGE_THRESHOLD;
    const int sw = C.w + 2, npx = C.w * C.h;
    for (int i = lane; i < sw * (C.h + 2); i += 64) s_score[i] = 0;
    __syncthreads();
    const int lowest = min(ini_th, min_th);
    for (int i = lane; i < npx; i += 64) {
        const int ix = i % C.w, iy = i / C.w;
        const int sc = fast_score(img + (int64_t)(C.y0 + iy) * L.step + C.x0 + ix, L.step);
        s_score[(iy + 1) * sw + ix + 1] = (uint8_t)(sc >= lowest ? sc : 0);
    }
    __syncthreads();
    int th = ini_th, total = 0;
    for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        const bool keep = i < npx && nms_keep(s_score + (i / C.w + 1) * sw + i % C.w + 1, sw, th);
        total += __popcll(__ballot(keep));
    }
    if (total == 0) {              // vKeysCell.empty() -> FAST(minThFAST)
        th = min_th;
        for (int i0 = 0; i0 < npx; i0 += 64) {
            const int i = i0 + lane;
            const bool keep = i < npx && nms_keep(s_score + (i / C.w + 1) * sw + i % C.w + 1, sw, th);
            total += __popcll(__ballot(keep));
        }
    }
    const int count = total < cap ? total : cap;
    // publish, then the sum of the counts before this cell
    if (lane == 0) __hip_atomic_store(status + cell, (epoch << 12) | (uint32_t)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int offset = 0;
    for (int b = 0; b < cell; b += 64) {
        const int i = b + lane;
        int v = 0;
        if (i < cell) {
            int spins = 0;
            uint32_t w = __hip_atomic_load(status + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((w >> 12) != epoch) {
                if (++spins > FAST_SPIN_LIMIT) { *reinterpret_cast<volatile int *>(err) = 1; break; }
                __builtin_amdgcn_s_sleep(1);
                w = __hip_atomic_load(status + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v = (int)(w & 0xfffu);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        offset += v;
    }
    if (lane == 0) {
        offsets[cell] = offset;
        if (cell == ncells - 1) offsets[ncells] = offset + count;
    }
    uint32_t *out = dense + offset;
    int base = 0;
    for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        const int ix = i % C.w, iy = i / C.w;
        const bool keep = i < npx && nms_keep(s_score + (iy + 1) * sw + ix + 1, sw, th);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int k = base + __popcll(m & ((1ull << lane) - 1ull));
            if (k < cap)
                out[k] = (uint32_t)(C.x0 + ix - 16) | ((uint32_t)(C.y0 + iy - 16) << 12) |
                         ((uint32_t)s_score[(iy + 1) * sw + ix + 1] << 24);
        }
        base += __popcll(m);
    }
}

// The three-launch form (launch mode bit 1 clear): per-cell slots, a scan of the counts, a compaction — no workgroup waits for another,
// which is what matters beside the network (see sivo_orb_set_launch_mode).  One wave per 30-px cell (ORBextractor.cc:775-819): scores of the cell's tested region in
// LDS, per-cell 3x3 NMS at iniTh, fall back to minTh when that leaves nothing, raster-order
// emission through ballot/popcount.  Slot word = x | y << 12 | score << 24 with (x, y)
// relative to the FAST border (level coordinate - 16), as vToDistributeKeys holds them.
__global__ __launch_bounds__(64) void fast_cells_slots_kernel(const uint8_t *pyr, LevelTable T, const CellInfo *cells,
                                                       int ini_th, int min_th, uint32_t *slots, int cap,
                                                       int *counts) {
    __shared__ uint8_t s_score[(CELL_MAX + 2) * (CELL_MAX + 2)];
    const CellInfo C = cells[blockIdx.x];
    const LevelInfo L = T.lv[C.level];
    const int lane = threadIdx.x;
    const uint8_t *img = pyr + L.off + (int64_t)EDGE_THRESHOLD * L.step + EDGE_THRESHOLD;
    const int sw = C.w + 2, npx = C.w * C.h;
    for (int i = lane; i < sw * (C.h + 2); i += 64) s_score[i] = 0;
    __syncthreads();
    const int lowest = min(ini_th, min_th);
    for (int i = lane; i < npx; i += 64) {
        const int ix = i % C.w, iy = i / C.w;
        const int sc = fast_score(img + (int64_t)(C.y0 + iy) * L.step + C.x0 + ix, L.step);
        s_score[(iy + 1) * sw + ix + 1] = (uint8_t)(sc >= lowest ? sc : 0);
    }
    __syncthreads();
    int th = ini_th, total = 0;
    for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        const bool keep = i < npx && nms_keep(s_score + (i / C.w + 1) * sw + i % C.w + 1, sw, th);
        total += __popcll(__ballot(keep));
    }
    if (total == 0) th = min_th;   // vKeysCell.empty() -> FAST(minThFAST)
    uint32_t *out = slots + (int64_t)blockIdx.x * cap;
    int base = 0;
    for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        const int ix = i % C.w, iy = i / C.w;
        const bool keep = i < npx && nms_keep(s_score + (iy + 1) * sw + ix + 1, sw, th);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int k = base + __popcll(m & ((1ull << lane) - 1ull));
            if (k < cap)
                out[k] = (uint32_t)(C.x0 + ix - 16) | ((uint32_t)(C.y0 + iy - 16) << 12) |
                         ((uint32_t)s_score[(iy + 1) * sw + ix + 1] << 24);
        }
        base += __popcll(m);
    }
    if (lane == 0) counts[blockIdx.x] = base < cap ? base : cap;
}

// Exclusive scan of the per-cell counts (single workgroup; cells are few thousand at most).
__global__ __launch_bounds__(256) void scan_counts_kernel(const int *counts, int n, int *offsets, int *mirror) {
    __shared__ int s[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 256) {
        const int i = b + threadIdx.x;
        const int v = i < n ? counts[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) { offsets[i] = carry + s[threadIdx.x] - v; mirror[i] = carry + s[threadIdx.x] - v; }
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { offsets[n] = carry; mirror[n] = carry; }          // (mirror: the host's copy of the offsets, total last)
}

__global__ __launch_bounds__(64) void compact_kernel(const uint32_t *slots, int cap, const int *counts,
                                                    const int *offsets, uint32_t *dense) {
    const int c = blockIdx.x, n = counts[c], o = offsets[c];
    for (int k = threadIdx.x; k < n; k += 64) dense[o + k] = slots[(int64_t)c * cap + k];
}

// ---------------------------------------------------------------- Gaussian blur
// GaussianBlur(7x7, sigma 2, REFLECT_101) on 8U: 8-bit fixed-point kernel (sum 257), row
// pass in int, column pass (sum + 2^15) >> 16.  32x32 output tile per workgroup, all levels
// in one launch (blockIdx.y = level); rows nlevels .. 2 nlevels - 1 of the grid write the reflect-101 borders of the
// pyramid levels (BORDER_BLOCKS workgroups each; the blur reflects for itself and reads interiors only, so the two do not meet).
__global__ __launch_bounds__(256) void blur_border_kernel(uint8_t *pyr, uint8_t *blur, LevelTable T, int k0, int k1,
                                                         int k2, int k3) {
    if ((int)blockIdx.y >= T.n) {
        if ((int)blockIdx.x < BORDER_BLOCKS) border_body(pyr, T.lv[blockIdx.y - T.n], blockIdx.x, BORDER_BLOCKS);
        return;
    }
    const LevelInfo L = T.lv[blockIdx.y];
    const int tiles_x = (L.cols + 31) / 32, tiles_y = (L.rows + 31) / 32;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int x0 = (blockIdx.x % tiles_x) * 32, y0 = (blockIdx.x / tiles_x) * 32;
    __shared__ uint8_t s_in[38][40];
    __shared__ int s_row[38][33];
    const uint8_t *img = pyr + L.off + (int64_t)EDGE_THRESHOLD * L.step + EDGE_THRESHOLD;
    for (int i = threadIdx.x; i < 38 * 38; i += 256) {
        const int px = i % 38, py = i / 38;
        const int gx = reflect101(min(x0 + px - 3, L.cols + 2), L.cols), gy = reflect101(min(y0 + py - 3, L.rows + 2), L.rows);
        s_in[py][px] = img[(int64_t)gy * L.step + gx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 38 * 32; i += 256) {
        const int x = i % 32, y = i / 32;
        const uint8_t *r = &s_in[y][x];
        s_row[y][x] = k0 * (r[0] + r[6]) + k1 * (r[1] + r[5]) + k2 * (r[2] + r[4]) + k3 * r[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int x = i % 32, y = i / 32;
        if (x0 + x >= L.cols || y0 + y >= L.rows) continue;
        const int s = k0 * (s_row[y][x] + s_row[y + 6][x]) + k1 * (s_row[y + 1][x] + s_row[y + 5][x]) +
                      k2 * (s_row[y + 2][x] + s_row[y + 4][x]) + k3 * s_row[y + 3][x];
        const int v = (s + (1 << 15)) >> 16;
        blur[L.blur_off + (int64_t)(y0 + y) * L.cols + x0 + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// ---------------------------------------------------------------- orientation + descriptor
struct DevKp { float x, y; int level; };

// cv::fastAtan2 (7th-order polynomial, degrees); every operation individually rounded.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.141592653589793238462643383279502884);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// IC_Angle (ORBextractor.cc:75-100) and computeOrbDescriptor (ORBextractor.cc:104-150) of a keypoint in ONE wave (two launches
// and two result copies until round 5).  Angle: lane = patch row v in [-15,15]; integer moments are order independent, so the wave
// reduction is exact.  Descriptor, on the blurred level image: lane l evaluates tests 4l..4l+3 (a nibble); lanes pair up into
// bytes.  Result record per keypoint: [angle f32 | 32 descriptor bytes] — one D2H copy for both.
constexpr int KP_REC = 36;
__global__ __launch_bounds__(256) void orient_describe_kernel(const uint8_t *pyr, const uint8_t *blur, LevelTable T, const DevKp *kps,
                                                             int n, uint8_t *rec) {
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const DevKp kp = kps[k];
    const LevelInfo L = T.lv[kp.level];
    const int kx = __float2int_rn(kp.x), ky = __float2int_rn(kp.y);
    float deg;
    {
        const uint8_t *center = pyr + L.off + (int64_t)(EDGE_THRESHOLD + ky) * L.step + EDGE_THRESHOLD + kx;
        int m10 = 0, m01 = 0;
        if (lane < 2 * HALF_PATCH + 1) {
            const int v = lane - HALF_PATCH;
            const int d = c_umax[v < 0 ? -v : v];
            const uint8_t *row = center + (int64_t)v * L.step;
            int s = 0;
            for (int u = -d; u <= d; ++u) { const int val = row[u]; m10 += u * val; s += val; }
            m01 = v * s;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
        deg = fast_atan2_deg((float)m01, (float)m10);          // (every lane holds the sums: the same value in every lane)
    }
    uint8_t *r = rec + (int64_t)k * KP_REC;
    if (lane == 0) *reinterpret_cast<float *>(r) = deg;
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    const float angle = __fmul_rn(deg, factorPI);
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const int step = L.cols;
    const uint8_t *center = blur + L.blur_off + (int64_t)ky * step + kx;
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int8_t *p = c_pattern + 4 * (4 * lane + t);
        const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = center[r0 * step + c0], t1 = center[r1 * step + c1];
        nib |= (t0 < t1) << t;
    }
    // four bytes per store: the record lives in the host's memory (one PCIe write of 36 bytes per keypoint, not 32 single bytes)
    const int hi = __shfl_down(nib, 1);
    const uint32_t byte = (uint32_t)(nib | (hi << 4)) & 0xffu;                // (valid on even lanes: byte lane / 2 of the descriptor)
    const uint32_t w = byte | (__shfl_down(byte, 2) << 8) | (__shfl_down(byte, 4) << 16) | (__shfl_down(byte, 6) << 24);
    if ((lane & 7) == 0) reinterpret_cast<uint32_t *>(r + 4)[lane >> 3] = w;
}

// ---- the separate forms (launch mode bits 2 / 3 clear): A/B against the merged kernels above
// copyMakeBorder(BORDER_REFLECT_101) of every level in one launch (blockIdx.y = level).
__global__ void border_kernel(uint8_t *pyr, LevelTable T) {
    const LevelInfo L = T.lv[blockIdx.y];
    const int PW = L.cols + 2 * EDGE_THRESHOLD, PH = L.rows + 2 * EDGE_THRESHOLD;
    uint8_t *base = pyr + L.off;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)PW * PH;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % PW), py = (int)(i / PW);
        const int x = px - EDGE_THRESHOLD, y = py - EDGE_THRESHOLD;
        if (x >= 0 && x < L.cols && y >= 0 && y < L.rows) continue;
        const int sx = reflect101(x, L.cols), sy = reflect101(y, L.rows);
        base[(int64_t)py * L.step + px] = base[(int64_t)(sy + EDGE_THRESHOLD) * L.step + sx + EDGE_THRESHOLD];
    }
}

// (blur alone)
__global__ __launch_bounds__(256) void blur_kernel(const uint8_t *pyr, uint8_t *blur, LevelTable T, int k0, int k1,
                                                  int k2, int k3) {
    const LevelInfo L = T.lv[blockIdx.y];
    const int tiles_x = (L.cols + 31) / 32, tiles_y = (L.rows + 31) / 32;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int x0 = (blockIdx.x % tiles_x) * 32, y0 = (blockIdx.x / tiles_x) * 32;
    __shared__ uint8_t s_in[38][40];
    __shared__ int s_row[38][33];
    const uint8_t *img = pyr + L.off + (int64_t)EDGE_THRESHOLD * L.step + EDGE_THRESHOLD;
    for (int i = threadIdx.x; i < 38 * 38; i += 256) {
        const int px = i % 38, py = i / 38;
        const int gx = reflect101(min(x0 + px - 3, L.cols + 2), L.cols), gy = reflect101(min(y0 + py - 3, L.rows + 2), L.rows);
        s_in[py][px] = img[(int64_t)gy * L.step + gx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 38 * 32; i += 256) {
        const int x = i % 32, y = i / 32;
        const uint8_t *r = &s_in[y][x];
        s_row[y][x] = k0 * (r[0] + r[6]) + k1 * (r[1] + r[5]) + k2 * (r[2] + r[4]) + k3 * r[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int x = i % 32, y = i / 32;
        if (x0 + x >= L.cols || y0 + y >= L.rows) continue;
        const int s = k0 * (s_row[y][x] + s_row[y + 6][x]) + k1 * (s_row[y + 1][x] + s_row[y + 5][x]) +
                      k2 * (s_row[y + 2][x] + s_row[y + 4][x]) + k3 * s_row[y + 3][x];
        const int v = (s + (1 << 15)) >> 16;
        blur[L.blur_off + (int64_t)(y0 + y) * L.cols + x0 + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// IC_Angle (ORBextractor.cc:75-100): one wave per keypoint, lane = patch row v in [-15,15];
// integer moments are order independent, so the wave reduction is exact.
__global__ __launch_bounds__(256) void angle_kernel(const uint8_t *pyr, LevelTable T, const DevKp *kps, int n,
                                                   float *angles) {
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const DevKp kp = kps[k];
    const LevelInfo L = T.lv[kp.level];
    const uint8_t *center = pyr + L.off + (int64_t)(EDGE_THRESHOLD + __float2int_rn(kp.y)) * L.step + EDGE_THRESHOLD +
                            __float2int_rn(kp.x);
    int m10 = 0, m01 = 0;
    if (lane < 2 * HALF_PATCH + 1) {
        const int v = lane - HALF_PATCH;
        const int d = c_umax[v < 0 ? -v : v];
        const uint8_t *row = center + (int64_t)v * L.step;
        int s = 0;
        for (int u = -d; u <= d; ++u) { const int val = row[u]; m10 += u * val; s += val; }
        m01 = v * s;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    if (lane == 0) angles[k] = fast_atan2_deg((float)m01, (float)m10);
}

// computeOrbDescriptor (ORBextractor.cc:104-150) on the blurred level image: one wave per
// keypoint, lane l evaluates tests 4l..4l+3 (a nibble); lanes pair up into bytes.
__global__ __launch_bounds__(256) void descriptor_kernel(const uint8_t *blur, LevelTable T, const DevKp *kps,
                                                        const float *angles, int n, uint8_t *desc) {
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const DevKp kp = kps[k];
    const LevelInfo L = T.lv[kp.level];
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    const float angle = __fmul_rn(angles[k], factorPI);
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const int step = L.cols;
    const uint8_t *center = blur + L.blur_off + (int64_t)__float2int_rn(kp.y) * step + __float2int_rn(kp.x);
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int8_t *p = c_pattern + 4 * (4 * lane + t);
        const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = center[r0 * step + c0], t1 = center[r1 * step + c1];
        nib |= (t0 < t1) << t;
    }
    const int hi = __shfl_down(nib, 1);
    if ((lane & 1) == 0) desc[(int64_t)k * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

// ---------------------------------------------------------------- stereo SAD (Frame.cc:543-583)
struct SadJob { int level, cy, cxl, cxr0; };
// One wave per job: 11 L1 distances between the centre-subtracted 11x11 windows of the left
// level image at (cxl, cy) and of the right one at (cxr0 + inc, cy), inc = -5..5.
__global__ __launch_bounds__(256) void stereo_sad_kernel(const uint8_t *pyrL, LevelTable TL, const uint8_t *pyrR,
                                                        LevelTable TR, const SadJob *jobs, int n, int *dists) {
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const SadJob J = jobs[k];
    const LevelInfo LL = TL.lv[J.level], LR = TR.lv[J.level];
    const uint8_t *iL = pyrL + LL.off + (int64_t)(EDGE_THRESHOLD + J.cy) * LL.step + EDGE_THRESHOLD + J.cxl;
    const uint8_t *iR = pyrR + LR.off + (int64_t)(EDGE_THRESHOLD + J.cy) * LR.step + EDGE_THRESHOLD + J.cxr0;
    const int cL = iL[0];
    int acc[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) acc[s] = 0;
    for (int i = lane; i < 121; i += 64) {
        const int dy = i / 11 - 5, dx = i % 11 - 5;
        const int l = (int)iL[(int64_t)dy * LL.step + dx] - cL;
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const int inc = s - 5;
            const int r = (int)iR[(int64_t)dy * LR.step + dx + inc] - (int)iR[inc];
            const int d = l - r;
            acc[s] += d < 0 ? -d : d;
        }
    }
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        int v = acc[s];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) dists[k * 11 + s] = v;
    }
}

}  // namespace sivo

// ================================================================== host side
using namespace sivo;

// Host memory that kernels read and write directly (zero-copy): COHERENT (fine-grained) pinned memory — the device does not cache it, a
// kernel's stores are in the host's memory when its stream has been synchronised with and the host's stores are what the next kernel
// reads.  (With hipHostMallocDefault one extraction in ~10 read cell offsets of which some were the previous frame's: round 6.)
constexpr unsigned ORB_PINNED = hipHostMallocCoherent | hipHostMallocMapped;

struct sivo_orb {
    // One image at a time: an extraction, and a stereo matching that reads this extractor's pyramid, hold this lock (a second thread
    // entering the same handle waits instead of interleaving with the first one's buffers).
    std::mutex mu;
    int device = 0;
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor;
    float scale[MAX_LEVELS], inv_scale[MAX_LEVELS], sigma2[MAX_LEVELS], inv_sigma2[MAX_LEVELS];
    int feat_per_level[MAX_LEVELS];
    int umax[16];
    int gk[4];                       // fixed-point Gaussian taps k0..k3 (k3 = centre)
    // geometry-dependent state
    int rows = 0, cols = 0;
    LevelTable table{};
    std::vector<CellInfo> cells;
    std::vector<int> level_cell_begin;   // nlevels + 1
    int cap = 0;
    uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_src = nullptr;
    size_t pyr_bytes = 0, blur_bytes = 0, src_bytes = 0;
    XTab *d_xt[MAX_LEVELS] = {};
    YTab *d_yt[MAX_LEVELS] = {};
    CellInfo *d_cells = nullptr;
    // FAST results: [offsets: ncells + 1 | error word, padded to off_words][dense candidates] in PINNED HOST memory that the kernels write
    // directly (zero-copy, ~70 KB per image): beside a network that keeps every CU busy a copy is one more dependent operation that
    // has to wait for a CU (blit kernel) or an engine, and the extraction is a chain of such waits (profiles/r06_ab_serial.log)
    int *h_fast = nullptr, *d_fast = nullptr;      // (d_fast: the device's address of h_fast)
    size_t off_words = 0, dense_cap = 0;
    uint32_t *d_status = nullptr;     // per cell: epoch << 12 | count (fast_cells_kernel)
    int *d_err = nullptr;
    uint32_t epoch = 0;
    // the pyramid in one launch (pyramid_kernel) when the footprints fit its staging buffers
    std::vector<XTab> h_xt[MAX_LEVELS];
    std::vector<YTab> h_yt[MAX_LEVELS];
    PyrTile *d_tiles = nullptr;
    int ntiles = 0;
    bool fused_pyramid = false;     // the footprints fit the kernel's LDS
    // sivo_orb_set_launch_mode: bit 0 the pyramid in one launch, bit 1 FAST + scan + emission in one launch, bit 2 blur + borders in one
    // launch, bit 3 IC angle + rBRIEF in one launch.  Default 15: four launches per image.  Measured inside the frame of bench.py (the
    // network keeps every CU busy; 8 runs each on one box, profiles/r06_orb_launch_modes.log): 15 -> 144.2 frames/s, 14 (pyramid level
    // by level: its resize kernels need no LDS and slip in beside the network's whole-LDS workgroups) -> 145.1, 0 (the round-5 kernels) ->
    // 146 / 145; round 5's code with its 12 copies per stereo pair -> 143.8.  Alone on the GPU 15 is the fastest (1.0 ms per stereo pair).
    int launch_mode = 15;
    uint32_t *d_slots = nullptr;    // three-launch FAST: ncells x cap candidate slots
    int *d_counts = nullptr, *d_offsets = nullptr;
    PyrTables pyr_tabs{};
    DevKp *h_kps = nullptr, *d_kps = nullptr;       // pinned; d_* = the device's address of the same memory: the kernel reads the kept keys
    uint8_t *h_rec = nullptr, *d_rec = nullptr;     // from the host's memory and writes the [angle | descriptor] records (KP_REC bytes per keypoint) there
    float *d_angles = nullptr;                      // separate angle / descriptor kernels (launch mode bit 3 clear)
    uint8_t *d_desc = nullptr;
    int kp_cap = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t ev_pyr = nullptr, ev_blur = nullptr;
    // profiling (sivo_orb_profile): HIP events around the kernel groups of one extraction, on the stream each runs on
    static constexpr int NPROF = 5;             // pyramid, blur + border, FAST cells (+ scan + emission), angle + descriptor, (unused since round 6: 0)
    bool prof = false;
    hipEvent_t pe0[NPROF] = {nullptr, nullptr, nullptr, nullptr, nullptr}, pe1[NPROF] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool pe_used[NPROF] = {false, false, false, false, false};
    double prof_ms[NPROF] = {0, 0, 0, 0, 0};
    int prof_calls = 0, prof_keys = 0;
    std::vector<std::vector<SivoKeyPoint>> last_candidates;
    bool have_pyramid = false;
    // grow-only device arena of sivo_stereo_match_begin (no hipMalloc / hipFree — and so no device-wide
    // synchronisation — on the per-frame path once it has reached its working size)
    void *d_match = nullptr;
    size_t match_cap = 0;
    // its pinned twin: the call's inputs are packed here and go over in ONE copy (four until round 5), its first results come back in one
    uint8_t *h_match = nullptr;
    size_t h_match_cap = 0;
    uint8_t *match_stage(size_t bytes) {
        if (bytes > h_match_cap) {
            if (h_match) (void)hipHostFree(h_match);
            h_match = nullptr;
            h_match_cap = bytes + bytes / 2;
            SIVO_HIP(hipHostMalloc((void **)&h_match, h_match_cap, ORB_PINNED));
        }
        return h_match;
    }
    void *match_arena(size_t bytes) {
        if (bytes > match_cap) {
            if (d_match) (void)hipFree(d_match);
            d_match = nullptr;
            match_cap = bytes + bytes / 2;
            SIVO_HIP(hipMalloc(&d_match, match_cap));
        }
        return d_match;
    }

    void free_geometry() {
        for (void *p : {(void *)d_pyr, (void *)d_blur, (void *)d_src, (void *)d_cells, (void *)d_status, (void *)d_tiles, (void *)d_slots, (void *)d_counts, (void *)d_offsets})
            if (p) (void)hipFree(p);
        for (int l = 0; l < MAX_LEVELS; ++l) {
            if (d_xt[l]) (void)hipFree(d_xt[l]);
            if (d_yt[l]) (void)hipFree(d_yt[l]);
            d_xt[l] = nullptr; d_yt[l] = nullptr;
        }
        if (h_fast) (void)hipHostFree(h_fast);
        d_pyr = d_blur = d_src = nullptr; d_cells = nullptr; d_fast = nullptr; h_fast = nullptr;
        d_status = nullptr; d_err = nullptr; d_tiles = nullptr; d_slots = nullptr; d_counts = nullptr; d_offsets = nullptr; ntiles = 0; fused_pyramid = false;
        src_bytes = 0;
    }
    ~sivo_orb() {
        free_geometry();
        if (d_match) (void)hipFree(d_match);
        if (h_match) (void)hipHostFree(h_match);

        if (d_angles) (void)hipFree(d_angles);
        if (d_desc) (void)hipFree(d_desc);
        if (h_kps) (void)hipHostFree(h_kps);
        if (h_rec) (void)hipHostFree(h_rec);
        for (int i = 0; i < NPROF; ++i) {
            if (pe0[i]) (void)hipEventDestroy(pe0[i]);
            if (pe1[i]) (void)hipEventDestroy(pe1[i]);
        }
        if (ev_pyr) (void)hipEventDestroy(ev_pyr);
        if (ev_blur) (void)hipEventDestroy(ev_blur);
        if (stream) (void)hipStreamDestroy(stream);
        if (stream2) (void)hipStreamDestroy(stream2);
    }
};

namespace {

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_roundf(float v) { return (int)std::lrintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// ORBextractor::ORBextractor (ORBextractor.cc:412-475): scale chain, features per level, umax.
void init_tables(sivo_orb &o, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    o.nfeatures = nfeatures; o.nlevels = nlevels; o.ini_th = ini_th; o.min_th = min_th;
    o.scale_factor = scale_factor;   // the reference keeps it in a double member
    o.scale[0] = 1.0f; o.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        o.scale[i] = (float)(o.scale[i - 1] * o.scale_factor);
        o.sigma2[i] = o.scale[i] * o.scale[i];
    }
    for (int i = 0; i < nlevels; ++i) { o.inv_scale[i] = 1.0f / o.scale[i]; o.inv_sigma2[i] = 1.0f / o.sigma2[i]; }
    const float factor = (float)(1.0f / o.scale_factor);
    float desired = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        o.feat_per_level[l] = cv_roundf(desired);
        sum += o.feat_per_level[l];
        desired *= factor;
    }
    o.feat_per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    const int vmax = cv_floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1), vmin = cv_ceil(HALF_PATCH * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (int v = 0; v <= vmax; ++v) o.umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
        o.umax[v] = v0;
        ++v0;
    }
    // getGaussianKernel(7, 2, CV_32F) quantised by convertTo(CV_32S, 256)
    float cf[7]; double s = 0;
    for (int i = 0; i < 7; ++i) { const double x = i - 3.0; cf[i] = (float)std::exp(-0.5 / 4.0 * x * x); s += cf[i]; }
    s = 1. / s;
    for (int i = 0; i < 4; ++i) o.gk[i] = cv_round((double)(float)(cf[i] * s) * 256.0);
}

// Everything that depends on the image size: level buffers, resize tables, FAST cell list.
void setup_geometry(sivo_orb &o, int rows, int cols) {
    if (rows == o.rows && cols == o.cols) return;
    if (rows >= 4096 || cols >= 4096) throw std::invalid_argument("images up to 4095 x 4095 are supported");
    o.free_geometry();
    o.rows = rows; o.cols = cols;
    o.table.n = o.nlevels;
    size_t poff = 0, boff = 0;
    for (int l = 0; l < o.nlevels; ++l) {
        LevelInfo &L = o.table.lv[l];
        L.cols = cv_roundf((float)cols * o.inv_scale[l]);
        L.rows = cv_roundf((float)rows * o.inv_scale[l]);
        L.step = L.cols + 2 * EDGE_THRESHOLD;
        L.off = (int64_t)poff; L.blur_off = (int64_t)boff;
        poff += ((size_t)L.step * (L.rows + 2 * EDGE_THRESHOLD) + 255) & ~(size_t)255;
        boff += ((size_t)L.cols * L.rows + 255) & ~(size_t)255;
        // The reference needs maxBorder - minBorder = size - 2 * (EDGE_THRESHOLD - 3) >= 1 on every level: below that its
        // DistributeOctTree sizes a vector with a negative count and aborts (ORBextractor.cc:553-560); from 33 px up a level
        // that is too small for a 30 px cell simply yields no keys (tests/test_pin_orb.py, 120 x 160 at 8 levels).
        if (L.cols < 2 * EDGE_THRESHOLD - 5 || L.rows < 2 * EDGE_THRESHOLD - 5)
            throw std::invalid_argument("image too small for the requested number of pyramid levels");
    }
    o.pyr_bytes = poff; o.blur_bytes = boff;
    o.d_pyr = dev_alloc<uint8_t>(poff);
    o.d_blur = dev_alloc<uint8_t>(boff);
    SIVO_HIP(hipMemset(o.d_pyr, 0, poff));
    // cv::resize coefficient tables (level l from level l-1)
    for (int l = 1; l < o.nlevels; ++l) {
        const LevelInfo &S = o.table.lv[l - 1], &D = o.table.lv[l];
        const double scale_x = 1. / ((double)D.cols / S.cols), scale_y = 1. / ((double)D.rows / S.rows);
        std::vector<XTab> &xt = o.h_xt[l];
        std::vector<YTab> &yt = o.h_yt[l];
        xt.assign(D.cols, XTab{}); yt.assign(D.rows, YTab{});
        for (int dx = 0; dx < D.cols; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = cv_floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= S.cols - 1) { fx = 0; sx = S.cols - 1; }
            xt[dx] = XTab{sx, (short)cv_roundf((1.f - fx) * 2048), (short)cv_roundf(fx * 2048)};
        }
        for (int dy = 0; dy < D.rows; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = cv_floor(fy);
            fy -= sy;
            const int sy0 = std::min(std::max(sy, 0), S.rows - 1), sy1 = std::min(std::max(sy + 1, 0), S.rows - 1);
            yt[dy] = YTab{sy0, sy1, (short)cv_roundf((1.f - fy) * 2048), (short)cv_roundf(fy * 2048)};
        }
        o.d_xt[l] = dev_alloc<XTab>(xt.size());
        o.d_yt[l] = dev_alloc<YTab>(yt.size());
        SIVO_HIP(hipMemcpy(o.d_xt[l], xt.data(), xt.size() * sizeof(XTab), hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(o.d_yt[l], yt.data(), yt.size() * sizeof(YTab), hipMemcpyHostToDevice));
        o.pyr_tabs.xt[l] = o.d_xt[l]; o.pyr_tabs.yt[l] = o.d_yt[l];
    }
    // tiles of pyramid_kernel with the footprint of every level below them (the walk down the tables), deepest levels first, and whether
    // every footprint and its table slices fit the kernel's LDS
    {
        std::vector<PyrTile> tiles;
        bool fits = true;
        for (int l = o.nlevels - 1; l >= 0; --l) {
            const LevelInfo &L = o.table.lv[l];
            // tile width by depth: levels 0 and 1 read the source image directly (no footprint to stage: wide tiles, few workgroups); deep
            // levels take narrow tiles (their footprint on level 1 — the one stage that reads memory — grows with 1.2^l)
            const int tw = l <= 1 ? 4 * PYR_TW : (l == 2 ? 2 * PYR_TW : (l >= 5 ? PYR_TW / 2 : PYR_TW));
            for (int y0 = 0; y0 < L.rows; y0 += PYR_TH)
                for (int x0 = 0; x0 < L.cols; x0 += tw) {
                    PyrTile t{};
                    t.level = (short)l;
                    int a0 = x0, a1 = std::min(x0 + tw, L.cols), b0 = y0, b1 = std::min(y0 + PYR_TH, L.rows);
                    t.r[l][0] = (short)a0; t.r[l][1] = (short)a1; t.r[l][2] = (short)b0; t.r[l][3] = (short)b1;
                    int xsum = a1 - a0, ysum = b1 - b0;
                    for (int j = l; j > 1; --j) {
                        const int sw = o.table.lv[j - 1].cols;
                        const int na0 = o.h_xt[j][a0].sx, hi = o.h_xt[j][a1 - 1].sx + 1;
                        const int nb0 = o.h_yt[j][b0].sy0, nb1 = o.h_yt[j][b1 - 1].sy1 + 1;
                        a0 = na0; a1 = (hi < sw ? hi : sw - 1) + 1; b0 = nb0; b1 = nb1;
                        t.r[j - 1][0] = (short)a0; t.r[j - 1][1] = (short)a1; t.r[j - 1][2] = (short)b0; t.r[j - 1][3] = (short)b1;
                        if ((int64_t)(a1 - a0) * (b1 - b0) > PYR_LDS) fits = false;
                        xsum += a1 - a0; ysum += b1 - b0;
                    }
                    if (xsum > PYR_XT || ysum > PYR_YT) fits = false;
                    tiles.push_back(t);
                }
        }
        o.fused_pyramid = fits;
        o.ntiles = (int)tiles.size();
        o.d_tiles = dev_alloc<PyrTile>(tiles.size());
        SIVO_HIP(hipMemcpy(o.d_tiles, tiles.data(), tiles.size() * sizeof(PyrTile), hipMemcpyHostToDevice));
    }
    // FAST cells (ORBextractor.cc:752-819)
    o.cells.clear();
    o.level_cell_begin.assign(o.nlevels + 1, 0);
    int cap = 1;
    for (int l = 0; l < o.nlevels; ++l) {
        o.level_cell_begin[l] = (int)o.cells.size();
        const LevelInfo &L = o.table.lv[l];
        const float W = 30;
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
        const int maxBX = L.cols - EDGE_THRESHOLD + 3, maxBY = L.rows - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                CellInfo c;
                c.level = l; c.x0 = (int)iniX + 3; c.y0 = (int)iniY + 3;
                c.w = (int)maxX - (int)iniX - 6; c.h = (int)maxY - (int)iniY - 6;
                if (c.w <= 0 || c.h <= 0) continue;   // cv::FAST on a view narrower than 7 finds nothing
                if (c.w > CELL_MAX || c.h > CELL_MAX) throw std::runtime_error("FAST cell larger than 64 px");
                cap = std::max(cap, ((c.w + 1) / 2) * ((c.h + 1) / 2));
                o.cells.push_back(c);
            }
        }
    }
    o.level_cell_begin[o.nlevels] = (int)o.cells.size();
    o.cap = cap;
    const size_t nc = o.cells.size();
    o.d_cells = dev_alloc<CellInfo>(nc);
    if (nc) SIVO_HIP(hipMemcpy(o.d_cells, o.cells.data(), nc * sizeof(CellInfo), hipMemcpyHostToDevice));
    o.dense_cap = std::max<size_t>(nc * cap, 1);
    o.off_words = (nc + 2 + 63) / 64 * 64;          // nc + 1 offsets, then the error word of the scan
    SIVO_HIP(hipHostMalloc((void **)&o.h_fast, (o.off_words + o.dense_cap) * sizeof(int), ORB_PINNED));
    std::memset(o.h_fast, 0, (o.off_words + o.dense_cap) * sizeof(int));
    SIVO_HIP(hipHostGetDevicePointer((void **)&o.d_fast, o.h_fast, 0));
    if (cap >= 4096) throw std::runtime_error("FAST cell capacity beyond the 12 bits of its status word");
    o.d_status = dev_alloc<uint32_t>(nc + 1);
    o.d_slots = dev_alloc<uint32_t>(std::max<size_t>(nc * cap, 1));
    o.d_counts = dev_alloc<int>(nc + 1);
    o.d_offsets = dev_alloc<int>(nc + 1);
    o.d_err = o.d_fast + nc + 1;                    // (inside the region the host copies: zero from the memset above)
    SIVO_HIP(hipMemset(o.d_status, 0, (nc + 1) * sizeof(uint32_t)));
    o.epoch = 0;
    o.have_pyramid = false;
}

void ensure_kp_capacity(sivo_orb &o, int n) {
    if (n <= o.kp_cap) return;
    if (o.h_kps) { (void)hipHostFree(o.h_kps); (void)hipHostFree(o.h_rec); }
    const int cap = std::max(n, 4096);
    if (o.d_angles) { (void)hipFree(o.d_angles); (void)hipFree(o.d_desc); }
    o.d_angles = dev_alloc<float>(cap); o.d_desc = dev_alloc<uint8_t>((size_t)cap * 32);
    SIVO_HIP(hipHostMalloc((void **)&o.h_kps, cap * sizeof(DevKp), ORB_PINNED));
    SIVO_HIP(hipHostMalloc((void **)&o.h_rec, (size_t)cap * KP_REC, ORB_PINNED));
    SIVO_HIP(hipHostGetDevicePointer((void **)&o.d_kps, o.h_kps, 0));
    SIVO_HIP(hipHostGetDevicePointer((void **)&o.d_rec, o.h_rec, 0));
    o.kp_cap = cap;
}

// The whole operator(): d_src is a device image (rows x cols, stride step).
int extract_locked(sivo_orb &o, const uint8_t *d_src, int rows, int cols, int step, SivoKeyPoint *keypoints,
                   uint8_t *descriptors, int capacity, int *n_out, hipStream_t user_stream) {
    setup_geometry(o, rows, cols);
    hipStream_t st = o.stream;
    if (user_stream) {   // order after the producer of d_src
        SIVO_HIP(hipEventRecord(o.ev_pyr, user_stream));
        SIVO_HIP(hipStreamWaitEvent(st, o.ev_pyr, 0));
    }
    const LevelTable &T = o.table;
    auto mark = [&](int k, bool end, hipStream_t on) {
        if (!o.prof) return;
        if (!o.pe0[k]) { SIVO_HIP(hipEventCreate(&o.pe0[k])); SIVO_HIP(hipEventCreate(&o.pe1[k])); }
        SIVO_HIP(hipEventRecord(end ? o.pe1[k] : o.pe0[k], on));
        if (end) o.pe_used[k] = true;
    };
    // ---- pyramid
    mark(0, false, st);
    if (o.fused_pyramid && (o.launch_mode & 1)) {
        hipLaunchKernelGGL(pyramid_kernel, dim3(o.ntiles), dim3(PYR_T), 0, st, d_src, step, o.d_pyr, T, o.pyr_tabs, o.d_tiles);
    } else {        // footprints beyond the staging buffers (more levels / larger scale steps than ORB-SLAM's): level by level
        const LevelInfo &L0 = T.lv[0];
        uint8_t *dst = o.d_pyr + L0.off + (size_t)EDGE_THRESHOLD * L0.step + EDGE_THRESHOLD;
        hipLaunchKernelGGL(copy_level0_kernel, dim3(cdiv(cols, 256), rows), dim3(256), 0, st, d_src, step, dst, L0.step, rows, cols);
        for (int l = 1; l < o.nlevels; ++l) {
            const LevelInfo &S = T.lv[l - 1], &D = T.lv[l];
            const uint8_t *src = o.d_pyr + S.off + (size_t)EDGE_THRESHOLD * S.step + EDGE_THRESHOLD;
            uint8_t *d = o.d_pyr + D.off + (size_t)EDGE_THRESHOLD * D.step + EDGE_THRESHOLD;
            hipLaunchKernelGGL(resize_kernel, dim3(cdiv(D.cols, 256), D.rows), dim3(256), 0, st, src, S.step, S.cols, d, D.step,
                               D.rows, D.cols, o.d_xt[l], o.d_yt[l]);
        }
    }
    mark(0, true, st);
    SIVO_HIP(hipEventRecord(o.ev_pyr, st));
    // ---- blur + borders on the second stream (they need interiors only), overlapping FAST + the host quadtree
    SIVO_HIP(hipStreamWaitEvent(o.stream2, o.ev_pyr, 0));
    {
        int max_tiles = BORDER_BLOCKS;
        for (int l = 0; l < o.nlevels; ++l) max_tiles = std::max(max_tiles, cdiv(T.lv[l].cols, 32) * cdiv(T.lv[l].rows, 32));
        mark(1, false, o.stream2);
        if (o.launch_mode & 4) {
            hipLaunchKernelGGL(blur_border_kernel, dim3(max_tiles, 2 * o.nlevels), dim3(256), 0, o.stream2, o.d_pyr, o.d_blur, T, o.gk[0], o.gk[1],
                               o.gk[2], o.gk[3]);
        } else {
            hipLaunchKernelGGL(blur_kernel, dim3(max_tiles, o.nlevels), dim3(256), 0, o.stream2, o.d_pyr, o.d_blur, T, o.gk[0], o.gk[1], o.gk[2], o.gk[3]);
            hipLaunchKernelGGL(border_kernel, dim3(64, o.nlevels), dim3(256), 0, o.stream2, o.d_pyr, T);
        }
        mark(1, true, o.stream2);
        SIVO_HIP(hipEventRecord(o.ev_blur, o.stream2));
    }
    // ---- FAST cells -> ordered candidate list, written by the kernels into the host's (pinned) memory: [cell offsets | candidates]
    const int nc = (int)o.cells.size();
    int total = 0;
    int *h_off = o.h_fast;
    const uint32_t *h_dense = reinterpret_cast<const uint32_t *>(o.h_fast + o.off_words);
    if (nc) {
        h_off[nc] = -1;
        mark(2, false, st);
        if (o.launch_mode & 2) {
            o.epoch = (o.epoch % FAST_EPOCH_MASK) + 1;          // 1 .. 2^20 - 1: never the zero the status words start with
            hipLaunchKernelGGL(fast_cells_kernel, dim3(nc), dim3(64), 0, st, o.d_pyr, T, o.d_cells, nc, o.ini_th, o.min_th, o.cap, o.d_status,
                               o.epoch, o.d_fast, reinterpret_cast<uint32_t *>(o.d_fast + o.off_words), o.d_err);
        } else {
            hipLaunchKernelGGL(fast_cells_slots_kernel, dim3(nc), dim3(64), 0, st, o.d_pyr, T, o.d_cells, o.ini_th, o.min_th, o.d_slots, o.cap, o.d_counts);
            hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(256), 0, st, o.d_counts, nc, o.d_offsets, o.d_fast);
            hipLaunchKernelGGL(compact_kernel, dim3(nc), dim3(64), 0, st, o.d_slots, o.cap, o.d_counts, o.d_offsets, reinterpret_cast<uint32_t *>(o.d_fast + o.off_words));
        }
        mark(2, true, st);
        SIVO_HIP(hipStreamSynchronize(st));                 // the kernels wrote offsets and candidates into h_fast themselves
        total = h_off[nc];
        if (total < 0 || (size_t)total > o.dense_cap || h_off[nc + 1] != 0)
            return fail(SIVO_ERR_RUNTIME, "FAST candidate scan did not complete (total %d, error word %d)", total, h_off[nc + 1]);
    }
    o.have_pyramid = true;
    // ---- host: quadtree per level (ORBextractor.cc:821-841)
    o.last_candidates.assign(o.nlevels, {});
    std::vector<SivoKeyPoint> kept, all;
    std::vector<int> level_count(o.nlevels, 0);
    for (int l = 0; l < o.nlevels; ++l) {
        const int c0 = o.level_cell_begin[l], c1 = o.level_cell_begin[l + 1];
        if (c0 == c1) continue;
        const int b = h_off[c0], e = h_off[c1];
        if (b < 0 || e < b || e > total) {
            // (diagnosis of an inconsistent offset table: what the device holds against what the host sees)
            std::vector<int> dev(nc + 1, -7);
            if (!(o.launch_mode & 2)) (void)hipMemcpy(dev.data(), o.d_offsets, (size_t)(nc + 1) * sizeof(int), hipMemcpyDeviceToHost);
            int bad = 0, first_bad = -1;
            for (int i = 0; i <= nc; ++i) if (dev[i] != h_off[i]) { if (first_bad < 0) first_bad = i; ++bad; }
            return fail(SIVO_ERR_RUNTIME, "FAST offsets inconsistent: level %d cells [%d, %d) offsets %d .. %d, total %d, ncells %d; host != device at %d entries (first %d: host %d device %d)",
                        l, c0, c1, b, e, total, nc, bad, first_bad, first_bad >= 0 ? h_off[first_bad] : 0, first_bad >= 0 ? dev[first_bad] : 0);
        }
        std::vector<SivoKeyPoint> &cand = o.last_candidates[l];
        cand.resize(e - b);
        for (int i = b; i < e; ++i) {
            const uint32_t w = h_dense[i];
            SivoKeyPoint &k = cand[i - b];
            k.x = (float)(w & 0xfff); k.y = (float)((w >> 12) & 0xfff);
            k.size = 7.f; k.angle = -1.f; k.response = (float)(w >> 24); k.octave = 0; k.class_id = -1;
        }
        if (cand.empty()) continue;
        const LevelInfo &L = T.lv[l];
        const int minB = EDGE_THRESHOLD - 3;
        distribute_quadtree(cand.data(), (int)cand.size(), minB, L.cols - EDGE_THRESHOLD + 3, minB, L.rows - EDGE_THRESHOLD + 3,
                            o.feat_per_level[l], kept);
        const int scaled_patch = (int)(PATCH_SIZE * o.scale[l]);
        for (SivoKeyPoint &k : kept) {
            k.x += minB; k.y += minB; k.octave = l; k.size = (float)scaled_patch;
            all.push_back(k);
        }
        level_count[l] = (int)kept.size();
    }
    const int n = (int)all.size();
    *n_out = n;
    if (n == 0) { SIVO_HIP(hipStreamSynchronize(o.stream2)); return SIVO_OK; }
    if (n > capacity) { SIVO_HIP(hipStreamSynchronize(o.stream2)); return fail(SIVO_ERR_CAPACITY, "%d keypoints, capacity %d", n, capacity); }
    // ---- device: orientation + descriptors
    ensure_kp_capacity(o, n);
    for (int i = 0; i < n; ++i) o.h_kps[i] = DevKp{all[i].x, all[i].y, all[i].octave};
    if (o.launch_mode & 8) {
        SIVO_HIP(hipStreamWaitEvent(st, o.ev_blur, 0));
        mark(3, false, st);
        hipLaunchKernelGGL(orient_describe_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, o.d_pyr, o.d_blur, T, o.d_kps, n, o.d_rec);
        mark(3, true, st);
        SIVO_HIP(hipStreamSynchronize(st));
    } else {
        mark(3, false, st);
        hipLaunchKernelGGL(angle_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, o.d_pyr, T, o.d_kps, n, o.d_angles);
        mark(3, true, st);
        SIVO_HIP(hipStreamWaitEvent(st, o.ev_blur, 0));
        mark(4, false, st);
        hipLaunchKernelGGL(descriptor_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, o.d_blur, T, o.d_kps, o.d_angles, n, o.d_desc);
        mark(4, true, st);
        // (into the record layout the assembly below reads: angles behind the first n records' worth of descriptors is not possible in one
        // copy — two copies, as until round 5)
        SIVO_HIP(hipMemcpyAsync(o.h_rec, o.d_angles, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        SIVO_HIP(hipMemcpyAsync(o.h_rec + (size_t)o.kp_cap * 4, o.d_desc, (size_t)n * 32, hipMemcpyDeviceToHost, st));
        SIVO_HIP(hipStreamSynchronize(st));
    }
    const bool recs = (o.launch_mode & 8) != 0;
    SIVO_HIP(hipGetLastError());
    if (o.prof) {
        SIVO_HIP(hipStreamSynchronize(o.stream2));
        for (int k = 0; k < sivo_orb::NPROF; ++k)
            if (o.pe_used[k]) {
                float ms = 0.f;
                SIVO_HIP(hipEventElapsedTime(&ms, o.pe0[k], o.pe1[k]));
                o.prof_ms[k] += ms;
                o.pe_used[k] = false;
            }
        ++o.prof_calls; o.prof_keys += n;
    }
    // ---- assemble (ORBextractor.cc:1068-1081): pt *= scale for level > 0
    for (int i = 0; i < n; ++i) {
        SivoKeyPoint k = all[i];
        std::memcpy(&k.angle, recs ? o.h_rec + (size_t)i * KP_REC : o.h_rec + (size_t)i * 4, 4);
        if (k.octave != 0) { const float s = o.scale[k.octave]; k.x *= s; k.y *= s; }
        keypoints[i] = k;
    }
    if (descriptors)
        for (int i = 0; i < n; ++i)
            std::memcpy(descriptors + (size_t)i * 32, recs ? o.h_rec + (size_t)i * KP_REC + 4 : o.h_rec + (size_t)o.kp_cap * 4 + (size_t)i * 32, 32);
    return SIVO_OK;
}

}  // namespace

// Which OpenCV's GaussianBlur(7x7, sigma 2) on 8U the descriptor image is blurred like (ORBextractor.cc:1060-1062; README.md:57 asks
// for "OpenCV > 3.2", and the 8.8 fixed-point taps changed inside that range): 0 = every tap round(g * 256): 18 34 49 55 49 34 18
// (OpenCV 3.2 - 3.4.12 and 4.0 - 4.5.0; the default), 1 = getGaussianKernelFixedPoint_ED of OpenCV >= 3.4.13 / >= 4.5.1: the
// rounding error carried from tap to tap, the centre takes the rest of 256: 18 34 48 56 48 34 18.  Same arithmetic either way
// (blur_kernel); oracle: orb_oracle.c orc_gaussian7_taps.
extern "C" int sivo_orb_set_launch_mode(sivo_orb_t h, int mode) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        if (mode < 0 || mode > 15) throw std::invalid_argument("launch mode: bits 0 - 3 (pyramid, FAST, blur + borders, angle + descriptor in one launch each)");
        h->launch_mode = mode;
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_set_gaussian(sivo_orb_t h, int variant) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        if (variant != 0 && variant != 1) throw std::invalid_argument("GaussianBlur variant: 0 (rounded taps) or 1 (error-diffused taps)");
        double g[7], sum = 0;
        float cf[7];
        for (int i = 0; i < 7; ++i) { const double x = i - 3.0; g[i] = std::exp(-0.5 / 4.0 * x * x); cf[i] = (float)g[i]; }
        if (variant == 0) {
            for (int i = 0; i < 7; ++i) sum += cf[i];
            sum = 1. / sum;
            for (int i = 0; i < 4; ++i) h->gk[i] = cv_round((double)(float)(cf[i] * sum) * 256.0);
        } else {
            for (int i = 0; i < 7; ++i) sum += g[i];
            double err = 0;
            int acc = 0;
            for (int i = 0; i < 3; ++i) {
                const double adj = g[i] / sum * 256.0 + err;
                h->gk[i] = cv_round(adj);
                err = adj - h->gk[i];
                acc += 2 * h->gk[i];
            }
            h->gk[3] = 256 - acc;
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                               int device, sivo_orb_t *out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("out is NULL");
        *out = nullptr;
        if (nfeatures < 1 || nlevels < 1 || nlevels > MAX_LEVELS || scale_factor <= 1.0f || ini_th_fast < 0 || min_th_fast < 0)
            throw std::invalid_argument("bad ORB parameters (nlevels <= 16, scale_factor > 1)");
        if (device < 0 || sivo_device_count() <= device)
            return fail(SIVO_ERR_RUNTIME, "HIP device %d is not available (%d visible): libsivo_hip has no CPU fallback", device,
                        sivo_device_count());
        std::unique_ptr<sivo_orb> o(new sivo_orb);
        o->device = device;
        init_tables(*o, nfeatures, scale_factor, nlevels, std::min(std::max(ini_th_fast, 0), 255), std::min(std::max(min_th_fast, 0), 255));
        DeviceGuard dg(device);
        SIVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), h_pattern, sizeof h_pattern));
        SIVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), o->umax, sizeof(int) * 16));
        // Highest stream priority: the extractor is ~0.5 ms of small latency-bound kernels with two host round trips
        // (candidate read-back, quadtree) that runs BESIDE the network (Frame.cc:126-129 threads); at default priority
        // each of its launches queues behind a chip-filling convolution and the frame waits for ORB, not the network.
        int prio_lo = 0, prio_hi = 0;
        SIVO_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        if (const char *e = SIVO_DIAG_ENV("SIVO_ORB_PRIO"))       // experiments: 0 = default priority, 1 = lowest
            prio_hi = std::atoi(e) == 0 ? 0 : std::atoi(e) == 1 ? prio_lo : prio_hi;
        SIVO_HIP(hipStreamCreateWithPriority(&o->stream, hipStreamNonBlocking, prio_hi));
        SIVO_HIP(hipStreamCreateWithPriority(&o->stream2, hipStreamNonBlocking, prio_hi));
        SIVO_HIP(hipEventCreateWithFlags(&o->ev_pyr, hipEventDisableTiming));
        SIVO_HIP(hipEventCreateWithFlags(&o->ev_blur, hipEventDisableTiming));
        *out = o.release();
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_destroy(sivo_orb_t h) {
    return guarded([&] {
        if (h) { DeviceGuard dg(h->device); delete h; }
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_tables(sivo_orb_t h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                               int32_t *features_per_level) {
    if (!h) return fail(SIVO_ERR_INVALID_ARGUMENT, "null handle");
    for (int l = 0; l < h->nlevels; ++l) {
        if (scale) scale[l] = h->scale[l];
        if (inv_scale) inv_scale[l] = h->inv_scale[l];
        if (sigma2) sigma2[l] = h->sigma2[l];
        if (inv_sigma2) inv_sigma2[l] = h->inv_sigma2[l];
        if (features_per_level) features_per_level[l] = h->feat_per_level[l];
    }
    return SIVO_OK;
}

extern "C" int sivo_orb_extract_dev(sivo_orb_t h, const uint8_t *d_gray, int rows, int cols, int step,
                                    SivoKeyPoint *keypoints, uint8_t *descriptors, int capacity, int *n_out,
                                    void *stream) {
    return guarded([&] {
        if (!h || !n_out) throw std::invalid_argument("null argument");
        *n_out = 0;
        if (!d_gray || rows <= 0 || cols <= 0) return SIVO_OK;   // _image.empty(): return silently (:1023-1024)
        if (step < cols || !keypoints) throw std::invalid_argument("bad step / null keypoints");
        DeviceGuard dg(h->device);
        std::lock_guard<std::mutex> one_image(h->mu);
        return extract_locked(*h, d_gray, rows, cols, step, keypoints, descriptors, capacity, n_out, (hipStream_t)stream);
    });
}

extern "C" int sivo_orb_extract_pair_dev(sivo_orb_t left, sivo_orb_t right, const uint8_t *d_left, const uint8_t *d_right, int rows, int cols,
                                         int step_left, int step_right, SivoKeyPoint *kp_left, uint8_t *desc_left, int capacity_left,
                                         int *n_left, SivoKeyPoint *kp_right, uint8_t *desc_right, int capacity_right, int *n_right,
                                         void *stream) {
    return guarded([&] {
        if (!left || !right || left == right || !n_left || !n_right) throw std::invalid_argument("two different extractors and both counts are needed");
        // threadRight (Frame.cc:127): the right image on a thread of its own; its error text travels back with its return code
        std::string right_error;
        auto fr = std::async(std::launch::async, [&] {
            const int rc = sivo_orb_extract_dev(right, d_right, rows, cols, step_right, kp_right, desc_right, capacity_right, n_right, stream);
            if (rc) right_error = sivo_last_error();
            return rc;
        });
        const int rc_l = sivo_orb_extract_dev(left, d_left, rows, cols, step_left, kp_left, desc_left, capacity_left, n_left, stream);
        const int rc_r = fr.get();
        if (rc_l) return rc_l;                      // (its message is this thread's last error already)
        if (rc_r) return fail(rc_r, "right image: %s", right_error.c_str());
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_extract(sivo_orb_t h, const uint8_t *gray, int rows, int cols, int step,
                                SivoKeyPoint *keypoints, uint8_t *descriptors, int capacity, int *n_out) {
    return guarded([&] {
        if (!h || !n_out) throw std::invalid_argument("null argument");
        *n_out = 0;
        if (!gray || rows <= 0 || cols <= 0) return SIVO_OK;
        if (step < cols || !keypoints) throw std::invalid_argument("bad step / null keypoints");
        DeviceGuard dg(h->device);
        std::lock_guard<std::mutex> one_image(h->mu);
        const size_t need = (size_t)rows * cols;
        if (h->src_bytes < need || h->rows != rows || h->cols != cols) {
            setup_geometry(*h, rows, cols);
            if (h->d_src) (void)hipFree(h->d_src);
            h->d_src = dev_alloc<uint8_t>(need);
            h->src_bytes = need;
        }
        SIVO_HIP(hipMemcpy2DAsync(h->d_src, cols, gray, step, cols, rows, hipMemcpyHostToDevice, h->stream));
        return extract_locked(*h, h->d_src, rows, cols, cols, keypoints, descriptors, capacity, n_out, nullptr);
    });
}

// Profiling: enable != 0 brackets the kernel groups of every following extraction with HIP events (and clears the
// accumulators); sivo_orb_profile_read returns the mean milliseconds per extraction of {pyramid (copy + resizes), blur +
// border, FAST cells + scan + compact, IC-angle, rBRIEF descriptors}, the extractions and the mean keypoints per extraction.
extern "C" int sivo_orb_profile(sivo_orb_t h, int enable) {
    return guarded([&] {
        if (!h) throw std::invalid_argument("null handle");
        h->prof = enable != 0;
        for (double &v : h->prof_ms) v = 0.0;
        h->prof_calls = 0; h->prof_keys = 0;
        return SIVO_OK;
    });
}
extern "C" int sivo_orb_profile_read(sivo_orb_t h, double ms_out[5], int *calls, double *mean_keys) {
    return guarded([&] {
        if (!h || !ms_out) throw std::invalid_argument("null argument");
        for (int k = 0; k < sivo_orb::NPROF; ++k) ms_out[k] = h->prof_calls ? h->prof_ms[k] / h->prof_calls : 0.0;
        if (calls) *calls = h->prof_calls;
        if (mean_keys) *mean_keys = h->prof_calls ? (double)h->prof_keys / h->prof_calls : 0.0;
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_level(sivo_orb_t h, int level, uint8_t *host_out, size_t capacity, int32_t *rows, int32_t *cols) {
    return guarded([&] {
        if (!h || level < 0 || level >= h->nlevels) throw std::invalid_argument("bad level");
        if (!h->have_pyramid) throw std::invalid_argument("no image has been extracted yet");
        const LevelInfo &L = h->table.lv[level];
        if (rows) *rows = L.rows;
        if (cols) *cols = L.cols;
        const size_t n = (size_t)L.step * (L.rows + 2 * EDGE_THRESHOLD);
        if (!host_out) return SIVO_OK;
        if (capacity < n) return fail(SIVO_ERR_CAPACITY, "level needs %zu bytes, capacity %zu", n, capacity);
        DeviceGuard dg(h->device);
        SIVO_HIP(hipStreamSynchronize(h->stream2));
        SIVO_HIP(hipMemcpy(host_out, h->d_pyr + L.off, n, hipMemcpyDeviceToHost));
        return SIVO_OK;
    });
}

extern "C" int sivo_orb_candidates(sivo_orb_t h, int level, SivoKeyPoint *out, int capacity, int *n_out) {
    return guarded([&] {
        if (!h || level < 0 || level >= h->nlevels || !n_out) throw std::invalid_argument("bad argument");
        if ((int)h->last_candidates.size() <= level) { *n_out = 0; return SIVO_OK; }
        const std::vector<SivoKeyPoint> &c = h->last_candidates[level];
        *n_out = (int)c.size();
        if (!out) return SIVO_OK;
        if ((int)c.size() > capacity) return fail(SIVO_ERR_CAPACITY, "%zu candidates, capacity %d", c.size(), capacity);
        std::copy(c.begin(), c.end(), out);
        return SIVO_OK;
    });
}

// Frame::ComputeStereoMatches (Frame.cc:444-629).  Host: row table, candidate lists and the
// float decision logic exactly as the reference orders it; device: Hamming argmin over the
// candidate lists and the 11 SAD windows per match on the two resident pyramids.
extern "C" int sivo_stereo_match_begin(sivo_orb_t left, sivo_orb_t right, const SivoKeyPoint *kpL, const uint8_t *descL,
                                       int nL, const SivoKeyPoint *kpR, const uint8_t *descR, int nR, float bf, float b,
                                       float *u_right, float *depth, int32_t *best_right, int32_t *sad_dist) {
    return guarded([&] {
        if (!left || !right || nL < 0 || nR < 0 || (nL && (!kpL || !descL || !u_right || !depth)) || (nR && (!kpR || !descR)))
            throw std::invalid_argument("bad argument");
        std::unique_lock<std::mutex> lock_l(left->mu, std::defer_lock), lock_r(right->mu, std::defer_lock);
        if (left == right) lock_l.lock(); else std::lock(lock_l, lock_r);
        if (!left->have_pyramid || !right->have_pyramid) throw std::invalid_argument("both extractors must hold a pyramid");
        if (left->nlevels != right->nlevels || left->rows != right->rows || left->cols != right->cols)
            throw std::invalid_argument("left/right extractors differ in geometry");
        if (nL && !sad_dist) throw std::invalid_argument("bad argument");
        for (int i = 0; i < nL; ++i) { u_right[i] = -1.f; depth[i] = -1.f; sad_dist[i] = -1; if (best_right) best_right[i] = -1; }
        if (nL == 0 || nR == 0) return SIVO_OK;
        DeviceGuard dg(left->device);
        const int TH_HIGH = 100, TH_LOW = 50, thOrbDist = (TH_HIGH + TH_LOW) / 2;
        const int nRows = left->table.lv[0].rows;
        // row table (:454-477); the reference indexes rows unchecked — clamped here.  Flat (count, then fill in key order: the order a
        // vector per row would hold) in per-thread storage: 352 small vectors per call were a fifth of the call's host time.
        static thread_local std::vector<int> row_off, row_idx, row_at;
        row_off.assign((size_t)nRows + 1, 0);
        auto span = [&](int iR, int &minr, int &maxr) {
            const float r = 2.0f * left->scale[kpR[iR].octave];
            maxr = std::min((int)std::ceil(kpR[iR].y + r), nRows - 1); minr = std::max((int)std::floor(kpR[iR].y - r), 0);
        };
        for (int iR = 0; iR < nR; ++iR) {
            int minr, maxr;
            span(iR, minr, maxr);
            for (int yi = minr; yi <= maxr; ++yi) ++row_off[yi + 1];
        }
        for (int y = 0; y < nRows; ++y) row_off[y + 1] += row_off[y];
        row_idx.resize((size_t)row_off[nRows]);
        row_at.assign(row_off.begin(), row_off.end() - 1);
        for (int iR = 0; iR < nR; ++iR) {
            int minr, maxr;
            span(iR, minr, maxr);
            for (int yi = minr; yi <= maxr; ++yi) row_idx[row_at[yi]++] = iR;
        }
        const float minZ = b, minD = 0, maxD = bf / minZ;
        static thread_local std::vector<int> off, idx;
        off.assign((size_t)nL + 1, 0); idx.clear();
        for (int iL = 0; iL < nL; ++iL) {
            off[iL] = (int)idx.size();
            const int row = (int)kpL[iL].y;
            if (row < 0 || row >= nRows) continue;
            const float uL = kpL[iL].x, minU = uL - maxD, maxU = uL - minD;
            if (row_off[row] == row_off[row + 1] || maxU < 0) continue;
            for (int j = row_off[row]; j < row_off[row + 1]; ++j) {
                const int iR = row_idx[j];
                if (kpR[iR].octave < kpL[iL].octave - 1 || kpR[iR].octave > kpL[iL].octave + 1) continue;
                if (kpR[iR].x >= minU && kpR[iR].x <= maxU) idx.push_back(iR);
            }
        }
        off[nL] = (int)idx.size();
        // Device work on the left extractor's own (non-blocking, high-priority) stream and arena: nothing here touches
        // the null stream, so the call can run beside a network forward that is queued on another stream.
        hipStream_t st = left->stream;
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t o_dl = 0, o_dr = o_dl + al((size_t)nL * 32), o_off = o_dr + al((size_t)nR * 32),
                     o_idx = o_off + al((size_t)(nL + 1) * 4), o_bi = o_idx + al(idx.size() * 4 + 4), o_bd = o_bi + al((size_t)nL * 4),
                     o_sd = o_bd + al((size_t)nL * 4), o_jobs = o_sd + al((size_t)nL * 4),
                     o_dists = o_jobs + al((size_t)nL * sizeof(SadJob)), total = o_dists + al((size_t)nL * 11 * 4);
        uint8_t *base = (uint8_t *)left->match_arena(total);
        // inputs [descL | descR | off | idx] in ONE staged copy (they are read many times: device memory); the results — best index / best
        // distance, later the SAD distances — are written by the kernels into the pinned twin directly, and the SAD jobs are read from it
        uint8_t *hs = left->match_stage(total), *hs_dev = nullptr;
        SIVO_HIP(hipHostGetDevicePointer((void **)&hs_dev, hs, 0));
        std::memcpy(hs + o_dl, descL, (size_t)nL * 32);
        std::memcpy(hs + o_dr, descR, (size_t)nR * 32);
        std::memcpy(hs + o_off, off.data(), (size_t)(nL + 1) * 4);
        if (!idx.empty()) std::memcpy(hs + o_idx, idx.data(), idx.size() * 4);
        SIVO_HIP(hipMemcpyAsync(base, hs, o_idx + idx.size() * 4, hipMemcpyHostToDevice, st));
        const int *bi = reinterpret_cast<const int *>(hs + o_bi), *bd = reinterpret_cast<const int *>(hs + o_bd);
        int rc = sivo_hamming_argmin2_dev(base + o_dl, nL, base + o_dr, (const int32_t *)(base + o_off), (const int32_t *)(base + o_idx),
                                          (int32_t *)(hs_dev + o_bi), (int32_t *)(hs_dev + o_bd), (int32_t *)(base + o_sd), nullptr, st);
        if (rc) return rc;
        SIVO_HIP(hipStreamSynchronize(st));
        // SAD jobs (:538-565)
        std::vector<SadJob> jobs;
        std::vector<int> jobL;
        const int w = 5, Lw = 5;
        for (int iL = 0; iL < nL; ++iL) {
            if (bd[iL] >= TH_HIGH || bi[iL] < 0) continue;   // bestDist starts at TH_HIGH
            if (best_right) best_right[iL] = bi[iL];
            if (bd[iL] >= thOrbDist) continue;
            const int lvl = kpL[iL].octave;
            const float sf = left->inv_scale[lvl];
            const float suL = std::round(kpL[iL].x * sf), svL = std::round(kpL[iL].y * sf), suR0 = std::round(kpR[bi[iL]].x * sf);
            const float iniu = suR0 + Lw - w, endu = suR0 + Lw + w + 1;
            if (iniu < 0 || endu >= right->table.lv[lvl].cols) continue;
            jobs.push_back(SadJob{lvl, (int)svL, (int)suL, (int)suR0});
            jobL.push_back(iL);
        }
        const int *dists = reinterpret_cast<const int *>(hs + o_dists);
        if (!jobs.empty()) {
            std::memcpy(hs + o_jobs, jobs.data(), jobs.size() * sizeof(SadJob));      // (each job is read once by its wave: from the host's memory)
            SadJob *dj = (SadJob *)(hs_dev + o_jobs);
            int *dd = (int *)(hs_dev + o_dists);
            SIVO_HIP(hipStreamSynchronize(left->stream2));       // the pyramids of both extractors are complete
            SIVO_HIP(hipStreamSynchronize(right->stream));
            SIVO_HIP(hipStreamSynchronize(right->stream2));
            hipLaunchKernelGGL(stereo_sad_kernel, dim3(cdiv((int)jobs.size(), 4)), dim3(256), 0, st, left->d_pyr, left->table,
                               right->d_pyr, right->table, dj, (int)jobs.size(), dd);
            SIVO_HIP(hipStreamSynchronize(st));
        }
        // decision logic (:567-628)
        for (size_t j = 0; j < jobs.size(); ++j) {
            const int iL = jobL[j], lvl = jobs[j].level;
            int bestDist = INT32_MAX, bestinc = 0;
            float vD[11];
            for (int inc = -Lw; inc <= Lw; ++inc) {
                const float dist = (float)dists[j * 11 + inc + Lw];
                if (dist < (float)bestDist) { bestDist = (int)dist; bestinc = inc; }
                vD[Lw + inc] = dist;
            }
            if (bestinc == -Lw || bestinc == Lw) continue;
            const float d1 = vD[Lw + bestinc - 1], d2 = vD[Lw + bestinc], d3 = vD[Lw + bestinc + 1];
            const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = left->scale[lvl] * ((float)jobs[j].cxr0 + (float)bestinc + deltaR);
            const float uL = kpL[iL].x;
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)(uL - 0.01); }
                depth[iL] = bf / disparity;
                u_right[iL] = bestuR;
                sad_dist[iL] = bestDist;
            }
        }
        return SIVO_OK;
    });
}

// Median cull of Frame::ComputeStereoMatches (Frame.cc:616-628) over the keypoints with keep[i] != 0 (all when keep
// is NULL): matches whose SAD distance is >= 1.5 * 1.4 * median are dropped; keypoints that are not kept get -1.
extern "C" int sivo_stereo_match_cull(int n, const uint8_t *keep, const int32_t *sad_dist, float *u_right, float *depth) {
    return guarded([&] {
        if (n < 0 || (n && (!sad_dist || !u_right || !depth))) throw std::invalid_argument("bad argument");
        std::vector<std::pair<int, int>> vDistIdx;
        for (int i = 0; i < n; ++i) {
            if (keep && !keep[i]) { u_right[i] = -1.f; depth[i] = -1.f; continue; }
            if (sad_dist[i] >= 0 && u_right[i] >= 0) vDistIdx.emplace_back(sad_dist[i], i);
        }
        if (!vDistIdx.empty()) {
            std::sort(vDistIdx.begin(), vDistIdx.end());
            const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
            const float thDist = 1.5f * 1.4f * median;
            for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
                if ((float)vDistIdx[i].first < thDist) break;
                u_right[vDistIdx[i].second] = -1;
                depth[vDistIdx[i].second] = -1;
            }
        }
        return SIVO_OK;
    });
}

extern "C" int sivo_stereo_match(sivo_orb_t left, sivo_orb_t right, const SivoKeyPoint *kpL, const uint8_t *descL,
                                 int nL, const SivoKeyPoint *kpR, const uint8_t *descR, int nR, float bf, float b,
                                 float *u_right, float *depth, int32_t *best_right) {
    std::vector<int32_t> sad((size_t)(nL > 0 ? nL : 1));
    const int rc = sivo_stereo_match_begin(left, right, kpL, descL, nL, kpR, descR, nR, bf, b, u_right, depth, best_right, sad.data());
    if (rc) return rc;
    return sivo_stereo_match_cull(nL, nullptr, sad.data(), u_right, depth);
}
