/*
 * oracle/orb_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of SIVO::ORBextractor (src/orbslam/ORBextractor.cc:70-150,
 * 412-486, 488-847, 1008-1122) together with the OpenCV primitives it calls.
 * OpenCV is NOT IN TREE (README.md:57 asks for OpenCV > 3.2); the primitives
 * are restated from the published OpenCV 3.2–3.4.0 algorithms
 * (SURVEY.md Appendix C):
 *   cvRound        round-half-to-even (lrint)
 *   cv::FAST       TYPE_9_16 with nonmax suppression (fast.cpp FAST_t<16>,
 *                  fast_score.cpp cornerScore<16>)
 *   cv::resize     INTER_LINEAR on 8UC1: 11-bit fixed-point separable
 *   copyMakeBorder BORDER_REFLECT_101
 *   GaussianBlur   7x7 sigma 2 on 8U: the 8-bit fixed-point separable path
 *                  (kernel round(g*256) = 18 34 49 55 49 34 18, row pass int,
 *                  column pass (sum + 2^15) >> 16); the scalar rounding rule
 *                  is used everywhere (OpenCV's SSE column path rounds ties to
 *                  even instead: +-1 LSB on ~1e-5 of pixels, build dependent)
 *   fastAtan2      the 7th-order polynomial in degrees
 * Floating-point expressions are evaluated WITHOUT fused multiply-add
 * (compile with -ffp-contract=off); a reference built with -march=native may
 * contract them — that is a compiler-dependent property of the reference.
 *
 * DistributeOctTree sorts pair<int, ExtractorNode*> (ORBextractor.cc:675), i.e.
 * it breaks size ties by heap address.  The stable rule used here: nodes
 * carry a creation sequence number and a later-created node compares greater.
 *
 * PARITY: the reference has no ORB tests and OpenCV is absent, but ORBextractor.cc itself compiles untouched
 * (oracle/Makefile `ref` -> oracle/_ref/libref_orb.so) once the OpenCV primitives above are supplied — by THIS file,
 * through oracle/ref_shims/opencv2/imgproc/imgproc.hpp.  tests/test_pin_orb.py: on 32 image x configuration cases
 * the extractor restated below equals the reference's own code in every cv::KeyPoint field, descriptor byte and
 * pyramid pixel (with heap addresses growing in creation order, see DistributeOctTree above).  So the EXTRACTOR logic
 * (constructor tables, pyramid, cell walk, octree, IC_Angle, pattern, steered BRIEF, scaling) is PINNED against the
 * reference's code; the OpenCV PRIMITIVES stay restatements on both sides of that comparison and are UNPINNED.
 * Known answers checked in tests/test_oracle_orb.py: features-per-level
 * [434,362,302,251,209,175,145,122], pyramid sizes, umax table, scale chain.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define EDGE_THRESHOLD 19

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } OrcKeyPoint;

static const int bit_pattern_31[256 * 4] = {
#include "orb_pattern.inc"
};

/* ---- OpenCV scalar helpers ------------------------------------------------ */
int orc_cvround(double v) { return (int)lrint(v); }
static inline int cv_roundf(float v) { return (int)lrintf(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

float orc_fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.141592653589793238462643383279502884);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.141592653589793238462643383279502884);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---- cv::resize INTER_LINEAR, 8UC1 ---------------------------------------- */
void orc_resize_linear_u8(const uint8_t *src, int sh, int sw, int sstep,
                          uint8_t *dst, int dh, int dw, int dstep) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *row0 = (int *)malloc(sizeof(int) * (size_t)dw), *row1 = (int *)malloc(sizeof(int) * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cv_roundf((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = (short)cv_roundf(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        const short b0 = (short)cv_roundf((1.f - fy) * 2048), b1 = (short)cv_roundf(fy * 2048);
        int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstep, *S1 = src + (size_t)sy1 * sstep;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;   /* weight is 0 there */
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        uint8_t *D = dst + (size_t)dy * dstep;
        for (int dx = 0; dx < dw; ++dx) {
            const int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

static inline int reflect101(int p, int len) {
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* copyMakeBorder(BORDER_REFLECT_101): `img` points at the interior (rows x cols,
 * stride step) of a buffer that has `b` pixels of room on every side. */
void orc_border101(uint8_t *img, int rows, int cols, int step, int b) {
    for (int y = -b; y < rows + b; ++y) {
        const int sy = reflect101(y, rows);
        for (int x = -b; x < cols + b; ++x) {
            if (y >= 0 && y < rows && x >= 0 && x < cols) continue;
            img[(ptrdiff_t)y * step + x] = img[(ptrdiff_t)sy * step + reflect101(x, cols)];
        }
    }
}

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on an isolated 8UC1 image: a separable 8.8 fixed-point kernel, row pass
 * sum k p (16 bits), column pass (sum k row + 2^15) >> 16, saturated.  What differs between OpenCV versions is the KERNEL
 * (restated from the published sources; OpenCV is not in the tree, so neither variant is pinned):
 *   variant 0 "rounded"  every tap round(g * 256): 18 34 49 55 49 34 18, sum 257.  The 8-bit separable filter of
 *                        OpenCV 3.2 - 3.4.0 (createSeparableLinearFilter, bits = 8; its SSE column path rounds ties to even
 *                        instead: +-1 LSB on ~1e-5 of the pixels, build dependent) and the bit-exact fixed-point path
 *                        (ufixedpoint16) of OpenCV 3.4.1 - 3.4.12 / 4.0 - 4.5.0, whose scalar and SIMD forms agree.
 *   variant 1 "ed"       getGaussianKernelFixedPoint_ED (OpenCV >= 3.4.13 / >= 4.5.1): the rounding error of each tap is
 *                        carried into the next one from the outside in and the centre takes what is left of 256:
 *                        18 34 48 56 48 34 18, sum 256 (no brightness gain).
 * Same arithmetic for both, so every u8 result is exact for the variant's kernel. */
void orc_gaussian7_taps(int variant, int kq[7]) {
    if (variant == 1) {
        /* getGaussianKernelBitExact (exp in double here; softdouble there: the taps are far from any rounding tie) + _ED at 8 fraction bits */
        double g[7], sum = 0, err = 0;
        int acc = 0;
        for (int i = 0; i < 7; ++i) { const double x = i - 3.0; g[i] = exp(-0.5 / 4.0 * x * x); sum += g[i]; }
        for (int i = 0; i < 3; ++i) {
            const double adj = g[i] / sum * 256.0 + err;
            const int v = orc_cvround(adj);
            err = adj - v;
            kq[i] = kq[6 - i] = v;
            acc += 2 * v;
        }
        kq[3] = 256 - acc;
        return;
    }
    /* getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256) */
    float cf[7]; double sum = 0;
    for (int i = 0; i < 7; ++i) { const double x = i - 3.0; cf[i] = (float)exp(-0.5 / 4.0 * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; ++i) { cf[i] = (float)(cf[i] * sum); kq[i] = orc_cvround((double)cf[i] * 256.0); }
}

void orc_gaussian7_u8_v(const uint8_t *src, int rows, int cols, int sstep, uint8_t *dst, int dstep, int variant) {
    int kq[7];
    orc_gaussian7_taps(variant, kq);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)rows * cols);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += kq[k + 3] * src[(size_t)y * sstep + reflect101(x + k, cols)];
            tmp[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += kq[k + 3] * tmp[(size_t)reflect101(y + k, rows) * cols + x];
            const int v = (s + (1 << 15)) >> 16;
            dst[(size_t)y * dstep + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    free(tmp);
}

void orc_gaussian7_u8(const uint8_t *src, int rows, int cols, int sstep, uint8_t *dst, int dstep) {
    orc_gaussian7_u8_v(src, rows, cols, sstep, dst, dstep, 0);
}

/* ---- cv::FAST TYPE_9_16 --------------------------------------------------- */
static const int fast_off16[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; ++q) a = a < d[k + q] ? a : d[k + q];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 5; ++q) b = b > d[k + q] ? b : d[k + q];
        if (b >= b0) continue;
        for (int q = 6; q <= 8; ++q) b = b > d[k + q] ? b : d[k + q];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

/* cv::FAST(img, kps, threshold, true).  Emits (x, y, score) in raster order.
 * Returns the count (at most max_out are written). */
int orc_fast9_16(const uint8_t *img, int rows, int cols, int step, int threshold,
                 int nonmax, int32_t *out_xy, uint8_t *out_score, int max_out) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = fast_off16[k][0] + fast_off16[k][1] * step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    if (rows < 7 || cols < 7) return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)rows * cols, 1);
    uint8_t *isc = (uint8_t *)calloc((size_t)rows * cols, 1);
    for (int i = 3; i < rows - 3; ++i) {
        const uint8_t *ptr = img + (size_t)i * step + 3;
        for (int j = 3; j < cols - 3; ++j, ++ptr) {
            const int v = ptr[0];
            int found = 0;
            {   int vt = v - threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                } }
            if (!found) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                } }
            if (found) {
                isc[(size_t)i * cols + j] = 1;
                score[(size_t)i * cols + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
            }
        }
    }
    int n = 0;
    for (int i = 3; i < rows - 3; ++i)
        for (int j = 3; j < cols - 3; ++j) {
            if (!isc[(size_t)i * cols + j]) continue;
            const uint8_t *s = score + (size_t)i * cols + j;
            const int sc = s[0];
            if (!nonmax || (sc > s[1] && sc > s[-1] && sc > s[-cols - 1] && sc > s[-cols] && sc > s[-cols + 1] &&
                            sc > s[cols - 1] && sc > s[cols] && sc > s[cols + 1])) {
                if (n < max_out) { out_xy[2 * n] = j; out_xy[2 * n + 1] = i; out_score[n] = (uint8_t)sc; }
                ++n;
            }
        }
    free(score); free(isc);
    return n;
}

/* ---- ORBextractor ---------------------------------------------------------- */
#define MAX_LEVELS 16
typedef struct OrcOrb {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;
    float mvScaleFactor[MAX_LEVELS], mvInvScaleFactor[MAX_LEVELS], mvLevelSigma2[MAX_LEVELS], mvInvLevelSigma2[MAX_LEVELS];
    int mnFeaturesPerLevel[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    /* pyramid: buffers with border; img = interior pointer */
    uint8_t *buf[MAX_LEVELS]; int rows[MAX_LEVELS], cols[MAX_LEVELS], step[MAX_LEVELS];
    /* per-level candidate dump (vToDistributeKeys) for stage parity */
    OrcKeyPoint *cand[MAX_LEVELS]; int ncand[MAX_LEVELS];
    int gaussian_variant;            /* orc_gaussian7_u8_v: 0 "rounded" (default), 1 "ed" */
} OrcOrb;

void orc_orb_set_gaussian(OrcOrb *o, int variant) { o->gaussian_variant = variant; }

/* ORBextractor::ORBextractor (ORBextractor.cc:412-475) */
OrcOrb *orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
    OrcOrb *o = (OrcOrb *)calloc(1, sizeof(OrcOrb));
    o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels;
    o->iniThFAST = iniThFAST; o->minThFAST = minThFAST;
    o->mvScaleFactor[0] = 1.0f; o->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        o->mvScaleFactor[i] = (float)(o->mvScaleFactor[i - 1] * o->scaleFactor);  /* float * double member */
        o->mvLevelSigma2[i] = o->mvScaleFactor[i] * o->mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; ++i) {
        o->mvInvScaleFactor[i] = 1.0f / o->mvScaleFactor[i];
        o->mvInvLevelSigma2[i] = 1.0f / o->mvLevelSigma2[i];
    }
    /* :440-452.  `scaleFactor` here is the double member initialised from the float argument. */
    float factor = (float)(1.0f / o->scaleFactor);
    float nDesired = (float)(nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        o->mnFeaturesPerLevel[l] = cv_roundf(nDesired);
        sum += o->mnFeaturesPerLevel[l];
        nDesired *= factor;
    }
    o->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* :460-474 umax */
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) o->umax[v] = orc_cvround(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    return o;
}

void orc_orb_destroy(OrcOrb *o) {
    if (!o) return;
    for (int l = 0; l < MAX_LEVELS; ++l) { free(o->buf[l]); free(o->cand[l]); }
    free(o);
}

void orc_orb_tables(const OrcOrb *o, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                    int32_t *feat_per_level, int32_t *umax) {
    for (int l = 0; l < o->nlevels; ++l) {
        scale[l] = o->mvScaleFactor[l]; inv_scale[l] = o->mvInvScaleFactor[l];
        sigma2[l] = o->mvLevelSigma2[l]; inv_sigma2[l] = o->mvInvLevelSigma2[l];
        feat_per_level[l] = o->mnFeaturesPerLevel[l];
    }
    for (int i = 0; i <= HALF_PATCH_SIZE; ++i) umax[i] = o->umax[i];
}

/* ComputePyramid (:1085-1122) */
static void compute_pyramid(OrcOrb *o, const uint8_t *image, int rows, int cols, int step) {
    for (int l = 0; l < o->nlevels; ++l) {
        const float scale = o->mvInvScaleFactor[l];
        const int w = cv_roundf((float)cols * scale), h = cv_roundf((float)rows * scale);
        const int W = w + EDGE_THRESHOLD * 2, H = h + EDGE_THRESHOLD * 2;
        free(o->buf[l]);
        o->buf[l] = (uint8_t *)calloc((size_t)W * H, 1);
        o->rows[l] = h; o->cols[l] = w; o->step[l] = W;
        uint8_t *img = o->buf[l] + (size_t)EDGE_THRESHOLD * W + EDGE_THRESHOLD;
        if (l != 0) {
            const uint8_t *prev = o->buf[l - 1] + (size_t)EDGE_THRESHOLD * o->step[l - 1] + EDGE_THRESHOLD;
            orc_resize_linear_u8(prev, o->rows[l - 1], o->cols[l - 1], o->step[l - 1], img, h, w, W);
        } else {
            for (int y = 0; y < h; ++y) memcpy(img + (size_t)y * W, image + (size_t)y * step, (size_t)w);
        }
        orc_border101(img, h, w, W, EDGE_THRESHOLD);
    }
}

/* level image accessor: interior pointer (rows x cols, stride step) */
const uint8_t *orc_orb_level(const OrcOrb *o, int level, int32_t *rows, int32_t *cols, int32_t *step) {
    *rows = o->rows[level]; *cols = o->cols[level]; *step = o->step[level];
    return o->buf[level] + (size_t)EDGE_THRESHOLD * o->step[level] + EDGE_THRESHOLD;
}

int orc_orb_candidates(const OrcOrb *o, int level, OrcKeyPoint *out, int max_out) {
    const int n = o->ncand[level];
    if (out) memcpy(out, o->cand[level], sizeof(OrcKeyPoint) * (size_t)(n < max_out ? n : max_out));
    return n;
}

/* -- quadtree ("OctTree") ---------------------------------------------------- */
typedef struct Node {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    OrcKeyPoint *keys; int nkeys, cap;
    int noMore;
    long seq;                 /* creation order: the stable stand-in for the heap address */
    struct Node *prev, *next; /* std::list links */
} Node;

typedef struct { Node *head, *tail; int size; long next_seq; } NodeList;

static Node *node_new(NodeList *L, int cap) {
    Node *n = (Node *)calloc(1, sizeof(Node));
    n->cap = cap > 0 ? cap : 1;
    n->keys = (OrcKeyPoint *)malloc(sizeof(OrcKeyPoint) * (size_t)n->cap);
    n->seq = L->next_seq++;
    return n;
}
static void node_push_key(Node *n, const OrcKeyPoint *k) {
    if (n->nkeys == n->cap) { n->cap *= 2; n->keys = (OrcKeyPoint *)realloc(n->keys, sizeof(OrcKeyPoint) * (size_t)n->cap); }
    n->keys[n->nkeys++] = *k;
}
static void node_free(Node *n) { free(n->keys); free(n); }
static void list_push_back(NodeList *L, Node *n) {
    n->prev = L->tail; n->next = NULL;
    if (L->tail) L->tail->next = n; else L->head = n;
    L->tail = n; L->size++;
}
static void list_push_front(NodeList *L, Node *n) {
    n->next = L->head; n->prev = NULL;
    if (L->head) L->head->prev = n; else L->tail = n;
    L->head = n; L->size++;
}
static Node *list_erase(NodeList *L, Node *n) { /* returns the following node */
    Node *nx = n->next;
    if (n->prev) n->prev->next = n->next; else L->head = n->next;
    if (n->next) n->next->prev = n->prev; else L->tail = n->prev;
    L->size--;
    node_free(n);
    return nx;
}

/* ExtractorNode::DivideNode (:488-542); children are returned detached. */
static void divide_node(NodeList *L, const Node *p, Node *c[4]) {
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int i = 0; i < 4; ++i) c[i] = node_new(L, p->nkeys);
    Node *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;

    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;

    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;

    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;

    for (int i = 0; i < p->nkeys; ++i) {
        const OrcKeyPoint *kp = &p->keys[i];
        if (kp->x < n1->URx) {
            if (kp->y < n1->BRy) node_push_key(n1, kp); else node_push_key(n3, kp);
        } else if (kp->y < n1->BRy) node_push_key(n2, kp);
        else node_push_key(n4, kp);
    }
    for (int i = 0; i < 4; ++i) if (c[i]->nkeys == 1) c[i]->noMore = 1;
}

typedef struct { int size; Node *node; } SizeNode;
static int cmp_sizenode(const void *a, const void *b) {
    const SizeNode *p = (const SizeNode *)a, *q = (const SizeNode *)b;
    if (p->size != q->size) return p->size < q->size ? -1 : 1;
    return p->node->seq < q->node->seq ? -1 : (p->node->seq > q->node->seq);
}

/* add the non-empty children to the front of the list, in n1..n4 order (:617-656) */
static void adopt_children(NodeList *L, Node *c[4], SizeNode *vec, int *nvec, int *nToExpand) {
    for (int i = 0; i < 4; ++i) {
        if (c[i]->nkeys > 0) {
            list_push_front(L, c[i]);
            if (c[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = c[i]->nkeys; vec[*nvec].node = c[i]; (*nvec)++;
            }
        } else node_free(c[i]);
    }
}

/* ORBextractor::DistributeOctTree (:544-750) */
static int distribute_octtree(const OrcKeyPoint *keys, int nkeys, int minX, int maxX, int minY, int maxY,
                              int N, int nfeatures, OrcKeyPoint *out) {
    (void)nfeatures;
    NodeList L = {0, 0, 0, 0};
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    const float hX = (float)(maxX - minX) / nIni;
    Node **ini = (Node **)malloc(sizeof(Node *) * (size_t)(nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; ++i) {
        Node *ni = node_new(&L, nkeys);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        list_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < nkeys; ++i) node_push_key(ini[(int)(keys[i].x / hX)], &keys[i]);
    free(ini);

    for (Node *it = L.head; it;) {
        if (it->nkeys == 1) { it->noMore = 1; it = it->next; }
        else if (it->nkeys == 0) it = list_erase(&L, it);
        else it = it->next;
    }

    int bFinish = 0;
    int veccap = 4 * (nkeys + 16);
    SizeNode *vec = (SizeNode *)malloc(sizeof(SizeNode) * (size_t)veccap);
    SizeNode *prevvec = (SizeNode *)malloc(sizeof(SizeNode) * (size_t)veccap);
    int nvec = 0;

    while (!bFinish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        for (Node *it = L.head; it;) {
            if (it->noMore) { it = it->next; continue; }
            Node *c[4];
            divide_node(&L, it, c);
            adopt_children(&L, c, vec, &nvec, &nToExpand);
            it = list_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) {
            bFinish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = L.size;
                const int nprev = nvec;
                memcpy(prevvec, vec, sizeof(SizeNode) * (size_t)nprev);
                nvec = 0;
                qsort(prevvec, (size_t)nprev, sizeof(SizeNode), cmp_sizenode);
                for (int j = nprev - 1; j >= 0; --j) {
                    Node *c[4];
                    divide_node(&L, prevvec[j].node, c);
                    adopt_children(&L, c, vec, &nvec, NULL);
                    list_erase(&L, prevvec[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vec); free(prevvec);

    int nout = 0;
    for (Node *it = L.head; it; it = it->next) {
        const OrcKeyPoint *best = &it->keys[0];
        float maxResponse = best->response;
        for (int k = 1; k < it->nkeys; ++k)
            if (it->keys[k].response > maxResponse) { best = &it->keys[k]; maxResponse = it->keys[k].response; }
        out[nout++] = *best;
    }
    for (Node *it = L.head; it;) { Node *nx = it->next; node_free(it); it = nx; }
    return nout;
}

/* exported for host-side quadtree parity tests */
int orc_distribute_octtree(const OrcKeyPoint *keys, int nkeys, int minX, int maxX, int minY, int maxY,
                           int N, OrcKeyPoint *out) {
    return distribute_octtree(keys, nkeys, minX, maxX, minY, maxY, N, 0, out);
}

/* IC_Angle (:75-100) */
static float ic_angle(const uint8_t *img, int step, float px, float py, const int *umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (ptrdiff_t)cv_roundf(py) * step + cv_roundf(px);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor (:104-150) */
static void orb_descriptor(const OrcKeyPoint *kpt, const uint8_t *img, int step, uint8_t *desc) {
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    const float angle = (float)kpt->angle * factorPI;
    const float a = (float)cos(angle), b = (float)sin(angle);
    const uint8_t *center = img + (ptrdiff_t)cv_roundf(kpt->y) * step + cv_roundf(kpt->x);
    const int *pattern = bit_pattern_31;
#define GET_VALUE(idx)                                                                     \
    center[cv_roundf(pattern[2 * (idx)] * b + pattern[2 * (idx) + 1] * a) * step +         \
           cv_roundf(pattern[2 * (idx)] * a - pattern[2 * (idx) + 1] * b)]
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int t = 0; t < 8; ++t) {
            const int t0 = GET_VALUE(2 * t), t1 = GET_VALUE(2 * t + 1);
            val |= (t0 < t1) << t;
        }
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

/* ComputeKeyPointsOctTree (:752-847) for one level; result in `out` (level coordinates). */
static int keypoints_level(OrcOrb *o, int level, OrcKeyPoint *out) {
    const float W = 30;
    const uint8_t *img = o->buf[level] + (size_t)EDGE_THRESHOLD * o->step[level] + EDGE_THRESHOLD;
    const int step = o->step[level];
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = o->cols[level] - EDGE_THRESHOLD + 3;
    const int maxBorderY = o->rows[level] - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);

    int cap = o->nfeatures * 10, n = 0;
    OrcKeyPoint *cand = (OrcKeyPoint *)malloc(sizeof(OrcKeyPoint) * (size_t)cap);
    const int cellcap = 64 * 64;
    int32_t *xy = (int32_t *)malloc(sizeof(int32_t) * 2 * cellcap);
    uint8_t *sc = (uint8_t *)malloc(cellcap);

    for (int i = 0; i < nRows; ++i) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; ++j) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const uint8_t *cell = img + (ptrdiff_t)(int)iniY * step + (int)iniX;
            const int ch = (int)maxY - (int)iniY, cw = (int)maxX - (int)iniX;
            int nk = orc_fast9_16(cell, ch, cw, step, o->iniThFAST, 1, xy, sc, cellcap);
            if (nk == 0) nk = orc_fast9_16(cell, ch, cw, step, o->minThFAST, 1, xy, sc, cellcap);
            for (int k = 0; k < nk; ++k) {
                if (n == cap) { cap *= 2; cand = (OrcKeyPoint *)realloc(cand, sizeof(OrcKeyPoint) * (size_t)cap); }
                OrcKeyPoint *kp = &cand[n++];
                kp->x = (float)xy[2 * k]; kp->y = (float)xy[2 * k + 1];
                kp->x += j * wCell; kp->y += i * hCell;
                kp->size = 7.f; kp->angle = -1.f; kp->response = (float)sc[k]; kp->octave = 0; kp->class_id = -1;
            }
        }
    }
    free(xy); free(sc);
    free(o->cand[level]); o->cand[level] = cand; o->ncand[level] = n;

    const int nk = n == 0 ? 0 : distribute_octtree(cand, n, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                       o->mnFeaturesPerLevel[level], o->nfeatures, out);
    const int scaledPatchSize = (int)(PATCH_SIZE * o->mvScaleFactor[level]);
    for (int k = 0; k < nk; ++k) {
        out[k].x += minBorderX; out[k].y += minBorderY;
        out[k].octave = level; out[k].size = (float)scaledPatchSize;
    }
    for (int k = 0; k < nk; ++k) out[k].angle = ic_angle(img, step, out[k].x, out[k].y, o->umax);
    return nk;
}

/* ORBextractor::operator() (:1019-1083).  Returns the keypoint count;
 * kps_out / desc_out must hold at least the returned count (call with NULL to size). */
int orc_orb_extract(OrcOrb *o, const uint8_t *gray, int rows, int cols, int step,
                    OrcKeyPoint *kps_out, uint8_t *desc_out, int max_out) {
    if (!gray || rows <= 0 || cols <= 0) return 0;
    compute_pyramid(o, gray, rows, cols, step);
    int total = 0;
    for (int level = 0; level < o->nlevels; ++level) {
        OrcKeyPoint *kps = (OrcKeyPoint *)malloc(sizeof(OrcKeyPoint) * (size_t)(o->nfeatures * 10 + 16 + o->rows[level] * o->cols[level] / 4));
        const int nk = keypoints_level(o, level, kps);
        if (nk > 0) {
            const int h = o->rows[level], w = o->cols[level];
            const uint8_t *img = o->buf[level] + (size_t)EDGE_THRESHOLD * o->step[level] + EDGE_THRESHOLD;
            uint8_t *work = (uint8_t *)malloc((size_t)h * w);
            orc_gaussian7_u8_v(img, h, w, o->step[level], work, w, o->gaussian_variant);
            for (int k = 0; k < nk; ++k) {
                if (total + k < max_out && desc_out) orb_descriptor(&kps[k], work, w, desc_out + 32 * (size_t)(total + k));
            }
            free(work);
            if (level != 0) {
                const float scale = o->mvScaleFactor[level];
                for (int k = 0; k < nk; ++k) { kps[k].x *= scale; kps[k].y *= scale; }
            }
            for (int k = 0; k < nk; ++k) if (total + k < max_out && kps_out) kps_out[total + k] = kps[k];
            total += nk;
        }
        free(kps);
    }
    return total;
}

/* stage-level entry points for kernel parity tests */
float orc_ic_angle(const uint8_t *img, int step, float px, float py, const int32_t *umax) {
    int um[HALF_PATCH_SIZE + 1];
    for (int i = 0; i <= HALF_PATCH_SIZE; ++i) um[i] = umax[i];
    return ic_angle(img, step, px, py, um);
}
void orc_orb_descriptor(const OrcKeyPoint *kp, const uint8_t *img, int step, uint8_t *desc) {
    orb_descriptor(kp, img, step, desc);
}
