/*
 * oracle/segnet_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the Bayesian-SegNet per-frame path of navganti/SIVO:
 *   - the Caffe-SegNet layer set named by the reference prototxts
 *     (config/bayesian_segnet/{basic,standard}/kitti/ prototxt files), with the
 *     layer semantics of navganti/caffe-segnet-cudnn7 (NOT IN TREE — commits
 *     named at reference README.md:49,94),
 *   - the Monte-Carlo post-processing of
 *     src/bayesian_segnet/bayesian_segnet.cpp:38-44,180-203,262-297.
 *
 * PARITY, two halves.  (1) The LAYERS (the forward pass) are UNPINNED: the reference ships no numeric fixture for this
 * path (tests/test_bayesian_segnet.cpp:152-168 pins output sizes only) and Caffe is absent, so they are checked against an
 * independent second opinion (PyTorch-CPU ops, tests/test_oracle_segnet.py) and known-answer constants only.
 * (2) Everything AROUND the forward pass — orc_preprocess, orc_mc_mean, orc_mc_finalize, orc_mc_variance — is PINNED
 * against the reference's own bayesian_segnet.cpp, compiled untouched into oracle/_ref/libref_segnet.so with a stand-in
 * network whose Forward() copies in given probabilities: input blob, classes (ties included), confidence, entropy and
 * variance bit for bit, up to the full 12 x 15 x 352 x 1024 size (tests/test_pin_segnet_post.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * All tensors are contiguous NCHW fp32, as Caffe blobs are
 * (bayesian_segnet.cpp:129-139 walks the input blob as N x C x H x W).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11 "Parallel random numbers: as easy as  */
/* 1, 2, 3") — the counter-based generator the test-time dropout masks are   */
/* keyed on.  Caffe's own RNG is unseeded in SIVO (non-reproducible by       */
/* design), so oracle and device share this published         */
/* generator instead; each side has its own implementation.                  */
/* ------------------------------------------------------------------------- */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* Keep-bit of element e (flat index inside ONE sample's C*H*W block) at
 * dropout site `site` for global MC sample `sample`:
 *   words = Philox4x32-10(ctr = {e >> 7, site, sample, 0}, key = {seed lo, hi})
 *   keep  = bit (e & 31) of words[(e >> 5) & 3]
 * One Philox call therefore yields 128 consecutive mask bits. */
static inline int dropout_keep(uint64_t e, uint32_t site, uint32_t sample, uint64_t seed) {
    uint32_t ctr[4] = {(uint32_t)(e >> 7), site, sample, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    return (int)((w[(e >> 5) & 3] >> (e & 31)) & 1u);
}

/* Dropout with sample_weights_test: true (prototxt standard:445-455 etc.):
 * Bernoulli(1-ratio) mask applied in TEST phase, survivors scaled by
 * 1/(1-ratio).  Only ratio 0.5 occurs in the reference nets; the mask bit is
 * the Philox bit above.  In-place on x (N,chw); sample0 = global index of
 * batch slot 0. */
void orc_dropout(float *x, int N, int64_t chw, uint32_t site, uint32_t sample0,
                 uint64_t seed, float ratio) {
    const float scale = 1.0f / (1.0f - ratio);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float *p = x + (int64_t)n * chw;
        for (int64_t e = 0; e < chw; ++e)
            p[e] = dropout_keep((uint64_t)e, site, sample0 + (uint32_t)n, seed) ? p[e] * scale : 0.0f;
    }
}

/* ------------------------------------------------------------------------- */
/* Convolution: cross-correlation, weights (Cout,Cin,k,k), bias per Cout,    */
/* stride 1, zero padding `pad` (Caffe ConvolutionLayer; every conv in both  */
/* prototxts is stride 1).  acc64 != 0 accumulates in double (the "oracle-   */
/* f64" figure of SURVEY.md 8d config 2).                                    */
/* ------------------------------------------------------------------------- */
void orc_conv2d(const float *in, int N, int Cin, int H, int W, const float *w,
                const float *bias, int Cout, int k, int pad, float *out, int acc64) {
    const int64_t plane = (int64_t)H * W;
    if (!acc64) {
        /* fp32 accumulation in the order (ci, ky, kx) per output element.  The plane is walked in blocks of
         * RB output rows so that the accumulator rows stay in L1 while the input channels stream by: a cache
         * blocking only — every output element still receives its terms in exactly the order above. */
        int RB = 6144 / (W > 0 ? W : 1);
        if (RB < 1) RB = 1;
        if (RB > H) RB = H;
        const int nrb = (H + RB - 1) / RB;
#pragma omp parallel for collapse(3) schedule(dynamic)
        for (int n = 0; n < N; ++n)
            for (int co = 0; co < Cout; ++co)
                for (int rb = 0; rb < nrb; ++rb) {
                    const int r0 = rb * RB, r1 = r0 + RB < H ? r0 + RB : H;
                    float *o = out + ((int64_t)n * Cout + co) * plane;
                    for (int64_t i = (int64_t)r0 * W; i < (int64_t)r1 * W; ++i) o[i] = 0.0f;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float *ip = in + ((int64_t)n * Cin + ci) * plane;
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx) {
                                const float wv = w[(((int64_t)co * Cin + ci) * k + ky) * k + kx];
                                const int dy = ky - pad, dx = kx - pad;
                                int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
                                const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
                                if (y0 < r0) y0 = r0;
                                if (y1 > r1) y1 = r1;
                                for (int y = y0; y < y1; ++y) {
                                    const float *ir = ip + (int64_t)(y + dy) * W + dx;
                                    float *orow = o + (int64_t)y * W;
                                    for (int x = x0; x < x1; ++x) orow[x] += wv * ir[x];
                                }
                            }
                    }
                    if (bias) {
                        const float b = bias[co];
                        for (int64_t i = (int64_t)r0 * W; i < (int64_t)r1 * W; ++i) o[i] += b;
                    }
                }
        return;
    }
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int n = 0; n < N; ++n) {
        for (int co = 0; co < Cout; ++co) {
            float *o = out + ((int64_t)n * Cout + co) * plane;
            double *acc = (double *)malloc(sizeof(double) * plane);
            for (int64_t i = 0; i < plane; ++i) acc[i] = 0.0;
            for (int ci = 0; ci < Cin; ++ci) {
                const float *ip = in + ((int64_t)n * Cin + ci) * plane;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const double wv = w[(((int64_t)co * Cin + ci) * k + ky) * k + kx];
                        const int dy = ky - pad, dx = kx - pad;
                        const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
                        const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
                        for (int y = y0; y < y1; ++y) {
                            const float *ir = ip + (int64_t)(y + dy) * W + dx;
                            double *ar = acc + (int64_t)y * W;
                            for (int x = x0; x < x1; ++x) ar[x] += wv * (double)ir[x];
                        }
                    }
            }
            const double b = bias ? bias[co] : 0.0;
            for (int64_t i = 0; i < plane; ++i) o[i] = (float)(acc[i] + b);
            free(acc);
        }
    }
}

/* BN with bn_mode: INFERENCE (prototxt standard:35-58): top = scale[c]*x + shift[c]. */
void orc_bn_inference(float *x, int N, int C, int64_t hw, const float *scale, const float *shift) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            float *p = x + ((int64_t)n * C + c) * hw;
            const float s = scale[c], b = shift[c];
            for (int64_t i = 0; i < hw; ++i) p[i] = s * p[i] + b;
        }
}

void orc_relu(float *x, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f;
}

/* Pooling MAX k x k stride s with a second top holding the argmax (prototxt
 * standard:127-137).  Caffe: Ho = ceil((H-k)/s)+1, windows clipped to the
 * plane, scan order row-major inside the window, strict '>' so the first
 * maximum wins; the mask stores the flat index h*W+w into the INPUT plane
 * (stored here as int32; Caffe stores it as float). */
void orc_maxpool(const float *in, int N, int C, int H, int W, int k, int s,
                 float *out, int32_t *mask, int Ho, int Wo) {
#pragma omp parallel for schedule(static)
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc) {
        const float *ip = in + nc * H * W;
        float *op = out + nc * Ho * Wo;
        int32_t *mp = mask + nc * Ho * Wo;
        for (int ph = 0; ph < Ho; ++ph)
            for (int pw = 0; pw < Wo; ++pw) {
                const int hs = ph * s, ws = pw * s;
                const int he = hs + k < H ? hs + k : H, we = ws + k < W ? ws + k : W;
                float best = -FLT_MAX;
                int32_t bi = -1;
                for (int h = hs; h < he; ++h)
                    for (int w_ = ws; w_ < we; ++w_)
                        if (ip[h * W + w_] > best) { best = ip[h * W + w_]; bi = h * W + w_; }
                op[ph * Wo + pw] = best;
                mp[ph * Wo + pw] = bi;
            }
    }
}

/* Upsample (SegNet max-unpool, prototxt standard:841-850): top (Ho,Wo) is
 * zero-filled and top.flat[mask[i]] = bottom.flat[i] per (n,c) plane. */
void orc_unpool(const float *in, const int32_t *mask, int N, int C, int H, int W,
                float *out, int Ho, int Wo) {
#pragma omp parallel for schedule(static)
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc) {
        const float *ip = in + nc * H * W;
        const int32_t *mp = mask + nc * H * W;
        float *op = out + nc * Ho * Wo;
        memset(op, 0, sizeof(float) * (size_t)Ho * Wo);
        for (int i = 0; i < H * W; ++i) op[mp[i]] = ip[i];
    }
}

/* LRN ACROSS_CHANNELS (prototxt basic:7-17): top = x / (1 + alpha/n * sum_{window} x^2)^beta,
 * window of local_size channels centred on c, clipped to existing channels. */
void orc_lrn(const float *in, int N, int C, int64_t hw, int local_size, float alpha,
             float beta, float *out) {
    const int half = (local_size - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const int c0 = c - half < 0 ? 0 : c - half;
            const int c1 = c + half >= C ? C - 1 : c + half;
            for (int64_t i = 0; i < hw; ++i) {
                float ss = 0.0f;
                for (int cc = c0; cc <= c1; ++cc) {
                    const float v = in[((int64_t)n * C + cc) * hw + i];
                    ss += v * v;
                }
                const float scale = 1.0f + (alpha / (float)local_size) * ss;
                out[((int64_t)n * C + c) * hw + i] = in[((int64_t)n * C + c) * hw + i] * powf(scale, -beta);
            }
        }
}

/* Softmax over the channel axis with max subtraction (Caffe SoftmaxLayer,
 * engine: CAFFE — prototxt standard:1632-1636). */
void orc_softmax(const float *in, int N, int C, int64_t hw, float *out) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int64_t i = 0; i < hw; ++i) {
            const float *p = in + (int64_t)n * C * hw + i;
            float *q = out + (int64_t)n * C * hw + i;
            float m = p[0];
            for (int c = 1; c < C; ++c) m = p[c * hw] > m ? p[c * hw] : m;
            float sum = 0.0f;
            for (int c = 0; c < C; ++c) { const float e = expf(p[c * hw] - m); q[c * hw] = e; sum += e; }
            for (int c = 0; c < C; ++c) q[c * hw] = q[c * hw] / sum;
        }
}

/* ------------------------------------------------------------------------- */
/* Monte-Carlo post-processing                                               */
/* ------------------------------------------------------------------------- */

/* bayesian_segnet.cpp:38-44 */
static inline double compute_entropy(double p) { return p == 0 ? 0 : -1.0 * p * log2(p); }

/* extractMeanConfidence (:278-297): cast the (T,C,H,W) f32 prob blob to f64
 * and take the mean over T.  Eigen's mean reducer divides the sum by the
 * count; the sum is taken here in slot order. */
void orc_mc_mean(const float *prob, int T, int C, int64_t hw, double *mean) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)C * hw; ++i) {
        double s = 0.0;
        for (int t = 0; t < T; ++t) s += (double)prob[(int64_t)t * C * hw + i];
        mean[i] = s / (double)T;
    }
}

/* computeClasses (:180-190) argmax over the class axis, first index wins on
 * ties (tests/test_bayesian_segnet.cpp:43-136); computeMaxConfidence
 * (:192-203); computeClassificationEntropy (:262-276). */
void orc_mc_finalize(const double *mean, int C, int64_t hw, uint8_t *classes,
                     double *confidence, double *entropy) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < hw; ++i) {
        int best = 0;
        double bv = mean[i], ent = compute_entropy(mean[i]);
        for (int c = 1; c < C; ++c) {
            const double v = mean[(int64_t)c * hw + i];
            if (v > bv) { bv = v; best = c; }
            ent += compute_entropy(v);
        }
        classes[i] = (uint8_t)best;
        confidence[i] = bv;
        entropy[i] = ent;
    }
}

/* computeVariance (:205-260; private and never called in the reference):
 * sample variance over T of the probability of the winning class. */
void orc_mc_variance(const float *prob, int T, int C, int64_t hw, const uint8_t *classes,
                     double *variance) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < hw; ++i) {
        const int c = classes[i];
        double avg = 0.0;
        for (int t = 0; t < T; ++t) avg += (double)prob[((int64_t)t * C + c) * hw + i];
        avg /= (double)T;
        double sum = 0.0;
        for (int t = 0; t < T; ++t) {
            const double d = (double)prob[((int64_t)t * C + c) * hw + i] - avg;
            sum += d * d;
        }
        variance[i] = sum / (double)(T - 1);
    }
}

/* preprocessImage (:164-178) + resizeImage (:142-162): centre crop of an
 * 8UC3 BGR image to (H,W), convertTo CV_32FC3 (no scaling, no mean), split
 * into planes, replicated into each of the T batch slots.  Returns 0 on
 * success, 1 when the image is smaller than the net (the reference then
 * yields an empty Mat). */
int orc_preprocess(const uint8_t *bgr_hwc, int ih, int iw, int T, int H, int W, float *blob) {
    if (ih < H || iw < W) return 1;
    const int x_tl = (ih == H && iw == W) ? 0 : iw / 2 - W / 2;
    const int y_tl = (ih == H && iw == W) ? 0 : ih / 2 - H / 2;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 3; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    blob[(((int64_t)t * 3 + c) * H + y) * W + x] =
                        (float)bgr_hwc[((int64_t)(y + y_tl) * iw + (x + x_tl)) * 3 + c];
    return 0;
}
