/* caffe_cpu.c — TEST INFRASTRUCTURE: the reference-equivalent CPU baseline of bench.py (`cpu_baseline.value`), nothing else.
 *
 * Only tests/ and bench.py's cpu_baseline leg call this file; the product path never does.
 *
 * What it restates: how Caffe's CPU ConvolutionLayer computes a forward pass — per image of the batch, im2col of the bottom
 * blob followed by one SGEMM  top[Cout][H W] = weights[Cout][Cin k k] x col[Cin k k][H W], then the bias — which is what
 * `network->Forward()` runs for every one of the T copies of the image in the reference (/root/reference/src/bayesian_segnet/
 * bayesian_segnet.cpp:174-177 fills the T slots, :310 runs them; Caffe itself — caffe-segnet-cudnn7, base_conv_layer.cpp
 * forward_cpu_gemm / im2col.cpp / math_functions.cpp caffe_cpu_gemm -> cblas_sgemm — is NOT IN TREE).  The SGEMM
 * here stands in for the BLAS Caffe links: a BLIS-style blocked kernel (operands packed into MR / NR panels, register-tiled
 * micro-kernel with FMA, K blocked for L1 / L2, OpenMP over the panels).  im2col writes the column matrix straight into the
 * packed-panel layout (the same bytes a BLAS would first write as `col` and then re-read to pack).
 *
 * The sums are taken in another order than orc_conv2d's (ci, ky, kx) chain and with fused multiply-adds: this file is a
 * TIMING baseline; tests/test_oracle_segnet.py holds it to the oracle within float round-off, it is not itself the oracle. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __AVX512F__
#define VB 64
#define MR 8
#else
#define VB 32
#define MR 6
#endif
#define VL (VB / 4)
#define NR (2 * VL)
#define KC 256
#define MCB (8 * MR) /* rows of C per task (whole MR strips) */

typedef float vf __attribute__((vector_size(VB), aligned(4)));

static inline int imin(int a, int b) { return a < b ? a : b; }

/* rows of the column matrix: r = (ci * k + ky) * k + kx; columns: output pixels p = y W + x.  Packed: [strip j][r][NR]. */
static void im2col_packed(const float *in, int H, int W, int k, int pad, float *Bp, int K, int nstrips) {
    const int64_t plane = (int64_t)H * W;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < nstrips; ++j) {
        const int64_t p0 = (int64_t)j * NR;
        const int y0 = (int)(p0 / W), x0 = (int)(p0 % W);
        const int n = (int)(plane - p0 < NR ? plane - p0 : NR);
        for (int r = 0; r < K; ++r) {
            const int ci = r / (k * k), dy = (r / k) % k - pad, dx = r % k - pad;
            const float *ip = in + (int64_t)ci * plane;
            float *dst = Bp + ((int64_t)j * K + r) * NR;
            if (x0 + NR <= W && n == NR) {                /* the strip lies in one image row: one bounds-checked run */
                const int y = y0 + dy;
                if (y < 0 || y >= H) { memset(dst, 0, NR * sizeof(float)); continue; }
                const float *row = ip + (int64_t)y * W;
                const int xa = x0 + dx;
                if (xa >= 0 && xa + NR <= W) { memcpy(dst, row + xa, NR * sizeof(float)); continue; }
                for (int e = 0; e < NR; ++e) { const int x = xa + e; dst[e] = (x >= 0 && x < W) ? row[x] : 0.f; }
                continue;
            }
            int y = y0, x = x0;
            for (int e = 0; e < NR; ++e) {
                float v = 0.f;
                if (e < n) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = ip[(int64_t)yy * W + xx];
                    if (++x == W) { x = 0; ++y; }
                }
                dst[e] = v;
            }
        }
    }
}

/* weights [M][K] row-major -> [strip i][r][MR], zero-padded rows */
static void pack_a(const float *A, int M, int K, float *Ap, int mstrips) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < mstrips; ++i)
        for (int r = 0; r < K; ++r)
            for (int e = 0; e < MR; ++e) {
                const int m = i * MR + e;
                Ap[((int64_t)i * K + r) * MR + e] = m < M ? A[(int64_t)m * K + r] : 0.f;
            }
}

/* C[MR][NR] (+)= Ap[kc][MR] * Bp[kc][NR] */
static inline void ukernel(int kc, const float *restrict Ap, const float *restrict Bp, float *restrict C, int64_t ldc, int mr, int nr, int accumulate) {
    vf acc[MR][2];
    for (int i = 0; i < MR; ++i) { acc[i][0] = (vf){0}; acc[i][1] = (vf){0}; }
    for (int p = 0; p < kc; ++p) {
        const vf b0 = *(const vf *)(Bp + (int64_t)p * NR), b1 = *(const vf *)(Bp + (int64_t)p * NR + VL);
#pragma GCC unroll 8
        for (int i = 0; i < MR; ++i) {
            const float a = Ap[(int64_t)p * MR + i];
            acc[i][0] += a * b0;
            acc[i][1] += a * b1;
        }
    }
    if (mr == MR && nr == NR) {
        for (int i = 0; i < MR; ++i) {
            vf *c0 = (vf *)(C + i * ldc), *c1 = (vf *)(C + i * ldc + VL);
            if (accumulate) { *c0 += acc[i][0]; *c1 += acc[i][1]; }
            else { *c0 = acc[i][0]; *c1 = acc[i][1]; }
        }
    } else {
        for (int i = 0; i < mr; ++i)
            for (int e = 0; e < nr; ++e) {
                const float v = e < VL ? acc[i][0][e] : acc[i][1][e - VL];
                if (accumulate) C[i * ldc + e] += v; else C[i * ldc + e] = v;
            }
    }
}

/* One Caffe ConvolutionLayer::Forward_cpu: N images, stride 1, zero padding.  ws: caller's workspace of cfc_conv_workspace() floats. */
int64_t cfc_conv_workspace(int Cin, int H, int W, int Cout, int k) {
    const int64_t K = (int64_t)Cin * k * k, P = (int64_t)H * W;
    const int64_t nstrips = (P + NR - 1) / NR, mstrips = (Cout + MR - 1) / MR;
    return nstrips * K * NR + mstrips * K * MR + 64;
}

void cfc_conv2d(const float *in, int N, int Cin, int H, int W, const float *w, const float *bias, int Cout, int k, int pad, float *out, float *ws) {
    const int K = Cin * k * k;
    const int64_t P = (int64_t)H * W;
    const int nstrips = (int)((P + NR - 1) / NR), mstrips = (Cout + MR - 1) / MR;
    float *Bp = (float *)(((uintptr_t)ws + 63) & ~(uintptr_t)63);
    float *Ap = Bp + (int64_t)nstrips * K * NR;
    pack_a(w, Cout, K, Ap, mstrips);
    const int mblocks = (Cout + MCB - 1) / MCB;
    for (int n = 0; n < N; ++n) {
        im2col_packed(in + (int64_t)n * Cin * P, H, W, k, pad, Bp, K, nstrips);
        float *C = out + (int64_t)n * Cout * P;
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int j = 0; j < nstrips; ++j)
            for (int mb = 0; mb < mblocks; ++mb) {
                const int m0 = mb * MCB, m1 = imin(Cout, m0 + MCB);
                const int nr = (int)imin(NR, (int)(P - (int64_t)j * NR));
                for (int p0 = 0; p0 < K; p0 += KC) {
                    const int kc = imin(KC, K - p0);
                    const float *bp = Bp + ((int64_t)j * K + p0) * NR;
                    for (int m = m0; m < m1; m += MR) {
                        const float *ap = Ap + ((int64_t)(m / MR) * K + p0) * MR;
                        ukernel(kc, ap, bp, C + (int64_t)m * P + (int64_t)j * NR, P, imin(MR, m1 - m), nr, p0 > 0);
                    }
                }
                if (bias)
                    for (int m = m0; m < m1; ++m) {
                        float *c = C + (int64_t)m * P + (int64_t)j * NR;
                        const float b = bias[m];
                        for (int e = 0; e < nr; ++e) c[e] += b;
                    }
            }
    }
}
