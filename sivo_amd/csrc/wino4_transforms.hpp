// wino4_transforms.hpp — the three matrices of Winograd F(4x4,3x3) of the three-kernel path (conv_wino4.hip: input / bridge / output
// transforms and the weight transform of every GEMM arithmetic):   Y = A^T [ (G g G^T) .* (B^T d B) ] A.
// (The fused fp32 kernel of the narrow layers, conv_wino4f.hip — a fallback that runs when f16x3 is off or paused — keeps Lavin's points
// with its own transforms and weights: its first-dimension transform is built around at most four non-zeros per row of B^T.)
//
// Interpolation points 0, 1, -1, 1/2, -2, infinity (round 6).  Rounds 1 - 5 used Lavin's 0, +-1, +-2: its B^T and A^T carry 4, 5 and 8,
// and in fp32 the cancellation behind those coefficients is most of what F(4x4) loses against a direct convolution.  With one point of
// the outer pair replaced by its reciprocal (Barabasz et al., "Error analysis and improving the accuracy of Winograd convolution for
// deep neural networks": the best five-point sets for F(4,3) mix p and 1 / p) every entry of B^T and A^T is a dyadic rational of modest
// size — still exact in fp32 — and the error of a layer falls by 2 - 3.7x (emulated on the CPU in fp32 on dense Gaussian and on sparse
// heavy-tailed data: 8.1e-6 -> 2.2e-6 and 1.8e-6 -> 1.0e-6 of the largest output; on the GPU: DESIGN 3.4 / NOTEBOOK 11.5).
//
//   B^T =  1  -3/2  -2    3/2   1    0        G =   1      0      0          A^T =  1  1   1   1     1   0
//          0  -1     1/2  5/2   1    0              1/3    1/3    1/3               0  1  -1   1/2  -2   0
//          0   1    -5/2  1/2   1    0             -1/3    1/3   -1/3               0  1   1   1/4   4   0
//          0  -2    -1    2     1    0            -16/15  -8/15  -4/15              0  1  -1   1/8  -8   1
//          0   1/2  -1   -1/2   1    0              1/15  -2/15   4/15
//          0   1    -3/2 -2     3/2  1              0      0      1
// (rows of B^T: the polynomials prod_{k != j} (x - p_k), last row prod_k (x - p_k); G row j: (1, p_j, p_j^2) / prod_{k != j} (p_j - p_k);
// A^T column j: (1, p_j, p_j^2, p_j^3).)
#pragma once
#include <hip/hip_runtime.h>

namespace sivo {

// 1-D input transform B^T d
__host__ __device__ __forceinline__ void wino4_bt(const float d0, const float d1, const float d2, const float d3, const float d4,
                                                  const float d5, float *t) {
    const float a = d4 - d2, b = d3 - d1;                     // rows 3 and 4 share them
    t[0] = (d0 + d4) + 1.5f * b - 2.f * d2;
    t[1] = (d4 - d1) + 0.5f * d2 + 2.5f * d3;
    t[2] = (d4 + d1) + 0.5f * d3 - 2.5f * d2;
    t[3] = a + 2.f * b;
    t[4] = a - 0.5f * b;
    t[5] = (d1 + d5) + 1.5f * a - 2.f * d3;
}

// 1-D output transform A^T m
__host__ __device__ __forceinline__ void wino4_at(const float m0, const float m1, const float m2, const float m3, const float m4,
                                                  const float m5, float *s) {
    const float p12 = m1 + m2, q12 = m1 - m2;
    s[0] = (m0 + p12) + (m3 + m4);
    s[1] = q12 + (0.5f * m3 - 2.f * m4);
    s[2] = p12 + (0.25f * m3 + 4.f * m4);
    s[3] = (q12 + m5) + (0.125f * m3 - 8.f * m4);
}

// weight transform, evaluated on the host in f64 and rounded once
static const double WINO4_G[6][3] = {{1.0, 0, 0},
                                     {1.0 / 3, 1.0 / 3, 1.0 / 3},
                                     {-1.0 / 3, 1.0 / 3, -1.0 / 3},
                                     {-16.0 / 15, -8.0 / 15, -4.0 / 15},
                                     {1.0 / 15, -2.0 / 15, 4.0 / 15},
                                     {0, 0, 1}};

}  // namespace sivo
