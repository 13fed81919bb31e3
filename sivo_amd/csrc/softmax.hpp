// softmax.hpp — the arithmetic of the Softmax layer (Caffe `Softmax`, engine CAFFE, axis 1 with max-subtraction: SURVEY.md A.3) as
// every device site evaluates it — mc_reduce_kernel, mc_reduce_finalize_kernel (segnet_kernels.hip) and the two fused classifier
// kernels (conv_cls_mc_softmax.inc) — so that maps computed inside a fused kernel equal the post-processing kernels' on the same
// logits bit for bit:
//     e_c = exp(x_c - max_c x)    as 2^((x_c - m) log2 e) on the transcendental unit (v_exp_f32, 1 ulp): TWO instructions per class;
//     den = sum_c e_c             fp32, classes in order;
//     p_c = e_c * (1 / den)       ONE correctly rounded reciprocal per pixel and a multiply per class.
// Until round 4 these were libm-style expf (about 20 instructions) and a correctly rounded division per class (about 12): 15 x 32
// instructions per pixel and sample, which measured as MORE than half of the fused classifier kernel (tools/cls_probe.py: 0.50 ms
// against a DMA floor of 0.29 ms; the layer's 108 MFMAs per stage are a fraction of that).  Error against the reference's
// expf / division: the exponent's rounding contributes |x - m| 2^-24 relative (at most ~4e-6 for a class 30 below the maximum,
// whose probability is e^-30), the rest is an ulp — against the 1e-5 the parity tests allow on probabilities.
#pragma once
#include <hip/hip_runtime.h>

namespace sivo {

__device__ __forceinline__ float softmax_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896341f); }      // d <= 0
__device__ __forceinline__ float softmax_rcp(float den) { return __frcp_rn(den); }

}  // namespace sivo
