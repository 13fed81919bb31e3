/* sivo_hip_debug.h — test / diagnostic entry points: single kernels of libsivo_hip.so on caller-supplied or random data.
 * NOT part of the product ABI: they live in libsivo_hip_dbg.so (sivo_amd/csrc/debug_entry.cpp, `make dbg`), a thin library
 * that links libsivo_hip.so and calls its kernels, so the tests measure the product's binary code; the ablation build
 * libsivo_hip_diag.so (`make diag`, tools/) carries them as well. */
#ifndef SIVO_HIP_DEBUG_H
#define SIVO_HIP_DEBUG_H
#include "sivo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic micro-benchmark of one convolution shape (random data); `variant` bit flags switch
 * parts of the kernel off to attribute time (0 = the production kernel).  Mean launch ms out. */
int sivo_debug_conv(int N, int Cin, int Cout, int H, int W, int ks, int iters, int variant, double *ms_out);

/* Diagnostic / test: the f16x3 GEMM alone on host operands.  V [36][C][Pp] fp32 (Pp = P rounded up to 128), U [36][C][Kp]
 * fp32, M [36][Kp][Pp] out (fp32, scales multiplied back out); C % 32 == 0, Kp % 128 == 0.  iters > 0: mean launch time
 * (ms) of `iters` further launches in *ms_out. */
int sivo_debug_h3_gemm(int C, int Kp, int P, const float *V, const float *U, float vscale, float *M, int iters, double *ms_out);

/* Diagnostic / test: the direct 3x3 convolution on the fp16 matrix cores (f16x3, conv3_h3.hip) alone.  d_in, d_mask, d_out:
 * DEVICE pointers — d_mask null: d_in is (N, Cin, H, W); else d_in is the pooled tensor (N, Cin, H/2, W/2) and d_mask its
 * u8 window codes (dy * 2 + dx), the layer reading through the Upsample as in the network.  Wt (Cout, Cin, 3, 3), scale,
 * shift (Cout): HOST arrays; out = act(scale * conv + shift).  vscale: the power of two the input is multiplied with before
 * it is split (max |in| * vscale well below 65504).  Cin % 16 == 0, Cin >= 32, Cout % 64 == 0.  *overflowed = 1 when a scaled
 * input left the fp16 range.  iters > 0: mean launch time (ms) of `iters` further launches in *ms_out. */
int sivo_debug_conv3_h3_dev(int N, int Cin, int Cout, int H, int W, const float *d_in, const uint8_t *d_mask, const float *Wt,
                            const float *scale, const float *shift, int relu, float vscale, float *d_out, int iters,
                            double *ms_out, int *overflowed);

/* The same kernel with its packed activation forms (conv3_h3.hip; pk_format.hip converts on the device).  mode bit 0: the
 * input goes in PACKED (d_in, fp32, is packed with vscale into a zero-bordered tensor first; with d_mask the pooled tensor is
 * packed and the window codes are re-laid per channel octet); bit 1: the output is written PACKED with out_vscale (and
 * unpacked into d_out, fp32, afterwards: (hi + lo) / out_vscale); bit 2: pad the packed planes by 3 extra rows / 5 extra
 * columns beyond what the tiling needs.  *overflowed = 1 when a value times its scale left the fp16 range.  iters: as above
 * (the conversion kernels are outside the timed launches). */
int sivo_debug_conv3_h3_pk_dev(int N, int Cin, int Cout, int H, int W, const float *d_in, const uint8_t *d_mask, const float *Wt,
                               const float *scale, const float *shift, int relu, float vscale, float out_vscale, int mode,
                               float *d_out, int iters, double *ms_out, int *overflowed);

/* The classifier convolution fused with the Monte-Carlo post-processing on the fp16 matrix cores (conv_cls_h3.hip) alone.
 * d_in: DEVICE fp32 (T, Cin, H, W), packed on the device with vscale first (Cin % 32 == 0, Cin <= 96); Wt (C, Cin, 3, 3), scale,
 * shift (C): HOST arrays; outputs on the DEVICE: d_logits (T, C, H, W) fp32 and the maps of the f64 mean over the T samples
 * (d_classes u8, d_confidence / d_entropy f64, each H x W).  iters > 0: mean launch time (ms) of `iters` further launches
 * without the logits output. */
int sivo_debug_conv_cls_h3_dev(int T, int Cin, int C, int H, int W, const float *d_in, const float *Wt, const float *scale,
                               const float *shift, int relu, float vscale, float *d_logits, uint8_t *d_classes,
                               double *d_confidence, double *d_entropy, int iters, double *ms_out);

/* The co-residency mitigation: dynamic LDS per CU the launchers of the kernels that issue LDS-DMA in inline assembly asked
 * for since the last reset — out[0] wino4_gemm_h3_kernel, [1] conv3_h3_kernel, [2] conv_cls_h3_kernel (both of its workgroups on a CU),
 * [3] conv7_h3_kernel; the smallest request of each, 0 = not launched.  163840 = nobody else's LDS fits on that CU. */
int sivo_debug_lds_claims(uint32_t out[4], int reset);

#ifdef __cplusplus
}
#endif
#endif
