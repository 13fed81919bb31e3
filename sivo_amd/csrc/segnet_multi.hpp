// segnet_multi.hpp — multi-device form of the SegNet handle (segnet_multi.cpp), used by segnet.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sivo_hip.h"

namespace sivo {
struct SegnetMulti;
SegnetMulti *segnet_multi_create(const char *text, size_t len, int t_total, const float *weights, size_t n_weights,
                                 const int *device_ids, int ndev, const SivoSegnetOptions *opts);
void segnet_multi_destroy(SegnetMulti *M);
void segnet_multi_shape(const SegnetMulti *M, int32_t *T, int32_t *H, int32_t *W, int32_t *classes, int32_t *ndev);
void segnet_multi_segment(SegnetMulti *M, const uint8_t *bgr, int rows, int cols, uint64_t seed, uint8_t *classes, double *confidence,
                          double *entropy);
// segnet.cpp: forward of n samples of a single-device handle, softmax + sum written pixel-chunk-major
// ([hw / chunk][classes][chunk]); chunk == hw is the plain [classes][hw] layout.
// The sums are the f64 accumulators themselves (not rounded to fp32): added over the devices in f64 they give the
// reference's f64 mean (bayesian_segnet.cpp:291-294) up to an f64 rounding, whatever the number of devices.
void segnet_forward_chunked(sivo_segnet_t h, const uint8_t *d_bgr, int n, int sample0, uint64_t seed, double *d_sum_chunked,
                            int64_t chunk, hipStream_t st, const void *d_slots = nullptr, int world = 0);
// true once after a frame of the handle raised the fp16 overflow flag of the f16x3 GEMM (the handle is on bf16x6 from then on)
// row bands of the sample-invariant prefix (segnet.cpp PrefixBands): slot size for `world` ranks (0: this net cannot be split, every
// device recomputes the prefix), one rank's band -> its slot, and the forward above on the gathered slots (slots != null)
size_t segnet_prefix_slot_bytes(sivo_segnet_t h, int world);
void segnet_prefix_band(sivo_segnet_t h, const uint8_t *d_bgr, int rank, int world, void *d_slot, hipStream_t st);
bool segnet_fp16_overflowed(sivo_segnet_t h, bool *backed_off);
void segnet_fp16_back_off(sivo_segnet_t h, bool already_backed_off);
}  // namespace sivo
