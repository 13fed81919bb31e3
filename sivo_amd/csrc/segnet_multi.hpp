// segnet_multi.hpp — multi-device form of the SegNet handle (segnet_multi.cpp), used by segnet.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sivo_hip.h"

namespace sivo {
struct SegnetMulti;
SegnetMulti *segnet_multi_create(const char *text, size_t len, int t_total, const float *weights, size_t n_weights,
                                 const int *device_ids, int ndev);
void segnet_multi_destroy(SegnetMulti *M);
void segnet_multi_shape(const SegnetMulti *M, int32_t *T, int32_t *H, int32_t *W, int32_t *classes, int32_t *ndev);
void segnet_multi_segment(SegnetMulti *M, const uint8_t *bgr, int rows, int cols, uint64_t seed, uint8_t *classes, double *confidence,
                          double *entropy);
// segnet.cpp: forward of n samples of a single-device handle, softmax + sum written pixel-chunk-major
// ([hw / chunk][classes][chunk]); chunk == hw is the plain [classes][hw] layout.
void segnet_forward_chunked(sivo_segnet_t h, const uint8_t *d_bgr, int n, int sample0, uint64_t seed, float *d_sum_chunked,
                            int64_t chunk, hipStream_t st);
}  // namespace sivo
