// pk_format.hip — the packed activation format between two f16x3 layers (conv3_h3.hip) and its converters.
//
//   P[n][C / 8][plane: hi, lo][Hp][Wp][8 halfs]
// One 16-byte piece holds channels 8 o .. 8 o + 7 of one pixel in one plane: exactly the B fragment of one lane of
// v_mfma_f32_32x32x16_f16 — what conv3_h3.hip keeps in LDS.  hi = fp16(x * scale), lo = fp16(x * scale - hi) with the
// consumer's calibrated power of two `scale`: the same 4 bytes per element as the fp32 blob, exact to 2^-22 |x|.  The image
// sits at rows / columns 1 .. of a zero-bordered (Hp, Wp) plane, so that a consumer's halo and the overhang of its partial
// items are plain in-bounds reads of zeros (its staging is address arithmetic only: LDS-DMA).  Producers write the interior only.
//
// The kernels here are the format's edge: fp32 NCHW -> packed for a producer that does not write the format itself,
// packed -> fp32 NCHW for sivo_segnet_blob and the tests, and the pooling window codes re-laid per channel octet for a
// consumer that reads a packed tensor through an Upsample.  All three are streaming kernels (HBM-bound, 8 B per element).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "segnet_kernels.hpp"

namespace sivo {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

size_t pk_bytes(int N, int C, int Hp, int Wp) { return (size_t)N * (size_t)(C / 8) * 2 * (size_t)Hp * Wp * 16; }

// thread = (sample, octet, y, x): 8 loads one channel plane apart (each coalesced over x), two 16-byte stores (coalesced over x)
__global__ __launch_bounds__(256) void pk_pack_kernel(const float *in, int64_t in_sample_stride, unsigned char *out, int N, int C, int H, int W,
                                                      int Hp, int Wp, float scale, uint32_t *h3_flag) {
    const int64_t total = (int64_t)N * (C / 8) * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int o = (int)(t % (C / 8));
    const int n = (int)(t / (C / 8));
    const float *src = in + (int64_t)n * in_sample_stride + ((int64_t)(o * 8) * H + y) * W + x;
    uint32_t pr[8];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs = src[(int64_t)e * H * W] * scale;
        const _Float16 hi = (_Float16)xs;
        const _Float16 lo = (_Float16)(xs - (float)hi);
        bad |= !(__builtin_fabsf(xs) <= 65504.f);
        pr[e] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
    }
    u32x4 hv, lv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hv[j] = __builtin_amdgcn_perm(pr[2 * j + 1], pr[2 * j], 0x05040100u);
        lv[j] = __builtin_amdgcn_perm(pr[2 * j + 1], pr[2 * j], 0x07060302u);
    }
    const int64_t plane = (int64_t)Hp * Wp * 16;
    unsigned char *dst = out + ((int64_t)(n * (C / 8) + o) * 2) * plane + ((int64_t)(y + 1) * Wp + x + 1) * 16;
    *reinterpret_cast<u32x4 *>(dst) = hv;
    *reinterpret_cast<u32x4 *>(dst + plane) = lv;
    if (bad && h3_flag) atomicOr(h3_flag, 1u);
}

__global__ __launch_bounds__(256) void pk_unpack_kernel(const unsigned char *in, float *out, int N, int C, int H, int W, int Hp, int Wp, float inv_scale) {
    const int64_t total = (int64_t)N * (C / 8) * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H); t /= H;
    const int o = (int)(t % (C / 8));
    const int n = (int)(t / (C / 8));
    const int64_t plane = (int64_t)Hp * Wp * 16;
    const unsigned char *src = in + ((int64_t)(n * (C / 8) + o) * 2) * plane + ((int64_t)(y + 1) * Wp + x + 1) * 16;
    const u32x4 hv = *reinterpret_cast<const u32x4 *>(src), lv = *reinterpret_cast<const u32x4 *>(src + plane);
    float *dst = out + (((int64_t)n * C + o * 8) * H + y) * W + x;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned short hb = (unsigned short)(hv[e >> 1] >> (16 * (e & 1))), lb = (unsigned short)(lv[e >> 1] >> (16 * (e & 1)));
        const float v = (float)__builtin_bit_cast(_Float16, hb) + (float)__builtin_bit_cast(_Float16, lb);      // exact in fp32
        dst[(int64_t)e * H * W] = v * inv_scale;
    }
}

// thread = (sample, octet, pooled y, pooled x): 8 code bytes in, one dword out (byte k: bit e set when channel 8 o + e has code k)
__global__ __launch_bounds__(256) void pool_bits_kernel(const uint8_t *codes, uint32_t *bits, int N, int C, int Hq, int Wq, int Hp, int Wp) {
    const int64_t total = (int64_t)N * (C / 8) * Hq * Wq;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wq);
    int64_t t = i / Wq;
    const int y = (int)(t % Hq); t /= Hq;
    const int o = (int)(t % (C / 8));
    const int n = (int)(t / (C / 8));
    const uint8_t *src = codes + (((int64_t)n * C + o * 8) * Hq + y) * Wq + x;
    uint32_t w = 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t k = src[(int64_t)e * Hq * Wq] & 3u;
        w |= 1u << (8 * k + e);
    }
    bits[((int64_t)(n * (C / 8) + o) * Hp + y + 1) * Wp + x + 1] = w;
}

void launch_pk_pack(const float *in, int64_t in_sample_stride, void *out, int N, int C, int H, int W, int Hp, int Wp, float scale, uint32_t *h3_flag,
                    hipStream_t s) {
    const int64_t total = (int64_t)N * (C / 8) * H * W;
    if (total <= 0) return;
    hipLaunchKernelGGL(pk_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_sample_stride, static_cast<unsigned char *>(out), N, C, H,
                       W, Hp, Wp, scale, h3_flag);
}

void launch_pk_unpack(const void *in, float *out, int N, int C, int H, int W, int Hp, int Wp, float scale, hipStream_t s) {
    const int64_t total = (int64_t)N * (C / 8) * H * W;
    if (total <= 0) return;
    hipLaunchKernelGGL(pk_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned char *>(in), out, N, C, H, W, Hp, Wp,
                       1.f / scale);
}

void launch_pool_bits(const uint8_t *codes, uint32_t *bits, int N, int C, int Hq, int Wq, int Hp, int Wp, hipStream_t s) {
    const int64_t total = (int64_t)N * (C / 8) * Hq * Wq;
    if (total <= 0) return;
    hipLaunchKernelGGL(pool_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, codes, bits, N, C, Hq, Wq, Hp, Wp);
}

}  // namespace sivo
