// h3_split.hpp — the fp32 -> (fp16 hi, fp16 lo) split of the f16x3 kernels (conv_wino4.hip's transform kernels,
// conv3_h3.hip's patch staging).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sivo {

// fp32 -> packed fp16 pair (hi | lo << 16) of x * scale: hi = fp16(xs), lo = fp16(xs - hi); exact to 2^-22 |xs|
__device__ __forceinline__ uint32_t wino4_pack_h3(float x, float scale, bool &bad) {
    const float xs = x * scale;
    const _Float16 hi = (_Float16)xs;
    const _Float16 lo = (_Float16)(xs - (float)hi);
    bad |= !(__builtin_fabsf(xs) <= 65504.f);
    return (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
}

}  // namespace sivo
