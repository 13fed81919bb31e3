// mfma_rate.hip — what a stream of independent v_mfma_f32_32x32x16_f16 sustains on gfx950, by where the accumulators live (arch VGPRs or
// AGPRs), how many of them rotate, and how many waves share a SIMD.  Stand-alone (no library): on the GPU box
//     hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Question behind it (NOTEBOOK 11.1): the f16x3 GEMM with NOTHING but its MFMAs (no loads, no LDS reads, no barrier) keeps the matrix cores
// busy 0.76 of the cycles — is that the kernel's bookkeeping or the instruction's ceiling with eight rotating 16-register accumulators?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool AG>
__global__ __launch_bounds__(512) void rate_kernel(const half8 *in, float *out, int iters) {
    half8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool AG>
static void run(const char *name, int threads, const half8 *d_in, float *d_out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<NACC, AG>), dim3(256), dim3(threads), 0, nullptr, d_in, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((rate_kernel<NACC, AG>), dim3(256), dim3(threads), 0, nullptr, d_in, d_out, iters);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 256.0 * (threads / 64) * (double)iters * 4 * NACC;
    const double flops = mfmas * 32768.0;
    // cycles of the matrix pipe this needs at 32 cycles per MFMA per SIMD: mfmas / 1024 SIMDs * 32
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.0f MFMA-cycles per SIMD; busy at 2.4 GHz: %.3f)\n", name, ms, flops / ms / 1e9, mfmas / 1024 * 32,
           mfmas / 1024 * 32 / (ms * 1e-3 * 2.4e9));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const bool zeros = argc > 2 && atoi(argv[2]) == 1;
    std::vector<_Float16> h(128 * 8);
    srand(1);
    for (auto &x : h) x = zeros ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.f);
    half8 *d_in; float *d_out;
    hipMalloc((void **)&d_in, h.size() * 2); hipMalloc((void **)&d_out, 256 * 512 * 4);
    hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("operands: %s; 256 workgroups; %d x 4 x NACC MFMAs per wave\n", zeros ? "all zero" : "random in [-1, 1]", iters);
    run<8, false>("8 accumulators in VGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    run<8, true>("8 accumulators in AGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    run<8, false>("8 accumulators in VGPRs, 1 wave / SIMD", 256, d_in, d_out, iters);
    run<8, true>("8 accumulators in AGPRs, 1 wave / SIMD", 256, d_in, d_out, iters);
    run<4, false>("4 accumulators in VGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    run<4, true>("4 accumulators in AGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    run<2, false>("2 accumulators in VGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    run<2, true>("2 accumulators in AGPRs, 2 waves / SIMD", 512, d_in, d_out, iters);
    return 0;
}
