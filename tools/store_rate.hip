// store_rate.hip — what a CU's store path sustains for the access patterns the f16x3 GEMM could use for M (NOTEBOOK 11.1: the M stores cost
// 24 - 39 % of the kernel and neither a start skew of the workgroups nor a relaxed vmcnt wait moves that).  Stand-alone; on the GPU box
//     hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
// 256 workgroups x 512 threads; every wave writes `items` x 32 KB in bursts of 128 (dword) or 32 (dwordx4) store instructions, as a GEMM
// wave does at the end of an item (4 tile blocks x 2 cout blocks x 16 accumulator registers); rows are Pp x 4 bytes apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// PATTERN 0: dword, lane ln -> 4 bytes of a 128-byte run, lh -> a second row (the product's: C/D of the 32 x 32 MFMA with tiles as columns)
//         1: dwordx4, 32 lanes -> 512 contiguous bytes of a row, lh -> a second row
//         2: dwordx4, lane ln -> 16 bytes of ITS OWN row (32 rows per instruction), lh -> the next 16 bytes (operands swapped: couts as columns)
//         3: dword, 64 lanes -> 256 contiguous bytes
template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(float *M, int64_t Pp, int items, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 31, lh = lane >> 5;
    const int wc = wave & 3, wt = wave >> 2;
    float v = (float)threadIdx.x;
    for (int it = 0; it < items; ++it) {
        // item (it, blockIdx.x): 256 couts x 256 tiles at rows k0.., columns p0..
        const int64_t p0 = ((int64_t)it * gridDim.x + blockIdx.x) * 256 % Pp;
        float *base = M + (int64_t)(((int64_t)it * gridDim.x + blockIdx.x) * 256 / Pp % 36) * 256 * Pp;      // "position": a 256-row slab
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k0 = (2 * wc + c) * 32, q0 = (wt * 4 + t) * 32;
                if (PATTERN == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) base[(int64_t)(k0 + 8 * (r >> 2) + 4 * lh + (r & 3)) * Pp + p0 + q0 + ln] = v;
                } else if (PATTERN == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int h = 0; h < 1; ++h)
                            *reinterpret_cast<f32x4 *>(base + (int64_t)(k0 + 8 * r + 4 * lh + (ln >> 3)) * Pp + p0 + q0 + 4 * (ln & 7)) = f32x4{v, v, v, v};
                } else if (PATTERN == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4 *>(base + (int64_t)(k0 + ln) * Pp + p0 + q0 + 8 * r + 4 * lh) = f32x4{v, v, v, v};
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) base[(int64_t)(k0 + 2 * r + (lane >> 5)) * Pp + p0 + q0 + ln] = v;
                }
            }
        // the next item's multiply phase: nothing but time (spin x 64 cycles), so that bursts can drain if the hardware lets them
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(1);
        v += 1.f;
    }
}

template <int PATTERN>
static void run(const char *name, float *M, int64_t Pp, int items, int spin) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((store_kernel<PATTERN>), dim3(256), dim3(512), 0, nullptr, M, Pp, 1, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((store_kernel<PATTERN>), dim3(256), dim3(512), 0, nullptr, M, Pp, items, spin);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * items * 256 * 256 * 4;
    printf("%-58s spin %5d: %8.3f ms  %7.1f GB/s  (%.1f us per item and CU)\n", name, spin, ms, bytes / ms / 1e6, ms * 1e3 / items);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 5;
    const int64_t Pp = 4224;
    float *M;
    hipMalloc((void **)&M, (size_t)36 * 512 * Pp * 4);
    for (int spin : {0, 300}) {          // 300 x 64 cycles ~ 8 us at 2.4 GHz between bursts
        run<0>("dword, 2 x 128-byte runs per instruction (product)", M, Pp, items, spin);
        run<1>("dwordx4, 8 lanes per row: 8 x 128-byte runs per instruction", M, Pp, items, spin);
        run<2>("dwordx4, a row per lane: 32 x 32-byte runs per instruction", M, Pp, items, spin);
        run<3>("dword, 2 x 128-byte runs in adjacent rows", M, Pp, items, spin);
    }
    return 0;
}
