// diag_kernels.hip — diagnostic build only (libsivo_hip_diag.so; empty in the product): a kernel that does nothing but hold a chosen
// amount of every CU's LDS for a chosen time, optionally with LDS traffic of its own, so that other kernels' workgroups are placed
// BESIDE it — at LDS bases they never see when the CU is theirs (tools/coresident_probe.py occupant).
#ifdef SIVO_DIAG
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "lds_dma.hpp"

namespace sivo {

// mode 0: sleeps; 1: ds_read / ds_write over its own LDS; 2: LDS-DMA (global_load_lds_dwordx4) from `src` into its own LDS
__global__ __launch_bounds__(512) void occupy_kernel(int lds_bytes, int mode, long long cycles, const uint32_t *src, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char occ_lds[];
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();        // 100 MHz
    uint32_t acc = 0;
    const int words = lds_bytes / 4;
    if (mode == 1) for (int i = threadIdx.x; i < words; i += 512) reinterpret_cast<uint32_t *>(occ_lds)[i] = (uint32_t)i;
    __syncthreads();
    const uint32_t base = lds_addr_uniform(occ_lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int k = 0;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < cycles) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else if (mode == 1) {
            const int i = (threadIdx.x * 4 + 2048 * k) % (words - 4);
            const uint4 v = *reinterpret_cast<const uint4 *>(occ_lds + (size_t)(i & ~3) * 4);
            acc += v.x ^ v.y ^ v.z ^ v.w;
            reinterpret_cast<uint32_t *>(occ_lds)[(i + 1) % words] = acc;
        } else {
            const int pieces = lds_bytes / 1024;
            const int piece = (wave + 8 * k) % pieces;
            lds_dma16_s(src, (uint32_t)(((piece * 64 + lane) * 16) % (1 << 20)), (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)piece * 1024)));
            if ((k & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ++k;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (mode && sink && acc == 0x12345678u) sink[0] = acc + occ_lds[threadIdx.x];
}

void launch_occupy(int lds_bytes, int mode, int microseconds, const uint32_t *src, uint32_t *sink, hipStream_t s) {
    static int attr_set[64] = {0};
    if (FirstUse once(attr_set); once)
        SIVO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(occupy_kernel, dim3(256), dim3(512), (size_t)lds_bytes, s, lds_bytes, mode, (long long)microseconds * 100, src, sink);   // s_memrealtime: 100 MHz
    SIVO_HIP(hipGetLastError());
}

}  // namespace sivo
#endif
