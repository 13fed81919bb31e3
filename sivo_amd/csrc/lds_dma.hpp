// lds_dma.hpp — LDS-DMA (global_load_lds_dwordx4) issued through inline assembly.
//
// hipcc's waitcnt insertion knows that __builtin_amdgcn_global_load_lds writes LDS, cannot tell WHERE for a kernel whose
// staging buffers live in one LDS array, and therefore puts `s_waitcnt vmcnt(0)` in front of the first LDS read that follows a
// DMA in the same wave (seen in the .s of every kernel here that stages its weight slab by DMA and then reads its patch:
// the wave drains all memory traffic it has just started for the NEXT K-chunk before it computes the current one, so no
// latency is ever hidden).  An asm DMA is invisible to that pass; the kernel waits for it itself with s_waitcnt vmcnt(N)
// (vector-memory loads complete in issue order) in front of the barrier that hands the stage over.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sivo {

// LDS byte address of a pointer into a __shared__ array, as a wave-uniform scalar (M0 operand of the DMA)
__device__ __forceinline__ uint32_t lds_addr_uniform(const void *p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(const __attribute__((address_space(3))) void *)p);
}

// 64 lanes x 16 bytes: lane l copies the 16 bytes at gsrc (its own pointer) to LDS address lds_byte_addr + 16 l.
// M0 is saved and restored inside the statement (the compiler keeps its own values there).
__device__ __forceinline__ void lds_dma16(const void *gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

// The same with a scalar base: lane l copies the 16 bytes at sbase + voff (its own byte offset) to lds_byte_addr + 16 l.
__device__ __forceinline__ void lds_dma16_s(const void *sbase, uint32_t voff, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

// 64 lanes x 4 bytes with a scalar base: lane l copies the dword at sbase + voff to LDS address lds_byte_addr + 4 l (used as an
// L2 prefetch probe: a load without a register destination — nothing the compiler could move or reuse while it is in flight).
__device__ __forceinline__ void lds_dma4_s(const void *sbase, uint32_t voff, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

// Workgroup barrier without the vmcnt(0) drain of __syncthreads(): this wave's LDS traffic done + s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace sivo
