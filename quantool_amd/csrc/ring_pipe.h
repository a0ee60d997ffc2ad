// Device helpers shared by the LDS-ring MFMA kernels (xtx.hip, gemm3_tn.hip): LDS-DMA issue from
// inline asm, transposing fragment reads, the 16-bit MFMA k-step and counted vmcnt waits.
//
// LDS-DMA from inline asm: hipcc does not track it, so it inserts no vmcnt drain in front of later
// ds_reads or barriers.  Every completion is ordered by hand (counted vmcnt + s_barrier in the kernels).
// saddr form: 64-bit scalar base + 32-bit per-lane byte offset; M0 = wave-uniform LDS destination.
// M0 is written in the statement that reads it and is not restored: nothing else in these kernels uses
// M0 (LDS instructions need none on gfx9+), which the build checks by grepping each translation unit's
// ISA for m0 outside the asm blocks (csrc/build.py:audit_m0).
#pragma once
#include "common.h"

// Both LDS-DMA instructions of one unit (A panel, B panel) in one statement; one source matrix.
__device__ __forceinline__ void glds16_pair(unsigned voffA, unsigned voffB, const void* sbase, unsigned ldsA,
                                            unsigned ldsB) {
    asm volatile(
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2"
        :
        : "v"(voffA), "v"(voffB), "s"(sbase), "s"(ldsA), "s"(ldsB)
        : "memory");
}
// The same with one scalar base per panel (A and B panels come from different matrices / planes).
__device__ __forceinline__ void glds16_pair2(unsigned voffA, unsigned voffB, const void* sbaseA, const void* sbaseB,
                                             unsigned ldsA, unsigned ldsB) {
    asm volatile(
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %2\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3"
        :
        : "v"(voffA), "v"(voffB), "s"(sbaseA), "s"(sbaseB), "s"(ldsA), "s"(ldsB)
        : "memory");
}
__device__ __forceinline__ void glds16_snapshot(unsigned voff, const void* sbase, unsigned lds_dst) {
    // 1 KiB of progress words -> LDS scratch; sc1: served by L2, never by this CU's L1 copy of the line
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1 sc1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

__device__ __forceinline__ s16x8 tr_load8(const char* lds_addr) {
    // two transposing reads: k rows +0..3 and +4..7 (rows are 256 B apart -> +1024 B)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS s16x4*)(lds_addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS s16x4*)(lds_addr + 1024));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one k-step (16 rows) of a 32x32 output tile; bf16 and fp16 products are both exact in the fp32
// accumulator and run at the same MFMA rate
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
