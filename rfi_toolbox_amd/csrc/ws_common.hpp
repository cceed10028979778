// Device helpers shared by the wave-specialised float32 (3 x bf16) kernels (conv_ws.hip, gemm_ws.hip).
#pragma once
#include "planes.hpp"

// RFI_DIAG_STAMPS builds (tools/build_diag.sh, never shipped): per-wave cycle sums of the phases of a kernel's main loop
#ifdef RFI_DIAG_STAMPS
#define WS_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define WS_ACC(i, a_, b_) st_[i] += (b_) - (a_)
#else
#define WS_T(v)
#define WS_ACC(i, a_, b_)
#endif

namespace rfi {
namespace ws {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ unsigned cvt_pair(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// v = h + m + l, each piece RNE-rounded to bf16 (exact: 3 x 8 significand bits cover float32's 24)
// (v_dot2c_f32_bf16 forms a residual "a - float(h.lo)" in one instruction, bit-identically (tools/probe_dot2_split.hip), but
// beside MFMA waves it is SLOWER than shift + subtract -- producer time per item 5.1 k -> 10.1 k cycles, round 3 -- the dot
// instructions appear to share the matrix pipe)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pair(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pair(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pair(sa, sb);
}
// pieces [0] = h, [1] = m, [2] = l; small terms first, the three products below 2^-24 |a b| dropped
__device__ __forceinline__ f32x16 mma3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// a bare s_barrier: no s_waitcnt in front of it (__syncthreads() would also wait for the consumers' output stores and the
// producers' prefetched loads); the "memory" clobber keeps the compiler from moving LDS accesses across it
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// s_waitcnt vmcnt(n) as the BUILTIN: the compiler's own wait bookkeeping sees it (an asm wait it would not), so the LDS
// writes that follow are not preceded by a conservative vmcnt(0) for the LDS-DMA issued before it
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xf) | (0x7 << 4) | (0xf << 8) | ((N >> 4) << 14));     // expcnt, lgkmcnt: no wait
}

}  // namespace ws
}  // namespace rfi
