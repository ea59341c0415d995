// f32-grade products on the bf16 matrix pipe with operands split IN REGISTERS on the way into the MFMA (mdbn_thin.hip,
// scripts/experiments/mdbn_oneshot.hip): the exact three-way split of gemm_bf16x6_kernel / the plane path, applied to fragments read from a
// float32 LDS image (or straight from global memory) instead of to tiles staged as planes.
#pragma once
#include <hip/hip_runtime.h>

namespace mdbn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tu32x4 __attribute__((ext_vector_type(4)));

// exact three-way split of two floats into packed bf16 pairs (low half = the first value): piece = the upper half of the
// f32 (which IS a bf16), remainder = value - piece (exact); three 8-bit significands cover the 24 bits.  Full-rate
// instructions only (v_and / v_sub / v_perm), as gemm_bf16x6_kernel's producers (mdbn_kernels.hip, x6_split2).
__device__ __forceinline__ void th_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3)
{
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u);
    const float rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
    const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u);
    const float sb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);            // (hi16(b) << 16) | hi16(a)
    p2 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
}

// eight consecutive-k floats -> the three bf16 fragments of v_mfma_f32_32x32x16_bf16 (element j = k 8 h + j)
__device__ __forceinline__ void th_split8(const float (&f)[8], tbf16x8 (&frag)[3])
{
    tu32x4 q[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b, c;
        th_split2(f[2 * e], f[2 * e + 1], a, b, c);
        q[0][e] = a; q[1][e] = b; q[2][e] = c;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) frag[pl] = __builtin_bit_cast(tbf16x8, q[pl]);
}

// acc += A B on the bf16 matrix pipe at f32 accuracy: AP = 3: the six piece products with i + j <= 4, smallest first
// (gemm_bf16x6_kernel's order); AP = 1: A is exactly one piece (0/1 samples): three products
template <int AP>
__device__ __forceinline__ void th_mma(f32x16& acc, const tbf16x8 (&a)[3], const tbf16x8 (&b)[3])
{
    if constexpr (AP == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
}


}  // namespace mdbn
