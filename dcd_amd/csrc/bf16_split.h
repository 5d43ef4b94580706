// Split-bf16 operands for the bf16 matrix cores (shared by conv.hip; dcn_v2.hip keeps its own copy in dcn_v2_bf16x3.inc).
// x = hi + lo + r, hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest-even), |r| <= 2^-17 |x|; a product is evaluated as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with fp32 accumulation: ~2^-16 relative error per product.
#pragma once

typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sp_u32x2 __attribute__((ext_vector_type(2)));
typedef short sp_s16x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x16 __attribute__((ext_vector_type(16)));

struct SpSplit8 {
    sp_bf16x8 hi, lo;
};

// (a, b) -> packed bf16 pairs (a in the low half)
__device__ __forceinline__ void sp_split_pair(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{a, b}, sp_bf16x2));
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{a - ah, b - bh}, sp_bf16x2));
}

__device__ __forceinline__ SpSplit8 sp_split8(const float (&v)[8])
{
    sp_u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hh, ll;
        sp_split_pair(v[2 * q], v[2 * q + 1], hh, ll);
        h[q] = hh;
        l[q] = ll;
    }
    SpSplit8 s;
    s.hi = __builtin_bit_cast(sp_bf16x8, h);
    s.lo = __builtin_bit_cast(sp_bf16x8, l);
    return s;
}

// eight fp32 values -> bf16 (round to nearest even), no low part: the operand of the ONE-product form (DCD_PREC_BF16)
__device__ __forceinline__ sp_bf16x8 sp_round8(const float (&v)[8])
{
    sp_u32x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{v[2 * q], v[2 * q + 1]}, sp_bf16x2));
    return __builtin_bit_cast(sp_bf16x8, h);
}

// acc += A * B over K = 16 (v_mfma_f32_32x32x16_bf16), small terms first
__device__ __forceinline__ sp_f32x16 sp_mfma_x3(const sp_bf16x8 &a_hi, const sp_bf16x8 &a_lo, const SpSplit8 &b, sp_f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.hi, acc, 0, 0, 0);
    return acc;
}
