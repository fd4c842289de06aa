// bf16 MFMA operand helpers shared by the low-precision kernels (enc_block_bf16.hip, dec_chain.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace msm {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));

// four floats -> four bf16 (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    const bf16x2_t lo = __builtin_convertvector(f32x2{a, b}, bf16x2_t), hi = __builtin_convertvector(f32x2{c, d}, bf16x2_t);
    const u32x2b u = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(bf16x4, u);
}
// v_mfma_f32_16x16x16_bf16: A lane (i = l & 15, kq = l >> 4) holds A[i][4 kq .. 4 kq + 3], B lane (j, kq) holds B[4 kq .. + 3][j],
// D lane (j = l & 15, rq = l >> 4) holds D[4 rq + r][j]
__device__ __forceinline__ f32x4b mfma_bf16(bf16x4 a, bf16x4 b, f32x4b c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// An fp32 fragment as TWO bf16 operands, x = hi + lo up to 2^-17 |x|.
struct Split4 {
    bf16x4 hi, lo;
};
__device__ __forceinline__ float bf16_hi_as_float(unsigned packed, int idx) { return __uint_as_float(idx ? (packed & 0xffff0000u) : (packed << 16)); }
__device__ __forceinline__ Split4 split4(float a, float b, float c, float d) {
    const bf16x2_t h0 = __builtin_convertvector(f32x2{a, b}, bf16x2_t), h1 = __builtin_convertvector(f32x2{c, d}, bf16x2_t);
    const unsigned u0 = __builtin_bit_cast(unsigned, h0), u1 = __builtin_bit_cast(unsigned, h1);
    Split4 r;
    r.hi = __builtin_bit_cast(bf16x4, u32x2b{u0, u1});
    r.lo = pack4(a - bf16_hi_as_float(u0, 0), b - bf16_hi_as_float(u0, 1), c - bf16_hi_as_float(u1, 0), d - bf16_hi_as_float(u1, 1));
    return r;
}

}  // namespace msm
