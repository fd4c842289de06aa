// bf16 MFMA operand helpers shared by the low-precision kernels (enc_block_split.hip, dec_chain.hip, attention.hip, kv_proj.hip).
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
// four floats -> four IEEE halves (round to nearest even): the tensors between the bf16 plan's encoder kernels are stored as
// fp16 -- same bytes as bf16, three more mantissa bits, and their values are O(1) (see csrc/enc_lp.hip)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }      // (no infinities from an outlier)
__device__ __forceinline__ u32x2b pack4h(float a, float b, float c, float d) {
    const f16x2_t lo = {(_Float16)clamp_h(a), (_Float16)clamp_h(b)}, hi = {(_Float16)clamp_h(c), (_Float16)clamp_h(d)};
    return u32x2b{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
// v_mfma_f32_16x16x32_f16: operand layout of the bf16 K = 32 form (below), IEEE half operands, the same issue rate
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 cvt8h(float a, float b, float c, float d, float e, float f, float g, float h) {
    return f16x8{(_Float16)clamp_h(a), (_Float16)clamp_h(b), (_Float16)clamp_h(c), (_Float16)clamp_h(d), (_Float16)clamp_h(e), (_Float16)clamp_h(f),
                 (_Float16)clamp_h(g), (_Float16)clamp_h(h)};
}
__device__ __forceinline__ f32x4b mfma_f16k32(f16x8 a, f16x8 b, f32x4b c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ u32x2b pack4h_nc(float a, float b, float c, float d) {          // no clamp: callers with bounded values (unit vectors)
    const f16x2_t lo = {(_Float16)a, (_Float16)b}, hi = {(_Float16)c, (_Float16)d};
    return u32x2b{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
__device__ __forceinline__ float half_lo(unsigned packed) { return (float)__builtin_bit_cast(f16x2_t, packed)[0]; }
__device__ __forceinline__ float half_hi(unsigned packed) { return (float)__builtin_bit_cast(f16x2_t, packed)[1]; }
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


// ---- exact three-term splits and K = 32 MFMAs (fp32 accuracy on the bf16 matrix pipe; see enc_block_split.hip) ------------
// v_mfma_f32_16x16x32_bf16: A lane (i = l & 15, kq = l >> 4) holds A[i][8 kq .. 8 kq + 7], B lane (j, kq) holds B[8 kq .. + 7][j],
// D as the K = 16 form.  It is the full-rate bf16 MFMA of gfx950; the K = 16 instruction issues at half its rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4b mfma_bf16k32(bf16x8 a, bf16x8 b, f32x4b c) {
#if defined(ES_EXP) && ES_EXP == 2      // tuning builds of enc_block_split.hip only: no MFMAs
    const u32x4b ua = __builtin_bit_cast(u32x4b, a), ub = __builtin_bit_cast(u32x4b, b);
    c[0] += __uint_as_float(ua.x ^ ub.x);
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) {
    const u32x2b ua = __builtin_bit_cast(u32x2b, a), ub = __builtin_bit_cast(u32x2b, b);
    return __builtin_bit_cast(bf16x8, u32x4b{ua.x, ua.y, ub.x, ub.y});
}
struct Split3 {                                // x = h + m + l, four values
    bf16x4 h, m, l;
};
struct Split3x8 {                              // the same for the eight values a lane feeds into one K = 32 MFMA
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split3 split3(float a, float b, float c, float d) {
    Split3 r;
    r.h = pack4(a, b, c, d);
    const u32x2b uh = __builtin_bit_cast(u32x2b, r.h);
    const float ra = a - bf16_hi_as_float(uh.x, 0), rb = b - bf16_hi_as_float(uh.x, 1), rc = c - bf16_hi_as_float(uh.y, 0),
                rd = d - bf16_hi_as_float(uh.y, 1);
    r.m = pack4(ra, rb, rc, rd);
    const u32x2b um = __builtin_bit_cast(u32x2b, r.m);
    r.l = pack4(ra - bf16_hi_as_float(um.x, 0), rb - bf16_hi_as_float(um.x, 1), rc - bf16_hi_as_float(um.y, 0), rd - bf16_hi_as_float(um.y, 1));
    return r;
}
__device__ __forceinline__ Split3x8 join(const Split3& a, const Split3& b) { return Split3x8{cat8(a.h, b.h), cat8(a.m, b.m), cat8(a.l, b.l)}; }
struct Frag3 {                                 // a weight fragment's three terms
    bf16x8 h, m, l;
};
// product term 0 .. 5 of mac6 on its own (callers interleave the terms of several accumulator chains)
__device__ __forceinline__ void mac_term(int term, f32x4b& lo, f32x4b& hi, const Frag3& w, const Split3x8& x) {
    switch (term) {
        case 0: lo = mfma_bf16k32(w.l, x.h, lo); break;
        case 1: lo = mfma_bf16k32(w.h, x.l, lo); break;
        case 2: lo = mfma_bf16k32(w.m, x.m, lo); break;
        case 3: lo = mfma_bf16k32(w.m, x.h, lo); break;
        case 4: lo = mfma_bf16k32(w.h, x.m, lo); break;
        default: hi = mfma_bf16k32(w.h, x.h, hi); break;
    }
}

}  // namespace msm
