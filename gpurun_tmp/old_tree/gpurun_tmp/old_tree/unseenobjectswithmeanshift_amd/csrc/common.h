// Shared helpers for the gfx950 kernels of libmsm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/msm_hip.h"

namespace msm {

void set_error(const char* fmt, ...);
// msm_set_option() value of MSM_OPT_* `key` (MSM_OPT_AUTO unless a tool or test set it)
int opt(int key);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, size): cheap on the hot path and
// keeps the call out of HIP-graph capture after warm-up.  Returns a hipError_t value.
int ensure_dynamic_lds(const void* kernel, size_t bytes);

#define MSM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::msm::set_error(__VA_ARGS__);     \
            return MSM_E_INVALID;              \
        }                                      \
    } while (0)

#define MSM_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ::msm::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return MSM_E_LAUNCH;                                                      \
        }                                                                             \
    } while (0)

#define MSM_CHECK_HIP(expr)                                                           \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            ::msm::set_error("%s failed: %s", #expr, hipGetErrorString(e__));         \
            return MSM_E_LAUNCH;                                                      \
        }                                                                             \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D = A(16x4) * B(4x16) + C, exact f32 (an fmaf chain over k).
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D:       lane l, reg r holds D[row = (l >> 4) * 4 + r][col = l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Sum over the 64 lanes, every lane gets the total: the butterfly v += v[lane ^ o], o = 32, 16, 8, 4, 2, 1 -- in that order, so the
// association (and the rounding) is that of the __shfl_xor loop it replaces -- on vector instructions: v_permlane32/16_swap (gfx950)
// for the two cross-row steps, DPP for the four steps inside a 16-lane row.  The __shfl_xor form compiles to six DEPENDENT
// ds_bpermute_b32 round trips through the LDS crossbar (~100 cycles each): a LayerNorm's mean -> variance chain was ~1200 cycles of
// latency on the critical path of every row-local decoder kernel.
__device__ __forceinline__ float wave_xor_dpp8(float v) {        // v[lane ^ 8]: rotate the 16-lane row by 8
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_xor_dpp4(float v) {        // v[lane ^ 4]: banks (quads) 0, 2 take lane + 4, banks 1, 3 lane - 4
    int t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, (int)__float_as_uint(v), 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    return __uint_as_float((unsigned)t);
}
__device__ __forceinline__ float wave_xor_dpp2(float v) {        // quad_perm [2, 3, 0, 1]
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4e, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_xor_dpp1(float v) {        // quad_perm [1, 0, 3, 2]
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xb1, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
    const u32x2w a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a.x) + __uint_as_float(a.y);
    const u32x2w b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(b.x) + __uint_as_float(b.y);
    v += wave_xor_dpp8(v);
    v += wave_xor_dpp4(v);
    v += wave_xor_dpp2(v);
    v += wave_xor_dpp1(v);
    return v;
}

// v summed (OR-ed) over the four lane rows (lanes l, l ^ 16, l ^ 32, l ^ 48) in the association of `v += shfl_xor(v, 16); v += shfl_xor(v, 32)`,
// on v_permlane16_swap / v_permlane32_swap instead of two ds_bpermute round trips (the key norm of the attention kernels sits on the
// dependent chain of every 16-key block).
__device__ __forceinline__ float sum_lane_rows(float v) {
    typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
    const u32x2r a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a.x) + __uint_as_float(a.y);
    const u32x2r b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b.x) + __uint_as_float(b.y);
}
__device__ __forceinline__ unsigned or_lane_rows(unsigned v) {
    typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
    const u32x2r a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = a.x | a.y;
    const u32x2r b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return b.x | b.y;
}

// max over the 64 lanes of a 64-bit key, every lane gets it (the seeding kernels' candidate reduction): the six butterfly stages on
// v_permlane*_swap / DPP word pairs instead of twelve dependent ds_bpermute round trips; a max is the same whatever the order.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    typedef unsigned u32x2m __attribute__((ext_vector_type(2)));
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    {
        const u32x2m a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const unsigned long long x = ((unsigned long long)b.x << 32) | a.x, y = ((unsigned long long)b.y << 32) | a.y;
        v = x > y ? x : y;
    }
    lo = (unsigned)v, hi = (unsigned)(v >> 32);
    {
        const u32x2m a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const unsigned long long x = ((unsigned long long)b.x << 32) | a.x, y = ((unsigned long long)b.y << 32) | a.y;
        v = x > y ? x : y;
    }
#define MSM_U64_STAGE(FN)                                                                                                     \
    {                                                                                                                         \
        const unsigned ol = __float_as_uint(FN(__uint_as_float((unsigned)v))), oh = __float_as_uint(FN(__uint_as_float((unsigned)(v >> 32)))); \
        const unsigned long long o = ((unsigned long long)oh << 32) | ol;                                                     \
        v = o > v ? o : v;                                                                                                    \
    }
    MSM_U64_STAGE(wave_xor_dpp8)
    MSM_U64_STAGE(wave_xor_dpp4)
    MSM_U64_STAGE(wave_xor_dpp2)
    MSM_U64_STAGE(wave_xor_dpp1)
#undef MSM_U64_STAGE
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace msm
