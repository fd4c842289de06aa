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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace msm
