// Elementwise glue of the backbones (the convolutions themselves stay MIOpen / hipBLASLt library calls) in one pass each.
//
// The ResNet-50 forward (detectron2 BottleneckBlock with frozen BatchNorm folded into the convolutions,
// Base-COCO-InstanceSegmentation.yaml:2-15) through stock torch ops spends a third of its bf16 time in elementwise launches around the
// convolutions: MIOpen adds a convolution's bias in a kernel of its own (SubTensorOpWithCastTensor1d, 15 us per 3x3 convolution at
// batch 8), F.relu is another (10 us), the residual add and its ReLU two more (16 + 10 us) -- 33 + 16 + 19 launches per pass.  Here:
//   msm_bias_act_nhwc    x = act(x + bias[c] (+ residual)) in place on a channels_last map (bf16 or fp32; fp32 arithmetic, one rounding)
//   msm_nhwc_to_nchw_f32 a channels_last bf16 / fp32 map -> NCHW fp32 planes (what the pixel decoder's input projections read), one pass
#include "bf16.h"
#include "common.h"

namespace msm {

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// sixteen bytes per thread: 8 bf16 or 4 fp32 values of one pixel's channel run (C % 8 == 0 / C % 4 == 0)
template <bool BF>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(void* __restrict__ xv, const void* __restrict__ biasv, const void* __restrict__ resv,
                                                            int relu, int64_t nvec, int C) {
    constexpr int V = BF ? 8 : 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)((i * V) % C);
        float v[V], r[V];
        if constexpr (BF) {
            const u32x4b xw = reinterpret_cast<const u32x4b*>(xv)[i];
            const u32x4b bw = *reinterpret_cast<const u32x4b*>(reinterpret_cast<const unsigned short*>(biasv) + c0);
            u32x4b rw = {0u, 0u, 0u, 0u};
            if (resv) rw = reinterpret_cast<const u32x4b*>(resv)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __uint_as_float(xw[j] << 16) + __uint_as_float(bw[j] << 16);
                v[2 * j + 1] = __uint_as_float(xw[j] & 0xffff0000u) + __uint_as_float(bw[j] & 0xffff0000u);
                r[2 * j] = __uint_as_float(rw[j] << 16);
                r[2 * j + 1] = __uint_as_float(rw[j] & 0xffff0000u);
            }
        } else {
            const float4 xw = reinterpret_cast<const float4*>(xv)[i];
            const float4 bw = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(biasv) + c0);
            float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (resv) rw = reinterpret_cast<const float4*>(resv)[i];
            v[0] = xw.x + bw.x, v[1] = xw.y + bw.y, v[2] = xw.z + bw.z, v[3] = xw.w + bw.w;
            r[0] = rw.x, r[1] = rw.y, r[2] = rw.z, r[3] = rw.w;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            v[j] += r[j];
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        if constexpr (BF) {
            const bf16x4 lo = pack4(v[0], v[1], v[2], v[3]), hi = pack4(v[4], v[5], v[6], v[7]);
            const u32x2b a = __builtin_bit_cast(u32x2b, lo), b = __builtin_bit_cast(u32x2b, hi);
            reinterpret_cast<u32x4b*>(xv)[i] = u32x4b{a.x, a.y, b.x, b.y};
        } else {
            reinterpret_cast<float4*>(xv)[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// in [B][HW][C] (bf16 or fp32) -> out [B][C][HW] fp32: 32 x 32 tiles through LDS
template <bool BF>
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const void* __restrict__ inv, float* __restrict__ out, int HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) {
            const int64_t idx = ((int64_t)b * HW + p) * C + c;
            v = BF ? bf2f(reinterpret_cast<const unsigned short*>(inv)[idx]) : reinterpret_cast<const float*>(inv)[idx];
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (p < HW && c < C) out[((int64_t)b * C + c) * HW + p] = tile[tx][k];
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_bias_act_nhwc(void* x, const void* bias, const void* residual, int relu, int64_t pixels, int C, int dtype, void* stream) {
    const char* who = "msm_bias_act_nhwc";
    MSM_REQUIRE(x && bias && pixels > 0 && C > 0, "%s: bad arguments", who);
    MSM_REQUIRE(dtype == 0 || dtype == 1, "%s: dtype=%d (0 = fp32, 1 = bf16)", who, dtype);
    const int V = dtype ? 8 : 4;
    MSM_REQUIRE(C % V == 0, "%s: C=%d must be a multiple of %d", who, C, V);
    MSM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)bias) | ((uintptr_t)residual)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    const int64_t nvec = pixels * C / V;
    const int grid = (int)(nvec / 256 + 1 > 8192 ? 8192 : nvec / 256 + 1);
    if (dtype) hipLaunchKernelGGL(bias_act_nhwc_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, residual, relu, nvec, C);
    else hipLaunchKernelGGL(bias_act_nhwc_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, residual, relu, nvec, C);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_nhwc_to_nchw_f32(const void* in, float* out, int B, int C, int HW, int dtype, void* stream) {
    const char* who = "msm_nhwc_to_nchw_f32";
    MSM_REQUIRE(in && out && B > 0 && B <= 65535 && C > 0 && HW > 0, "%s: bad arguments", who);
    MSM_REQUIRE(dtype == 0 || dtype == 1, "%s: dtype=%d (0 = fp32, 1 = bf16)", who, dtype);
    MSM_REQUIRE(cdiv(HW, 32) <= 65535, "%s: H*W=%d too large", who, HW);
    dim3 grid(cdiv(C, 32), cdiv(HW, 32), B), block(256);
    if (dtype) hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<true>, grid, block, 0, (hipStream_t)stream, in, out, HW, C);
    else hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<false>, grid, block, 0, (hipStream_t)stream, in, out, HW, C);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
