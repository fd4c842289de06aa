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
// T: 0 fp32, 1 bf16, 2 IEEE half
template <int T>
__device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    if constexpr (T == 1) {
        lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xffff0000u);
    } else {
        lo = half_lo(w), hi = half_hi(w);
    }
}
template <int T>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(void* __restrict__ xv, const void* __restrict__ biasv, const void* __restrict__ resv,
                                                            int relu, int64_t nvec, int C) {
    constexpr bool BF = T != 0;
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
                float x0, x1, b0, b1;
                unpack2<T>(xw[j], x0, x1), unpack2<T>(bw[j], b0, b1), unpack2<T>(rw[j], r[2 * j], r[2 * j + 1]);
                v[2 * j] = x0 + b0, v[2 * j + 1] = x1 + b1;
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
        if constexpr (T == 1) {
            const bf16x4 lo = pack4(v[0], v[1], v[2], v[3]), hi = pack4(v[4], v[5], v[6], v[7]);
            const u32x2b a = __builtin_bit_cast(u32x2b, lo), b = __builtin_bit_cast(u32x2b, hi);
            reinterpret_cast<u32x4b*>(xv)[i] = u32x4b{a.x, a.y, b.x, b.y};
        } else if constexpr (T == 2) {
            const u32x2b a = pack4h(v[0], v[1], v[2], v[3]), b = pack4h(v[4], v[5], v[6], v[7]);       // (clamped to the half range)
            reinterpret_cast<u32x4b*>(xv)[i] = u32x4b{a.x, a.y, b.x, b.y};
        } else {
            reinterpret_cast<float4*>(xv)[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// in [B][HW][C] (bf16 or fp32) -> out [B][C][HW] fp32: 32 x 32 tiles through LDS
template <int T>
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
            if constexpr (T == 0) v = reinterpret_cast<const float*>(inv)[idx];
            else if constexpr (T == 1) v = bf2f(reinterpret_cast<const unsigned short*>(inv)[idx]);
            else v = half_lo((unsigned)reinterpret_cast<const unsigned short*>(inv)[idx]);
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (p < HW && c < C) out[((int64_t)b * C + c) * HW + p] = tile[tx][k];
    }
}

// The tail of the UCN RGB-D backbone in one pass (SEG.py:97-117: upsample_bilinear of each tower's 1/8-resolution embedding, add fusion,
// F.normalize over the channels; pretrained_meanshiftformer_model.py:298-300 normalises once more):
//   out[b][c][y][x] = N(...N(up(a)[c] + up(b2)[c])),  up = bilinear, align_corners=True (nn.functional.upsample_bilinear), N(v) = v / max(|v|_2, eps)
// a, b2: [B][h][w][64] fp32 (channels_last maps), b2 nullable; out NCHW fp32.  Through torch ops this is two upsamples, an add, a norm
// reduction, a division, the second normalisation and their copies -- eight passes over 157 MB at batch 2 of 480x640 (1.2 ms); here the
// 2.4 MB of low-resolution maps are read from cache and the output is written once.
// Block: 64 consecutive x of one row; thread (px = tid & 63, cq = tid >> 6) owns channels 16 cq .. + 15 of its pixel.
__global__ __launch_bounds__(256) void ucn_tail_kernel(const float* __restrict__ a, const float* __restrict__ b2, float* __restrict__ out, int h, int w,
                                                       int H, int W, float ry, float rx, int norms, float eps) {
    __shared__ float red[2][4][64];
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + px, y = blockIdx.y, b = blockIdx.z;
    const bool live = x < W;
    // at::native upsample_bilinear2d, align_corners=True: source = scale * dst, scale = (in - 1) / (out - 1)
    const float sy = ry * (float)y, sx = rx * (float)min(x, W - 1);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* src = t == 0 ? a : b2;
        if (src == nullptr) continue;
        const float* base = src + (int64_t)b * h * w * 64 + cq * 16;
        const float4* p00 = reinterpret_cast<const float4*>(base + ((int64_t)y0 * w + x0) * 64);
        const float4* p01 = reinterpret_cast<const float4*>(base + ((int64_t)y0 * w + x1) * 64);
        const float4* p10 = reinterpret_cast<const float4*>(base + ((int64_t)y1 * w + x0) * 64);
        const float4* p11 = reinterpret_cast<const float4*>(base + ((int64_t)y1 * w + x1) * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 q00 = p00[j], q01 = p01[j], q10 = p10[j], q11 = p11[j];
            v[4 * j + 0] += hy * (hx * q00.x + lx * q01.x) + ly * (hx * q10.x + lx * q11.x);
            v[4 * j + 1] += hy * (hx * q00.y + lx * q01.y) + ly * (hx * q10.y + lx * q11.y);
            v[4 * j + 2] += hy * (hx * q00.z + lx * q01.z) + ly * (hx * q10.z + lx * q11.z);
            v[4 * j + 3] += hy * (hx * q00.w + lx * q01.w) + ly * (hx * q10.w + lx * q11.w);
        }
    }
    for (int n = 0; n < norms; ++n) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) ss += v[i] * v[i];
        red[n & 1][cq][px] = ss;
        __syncthreads();
        const float tot = (red[n & 1][0][px] + red[n & 1][1][px]) + (red[n & 1][2][px] + red[n & 1][3][px]);
        const float inv = 1.f / fmaxf(sqrtf(tot), eps);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= inv;
    }
    if (live) {
        float* o = out + (((int64_t)b * 64 + cq * 16) * H + y) * W + x;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[(int64_t)i * H * W] = v[i];
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_bias_act_nhwc(void* x, const void* bias, const void* residual, int relu, int64_t pixels, int C, int dtype, void* stream) {
    const char* who = "msm_bias_act_nhwc";
    MSM_REQUIRE(x && bias && pixels > 0 && C > 0, "%s: bad arguments", who);
    MSM_REQUIRE(dtype >= 0 && dtype <= 2, "%s: dtype=%d (0 = fp32, 1 = bf16, 2 = fp16)", who, dtype);
    const int V = dtype ? 8 : 4;
    MSM_REQUIRE(C % V == 0, "%s: C=%d must be a multiple of %d", who, C, V);
    MSM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)bias) | ((uintptr_t)residual)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    const int64_t nvec = pixels * C / V;
    const int grid = (int)(nvec / 256 + 1 > 8192 ? 8192 : nvec / 256 + 1);
    if (dtype == 1) hipLaunchKernelGGL(bias_act_nhwc_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, residual, relu, nvec, C);
    else if (dtype == 2) hipLaunchKernelGGL(bias_act_nhwc_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, residual, relu, nvec, C);
    else hipLaunchKernelGGL(bias_act_nhwc_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, residual, relu, nvec, C);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_nhwc_to_nchw_f32(const void* in, float* out, int B, int C, int HW, int dtype, void* stream) {
    const char* who = "msm_nhwc_to_nchw_f32";
    MSM_REQUIRE(in && out && B > 0 && B <= 65535 && C > 0 && HW > 0, "%s: bad arguments", who);
    MSM_REQUIRE(dtype >= 0 && dtype <= 2, "%s: dtype=%d (0 = fp32, 1 = bf16, 2 = fp16)", who, dtype);
    MSM_REQUIRE(cdiv(HW, 32) <= 65535, "%s: H*W=%d too large", who, HW);
    dim3 grid(cdiv(C, 32), cdiv(HW, 32), B), block(256);
    if (dtype == 1) hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<1>, grid, block, 0, (hipStream_t)stream, in, out, HW, C);
    else if (dtype == 2) hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<2>, grid, block, 0, (hipStream_t)stream, in, out, HW, C);
    else hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<0>, grid, block, 0, (hipStream_t)stream, in, out, HW, C);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_ucn_embedding_tail(const float* a, const float* b2, float* out, int B, int h, int w, int H, int W, int norms, float eps, void* stream) {
    const char* who = "msm_ucn_embedding_tail";
    MSM_REQUIRE(a && out && B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && H <= 65535 && W > 0, "%s: bad arguments", who);
    MSM_REQUIRE(norms >= 0 && norms <= 2, "%s: norms=%d (0, 1 or 2 normalisations)", who, norms);
    MSM_REQUIRE(((((uintptr_t)a) | ((uintptr_t)b2)) & 15) == 0, "%s: the low-resolution maps must be 16-byte aligned", who);
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(ucn_tail_kernel, dim3(cdiv(W, 64), H, B), dim3(256), 0, (hipStream_t)stream, a, b2, out, h, w, H, W, ry, rx, norms, eps);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
