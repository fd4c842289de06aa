// Multi-scale deformable attention, reference ABI, for every scalar type and channel count the reference's op takes.
//
// The reference dispatches its kernels over float AND double (AT_DISPATCH_FLOATING_TYPES, ops/src/cuda/ms_deform_attn_cuda.cu:69
// forward, :139 backward) and its own test drives the double instantiation: the exact forward check (ops/test.py:33-43) and
// gradcheck over D in {30, 32, 64, 71, 1025, 2048, 3096} (ops/test.py:66-89).  The tuned fp32 kernels of msda.hip cover the
// shapes the pixel decoder runs (D <= 64); this file is the shape- and type-generic form behind the same entry points:
//
//   forward   one lane per output scalar (b, q, m, d) like ms_deformable_im2col_gpu_kernel (cuh:242-304), but with d fastest
//             across lanes so that a wave reads 64 consecutive channels of each bilinear tap (one or two cache lines per tap
//             instead of 64 scattered ones), level geometry read once per lane;
//   backward  one WAVE per (b, q, m): lanes stride over the D channels, grad_value is accumulated with hardware atomics as
//             the reference does (cuh:128-160: its order is not fixed there either), and the channel sums that form
//             grad_sampling_loc / grad_attn_weight (the reference's shared-memory reductions, cuh:368-386) are a fixed
//             shuffle butterfly over the wave, so those two outputs are bit-reproducible from run to run -- gradcheck's
//             re-entrancy comparison (nondet_tol = 0) relies on that.
//
// Arithmetic follows cuh:38-89 (forward bilinear), cuh:92-239 (backward) in the tensor's own type.
#include "common.h"

namespace msm {

template <typename T>
__device__ __forceinline__ T wave_sum_t(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void msda_any_fwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                                                           const T* __restrict__ wgt, T* __restrict__ out, int64_t total,
                                                           int S, int M, int D, int L, int Lq, int P) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int d = (int)(idx % D);
        int64_t t = idx / D;
        const int m = (int)(t % M);
        t /= M;
        const int q = (int)(t % Lq);
        const int64_t b = t / Lq;
        const int64_t pix = (int64_t)M * D;
        const T* vb = value + b * S * pix + (int64_t)m * D + d;
        const int64_t base = ((b * Lq + q) * M + m) * L * P;
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const T* vl = vb + lstart[l] * pix;
            for (int p = 0; p < P; ++p) {
                const int64_t i = base + (int64_t)l * P + p;
                const T lx = loc[2 * i], ly = loc[2 * i + 1], aw = wgt[i];
                const T h_im = ly * (T)H - (T)0.5, w_im = lx * (T)W - (T)0.5;            // cuh:290-291
                if (!(h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W)) continue;   // cuh:293
                const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const T lh = h_im - (T)h_low, lw = w_im - (T)w_low, hh = (T)1 - lh, hw = (T)1 - lw;
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h_low >= 0 && w_low >= 0) v1 = vl[((int64_t)h_low * W + w_low) * pix];
                if (h_low >= 0 && w_high <= W - 1) v2 = vl[((int64_t)h_low * W + w_high) * pix];
                if (h_high <= H - 1 && w_low >= 0) v3 = vl[((int64_t)h_high * W + w_low) * pix];
                if (h_high <= H - 1 && w_high <= W - 1) v4 = vl[((int64_t)h_high * W + w_high) * pix];
                acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * aw;  // cuh:86-88, 295
            }
        }
        out[idx] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void msda_any_bwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                                                           const T* __restrict__ wgt, const T* __restrict__ gout,
                                                           T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gwgt,
                                                           int64_t units, int S, int M, int D, int L, int Lq, int P) {
    const int lane = threadIdx.x & 63;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += (int64_t)gridDim.x * 4) {
        const int m = (int)(u % M);
        const int64_t bq = u / M;                 // b * Lq + q
        const int64_t b = bq / Lq;
        const int64_t pix = (int64_t)M * D;
        const int64_t voff = b * S * pix + (int64_t)m * D;
        const T* go = gout + (bq * M + m) * D;
        const int64_t base = (bq * M + m) * L * P;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t lvl = voff + lstart[l] * pix;
            for (int p = 0; p < P; ++p) {
                const int64_t i = base + (int64_t)l * P + p;
                const T lx = loc[2 * i], ly = loc[2 * i + 1], aw = wgt[i];
                const T h_im = ly * (T)H - (T)0.5, w_im = lx * (T)W - (T)0.5;
                T g_w = 0, g_x = 0, g_y = 0;
                if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {              // cuh:352 (wave-uniform)
                    const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const T lh = h_im - (T)h_low, lw = w_im - (T)w_low, hh = (T)1 - lh, hw = (T)1 - lw;
                    const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
                    const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
                    const int64_t o1 = lvl + ((int64_t)h_low * W + w_low) * pix, o2 = o1 + pix;
                    const int64_t o3 = o1 + (int64_t)W * pix, o4 = o3 + pix;
                    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                    for (int d = lane; d < D; d += 64) {
                        const T g = go[d];
                        const T tg = g * aw;                                                   // top_grad_value, cuh:117
                        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                        if (ok1) { v1 = value[o1 + d]; unsafeAtomicAdd(gvalue + o1 + d, w1 * tg); }   // cuh:128-160
                        if (ok2) { v2 = value[o2 + d]; unsafeAtomicAdd(gvalue + o2 + d, w2 * tg); }
                        if (ok3) { v3 = value[o3 + d]; unsafeAtomicAdd(gvalue + o3 + d, w3 * tg); }
                        if (ok4) { v4 = value[o4 + d]; unsafeAtomicAdd(gvalue + o4 + d, w4 * tg); }
                        g_w += g * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                    // cuh:164
                        g_x += tg * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);                  // grad_w_weight
                        g_y += tg * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);                  // grad_h_weight
                    }
                }
                g_w = wave_sum_t(g_w);
                g_x = wave_sum_t(g_x) * (T)W;                                                  // cuh:165-166
                g_y = wave_sum_t(g_y) * (T)H;
                if (lane == 0) {
                    gwgt[i] = g_w;
                    gloc[2 * i] = g_x;
                    gloc[2 * i + 1] = g_y;
                }
            }
        }
    }
}

template <typename T>
static int launch_any_fwd(const char* name, const T* value, const int64_t* shapes, const int64_t* lstart, const T* loc,
                          const T* wgt, T* out, int B, int S, int M, int D, int L, int Lq, int P, void* stream) {
    MSM_REQUIRE(value && shapes && lstart && loc && wgt && out, "%s: null pointer", name);
    MSM_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, "%s: bad sizes", name);
    const int64_t total = (int64_t)B * Lq * M * D;
    const unsigned grid = (unsigned)min((int64_t)65536, (total + 255) / 256);
    hipLaunchKernelGGL((msda_any_fwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, value, shapes, lstart, loc,
                       wgt, out, total, S, M, D, L, Lq, P);
    MSM_CHECK_LAUNCH(name);
    return MSM_OK;
}

template <typename T>
static int launch_any_bwd(const char* name, const T* value, const int64_t* shapes, const int64_t* lstart, const T* loc,
                          const T* wgt, const T* gout, T* gvalue, T* gloc, T* gwgt, int B, int S, int M, int D, int L, int Lq,
                          int P, void* stream) {
    MSM_REQUIRE(value && shapes && lstart && loc && wgt && gout && gvalue && gloc && gwgt, "%s: null pointer", name);
    MSM_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, "%s: bad sizes", name);
    hipStream_t st = (hipStream_t)stream;
    MSM_CHECK_HIP(hipMemsetAsync(gvalue, 0, sizeof(T) * (size_t)B * S * M * D, st));
    const int64_t units = (int64_t)B * Lq * M;
    const unsigned grid = (unsigned)min((int64_t)65536, (units + 3) / 4);
    hipLaunchKernelGGL((msda_any_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, value, shapes, lstart, loc, wgt, gout, gvalue,
                       gloc, gwgt, units, S, M, D, L, Lq, P);
    MSM_CHECK_LAUNCH(name);
    return MSM_OK;
}

// fp32 shapes outside the tuned kernels' range (msda.hip routes here)
int msda_any_fwd_f32(const float* value, const int64_t* shapes, const int64_t* lstart, const float* loc, const float* wgt,
                     float* out, int B, int S, int M, int D, int L, int Lq, int P, void* stream) {
    return launch_any_fwd<float>("msm_msdeform_attn_fwd", value, shapes, lstart, loc, wgt, out, B, S, M, D, L, Lq, P, stream);
}
int msda_any_bwd_f32(const float* value, const int64_t* shapes, const int64_t* lstart, const float* loc, const float* wgt,
                     const float* gout, float* gvalue, float* gloc, float* gwgt, int B, int S, int M, int D, int L, int Lq,
                     int P, void* stream) {
    return launch_any_bwd<float>("msm_msdeform_attn_bwd", value, shapes, lstart, loc, wgt, gout, gvalue, gloc, gwgt, B, S, M,
                                 D, L, Lq, P, stream);
}

}  // namespace msm

using namespace msm;

extern "C" int msm_msdeform_attn_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                         const double* sampling_loc, const double* attn_weight, double* out, int B, int S,
                                         int M, int D, int L, int Lq, int P, void* stream) {
    return launch_any_fwd<double>("msm_msdeform_attn_fwd_f64", value, spatial_shapes, level_start_index, sampling_loc,
                                  attn_weight, out, B, S, M, D, L, Lq, P, stream);
}

extern "C" int msm_msdeform_attn_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                         const double* sampling_loc, const double* attn_weight, const double* grad_output,
                                         double* grad_value, double* grad_sampling_loc, double* grad_attn_weight, int B, int S,
                                         int M, int D, int L, int Lq, int P, void* stream) {
    return launch_any_bwd<double>("msm_msdeform_attn_bwd_f64", value, spatial_shapes, level_start_index, sampling_loc,
                                  attn_weight, grad_output, grad_value, grad_sampling_loc, grad_attn_weight, B, S, M, D, L, Lq,
                                  P, stream);
}
