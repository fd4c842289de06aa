// Label-image statistics for the two-stage harness (depth filter, ROI boxes, overlap rejection).
//
// The reference walks the label image once per label on the host side of torch (lib/fcn/test_dataset.py:62-112
// crop_rois / mask_to_tight_box, :116-131 overlap test, :183-198 filter_labels_depth): unique() + a masked
// reduction + .item() per label.  Here ONE pass over the image produces, for every label value v in [0, k):
//   area[v], sum of weight over the pixels of v, and the tight box (x_min, y_min, x_max, y_max)
// with per-workgroup tables in LDS (a label image has a dozen labels over 10^5 pixels: global atomics would all
// land on the same few addresses) and a wave-uniform fast path (a wave of 64 consecutive pixels almost always
// carries one label: one lane updates the table for the whole wave).
//
// Integer results are exact.  The weight sum is an fp32 sum in unspecified order: exact for the 0/1 weights the
// harness passes (valid-depth mask, first-stage mask), i.e. equal to the reference's counts.
#include "common.h"

namespace {

constexpr int LS_THREADS = 256;
constexpr int LS_COLS = 6;          // LDS row: area, xmin, ymin, xmax, ymax, wsum (float bits)

__global__ void label_stats_init_kernel(int* __restrict__ stats, float* __restrict__ wsum, int* __restrict__ overflow,
                                        int bins, int B, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bins) {
        int* s = stats + (size_t)i * 5;
        s[0] = 0; s[1] = W; s[2] = H; s[3] = -1; s[4] = -1;
        wsum[i] = 0.f;
    }
    if (i < B) overflow[i] = 0;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(LS_THREADS) void label_stats_kernel(const float* __restrict__ labels, const float* __restrict__ weight,
                                                                int* __restrict__ stats, float* __restrict__ wsum,
                                                                int* __restrict__ overflow, int H, int W, int k) {
    extern __shared__ int tab[];                       // k rows of LS_COLS
    __shared__ int over;
    const int b = blockIdx.y;
    const int n = H * W;
    for (int i = threadIdx.x; i < k; i += LS_THREADS) {
        int* r = tab + i * LS_COLS;
        r[0] = 0; r[1] = W; r[2] = H; r[3] = -1; r[4] = -1; r[5] = 0;   // 0 == 0.0f
    }
    if (threadIdx.x == 0) over = 0;
    __syncthreads();
    const float* lab = labels + (size_t)b * n;
    const float* wgt = weight ? weight + (size_t)b * n : nullptr;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(n, p0 + per);
    const int lane = threadIdx.x & 63;
    for (int base = p0 + (threadIdx.x & ~63); base < p1; base += LS_THREADS) {      // wave-uniform trip count
        const int p = base + lane;
        const bool valid = p < p1;
        int v = 0;
        float w = 0.f;
        if (valid) {
            const float f = lab[p];
            v = (int)f;
            if (wgt) w = wgt[p];
            if (!(f >= 0.f) || v >= k) { atomicAdd(&over, 1); v = min(max(v, 0), k - 1); }
        }
        const int y = p / W, x = p - y * W;
        const int v0 = __builtin_amdgcn_readfirstlane(v);
        const unsigned long long m = __ballot(valid), same = __ballot(valid && v == v0);
        if (same == m) {                                   // one label in the whole wave
            const int xmin = wave_min_i(valid ? x : W), ymin = wave_min_i(valid ? y : H);
            const int xmax = wave_max_i(valid ? x : -1), ymax = wave_max_i(valid ? y : -1);
            const float ws = msm::wave_sum(w);
            if (lane == 0) {
                int* r = tab + v0 * LS_COLS;
                atomicAdd(r, __popcll(m));
                atomicMin(r + 1, xmin); atomicMin(r + 2, ymin); atomicMax(r + 3, xmax); atomicMax(r + 4, ymax);
                if (wgt) atomicAdd(reinterpret_cast<float*>(r + 5), ws);
            }
        } else if (valid) {
            int* r = tab + v * LS_COLS;
            atomicAdd(r, 1);
            atomicMin(r + 1, x); atomicMin(r + 2, y); atomicMax(r + 3, x); atomicMax(r + 4, y);
            if (wgt) atomicAdd(reinterpret_cast<float*>(r + 5), w);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += LS_THREADS) {
        const int* r = tab + i * LS_COLS;
        if (r[0] > 0) {
            int* s = stats + ((size_t)b * k + i) * 5;
            atomicAdd(s, r[0]);
            atomicMin(s + 1, r[1]); atomicMin(s + 2, r[2]); atomicMax(s + 3, r[3]); atomicMax(s + 4, r[4]);
            if (wgt) atomicAdd(wsum + (size_t)b * k + i, __int_as_float(r[5]));
        }
    }
    if (threadIdx.x == 0 && over) atomicAdd(overflow + b, over);
}

}  // namespace

extern "C" int msm_label_stats(const float* labels, const float* weight, int32_t* stats, float* wsum, int32_t* overflow,
                               int B, int H, int W, int k, void* stream) {
    MSM_REQUIRE(labels && stats && wsum && overflow, "msm_label_stats: null pointer");
    MSM_REQUIRE(B >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1 << 30), "msm_label_stats: bad shape B=%d H=%d W=%d", B, H, W);
    MSM_REQUIRE(k >= 1 && k <= 2048, "msm_label_stats: k=%d outside [1, 2048] (LDS table)", k);
    if (B == 0) return MSM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int bins = B * k;
    hipLaunchKernelGGL(label_stats_init_kernel, dim3(msm::cdiv(max(bins, B), 256)), dim3(256), 0, s, stats, wsum, overflow, bins, B, H, W);
    const int n = H * W;
    int gx = msm::cdiv(n, LS_THREADS * 8);                    // >= 8 pixels per thread
    gx = max(1, min(gx, max(1, 512 / B)));
    hipLaunchKernelGGL(label_stats_kernel, dim3(gx, B), dim3(LS_THREADS), (size_t)k * LS_COLS * sizeof(int), s, labels, weight, stats,
                       wsum, overflow, H, W, k);
    MSM_CHECK_LAUNCH("msm_label_stats");
    return MSM_OK;
}
