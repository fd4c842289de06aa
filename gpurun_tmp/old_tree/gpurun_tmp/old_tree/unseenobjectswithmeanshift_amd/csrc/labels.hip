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

// ---- batched two-stage harness (round 3): label images, ROI crops and paste-back for a whole batch of frames ------------
// lib/fcn/test_utils.py:375-406 walks frames and crops one at a time; the three kernels below let the harness run the first
// stage on all frames at once, cut every frame's ROIs in one launch and paste every frame's refined labels in one launch.

// label image of test_utils.py:93-112 (combine_masks after get_confident_instances :35-52): instance i, if kept, carries
// label lab[b][i] = 2 + (kept instances before it), 0 when dropped; "later instances overwrite earlier ones" with labels
// growing in instance order = the per-pixel maximum.
__global__ __launch_bounds__(256) void label_image_kernel(const float* __restrict__ masks, const float* __restrict__ lab,
                                                          float* __restrict__ out, int K, int64_t hw4) {
    const int b = blockIdx.y;
    const float4* m = reinterpret_cast<const float4*>(masks) + (int64_t)b * K * hw4;
    const float* lb = lab + (int64_t)b * K;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < hw4; p += (int64_t)gridDim.x * 256) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
            const float l = lb[k];
            if (l == 0.f) continue;                              // uniform
            const float4 v = m[(int64_t)k * hw4 + p];
            r.x = fmaxf(r.x, v.x != 0.f ? l : 0.f);
            r.y = fmaxf(r.y, v.y != 0.f ? l : 0.f);
            r.z = fmaxf(r.z, v.z != 0.f ? l : 0.f);
            r.w = fmaxf(r.w, v.w != 0.f ? l : 0.f);
        }
        reinterpret_cast<float4*>(out)[(int64_t)b * hw4 + p] = r;
    }
}

// ROI table row (8 x int32): frame, label, x0, y0, x1, y1 (inclusive), 2 unused
// crop_rois (lib/fcn/test_dataset.py:62-112): rgb / xyz resized with F.upsample_bilinear (align_corners=True, :104,109), the
// mask (label == mask_id) with nearest (:106).  Index arithmetic as ATen's upsample kernels: scale = (in - 1) / (out - 1) in
// fp32, src = scale * dst, i0 = (int)src, lambda = src - i0, second tap i0 + (i0 < in - 1); nearest: min((int)floorf(dst *
// (in / out)), in - 1).
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ rgb, const float* __restrict__ depth,
                                                          const float* __restrict__ labels, const int32_t* __restrict__ table,
                                                          float* __restrict__ rgb_out, float* __restrict__ depth_out,
                                                          float* __restrict__ mask_out, int H, int W, int S) {
    const int n = blockIdx.y;
    const int32_t* t = table + (int64_t)n * 8;
    const int f = t[0], label = t[1], x0 = t[2], y0 = t[3], x1 = t[4], y1 = t[5];
    const int ih = y1 - y0 + 1, iw = x1 - x0 + 1;
    const float rh = S > 1 ? (float)(ih - 1) / (float)(S - 1) : 0.f, rw = S > 1 ? (float)(iw - 1) / (float)(S - 1) : 0.f;
    const float nh = (float)ih / (float)S, nw = (float)iw / (float)S;
    const int64_t plane = (int64_t)H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < S * S; p += gridDim.x * 256) {
        const int oy = p / S, ox = p - oy * S;
        const float h1r = rh * (float)oy, w1r = rw * (float)ox;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < ih - 1 ? 1 : 0, w1p = w1 < iw - 1 ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
        const int64_t o00 = (int64_t)(y0 + h1) * W + (x0 + w1), o01 = o00 + w1p, o10 = o00 + (int64_t)h1p * W, o11 = o10 + w1p;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* src = rgb + ((int64_t)f * 3 + c) * plane;
            rgb_out[((int64_t)n * 3 + c) * S * S + p] = h0l * (w0l * src[o00] + w1l * src[o01]) + h1l * (w0l * src[o10] + w1l * src[o11]);
        }
        if (depth) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* src = depth + ((int64_t)f * 3 + c) * plane;
                depth_out[((int64_t)n * 3 + c) * S * S + p] = h0l * (w0l * src[o00] + w1l * src[o01]) + h1l * (w0l * src[o10] + w1l * src[o11]);
            }
        }
        const int sy = min((int)floorf((float)oy * nh), ih - 1), sx = min((int)floorf((float)ox * nw), iw - 1);
        mask_out[(int64_t)n * S * S + p] = labels[(int64_t)f * plane + (int64_t)(y0 + sy) * W + (x0 + sx)] == (float)label ? 1.f : 0.f;
    }
}

// paste-back of match_label_crop (lib/fcn/test_dataset.py:160-177): per frame the crops are pasted in `order`, each resized
// (nearest) to its ROI, non-zero pixels overwriting what is there -- i.e. a pixel takes the LAST crop in order that covers
// it with a non-zero value.  order[frame_start[f] .. frame_start[f+1]) lists frame f's crops in paste order.
__global__ __launch_bounds__(256) void paste_labels_kernel(const float* __restrict__ renum, const int32_t* __restrict__ table,
                                                           const int32_t* __restrict__ order, const int32_t* __restrict__ frame_start,
                                                           float* __restrict__ refined, int H, int W, int S) {
    const int f = blockIdx.y;
    const int i0 = frame_start[f], i1 = frame_start[f + 1];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int y = p / W, x = p - y * W;
        float r = 0.f;
        for (int i = i1 - 1; i >= i0; --i) {
            const int n = order[i];
            const int32_t* t = table + (int64_t)n * 8;
            const int x0 = t[2], y0 = t[3], x1 = t[4], y1 = t[5];
            if (x < x0 || x > x1 || y < y0 || y > y1) continue;
            const int oh = y1 - y0 + 1, ow = x1 - x0 + 1;
            const int sy = min((int)floorf((float)(y - y0) * ((float)S / (float)oh)), S - 1);
            const int sx = min((int)floorf((float)(x - x0) * ((float)S / (float)ow)), S - 1);
            const float v = renum[((int64_t)n * S + sy) * S + sx];
            if (v != 0.f) { r = v; break; }
        }
        refined[(int64_t)f * H * W + p] = r;
    }
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

extern "C" int msm_label_image(const float* masks, const float* inst_labels, float* out, int B, int K, int H, int W, void* stream) {
    MSM_REQUIRE(masks && inst_labels && out, "msm_label_image: null pointer");
    MSM_REQUIRE(B >= 0 && K >= 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0, "msm_label_image: bad shape (H*W must be a multiple of 4)");
    MSM_REQUIRE(((((uintptr_t)masks) | ((uintptr_t)out)) & 15) == 0, "msm_label_image: pointers must be 16-byte aligned");
    if (B == 0) return MSM_OK;
    const int64_t hw4 = (int64_t)H * W / 4;
    const int gx = (int)max((int64_t)1, min((hw4 + 255) / 256, (int64_t)max(1, 2048 / B)));
    hipLaunchKernelGGL(label_image_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, masks, inst_labels, out, K, hw4);
    MSM_CHECK_LAUNCH("msm_label_image");
    return MSM_OK;
}

extern "C" int msm_crop_resize(const float* rgb, const float* depth, const float* labels, const int32_t* table, float* rgb_out,
                               float* depth_out, float* mask_out, int N, int H, int W, int S, void* stream) {
    MSM_REQUIRE(rgb && labels && table && rgb_out && mask_out, "msm_crop_resize: null pointer");
    MSM_REQUIRE((depth == nullptr) == (depth_out == nullptr), "msm_crop_resize: depth and depth_out go together");
    MSM_REQUIRE(N >= 0 && H > 0 && W > 0 && S > 0, "msm_crop_resize: bad sizes");
    if (N == 0) return MSM_OK;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(min(msm::cdiv(S * S, 256), 64), N), dim3(256), 0, (hipStream_t)stream, rgb, depth, labels, table,
                       rgb_out, depth_out, mask_out, H, W, S);
    MSM_CHECK_LAUNCH("msm_crop_resize");
    return MSM_OK;
}

extern "C" int msm_paste_labels(const float* renum, const int32_t* table, const int32_t* order, const int32_t* frame_start,
                                float* refined, int F, int H, int W, int S, void* stream) {
    MSM_REQUIRE(renum && table && order && frame_start && refined, "msm_paste_labels: null pointer");
    MSM_REQUIRE(F >= 0 && H > 0 && W > 0 && S > 0, "msm_paste_labels: bad sizes");
    if (F == 0) return MSM_OK;
    hipLaunchKernelGGL(paste_labels_kernel, dim3(min(msm::cdiv(H * W, 256), 256), F), dim3(256), 0, (hipStream_t)stream, renum, table, order,
                       frame_start, refined, H, W, S);
    MSM_CHECK_LAUNCH("msm_paste_labels");
    return MSM_OK;
}
