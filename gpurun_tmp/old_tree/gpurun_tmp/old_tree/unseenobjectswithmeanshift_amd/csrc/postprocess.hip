// Instance post-processing (see include/msm_hip.h: msm_topk_class_scores, msm_instance_postprocess).
//
// Reference: MSMFormer/meanshiftformer/pretrained_meanshiftformer_model.py:337-343 upsamples ALL Q
// low-resolution masks to image size (123 MB per 640x480 image), :355-376 post-processes them per
// image and :461-497 (instance_inference) keeps top-k of them, thresholds at 0, scores each mask by
// its mean sigmoid and takes boxes from the binary masks.  Here the top-k runs first on the tiny
// class-score matrix and only the T selected masks are upsampled: one pass reads T low-res maps and
// writes T binary maps (HBM-bound: 32 MB instead of >= 250 MB per image), with the score and box
// reductions fused into it.
#include <stdlib.h>

#include "common.h"

namespace msm {

// gridDim.y workgroups per image share the ranking: every one recomputes the Q*K scores (a few hundred exponentials), ranks the
// candidates i = blockIdx.y*256 + thread (+ gridDim.y*256 ...) against all of them and writes / gathers the winners among its own
// (one workgroup per image ranked 600 candidates of configs[4] in 67 us: 1800 compare rounds on a single CU).
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ logits, int Q, int K1, int T,
                                                   float* __restrict__ scores_out, int64_t* __restrict__ classes_out,
                                                   int32_t* __restrict__ qidx_out, const float* __restrict__ gsrc, int64_t gld, int gcols,
                                                   float* __restrict__ gout) {
    extern __shared__ float sc[];  // Q*K scores, then (rank, query) of this workgroup's winners
    __shared__ int n_win;
    const int b = blockIdx.x;
    const int K = K1 - 1, n = Q * K;
    const float* lg = logits + (int64_t)b * Q * K1;
    if (threadIdx.x == 0) n_win = 0;
    for (int qi = threadIdx.x; qi < Q; qi += 256) {
        float mx = -INFINITY;
        for (int c = 0; c < K1; ++c) mx = fmaxf(mx, lg[qi * K1 + c]);
        float den = 0.f;
        for (int c = 0; c < K1; ++c) den += expf(lg[qi * K1 + c] - mx);
        // NaN logits rank as -inf (a total order: every slot of the outputs is written, the reference's topk would
        // return NaN scores in an arbitrary order); the NaN itself is what is reported as the score
        for (int c = 0; c < K; ++c) sc[qi * K + c] = expf(lg[qi * K1 + c] - mx) / den;
    }
    __syncthreads();
    int2* win = reinterpret_cast<int2*>(sc + n + (n & 1));
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
        const float s = sc[i];
        const float sk = s != s ? -INFINITY : s;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            float o = sc[j];
            o = o != o ? -INFINITY : o;
            rank += (o > sk || (o == sk && j < i)) ? 1 : 0;
        }
        if (rank < T) {
            scores_out[(int64_t)b * T + rank] = s;
            classes_out[(int64_t)b * T + rank] = (int64_t)(i % K);
            qidx_out[(int64_t)b * T + rank] = i / K;
            if (gsrc) win[atomicAdd(&n_win, 1)] = make_int2(rank, i / K);
        }
    }
    if (!gsrc) return;
    // the selected rows of the per-query matrix (uniform branch; every rank 0..T-1 is written exactly once over the image's workgroups)
    __syncthreads();
    const int nw = n_win;
    for (int i = threadIdx.x; i < nw * gcols; i += 256) {
        const int t = i / gcols, c = i - t * gcols;
        gout[((int64_t)b * T + win[t].x) * gcols + c] = gsrc[((int64_t)b * Q + win[t].y) * gld + c];
    }
}

struct InstAcc {          // 32 bytes per (image, instance)
    double sum_sig;
    unsigned int cnt;
    int xmin, ymin, xmax, ymax;
    int pad;
};

__device__ __forceinline__ void src_index(int dst, float scale, int in, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;   // align_corners=False
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

// One partial per workgroup, no atomics: 90 workgroups of an instance used to queue six same-line atomics each behind one
// another at L2 (that queue, not the 197 MB of stores, set the kernel's 81 us); the finish kernel adds the partials of an
// instance in strip order (so the score no longer depends on the arrival order either).  The wave results meet in LDS.
__device__ __forceinline__ void inst_store_partial(InstAcc* __restrict__ acc, int bt, double sum, unsigned int cnt, int xmin,
                                                   int ymin, int xmax, int ymax) {
    __shared__ InstAcc wacc[4];
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) wacc[wave] = InstAcc{sum, cnt, xmin, ymin, xmax, ymax, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        InstAcc a = wacc[0];
        for (int w = 1; w < nw; ++w) {
            a.sum_sig += wacc[w].sum_sig;
            a.cnt += wacc[w].cnt;
            a.xmin = min(a.xmin, wacc[w].xmin);
            a.ymin = min(a.ymin, wacc[w].ymin);
            a.xmax = max(a.xmax, wacc[w].xmax);
            a.ymax = max(a.ymax, wacc[w].ymax);
        }
        const int parts = gridDim.x * gridDim.y;
        acc[(int64_t)bt * parts + blockIdx.y * gridDim.x + blockIdx.x] = a;
    }
}

// grid (tiles_x, rows/ROWS, B*T); each block upsamples a strip of the selected mask
__global__ __launch_bounds__(256) void inst_upsample_kernel(const float* __restrict__ logits, const int32_t* __restrict__ qidx,
                                                            float* __restrict__ masks, InstAcc* __restrict__ acc, int Q, int T,
                                                            int h, int w, int H, int W, int Hs, int Ws, int rows_per_block) {
    const int bt = blockIdx.z;
    const int b = bt / T;
    const int q = min(max(qidx[bt], 0), Q - 1);          // caller-supplied indices never address outside the logits
    const float* src = logits + ((int64_t)b * Q + q) * h * w;
    float* dst = masks + (int64_t)bt * H * W;
    const float sy = (float)h / (float)Hs, sx = (float)w / (float)Ws;     // scale of the (padded) frame; rows/cols >= H/W are cropped
    const int y0 = blockIdx.y * rows_per_block, y1 = min(H, y0 + rows_per_block);
    double sum = 0.0;
    unsigned int cnt = 0;
    int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
    const bool vec = (W % 4) == 0;     // 4 pixels per thread, one 16-byte store
    const int step = vec ? 4 : 1;
    for (int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * step; x0 < W; x0 += gridDim.x * blockDim.x * step) {
        // the column taps of this thread's pixels are the same for every row of the strip
        int xa[4], xb[4];
        float lx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) src_index(min(x0 + e, W - 1), sx, w, xa[e], xb[e], lx[e]);
        float fsum = 0.f;              // <= 4 * rows_per_block sigmoids in fp32, then into the double total
        int xhit_min = 0x7fffffff, xhit_max = -1;
        for (int y = y0; y < y1; ++y) {
            int ya, yb;
            float ly;
            src_index(y, sy, h, ya, yb, ly);
            const float hy = 1.f - ly;
            const float* ra = src + ya * w;
            const float* rb = src + yb * w;
            float o[4];
            bool any = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = 0.f;
                if (e < step) {
                    const float hx = 1.f - lx[e];
                    const float m = hy * (hx * ra[xa[e]] + lx[e] * ra[xb[e]]) + ly * (hx * rb[xa[e]] + lx[e] * rb[xb[e]]);
                    if (m > 0.f) {
                        o[e] = 1.f;
                        // sigmoid through v_exp_f32 / v_rcp_f32 (relative error ~1e-6; the score is a mean over the mask)
                        fsum += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));
                        cnt += 1;
                        xhit_min = min(xhit_min, x0 + e);
                        xhit_max = max(xhit_max, x0 + e);
                        any = true;
                    }
                }
            }
            if (any) {
                ymin = min(ymin, y);
                ymax = max(ymax, y);
            }
            if (vec) *reinterpret_cast<float4*>(dst + (int64_t)y * W + x0) = make_float4(o[0], o[1], o[2], o[3]);
            else dst[(int64_t)y * W + x0] = o[0];
        }
        sum += (double)fsum;
        xmin = min(xmin, xhit_min);
        xmax = max(xmax, xhit_max);
    }
    // wave reduce, then one set of atomics per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
        xmin = min(xmin, __shfl_xor(xmin, o, 64));
        ymin = min(ymin, __shfl_xor(ymin, o, 64));
        xmax = max(xmax, __shfl_xor(xmax, o, 64));
        ymax = max(ymax, __shfl_xor(ymax, o, 64));
    }
    inst_store_partial(acc, bt, sum, cnt, xmin, ymin, xmax, ymax);
}

// The same strip for the 4x case every shipped configuration hits (mask logits at 1/4 of the padded frame, W % 4 == 0):
// a thread's four output pixels 4k..4k+3 read source columns k-1, k, k+1 only and the 16 output rows of a strip read
// six source rows, so a strip costs 18 loads per thread instead of 256 (the generic kernel is bound by its tap loads,
// not by the 197 MB of masks it writes).  Same weights, same expression, same clamped taps: identical results.
__global__ __launch_bounds__(256) void inst_upsample4_kernel(const float* __restrict__ logits, const int32_t* __restrict__ qidx,
                                                             float* __restrict__ masks, InstAcc* __restrict__ acc, int Q, int T,
                                                             int h, int w, int H, int W, int Hs, int Ws) {
    const int bt = blockIdx.z;
    const int b = bt / T;
    const int q = min(max(qidx[bt], 0), Q - 1);          // caller-supplied indices never address outside the logits
    const float* src = logits + ((int64_t)b * Q + q) * h * w;
    float* dst = masks + (int64_t)bt * H * W;
    const float sy = (float)h / (float)Hs, sx = (float)w / (float)Ws;
    const int y0 = blockIdx.y * 16;
    const int j0 = y0 >> 2;
    double sum = 0.0;
    unsigned int cnt = 0;
    int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; 4 * k < W; k += gridDim.x * blockDim.x) {
        const int x0 = 4 * k;
        float lx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int xa, xb;
            src_index(x0 + e, sx, w, xa, xb, lx[e]);
        }
        const int c0 = max(k - 1, 0), c2 = min(k + 1, w - 1);
        float v[6][3];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float* row = src + min(max(j0 - 1 + i, 0), h - 1) * w;
            v[i][0] = row[c0];
            v[i][1] = row[k];
            v[i][2] = row[c2];
        }
        // the horizontal interpolation of the six source rows once per strip: exactly the inner terms of
        // hy * (hx*a + lx*b) + ly * (hx*c + lx*d), so the result is unchanged while a pixel costs 3 instead of 7 flops
        float hrow[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ca = e < 2 ? 0 : 1;                          // columns (k-1, k) for pixels 4k, 4k+1; (k, k+1) for 4k+2, 4k+3
                hrow[i][e] = (1.f - lx[e]) * v[i][ca] + lx[e] * v[i][ca + 1];
            }
        float fsum = 0.f;
        bool hit[4] = {false, false, false, false};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int y = y0 + t;
            if (y < H) {
                int ya, yb;
                float ly;
                src_index(y, sy, h, ya, yb, ly);
                const float hy = 1.f - ly;
                const int a = (t >> 2) + ((t & 3) < 2 ? 0 : 1);      // rows (j-1, j) for the upper half of a source row, (j, j+1) below
                float o[4];
                bool any = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = hy * hrow[a][e] + ly * hrow[a + 1][e];
                    o[e] = 0.f;
                    if (m > 0.f) {
                        o[e] = 1.f;
                        // sigmoid through v_exp_f32 / v_rcp_f32 (relative error ~1e-6; the score is a mean over the mask)
                        fsum += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));
                        cnt += 1;
                        hit[e] = true;
                        any = true;
                    }
                }
                if (any) {
                    ymin = min(ymin, y);
                    ymax = max(ymax, y);
                }
                *reinterpret_cast<float4*>(dst + (int64_t)y * W + x0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        int xhit_min = 0x7fffffff, xhit_max = -1;
#pragma unroll
        for (int e = 3; e >= 0; --e) xhit_min = hit[e] ? x0 + e : xhit_min;
#pragma unroll
        for (int e = 0; e < 4; ++e) xhit_max = hit[e] ? x0 + e : xhit_max;
        sum += (double)fsum;
        xmin = min(xmin, xhit_min);
        xmax = max(xmax, xhit_max);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
        xmin = min(xmin, __shfl_xor(xmin, o, 64));
        ymin = min(ymin, __shfl_xor(ymin, o, 64));
        xmax = max(xmax, __shfl_xor(xmax, o, 64));
        ymax = max(ymax, __shfl_xor(ymax, o, 64));
    }
    inst_store_partial(acc, bt, sum, cnt, xmin, ymin, xmax, ymax);
}

__global__ void inst_finish_kernel(const InstAcc* __restrict__ acc, int parts, const float* __restrict__ class_scores,
                                   float* __restrict__ score, float* __restrict__ boxes, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    InstAcc a = InstAcc{0.0, 0u, 0x7fffffff, 0x7fffffff, -1, -1, 0};
    for (int p = 0; p < parts; ++p) {
        const InstAcc t = acc[(int64_t)i * parts + p];
        a.sum_sig += t.sum_sig;
        a.cnt += t.cnt;
        a.xmin = min(a.xmin, t.xmin);
        a.ymin = min(a.ymin, t.ymin);
        a.xmax = max(a.xmax, t.xmax);
        a.ymax = max(a.ymax, t.ymax);
    }
    const float ms = (float)a.sum_sig / ((float)a.cnt + 1e-6f);
    score[i] = class_scores ? class_scores[i] * ms : ms;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.cnt > 0) bx = make_float4((float)a.xmin, (float)a.ymin, (float)(a.xmax + 1), (float)(a.ymax + 1));
    reinterpret_cast<float4*>(boxes)[i] = bx;
}

}  // namespace msm

using namespace msm;

static int topk_impl(const char* who, const float* pred_logits, int B, int Q, int K1, int T, float* scores_out, int64_t* classes_out,
                     int32_t* query_index_out, const float* gather_src, int64_t gather_ld, int gather_cols, float* gather_out, void* stream) {
    MSM_REQUIRE(pred_logits && scores_out && classes_out && query_index_out, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && Q > 0 && K1 >= 2, "%s: bad sizes", who);
    const int n = Q * (K1 - 1);
    MSM_REQUIRE(n <= 4096 && T > 0 && T <= n, "%s: need T <= Q*K <= 4096 (T=%d, Q*K=%d)", who, T, n);
    MSM_REQUIRE(!gather_src || (gather_out && gather_cols > 0 && gather_ld >= gather_cols), "%s: bad gather arguments", who);
    // (the winners' list is 8-byte aligned: n rounded up to an even count of floats in front of it)
    const int parts = min(16, cdiv(n, 256));          // one candidate per thread: the 600 of configs[4] are three workgroups, one round of compares each
    hipLaunchKernelGGL(topk_kernel, dim3(B, parts), dim3(256), sizeof(float) * (n + (n & 1)) + sizeof(int) * 2 * (gather_src ? min(T, 256) : 0),
                       (hipStream_t)stream, pred_logits, Q, K1, T, scores_out, classes_out, query_index_out, gather_src, gather_ld, gather_cols,
                       gather_out);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_topk_class_scores(const float* pred_logits, int B, int Q, int K1, int T, float* scores_out, int64_t* classes_out,
                                     int32_t* query_index_out, void* stream) {
    return topk_impl("msm_topk_class_scores", pred_logits, B, Q, K1, T, scores_out, classes_out, query_index_out, nullptr, 0, 0, nullptr, stream);
}

extern "C" int msm_topk_class_scores_gather(const float* pred_logits, int B, int Q, int K1, int T, float* scores_out, int64_t* classes_out,
                                            int32_t* query_index_out, const float* gather_src, int64_t gather_ld, int gather_cols,
                                            float* gather_out, void* stream) {
    MSM_REQUIRE(gather_src && gather_out, "msm_topk_class_scores_gather: null gather pointer");
    return topk_impl("msm_topk_class_scores_gather", pred_logits, B, Q, K1, T, scores_out, classes_out, query_index_out, gather_src, gather_ld,
                     gather_cols, gather_out, stream);
}

extern "C" int64_t msm_instance_postprocess_workspace(int B, int T, int H, int W) {
    const int cols = (W % 4 == 0) ? W / 4 : W;
    const int threads = min(256, cdiv(cols, 64) * 64);
    return (int64_t)B * T * cdiv(cols, threads) * cdiv(H, 16) * (int64_t)(sizeof(InstAcc) / sizeof(float));
}

extern "C" int msm_instance_postprocess(const float* mask_logits, const int32_t* query_index,
                                        const float* class_scores, float* pred_masks,
                                        float* mask_score, float* boxes, int B, int Q, int T, int h, int w, int H, int W,
                                        int Hs, int Ws, float* workspace, void* stream) {
    MSM_REQUIRE(mask_logits && query_index && pred_masks && mask_score && boxes && workspace,
                "msm_instance_postprocess: null pointer");
    MSM_REQUIRE(B > 0 && Q > 0 && T > 0 && h > 0 && w > 0 && H > 0 && W > 0, "msm_instance_postprocess: bad sizes");
    MSM_REQUIRE(Hs >= H && Ws >= W, "msm_instance_postprocess: frame %dx%d smaller than the output %dx%d", Hs, Ws, H, W);
    MSM_REQUIRE((((uintptr_t)workspace) & 7) == 0 && (((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)pred_masks) & 15) == 0,
                "msm_instance_postprocess: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    InstAcc* acc = reinterpret_cast<InstAcc*>(workspace);
    const int n = B * T;
    const int rows = 16;
    const int cols = (W % 4 == 0) ? W / 4 : W;                 // threads needed across a row
    const int threads = min(256, cdiv(cols, 64) * 64);          // whole waves, no idle wave (640 px -> 192 threads)
    dim3 grid(cdiv(cols, threads), cdiv(H, rows), n);
    if (Hs == 4 * h && Ws == 4 * w && W % 4 == 0 && opt(MSM_OPT_POST_GENERIC) != 1)
        hipLaunchKernelGGL(inst_upsample4_kernel, grid, dim3(threads), 0, st, mask_logits, query_index, pred_masks, acc, Q, T, h, w,
                           H, W, Hs, Ws);
    else
        hipLaunchKernelGGL(inst_upsample_kernel, grid, dim3(threads), 0, st, mask_logits, query_index, pred_masks, acc, Q, T, h, w,
                           H, W, Hs, Ws, rows);
    hipLaunchKernelGGL(inst_finish_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, acc, (int)(grid.x * grid.y), class_scores, mask_score, boxes,
                       n);
    MSM_CHECK_LAUNCH("msm_instance_postprocess");
    return MSM_OK;
}
