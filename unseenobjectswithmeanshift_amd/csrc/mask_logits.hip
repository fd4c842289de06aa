// Query x pixel-embedding mask step with the attention-mask derivation fused into the epilogue
// (see include/msm_hip.h: msm_mask_logits_fwd).
//
// Reference: forward_prediction_heads, meanshiftformer_transformer_decoder.py:668 (einsum
// "bqc,bchw->bqhw") and :675-680 (bilinear downsample to the next level, sigmoid < 0.5, repeat
// over heads).  F.interpolate(align_corners=False) from H x W to (H/s) x (W/s), s in {2,4,8}, samples
// at s*i + s/2 - 0.5, i.e. the exact average of the 2x2 block at rows/cols {s*i+s/2-1, s*i+s/2};
// with all four weights 0.25 the result is 0.25*((a+b)+(c+d)) bit-for-bit, and sigmoid(x) < 0.5
// <=> x < 0 (up to |x| < 6e-8 where fp32 sigmoid rounds to 0.5).  The mask bit is therefore
// sign((a+b)+(c+d)) of four accumulators that already sit in registers.
//
// Mapping (fp32 is MFMA-bound here: AI 35.8 FLOP/B against a ridge of ~20):
//   * one workgroup = 4 waves, one image b, one chunk of <=112 queries (7 MFMA row blocks);
//     the chunk's mask_embed rows live in LDS ([112][C+2], conflict-free ds_read_b32) for the
//     whole workgroup lifetime;
//   * one wave tile = 2 image rows x 32 columns: lane (j = l&15, kq = l>>4) streams
//     mask_feat[k0+kq][row][c0+2j..+1] as float2 for both rows straight from HBM/L2 into the B
//     operand (16 lanes x 8 B = one 128 B line per k-row), so the four 16-column MFMA tiles of a
//     wave are {top even cols, top odd cols, bottom even, bottom odd} and a 2x2 tap block is
//     lane-local (s=2) or one lane away (s=4,8);
//   * 7 x 4 accumulators (112 VGPRs), K-loop in register-prefetched groups of 8 k-steps.
#include "common.h"

namespace msm {

constexpr int QB = 7;           // 16-row MFMA blocks per query chunk
constexpr int QCH = QB * 16;    // 112 queries per chunk
constexpr int KU = 8;           // k-steps (of 4) per prefetch group

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 ld_f2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

template <int POOL, bool WRITE>
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ emb, const float* __restrict__ feat,
                                                          float* __restrict__ mask_out, uint8_t* __restrict__ attn_out,
                                                          int32_t* __restrict__ row_any, int Q, int C, int H, int W,
                                                          int th, int tw, int ypar, int n_rowpairs, int rp_step,
                                                          int rp_first, int feat_bytes) {
    extern __shared__ __attribute__((aligned(16))) float Es[];
    const int SE = C + 2;
    const int b = blockIdx.z, qc = blockIdx.y;
    const int q0 = qc * QCH;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave id is wave-uniform but not provably so to hipcc: readfirstlane keeps the tile loop, the
    // row/column bookkeeping and the buffer descriptor in SGPRs (otherwise every buffer load is wrapped
    // in a waterfall loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int HW = H * W;

    // stage this chunk of mask_embed: rows >= Q are zero
    const float* eb = emb + ((int64_t)b * Q + q0) * C;
    for (int idx = tid; idx < QCH * (C / 4); idx += 256) {
        const int r = idx / (C / 4), c4 = (idx - r * (C / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)r * C + c4);
        float2* d = reinterpret_cast<float2*>(&Es[r * SE + c4]);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    __syncthreads();

    const int ctiles = (W + 31) / 32;
    const int ntiles = n_rowpairs * ctiles;
    const float* fb = feat + (int64_t)b * C * HW;
    // buffer descriptor over this image's feature map, held in SGPRs
    const uint64_t fbu = (uint64_t)fb;   // readfirstlane makes the uniformity provable (no waterfall loops)
    // (readfirstlane returns int: go through unsigned, or a low half with bit 31 set sign-extends into the high half)
    const uint64_t fbs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(fbu >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fbu);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)fbs, 0, feat_bytes, 0x00020000);   // feat_bytes = C*H*W*4 from the host: stays scalar

    for (int t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);  // may be -1 (odd pairing): clamp loads
        const int ybot = ytop + 1;                               // may be H
        const int c = ct * 32 + 2 * lj;
        const bool col_ok = c < W;  // W is even
        const int cl = col_ok ? c : 0;
        // per-lane byte offsets of this lane's two pixels in k-row `lq`; the k-group part of the
        // address is wave-uniform and travels in the buffer instruction's SGPR soffset, so the
        // loads need no per-lane 64-bit address arithmetic at all
        const unsigned voff_top = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 4);
        const unsigned voff_bot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 4);

        f32x4 acc[QB][4];
#pragma unroll
        for (int m = 0; m < QB; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // K loop: groups of KU k-steps, two register buffers (A/B) in ping-pong.  The loads of the
        // next group are issued BEFORE the MFMAs of the current one and pinned there with
        // sched_barrier (left alone, hipcc sinks them behind the MFMAs and waits at once); no
        // buffer copies, so the only vmcnt waits are the counted ones at each buffer's first use.
        float2 tA[KU], bA[KU], tB[KU], bB[KU];
        auto load_group = [&](float2(&t)[KU], float2(&bt)[KU], int kbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const unsigned soff = (unsigned)(kbase + u * 4) * (unsigned)HW * 4u;
                t[u] = ld_f2(rsrc, voff_top, soff);
                bt[u] = ld_f2(rsrc, voff_bot, soff);
            }
        };
        auto compute_group = [&](const float2(&t)[KU], const float2(&bt)[KU], int kbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float* er = &Es[lj * SE + kbase + u * 4 + lq];
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    const float a = er[m * 16 * SE];
                    acc[m][0] = mfma16(a, t[u].x, acc[m][0]);
                    acc[m][1] = mfma16(a, t[u].y, acc[m][1]);
                    acc[m][2] = mfma16(a, bt[u].x, acc[m][2]);
                    acc[m][3] = mfma16(a, bt[u].y, acc[m][3]);
                }
            }
        };
        const int G = C / (4 * KU);
        load_group(tA, bA, 0);
        int g = 0;
        for (; g + 1 < G; g += 2) {   // straight-line body: nothing for LLVM to sink the loads into
            load_group(tB, bB, (g + 1) * (4 * KU));
            __builtin_amdgcn_sched_barrier(0);
            compute_group(tA, bA, g * (4 * KU));
            __builtin_amdgcn_sched_barrier(0);
            load_group(tA, bA, min(g + 2, G - 1) * (4 * KU));
            __builtin_amdgcn_sched_barrier(0);
            compute_group(tB, bB, (g + 1) * (4 * KU));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < G) compute_group(tA, bA, g * (4 * KU));   // odd number of groups

        // ---- epilogue: lane holds queries q0 + m*16 + lq*4 + r, columns c (tiles 0,2) and c+1 (1,3)
        if constexpr (WRITE) {
            if (col_ok) {
#pragma unroll
                for (int m = 0; m < QB; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + m * 16 + lq * 4 + r;
                        if (q < Q) {
                            float* o = mask_out + ((int64_t)b * Q + q) * HW + c;
                            if (ytop >= 0) *reinterpret_cast<float2*>(o + (int64_t)ytop * W) = make_float2(acc[m][0][r], acc[m][1][r]);
                            if (ybot < H) *reinterpret_cast<float2*>(o + (int64_t)ybot * W) = make_float2(acc[m][2][r], acc[m][3][r]);
                        }
                    }
                }
            }
        }
        if constexpr (POOL != 0) {
            // tap rows are (POOL*i + POOL/2 - 1, +1): the pair (ytop, ybot) is a tap pair iff
            // ytop % POOL == POOL/2 - 1 (always true for POOL == 2 with even pairing)
            const bool row_tap = (ytop >= 0) && (ybot < H) && ((ytop % POOL) == POOL / 2 - 1);
            // column taps (POOL*i + POOL/2 - 1, +1): POOL 2 -> (c, c+1) in-lane;
            // POOL 4/8 -> (c+1 of this lane, c of lane+1) when (c+1) % POOL == POOL/2 - 1
            bool col_tap;
            int tx;
            if constexpr (POOL == 2) {
                col_tap = col_ok;
                tx = c >> 1;
            } else {
                col_tap = col_ok && (((c + 1) % POOL) == POOL / 2 - 1) && (c + 2 < W);
                tx = (c + 1) / POOL;
            }
            const int ty = (ytop >= 0 ? ytop : 0) / POOL;
            const bool wave_row_tap = row_tap;  // uniform per wave
#pragma unroll
            for (int m = 0; m < QB; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s;
                    if constexpr (POOL == 2) {
                        s = (acc[m][0][r] + acc[m][1][r]) + (acc[m][2][r] + acc[m][3][r]);
                    } else {
                        const float n0 = __shfl_down(acc[m][0][r], 1, 64);
                        const float n2 = __shfl_down(acc[m][2][r], 1, 64);
                        s = (acc[m][1][r] + n0) + (acc[m][3][r] + n2);
                    }
                    const int q = q0 + m * 16 + lq * 4 + r;
                    if (wave_row_tap && col_tap && q < Q && tx < tw && ty < th) {
                        const bool masked = s < 0.f;
                        attn_out[((int64_t)b * Q + q) * (th * tw) + ty * tw + tx] = masked ? 1 : 0;
                        if (!masked) row_any[(int64_t)b * Q + q] = 1;
                    }
                }
            }
        }
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_mask_logits_fwd(const float* mask_embed, const float* mask_feat, float* mask_out,
                                   uint8_t* attn_out, int32_t* row_any, int B, int Q, int C, int H, int W,
                                   int th, int tw, int sparse, void* stream) {
    MSM_REQUIRE(mask_embed && mask_feat, "msm_mask_logits_fwd: null input");
    MSM_REQUIRE(mask_out || attn_out, "msm_mask_logits_fwd: nothing to produce");
    MSM_REQUIRE(B > 0 && Q > 0 && H > 1 && W > 1, "msm_mask_logits_fwd: bad sizes");
    MSM_REQUIRE(C % 32 == 0 && C >= 32 && C <= 320, "msm_mask_logits_fwd: C=%d must be a multiple of 32 and <= 320", C);
    MSM_REQUIRE(W % 2 == 0 && H % 2 == 0, "msm_mask_logits_fwd: H=%d W=%d must be even", H, W);
    MSM_REQUIRE((int64_t)C * H * W * 4 < (int64_t)1 << 31, "msm_mask_logits_fwd: one image of mask_feat must be < 2 GiB");
    MSM_REQUIRE((((uintptr_t)mask_embed) & 15) == 0 && (((uintptr_t)mask_feat) & 7) == 0 &&
                    (!mask_out || (((uintptr_t)mask_out) & 7) == 0),
                "msm_mask_logits_fwd: misaligned pointer");
    int pool = 0;
    if (attn_out) {
        MSM_REQUIRE(row_any, "msm_mask_logits_fwd: row_any required with attn_out");
        MSM_REQUIRE(th > 0 && tw > 0 && H % th == 0 && W % tw == 0 && H / th == W / tw,
                    "msm_mask_logits_fwd: target %dx%d incompatible with %dx%d", th, tw, H, W);
        pool = H / th;
        MSM_REQUIRE(pool == 2 || pool == 4 || pool == 8, "msm_mask_logits_fwd: pool factor %d not in {2,4,8}", pool);
    }
    hipStream_t st = (hipStream_t)stream;
    if (attn_out) MSM_CHECK_HIP(hipMemsetAsync(row_any, 0, sizeof(int32_t) * (size_t)B * Q, st));

    // row pairing: even (rows 2i, 2i+1) unless the taps need odd pairs (POOL 4/8 -> rows 4i+1,4i+2 / 8i+3,8i+4)
    int ypar = 0, n_rowpairs = H / 2, rp_step = 1, rp_first = 0;
    if (pool == 4 || pool == 8) {
        ypar = -1;
        n_rowpairs = H / 2 + 1;  // (-1,0), (1,2), ..., (H-1,H)
        if (sparse && !mask_out) {
            // only the tap pairs: ytop = pool*i + pool/2 - 1 = -1 + 2*(pool/2*i + pool/4)
            rp_step = pool / 2;
            rp_first = pool / 4;
            n_rowpairs = H / pool;
        }
    }
    const int ctiles = (W + 31) / 32;
    const int ntiles = n_rowpairs * ctiles;
    const int qchunks = cdiv(Q, QCH);
    // persistent-ish grid: enough workgroups per (image, chunk) to cover the chip once
    int wg_per = cdiv(ntiles, 4);
    const int target = cdiv(256, B * qchunks);
    if (wg_per > target) wg_per = max(target, 1);
    dim3 grid(wg_per, qchunks, B), block(256);
    const size_t lds = sizeof(float) * (size_t)QCH * (C + 2);
    void (*kern)(const float*, const float*, float*, uint8_t*, int32_t*, int, int, int, int, int, int, int, int, int, int, int);
    const bool wr = mask_out != nullptr;
    switch (pool) {
        case 0: kern = mask_logits_kernel<0, true>; break;
        case 2: kern = wr ? mask_logits_kernel<2, true> : mask_logits_kernel<2, false>; break;
        case 4: kern = wr ? mask_logits_kernel<4, true> : mask_logits_kernel<4, false>; break;
        default: kern = wr ? mask_logits_kernel<8, true> : mask_logits_kernel<8, false>; break;
    }
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, mask_embed, mask_feat, mask_out, attn_out, row_any, Q, C, H, W, th, tw,
                       ypar, n_rowpairs, rp_step, rp_first, (int)((int64_t)C * H * W * 4));
    MSM_CHECK_LAUNCH("msm_mask_logits_fwd");
    return MSM_OK;
}
