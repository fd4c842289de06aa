// The decoder's attention masks computed at the resolution they are used at (see include/msm_hip.h: msm_pool_mask_taps,
// msm_attn_mask_pooled).
//
// Reference, forward_prediction_heads (meanshiftformer_transformer_decoder.py:668-680):
//     outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)                      (120 x 160 at 640 x 480)
//     attn_mask    = interpolate(outputs_mask, size = next level, bilinear, align_corners=False).sigmoid() < 0.5
// Only the last of the ten predictions is an output of inference; the other nine exist for their attention masks (15 x 20,
// 30 x 40, 60 x 80).  Bilinear interpolation acts on the spatial axes, the contraction on the channel axis: they commute,
//     interpolate(einsum(e, F)) = einsum(e, interpolate(F)),
// and with the factored mask features F = Wm a + bm (FoldedMaskFeatures; the interpolation weights sum to 1)
//     attention logits = einsum(e Wm, interpolate(a)) + e.bm.
// For an integer ratio p = 2, 4, 8 align_corners=False puts every target pixel exactly between source rows p*y + p/2 - 1, p*y + p/2
// (weights 1/2, 1/2; columns alike): interpolate(a) is the mean of the four centre taps.  So the 64-channel activation is pooled
// ONCE per forward to the three key resolutions (6300 tokens per image instead of 19 200 pixels) and each of the nine
// intermediate mask steps contracts 100 queries with 300 / 1200 / 4800 pooled tokens: 1/64, 1/16, 1/4 of the full-resolution
// step's FLOPs (SURVEY 8d names the shortcut -- "compute only the taps needed for the next attention mask on 9 of 10 calls" --
// and how to report it: executed and reference FLOPs side by side).  Same arithmetic up to fp32 summation order (the mean of the
// taps is taken before the contraction instead of after it).
#include "bf16.h"
#include "common.h"

#ifndef AM_EXP
#define AM_EXP 0   // tuning experiments only: 1 = no mask stores
#endif

namespace msm {

constexpr int AM_C = 64;             // channels of the factored activation
constexpr int AM_MAXL = 4;

struct PoolLevels {
    int n;
    int pool[AM_MAXL], th[AM_MAXL], tw[AM_MAXL];
    float* out[AM_MAXL];             // [B][th*tw][64] token-major
};

// out_l[b][ty*tw + tx][c] = mean of act[b][c][p*ty + p/2 - 1 .. + 1][p*tx + p/2 - 1 .. + 1].  grid (blocks of 16 targets, B, level)
__global__ __launch_bounds__(256) void pool_taps_kernel(const float* __restrict__ act, PoolLevels lv, int H, int W,
                                                        int32_t* __restrict__ zero_buf, int64_t zero_count) {
    __shared__ __attribute__((aligned(16))) float tile[16][AM_C + 4];
    const int l = blockIdx.z, b = blockIdx.y;
    if (zero_buf) {     // the row flags of the first attention-mask step, cleared here instead of by a fill launch
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * gridDim.z * 256;
        for (int64_t i = (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_count; i += nthreads)
            zero_buf[i] = 0;
    }
    const int p = lv.pool[l], th = lv.th[l], tw = lv.tw[l], T = th * tw;
    const int t0 = blockIdx.x * 16;
    if (t0 >= T) return;                                   // the coarser levels have fewer blocks (uniform exit)
    const int tid = threadIdx.x, tl = tid & 15, cg = tid >> 4;
    const int t = min(t0 + tl, T - 1);
    const int ty = t / tw, tx = t - ty * tw;
    const int y0 = p * ty + p / 2 - 1, x0 = p * tx + p / 2 - 1;
    const float* a = act + ((int64_t)b * AM_C * H + y0) * W + x0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cg + 16 * i;
        const float* q = a + (int64_t)c * H * W;
        tile[tl][c] = 0.25f * ((q[0] + q[1]) + (q[W] + q[W + 1]));
    }
    __syncthreads();
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    if (t0 + r < T)
        *reinterpret_cast<float4*>(lv.out[l] + ((int64_t)b * T + t0 + r) * AM_C + c4) = *reinterpret_cast<const float4*>(&tile[r][c4]);
}

// attn[b][q][t] = (sum_c embed[b][q][c] * pooled[b][t][c] + qbias[b][q]) < 0;  row_any[b][q] = 1 if some key of the row is not
// masked (cleared by the caller or here).  A wave holds the embeddings of TWO 16-query blocks (blockIdx.z picks the pair) as B
// operands -- eight loads in front of its first MFMA instead of twenty-eight for all seven blocks: the launch sits on the decoder's
// critical path and is all latency at 300 keys -- and walks 16-key blocks: D[key 4 lq + r][query lj], so a lane ends with four
// consecutive keys of one query -- one 4-byte store.  K order: step s of the 16 carries channel lq*16 + s on both operands (a
// lane reads 16 consecutive floats of its row).
constexpr int AM_NQ = 2;             // query blocks per wave
// BITS (round 5): the mask leaves bit-packed and blocked for msm_hypersphere_attn_fused_kv_fwd -- per 16-key block 256 bytes = [query lj][8 query
// blocks of the 112-query chunk] 16-bit words, bit k = key 16 kb + k (msm_attn_pack_mask_bits' layout, T % 16 == 0) -- instead of bytes: the
// four lane quarters of a query OR their nibbles together and one of them stores the word.
// F16 (16-bit plans, round 5): the 64-channel contraction as two v_mfma_f32_16x16x32_f16 per (query block, key block) instead of sixteen
// dependent v_mfma_f32_16x16x4_f32 -- embedding and pooled activation rounded to IEEE half (clamped) as the plans' full-resolution mask
// step rounds them; the fp32 chain is ~1000 cycles of matrix-pipe latency per key block and the launch is all latency.
// F16 = 2 (round 6, the 16-bit plans' default): both operands as hi + lo IEEE-half pairs (x = h + l up to 2^-22 |x|), three terms per product
// (l h, h l, h h: six K = 32 MFMAs per key block, 96 cycles of the matrix pipe against the fp32 chain's 512) -- fp32-class logits, so the mask
// bits that feed back into the attention keep the accuracy the fp32 form was kept for; 12.0 -> ~7 us at 4800 keys.
__device__ __forceinline__ void split8h(const float4& p, const float4& q, f16x8& h, f16x8& l) {
    h = cvt8h(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w);
    l = f16x8{(_Float16)(p.x - (float)h[0]), (_Float16)(p.y - (float)h[1]), (_Float16)(p.z - (float)h[2]), (_Float16)(p.w - (float)h[3]),
              (_Float16)(q.x - (float)h[4]), (_Float16)(q.y - (float)h[5]), (_Float16)(q.z - (float)h[6]), (_Float16)(q.w - (float)h[7])};
}
template <bool VEC, bool BITS = false, int F16 = 0>
__global__ __launch_bounds__(256) void attn_mask_pooled_kernel(const float* __restrict__ embed, int64_t embed_ld, const float* __restrict__ qbias,
                                                               int64_t qbias_ld, const float* __restrict__ pooled, uint8_t* __restrict__ attn,
                                                               int32_t* __restrict__ row_any, int Q, int T) {
    // XCD-aware grid (round 6): (image, key-block group, query-block pair) -- linear id % 8 = image % 8 when B is a multiple of 8, so the
    // workgroups that walk one image's pooled tokens (up to 1.2 MB at 4800 keys) share ONE XCD's L2 instead of fetching them on several
    const int b = blockIdx.x;
    const int qb0 = blockIdx.z * AM_NQ;                     // first query block of this wave
    if (qb0 * 16 >= Q) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    // a lane's 16 of the 64 channels: fp32 form channels 16 lq .. + 15; F16 form 8 lq .. + 7 and 32 + 8 lq .. + 7 (the two K = 32 steps)
    constexpr int LQW = F16 ? 8 : 16;
    auto off = [](int u) { return F16 ? (u >> 1) * 32 + (u & 1) * 4 : u * 4; };
    float4 w[AM_NQ][4];
    f16x8 wh[AM_NQ][2], wl[AM_NQ][2];
    float qb[AM_NQ];
#pragma unroll
    for (int m = 0; m < AM_NQ; ++m) {
        const int q = (qb0 + m) * 16 + lj;
        const float* ep = embed + ((int64_t)b * Q + min(q, Q - 1)) * embed_ld + lq * LQW;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[m][u] = q < Q ? *reinterpret_cast<const float4*>(ep + off(u)) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (F16 == 2) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) split8h(w[m][2 * s_], w[m][2 * s_ + 1], wh[m][s_], wl[m][s_]);
        } else if constexpr (F16 == 1) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
                wh[m][s_] = cvt8h(w[m][2 * s_].x, w[m][2 * s_].y, w[m][2 * s_].z, w[m][2 * s_].w, w[m][2 * s_ + 1].x, w[m][2 * s_ + 1].y, w[m][2 * s_ + 1].z,
                                  w[m][2 * s_ + 1].w);
        }
        qb[m] = (qbias && q < Q) ? qbias[(int64_t)b * Q * qbias_ld + (int64_t)q * qbias_ld] : 0.f;
    }
    const int nq = min(AM_NQ, (Q - qb0 * 16 + 15) / 16);    // query blocks that hold a query (uniform)
    unsigned anyu = 0;                                      // bit m: some key of query (qb0 + m)*16 + lj was left unmasked by this lane
    const int nkb = (T + 15) / 16;
    const float* pb = pooled + (int64_t)b * T * AM_C + lq * LQW;
    uint8_t* ab = attn + (int64_t)b * Q * T;
    int kb = blockIdx.y * 4 + wave;
    float4 a[4], an[4];
    if (kb < nkb) {
        const float* ap = pb + (int64_t)min(kb * 16 + lj, T - 1) * AM_C;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(ap + off(u));
    }
    for (; kb < nkb; kb += gridDim.y * 4) {
        // the next block's keys are requested before this block's MFMAs (clamped: the last trip re-reads its own block)
        const int kn = min(kb + (int)gridDim.y * 4, nkb - 1);
        const float* apn = pb + (int64_t)min(kn * 16 + lj, T - 1) * AM_C;
#pragma unroll
        for (int u = 0; u < 4; ++u) an[u] = *reinterpret_cast<const float4*>(apn + off(u));
        const int key0 = kb * 16 + lq * 4;
#pragma unroll
        for (int m = 0; m < AM_NQ; ++m) {
            if (m >= nq) break;
            f32x4 acc = f32x4{qb[m], qb[m], qb[m], qb[m]};
            if constexpr (F16 == 2) {
                f16x8 ah0, al0, ah1, al1;                        // (split once per key block would cost registers the two query blocks need)
                split8h(a[0], a[1], ah0, al0);
                split8h(a[2], a[3], ah1, al1);
                f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f};            // the small terms in their own accumulator, added last
                lo = mfma_f16k32(al0, wh[m][0], lo);
                lo = mfma_f16k32(ah0, wl[m][0], lo);
                lo = mfma_f16k32(al1, wh[m][1], lo);
                lo = mfma_f16k32(ah1, wl[m][1], lo);
                acc = mfma_f16k32(ah0, wh[m][0], acc);
                acc = mfma_f16k32(ah1, wh[m][1], acc);
                acc = acc + lo;
            } else if constexpr (F16 == 1) {
                acc = mfma_f16k32(cvt8h(a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w), wh[m][0], acc);
                acc = mfma_f16k32(cvt8h(a[2].x, a[2].y, a[2].z, a[2].w, a[3].x, a[3].y, a[3].z, a[3].w), wh[m][1], acc);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = mfma16(a[u].x, w[m][u].x, acc);
                    acc = mfma16(a[u].y, w[m][u].y, acc);
                    acc = mfma16(a[u].z, w[m][u].z, acc);
                    acc = mfma16(a[u].w, w[m][u].w, acc);
                }
            }
            const int q = (qb0 + m) * 16 + lj;
            // sigmoid(x) < 0.5  <=>  x < 0 (DEC:677)
            const unsigned m0 = acc[0] < 0.f, m1 = acc[1] < 0.f, m2 = acc[2] < 0.f, m3 = acc[3] < 0.f;
            if constexpr (BITS) {
                unsigned nib = (m0 | (m1 << 1) | (m2 << 2) | (m3 << 3)) << (4 * lq);
                nib = or_lane_rows(nib);
                if ((m0 & m1 & m2 & m3) == 0) anyu |= 1u << m;
                const int gq = qb0 + m, qc = gq / 7, mb = gq - qc * 7;            // 112-query chunk and block within it (attention.hip: AQB = 7)
                const int qchunks = (Q + 111) / 112;
                if (lq == 0 && q < Q)
                    reinterpret_cast<unsigned short*>(attn)[((((int64_t)b * qchunks + qc) * nkb + kb) * 16 + lj) * 8 + mb] = (unsigned short)nib;
            } else if (q < Q) {
                if (VEC && key0 + 3 < T) {
#if AM_EXP == 1
                    if (m0 + m1 + m2 + m3 == 77)
#endif
                    *reinterpret_cast<uint32_t*>(ab + (int64_t)q * T + key0) = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
                    if ((m0 & m1 & m2 & m3) == 0) anyu |= 1u << m;
                } else {
                    const unsigned mm[4] = {m0, m1, m2, m3};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + r < T) {
                            ab[(int64_t)q * T + key0 + r] = (uint8_t)mm[r];
                            if (!mm[r]) anyu |= 1u << m;
                        }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = an[u];
    }
    // rows with at least one unmasked key keep their mask (DEC:618 resets the others): same flag value from every writer
    anyu = or_lane_rows(anyu);
    if (lq == 0) {
#pragma unroll
        for (int m = 0; m < AM_NQ; ++m)
            if (((anyu >> m) & 1u) && (qb0 + m) * 16 + lj < Q) row_any[(int64_t)b * Q + (qb0 + m) * 16 + lj] = 1;
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_pool_mask_taps(const float* act, int B, int H, int W, int n_levels, const int32_t* th, const int32_t* tw,
                                  float* const* out, int32_t* zero_buf, int64_t zero_count, void* stream) {
    MSM_REQUIRE(!zero_buf || zero_count > 0, "msm_pool_mask_taps: zero_buf needs a positive count");
    MSM_REQUIRE(act && th && tw && out && n_levels >= 1 && n_levels <= AM_MAXL, "msm_pool_mask_taps: bad arguments (1..%d levels)", AM_MAXL);
    MSM_REQUIRE(B > 0 && H > 1 && W > 1, "msm_pool_mask_taps: bad sizes");
    PoolLevels lv;
    lv.n = n_levels;
    int maxT = 0;
    for (int l = 0; l < AM_MAXL; ++l) {
        if (l < n_levels) {
            MSM_REQUIRE(out[l] && th[l] > 0 && tw[l] > 0 && H % th[l] == 0 && W % tw[l] == 0 && H / th[l] == W / tw[l],
                        "msm_pool_mask_taps: level %d (%dx%d) is not an integer reduction of %dx%d", l, th[l], tw[l], H, W);
            const int p = H / th[l];
            MSM_REQUIRE(p == 2 || p == 4 || p == 8, "msm_pool_mask_taps: ratio %d not in {2, 4, 8}", p);
            MSM_REQUIRE((((uintptr_t)out[l]) & 15) == 0, "msm_pool_mask_taps: misaligned output");
            lv.pool[l] = p; lv.th[l] = th[l]; lv.tw[l] = tw[l]; lv.out[l] = out[l];
            maxT = max(maxT, th[l] * tw[l]);
        } else {
            lv.pool[l] = 2; lv.th[l] = lv.tw[l] = 0; lv.out[l] = nullptr;
        }
    }
    hipLaunchKernelGGL(pool_taps_kernel, dim3(cdiv(maxT, 16), B, n_levels), dim3(256), 0, (hipStream_t)stream, act, lv, H, W, zero_buf, zero_count);
    MSM_CHECK_LAUNCH("msm_pool_mask_taps");
    return MSM_OK;
}

extern "C" int msm_attn_mask_pooled(const float* embed, int64_t embed_ld, const float* qbias, int64_t qbias_ld, const float* pooled,
                                    uint8_t* attn, int32_t* row_any, int row_any_cleared, int flags, int B, int Q, int T, void* stream) {
    const int bits = flags & 1;
    MSM_REQUIRE((flags & ~7) == 0 && (flags & 6) != 6, "msm_attn_mask_pooled: flags=%d (1 = bit-packed output, 2 = IEEE-half operands, 4 = hi + lo IEEE-half operands)", flags);
    MSM_REQUIRE(embed && pooled && attn && row_any, "msm_attn_mask_pooled: null pointer");
    MSM_REQUIRE(!bits || (T % 16 == 0 && (((uintptr_t)attn) & 15) == 0), "msm_attn_mask_pooled: the bit-packed mask needs T %% 16 == 0 and a 16-byte aligned buffer");
    MSM_REQUIRE(B > 0 && Q > 0 && Q <= 65535 * 16 * AM_NQ && T > 0, "msm_attn_mask_pooled: bad sizes");
    MSM_REQUIRE(embed_ld >= AM_C && embed_ld % 4 == 0 && (((uintptr_t)embed) & 15) == 0 && (((uintptr_t)pooled) & 15) == 0,
                "msm_attn_mask_pooled: embed / pooled must be 16-byte aligned rows of 64 floats");
    MSM_REQUIRE(!qbias || qbias_ld >= 1, "msm_attn_mask_pooled: bad qbias stride");
    hipStream_t st = (hipStream_t)stream;
    if (!row_any_cleared) MSM_CHECK_HIP(hipMemsetAsync(row_any, 0, sizeof(int32_t) * (size_t)B * Q, st));
    const int nkb = cdiv(T, 16);
    const int zq = cdiv(cdiv(Q, 16), AM_NQ);                  // query-block pairs
    // about one wave per SIMD of the chip over (images, pairs): 1024 / (B * zq) waves walk an image's key blocks for a pair
    const int wgs = max(1, min(cdiv(nkb, 4), max(1, 256 / (B * zq))));     // (128 / 512 / 1024 measured slower at 4800 keys: 20.4 / 12.9 / 14.8 against 12.6 us)
    const bool vec = T % 4 == 0 && (((uintptr_t)attn) & 3) == 0;
    const int f16 = (flags & 4) ? 2 : ((flags & 2) ? 1 : 0);
#define AM_LAUNCH(V_, B_, F_)                                                                                                                          \
    hipLaunchKernelGGL((attn_mask_pooled_kernel<V_, B_, F_>), dim3(B, wgs, zq), dim3(256), 0, st, embed, embed_ld, qbias, qbias_ld, pooled, attn, row_any, \
                       Q, T)
#define AM_FORMS(V_, B_)                      \
    if (f16 == 2) AM_LAUNCH(V_, B_, 2);       \
    else if (f16 == 1) AM_LAUNCH(V_, B_, 1);  \
    else AM_LAUNCH(V_, B_, 0)
    if (bits) {
        AM_FORMS(true, true);
    } else if (vec) {
        AM_FORMS(true, false);
    } else {
        AM_FORMS(false, false);
    }
#undef AM_FORMS
#undef AM_LAUNCH
    MSM_CHECK_LAUNCH("msm_attn_mask_pooled");
    return MSM_OK;
}
