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
#include <stdlib.h>

#include "common.h"

namespace msm {

constexpr int QB = 7;           // 16-row MFMA blocks per query chunk
constexpr int QCH = QB * 16;    // 112 queries per chunk
#ifndef MSM_MASK_KU
#define MSM_MASK_KU 4
#endif
#ifndef MSM_MASK_MW
#define MSM_MASK_MW 8
#endif
constexpr int KU = MSM_MASK_KU;  // k-steps (of 4) per prefetch group
constexpr int MW = MSM_MASK_MW;  // waves per workgroup: the 116 KB mask_embed chunk allows ONE workgroup per CU, so 8 waves
                                // give every SIMD two instruction streams (one wave alone cannot hide its own ds_read /
                                // buffer-load issue and waitcnt bubbles behind its MFMAs)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// NC consecutive columns per lane: NC == 2 -> one 8-byte load per (k-row, image row), wave tile 2 x 32;
// NC == 1 -> 4-byte loads, wave tile 2 x 16 (finer tiles: less quantisation loss when the tile count per
// SIMD is small, e.g. 2.34 -> 3 rounds with 2 x 32 but 4.69 -> 5 half-rounds with 2 x 16 at B = 8).
template <int NC>
struct Cols {
    float v[NC];
};
template <int NC>
__device__ __forceinline__ Cols<NC> ld_cols(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    Cols<NC> c;
    if constexpr (NC == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        c.v[0] = __uint_as_float(t.x);
        c.v[1] = __uint_as_float(t.y);
    } else {
        c.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
    return c;
}

__device__ __forceinline__ float dpp_next_in_row(float v) {   // lane i <- lane i + 1 within its row of 16 lanes (row_shl:1)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x101, 0xf, 0xf, true));
}

// Per-tile epilogue shared by the fp32 and bf16 kernels: lane holds queries q0 + m*16 + lq*4 + r, columns c..c+NC-1 of
// rows ytop (acc[.][cc]) and ybot (acc[.][NC+cc]).
template <int POOL, bool WRITE, int NC>
__device__ __forceinline__ void mask_tile_epilogue(const f32x4 (&acc)[QB][2 * NC], float* __restrict__ mask_out,
                                                   uint8_t* __restrict__ attn_out, int* __restrict__ any_flags, int b, int Q,
                                                   int q0, int H, int W, int th, int tw, int ytop, int ybot, int c, bool col_ok,
                                                   int lj, int lq) {
    const int HW = H * W;
    (void)lj;
    // ---- epilogue: lane holds queries q0 + m*16 + lq*4 + r, columns c..c+NC-1 of rows ytop (acc[.][cc])
    //      and ybot (acc[.][NC+cc]).  The query offset is made opaque here: the 28 per-query output base addresses
    //      depend only on the lane, so LICM would otherwise hoist them out of the tile loop and hold 56 VGPRs
    //      across the K loop (209 vs 157 VGPRs; the difference decides between 2 and 3 waves per SIMD).
    int qlane = lq * 4;
    asm volatile("" : "+v"(qlane));
    if constexpr (WRITE) {
        if (col_ok) {
#pragma unroll
            for (int m = 0; m < QB; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + m * 16 + qlane + r;
                    if (q < Q) {
                        float* o = mask_out + ((int64_t)b * Q + q) * HW + c;
                        if constexpr (NC == 2) {
                            if (ytop >= 0) *reinterpret_cast<float2*>(o + (int64_t)ytop * W) = make_float2(acc[m][0][r], acc[m][1][r]);
                            if (ybot < H) *reinterpret_cast<float2*>(o + (int64_t)ybot * W) = make_float2(acc[m][2][r], acc[m][3][r]);
                        } else {
                            if (ytop >= 0) o[(int64_t)ytop * W] = acc[m][0][r];
                            if (ybot < H) o[(int64_t)ybot * W] = acc[m][1][r];
                        }
                    }
                }
            }
        }
    }
    if constexpr (POOL == 1) {
        // mask at full resolution: one bit per logit
        if (col_ok) {
#pragma unroll
            for (int m = 0; m < QB; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + m * 16 + qlane + r;
                    if (q >= Q) continue;
                    uint8_t* o = attn_out + ((int64_t)b * Q + q) * HW + c;
                    bool any = false;
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        if (ytop >= 0) { const bool mk = acc[m][cc][r] < 0.f; o[(int64_t)ytop * W + cc] = mk; any |= !mk; }
                        if (ybot < H) { const bool mk = acc[m][NC + cc][r] < 0.f; o[(int64_t)ybot * W + cc] = mk; any |= !mk; }
                    }
                    if (any) any_flags[q - q0] = 1;
                }
            }
        }
    } else if constexpr (POOL != 0) {
        // tap rows are (POOL*i + POOL/2 - 1, +1): the pair (ytop, ybot) is a tap pair iff
        // ytop % POOL == POOL/2 - 1 (always true for POOL == 2 with even pairing)
        const bool row_tap = (ytop >= 0) && (ybot < H) && ((ytop % POOL) == POOL / 2 - 1);   // wave-uniform
        // tap columns are (POOL*i + POOL/2 - 1, +1).  NC == 2, POOL == 2: both in this lane.  Otherwise
        // the left tap is this lane's LAST column and the right tap the next lane's first.
        constexpr bool IN_LANE = (NC == 2 && POOL == 2);
        const int cleft = IN_LANE ? c : c + NC - 1;
        const bool col_tap = col_ok && ((cleft % POOL) == POOL / 2 - 1) && (cleft + 1 < W);
        const int tx = cleft / POOL;
        const int ty = (ytop >= 0 ? ytop : 0) / POOL;
        if (!row_tap) return;                     // wave-uniform: this row pair feeds no tap (every other pair at POOL 4, 3 of 4 at 8)
#pragma unroll
        for (int m = 0; m < QB; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s;
                if constexpr (IN_LANE) {
                    s = (acc[m][0][r] + acc[m][1][r]) + (acc[m][2][r] + acc[m][3][r]);
                } else {
                    // the right tap is the next lane of the same 16-lane row (a left tap is never lane 15: cleft % POOL ==
                    // POOL/2 - 1 is odd): a DPP row shift instead of a ds_bpermute through the LDS crossbar
                    const float n0 = dpp_next_in_row(acc[m][0][r]);
                    const float n2 = dpp_next_in_row(acc[m][NC][r]);
                    s = (acc[m][NC - 1][r] + n0) + (acc[m][2 * NC - 1][r] + n2);
                }
                const int q = q0 + m * 16 + qlane + r;
                if (row_tap && col_tap && q < Q && tx < tw && ty < th) {
                    const bool masked = s < 0.f;
                    attn_out[((int64_t)b * Q + q) * (th * tw) + ty * tw + tx] = masked ? 1 : 0;
                    if (!masked) any_flags[q - q0] = 1;      // LDS: flushed to row_any once per workgroup (not one hot global store per tile)
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one row block at a time: keeps the epilogue's live set (shuffled
                                                 // neighbours, addresses) from setting the kernel's VGPR count
        }
    }
}

// POOL: 0 = no attention mask; 1 = mask at the resolution of the logits (single-level decoder,
// meanshiftformer_transformer_decoder.py:1012-1035 with target size == mask size: interpolate is the
// identity); 2/4/8 = 2x2-tap average of a bilinear downsample by that factor.
template <int POOL, bool WRITE, int NC>
__global__ __launch_bounds__(MW * 64) void mask_logits_kernel(const float* __restrict__ emb, const float* __restrict__ feat,
                                                          float* __restrict__ mask_out, uint8_t* __restrict__ attn_out,
                                                          int32_t* __restrict__ row_any, int Q, int C, int H, int W,
                                                          int th, int tw, int ypar, int n_rowpairs, int rp_step,
                                                          int rp_first, int feat_bytes, int64_t emb_ld,
                                                          const float* __restrict__ qbias, int64_t qbias_ld) {
    extern __shared__ __attribute__((aligned(16))) float Es[];   // [QCH][C + 2] embeddings, then [QCH] per-query biases
    constexpr int TW = 16 * NC;      // tile width in columns
    constexpr int NA = 2 * NC;       // accumulator column blocks: [row (top,bottom)][cc]
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

    // stage this chunk of mask_embed: rows >= Q are zero; the per-query bias (the folded mask_features bias) starts every
    // accumulator of its row
    const float* eb = emb + ((int64_t)b * Q + q0) * emb_ld;
    for (int idx = tid; idx < QCH * (C / 4); idx += MW * 64) {
        const int r = idx / (C / 4), c4 = (idx - r * (C / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)r * emb_ld + c4);
        float2* d = reinterpret_cast<float2*>(&Es[r * SE + c4]);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    float* qb = Es + QCH * SE;
    int* any_flags = reinterpret_cast<int*>(qb + QCH);
    for (int r = tid; r < QCH; r += MW * 64) {
        qb[r] = (qbias && q0 + r < Q) ? qbias[((int64_t)b * Q + q0 + r) * qbias_ld] : 0.f;
        any_flags[r] = 0;
    }
    __syncthreads();

    const int ctiles = (W + TW - 1) / TW;
    const int ntiles = n_rowpairs * ctiles;
    const float* fb = feat + (int64_t)b * C * HW;
    // buffer descriptor over this image's feature map, held in SGPRs
    const uint64_t fbu = (uint64_t)fb;   // readfirstlane makes the uniformity provable (no waterfall loops)
    // (readfirstlane returns int: go through unsigned, or a low half with bit 31 set sign-extends into the high half)
    const uint64_t fbs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(fbu >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fbu);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)fbs, 0, feat_bytes, 0x00020000);   // feat_bytes = C*H*W*4 from the host: stays scalar

    // Tile schedule: full rounds go to all MW waves of every workgroup; the leftover tiles go first to waves 0..3
    // (one per SIMD) of every workgroup, then to waves 4..7, and so on, so no SIMD gets two leftover tiles
    // while another gets none (waves w, w+4, w+8, ... share a SIMD).
    const int slots = gridDim.x * MW;
    const int full_rounds = ntiles / slots;
    const int left = ntiles - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int my_tiles = full_rounds + (left_slot < left ? 1 : 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int t = (it < full_rounds) ? it * slots + (int)blockIdx.x * MW + wave : full_rounds * slots + left_slot;
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);  // may be -1 (odd pairing): clamp loads
        const int ybot = ytop + 1;                               // may be H
        const int c = ct * TW + NC * lj;
        const bool col_ok = c < W;  // W is a multiple of NC
        const int cl = col_ok ? c : 0;
        // per-lane byte offsets of this lane's pixels in k-row `lq`; the k-group part of the
        // address is wave-uniform and travels in the buffer instruction's SGPR soffset, so the
        // loads need no per-lane 64-bit address arithmetic at all
        const unsigned voff_top = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 4);
        const unsigned voff_bot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 4);

        f32x4 acc[QB][NA];
#pragma unroll
        for (int m = 0; m < QB; ++m)
#pragma unroll
            for (int n = 0; n < NA; ++n) acc[m][n] = *reinterpret_cast<const f32x4*>(qb + m * 16 + lq * 4);

        // K loop: groups of KU k-steps, two register buffers (A/B) in ping-pong.  The loads of the
        // next group are issued BEFORE the MFMAs of the current one and pinned there with
        // sched_barrier (left alone, hipcc sinks them behind the MFMAs and waits at once); no
        // buffer copies, so the only vmcnt waits are the counted ones at each buffer's first use.
        Cols<NC> tA[KU], bA[KU], tB[KU], bB[KU];
        auto load_group = [&](Cols<NC>(&t)[KU], Cols<NC>(&bt)[KU], int kbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const unsigned soff = (unsigned)(kbase + u * 4) * (unsigned)HW * 4u;
                t[u] = ld_cols<NC>(rsrc, voff_top, soff);
                bt[u] = ld_cols<NC>(rsrc, voff_bot, soff);
            }
        };
        auto compute_group = [&](const Cols<NC>(&t)[KU], const Cols<NC>(&bt)[KU], int kbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float* er = &Es[lj * SE + kbase + u * 4 + lq];
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    const float a = er[m * 16 * SE];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        acc[m][cc] = mfma16(a, t[u].v[cc], acc[m][cc]);
                        acc[m][NC + cc] = mfma16(a, bt[u].v[cc], acc[m][NC + cc]);
                    }
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

        mask_tile_epilogue<POOL, WRITE, NC>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c, col_ok, lj, lq);
    }
    if constexpr (POOL != 0) {
        __syncthreads();
        for (int r = tid; r < QCH; r += MW * 64)
            if (any_flags[r] && q0 + r < Q) row_any[(int64_t)b * Q + q0 + r] = 1;
    }
}

// ---- bf16 variant (BASELINE configs 3 and 5) ------------------------------------------------------------------------
// Same product with bf16 operands and fp32 accumulation (v_mfma_f32_16x16x16_bf16): at 2.5 PFLOP/s the 7.9 GFLOP of a
// launch are ~4 us of MFMA, so the step becomes a stream over the feature map -- HBM-bound (AI 71.6 FLOP/B
// against a bf16 ridge of ~312).  The features are kept in a channel-quad packed layout [B][C/4][HW][4] bf16
// (msm_pack_mask_features_bf16): the MFMA B operand of lane (pixel lj, k-group lq) is then ONE 8-byte load and the 16
// pixels of a k-group are a 128-byte line.  A whole tile's operands (32 loads) are requested while the previous tile
// is being multiplied.  Tile shape, schedule and the fused attention-mask epilogue are those of the fp32 kernel.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKS = 16;            // 16-channel k-steps held per tile: C <= 256

__device__ __forceinline__ unsigned short f2bf(float x) {   // round to nearest even
    const unsigned int u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int POOL, bool WRITE>
__global__ __launch_bounds__(MW * 64) void mask_logits_bf16_kernel(const float* __restrict__ emb, const unsigned short* __restrict__ featp,
                                                               float* __restrict__ mask_out, uint8_t* __restrict__ attn_out,
                                                               int32_t* __restrict__ row_any, int Q, int C, int H, int W, int th,
                                                               int tw, int ypar, int n_rowpairs, int rp_step, int rp_first,
                                                               int feat_bytes, int64_t emb_ld, const float* __restrict__ qbias,
                                                               int64_t qbias_ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned short Eb[];   // [QCH][C + 8] bf16, then [QCH] fp32 per-query biases
    const int SEb = C + 8;           // 132 dwords per row at C = 256: ds_read_b64 of (lj, lq) hits 64 distinct banks
    const int b = blockIdx.z, qc = blockIdx.y;
    const int q0 = qc * QCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int HW = H * W;
    const int nks = C / 16;

    const float* eb = emb + ((int64_t)b * Q + q0) * emb_ld;
    float* qb = reinterpret_cast<float*>(Eb + QCH * SEb);
    int* any_flags = reinterpret_cast<int*>(qb + QCH);
    for (int r = tid; r < QCH; r += MW * 64) {
        qb[r] = (qbias && q0 + r < Q) ? qbias[((int64_t)b * Q + q0 + r) * qbias_ld] : 0.f;
        any_flags[r] = 0;
    }
    for (int idx = tid; idx < QCH * (C / 4); idx += MW * 64) {
        const int r = idx / (C / 4), c4 = (idx - r * (C / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)r * emb_ld + c4);
        u32x2 pk;
        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
        *reinterpret_cast<u32x2*>(&Eb[r * SEb + c4]) = pk;
    }
    __syncthreads();

    const int ctiles = (W + 15) / 16;
    const int ntiles = n_rowpairs * ctiles;
    const uint64_t fbu = (uint64_t)(featp + (int64_t)b * C * HW);
    const uint64_t fbs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(fbu >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fbu);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fbs, 0, feat_bytes, 0x00020000);

    const int slots = gridDim.x * MW;
    const int full_rounds = ntiles / slots;
    const int left = ntiles - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int my_tiles = full_rounds + (left_slot < left ? 1 : 0);
    auto tile_of = [&](int it) {
        return (it < full_rounds) ? it * slots + (int)blockIdx.x * MW + wave : full_rounds * slots + left_slot;
    };
    struct TileRegs {
        u32x2 t[BKS], bt[BKS];
    };
    auto load_tile = [&](int t, TileRegs& r) {
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step), ybot = ytop + 1;
        const int c = ct * 16 + lj;
        const int cl = c < W ? c : 0;
        // packed element (k-quad, pixel): 8 bytes at ((kq * HW) + pixel) * 8; the k-step part travels in soffset
        const unsigned voff_top = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 8);
        const unsigned voff_bot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 8);
#pragma unroll
        for (int ks = 0; ks < BKS; ++ks) {
            const unsigned soff = (unsigned)min(ks, nks - 1) * 4u * (unsigned)HW * 8u;   // clamped: C < 256 re-reads, never faults
            r.t[ks] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_top, soff, 0);
            r.bt[ks] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_bot, soff, 0);
        }
    };
    TileRegs cur, nxt;
    if (my_tiles > 0) load_tile(tile_of(0), cur);
    for (int it = 0; it < my_tiles; ++it) {
        const int t = tile_of(it);
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);
        const int ybot = ytop + 1;
        const int c = ct * 16 + lj;
        const bool col_ok = c < W;
        load_tile(tile_of(min(it + 1, my_tiles - 1)), nxt);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[QB][2];
#pragma unroll
        for (int m = 0; m < QB; ++m) acc[m][0] = acc[m][1] = *reinterpret_cast<const f32x4*>(qb + m * 16 + lq * 4);
#pragma unroll
        for (int ks = 0; ks < BKS; ++ks) {
            if (ks < nks) {                                              // wave-uniform
                const unsigned short* er = &Eb[lj * SEb + ks * 16 + lq * 4];
                const bf16x4 bt_ = __builtin_bit_cast(bf16x4, cur.t[ks]);
                const bf16x4 bb_ = __builtin_bit_cast(bf16x4, cur.bt[ks]);
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    const bf16x4 a = __builtin_bit_cast(bf16x4, *reinterpret_cast<const u32x2*>(er + m * 16 * SEb));
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bt_, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bb_, acc[m][1], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mask_tile_epilogue<POOL, WRITE, 1>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c, col_ok, lj, lq);
#pragma unroll
        for (int ks = 0; ks < BKS; ++ks) {
            cur.t[ks] = nxt.t[ks];
            cur.bt[ks] = nxt.bt[ks];
        }
    }
    if constexpr (POOL != 0) {
        __syncthreads();
        for (int r = tid; r < QCH; r += MW * 64)
            if (any_flags[r] && q0 + r < Q) row_any[(int64_t)b * Q + q0 + r] = 1;
    }
}

// fp32 NCHW [B][C][HW] -> bf16 channel-quad packed [B][C/4][HW][4]
__global__ __launch_bounds__(256) void pack_mask_features_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                                      int64_t total, int C4, int HW) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;                 // b * C4 + c4
        const float* src = in + (r * 4) * HW + p;
        u32x2 pk;
        pk.x = (unsigned)f2bf(src[0]) | ((unsigned)f2bf(src[HW]) << 16);
        pk.y = (unsigned)f2bf(src[2 * (int64_t)HW]) | ((unsigned)f2bf(src[3 * (int64_t)HW]) << 16);
        *reinterpret_cast<u32x2*>(out + i * 4) = pk;
    }
}

}  // namespace msm

using namespace msm;

static int mask_embed_check(const char* who, const float* mask_embed, int64_t& embed_ld, const float* qbias, int64_t& qbias_ld, int C) {
    if (embed_ld == 0) embed_ld = C;
    if (qbias && qbias_ld == 0) qbias_ld = 1;
    MSM_REQUIRE(embed_ld >= C && embed_ld % 4 == 0 && (((uintptr_t)mask_embed) & 15) == 0, "%s: mask_embed row stride %lld must be >= C, a multiple of 4, base 16-byte aligned",
                who, (long long)embed_ld);
    MSM_REQUIRE(!qbias || qbias_ld >= 1, "%s: bad qbias stride", who);
    return MSM_OK;
}

extern "C" int msm_mask_logits_fwd(const float* mask_embed, const float* mask_feat, float* mask_out,
                                   uint8_t* attn_out, int32_t* row_any, int B, int Q, int C, int H, int W,
                                   int th, int tw, int flags, int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream) {
    const int sparse = flags & MSM_MASK_SPARSE;
    MSM_REQUIRE(mask_embed && mask_feat, "msm_mask_logits_fwd: null input");
    MSM_REQUIRE(mask_out || attn_out, "msm_mask_logits_fwd: nothing to produce");
    MSM_REQUIRE(B > 0 && Q > 0 && H > 1 && W > 1, "msm_mask_logits_fwd: bad sizes");
    MSM_REQUIRE(C % 32 == 0 && C >= 32 && C <= 320, "msm_mask_logits_fwd: C=%d must be a multiple of 32 and <= 320", C);
    if (int rc = mask_embed_check("msm_mask_logits_fwd", mask_embed, embed_ld, qbias, qbias_ld, C)) return rc;
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
        MSM_REQUIRE(pool == 1 || pool == 2 || pool == 4 || pool == 8, "msm_mask_logits_fwd: pool factor %d not in {1,2,4,8}", pool);
    }
    hipStream_t st = (hipStream_t)stream;
    if (attn_out && !(flags & MSM_MASK_ROW_ANY_CLEARED)) MSM_CHECK_HIP(hipMemsetAsync(row_any, 0, sizeof(int32_t) * (size_t)B * Q, st));

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
    const int qchunks = cdiv(Q, QCH);
    // tile width: 2 x 32 (8-byte loads) or 2 x 16 (4-byte loads).  One tile keeps a SIMD busy for C/4*28 (or
    // *14) MFMAs; with T tiles over the 1024 SIMDs the makespan is ceil(T/1024) tile times, so take the
    // narrow tile when it shortens the makespan by more than the cost of the narrower loads.
    const int64_t t32 = (int64_t)n_rowpairs * cdiv(W, 32) * B * qchunks;
    const int64_t t16 = (int64_t)n_rowpairs * cdiv(W, 16) * B * qchunks;
    const double cost32 = (double)cdiv(t32, 1024), cost16 = 0.5 * (double)cdiv(t16, 1024);
    int nc = (cost16 * 1.04 < cost32) ? 1 : 2;
    if (const int o = opt(MSM_OPT_MASK_NC); o != MSM_OPT_AUTO) nc = o == 1 ? 1 : 2;
    const int ctiles = cdiv(W, 16 * nc);
    const int ntiles = n_rowpairs * ctiles;
    // persistent-ish grid: enough workgroups per (image, chunk) to cover the chip once
    int wg_per = cdiv(ntiles, MW);
    const int target = cdiv(256, B * qchunks);
    if (wg_per > target) wg_per = max(target, 1);
    dim3 grid(wg_per, qchunks, B), block(MW * 64);
    const size_t lds = sizeof(float) * ((size_t)QCH * (C + 2) + 2 * QCH);
    typedef void (*kern_t)(const float*, const float*, float*, uint8_t*, int32_t*, int, int, int, int, int, int, int, int, int, int, int, int64_t,
                           const float*, int64_t);
    kern_t kern;
    const bool wr = mask_out != nullptr;
#define MASK_PICK(P)                                                                                   \
    (nc == 2 ? (wr ? (kern_t)mask_logits_kernel<P, true, 2> : (kern_t)mask_logits_kernel<P, false, 2>)   \
             : (wr ? (kern_t)mask_logits_kernel<P, true, 1> : (kern_t)mask_logits_kernel<P, false, 1>))
    switch (pool) {
        case 0: kern = nc == 2 ? (kern_t)mask_logits_kernel<0, true, 2> : (kern_t)mask_logits_kernel<0, true, 1>; break;
        case 1: kern = MASK_PICK(1); break;
        case 2: kern = MASK_PICK(2); break;
        case 4: kern = MASK_PICK(4); break;
        default: kern = MASK_PICK(8); break;
    }
#undef MASK_PICK
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, mask_embed, mask_feat, mask_out, attn_out, row_any, Q, C, H, W, th, tw,
                       ypar, n_rowpairs, rp_step, rp_first, (int)((int64_t)C * H * W * 4), embed_ld, qbias, qbias_ld);
    MSM_CHECK_LAUNCH("msm_mask_logits_fwd");
    return MSM_OK;
}

extern "C" int msm_pack_mask_features_bf16(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream) {
    MSM_REQUIRE(mask_feat && packed, "msm_pack_mask_features_bf16: null pointer");
    MSM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "msm_pack_mask_features_bf16: C=%d must be a multiple of 4", C);
    MSM_REQUIRE((((uintptr_t)packed) & 7) == 0, "msm_pack_mask_features_bf16: packed must be 8-byte aligned");
    const int64_t total = (int64_t)B * (C / 4) * HW;
    hipLaunchKernelGGL(pack_mask_features_bf16_kernel, dim3((unsigned)min((int64_t)4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, mask_feat, packed, total, C / 4, HW);
    MSM_CHECK_LAUNCH("msm_pack_mask_features_bf16");
    return MSM_OK;
}

extern "C" int msm_mask_logits_bf16_fwd(const float* mask_embed, const uint16_t* mask_feat_packed, float* mask_out,
                                        uint8_t* attn_out, int32_t* row_any, int B, int Q, int C, int H, int W, int th, int tw,
                                        int flags, int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream) {
    const int sparse = flags & MSM_MASK_SPARSE;
    MSM_REQUIRE(mask_embed && mask_feat_packed, "msm_mask_logits_bf16_fwd: null input");
    MSM_REQUIRE(mask_out || attn_out, "msm_mask_logits_bf16_fwd: nothing to produce");
    MSM_REQUIRE(B > 0 && Q > 0 && H > 1 && W > 1, "msm_mask_logits_bf16_fwd: bad sizes");
    MSM_REQUIRE(C % 16 == 0 && C >= 16 && C <= 16 * BKS, "msm_mask_logits_bf16_fwd: C=%d must be a multiple of 16 and <= %d", C, 16 * BKS);
    if (int rc = mask_embed_check("msm_mask_logits_bf16_fwd", mask_embed, embed_ld, qbias, qbias_ld, C)) return rc;
    MSM_REQUIRE(W % 2 == 0 && H % 2 == 0, "msm_mask_logits_bf16_fwd: H=%d W=%d must be even", H, W);
    MSM_REQUIRE((int64_t)C * H * W * 2 < (int64_t)1 << 31, "msm_mask_logits_bf16_fwd: one image of mask_feat must be < 2 GiB");
    MSM_REQUIRE((((uintptr_t)mask_embed) & 15) == 0 && (((uintptr_t)mask_feat_packed) & 7) == 0, "msm_mask_logits_bf16_fwd: misaligned pointer");
    int pool = 0;
    if (attn_out) {
        MSM_REQUIRE(row_any, "msm_mask_logits_bf16_fwd: row_any required with attn_out");
        MSM_REQUIRE(th > 0 && tw > 0 && H % th == 0 && W % tw == 0 && H / th == W / tw,
                    "msm_mask_logits_bf16_fwd: target %dx%d incompatible with %dx%d", th, tw, H, W);
        pool = H / th;
        MSM_REQUIRE(pool == 1 || pool == 2 || pool == 4 || pool == 8, "msm_mask_logits_bf16_fwd: pool factor %d not in {1,2,4,8}", pool);
    }
    hipStream_t st = (hipStream_t)stream;
    if (attn_out && !(flags & MSM_MASK_ROW_ANY_CLEARED)) MSM_CHECK_HIP(hipMemsetAsync(row_any, 0, sizeof(int32_t) * (size_t)B * Q, st));
    int ypar = 0, n_rowpairs = H / 2, rp_step = 1, rp_first = 0;      // row pairing exactly as msm_mask_logits_fwd
    if (pool == 4 || pool == 8) {
        ypar = -1;
        n_rowpairs = H / 2 + 1;
        if (sparse && !mask_out) {
            rp_step = pool / 2;
            rp_first = pool / 4;
            n_rowpairs = H / pool;
        }
    }
    const int qchunks = cdiv(Q, QCH);
    const int ntiles = n_rowpairs * cdiv(W, 16);
    // one workgroup per CU: with more, re-staging mask_embed (100 KB per workgroup) costs more than the extra loads in
    // flight gain (measured 30 us at 256 workgroups, 41 us at 512, 46 us at 1024)
    int wg_per = cdiv(ntiles, MW);
    const int tgt_total = opt(MSM_OPT_MASKB_TARGET) > 0 ? opt(MSM_OPT_MASKB_TARGET) : 256;
    const int target = cdiv(tgt_total, B * qchunks);
    if (wg_per > target) wg_per = max(target, 1);
    dim3 grid(wg_per, qchunks, B), block(MW * 64);
    const size_t lds = sizeof(unsigned short) * (size_t)QCH * (C + 8) + sizeof(float) * 2 * QCH;
    typedef void (*kern_t)(const float*, const unsigned short*, float*, uint8_t*, int32_t*, int, int, int, int, int, int, int, int, int, int, int,
                           int64_t, const float*, int64_t);
    const bool wr = mask_out != nullptr;
    kern_t kern;
#define MASKB_PICK(P) (wr ? (kern_t)mask_logits_bf16_kernel<P, true> : (kern_t)mask_logits_bf16_kernel<P, false>)
    switch (pool) {
        case 0: kern = (kern_t)mask_logits_bf16_kernel<0, true>; break;
        case 1: kern = MASKB_PICK(1); break;
        case 2: kern = MASKB_PICK(2); break;
        case 4: kern = MASKB_PICK(4); break;
        default: kern = MASKB_PICK(8); break;
    }
#undef MASKB_PICK
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, mask_embed, mask_feat_packed, mask_out, attn_out, row_any, Q, C, H, W, th, tw, ypar,
                       n_rowpairs, rp_step, rp_first, (int)((int64_t)C * H * W * 2), embed_ld, qbias, qbias_ld);
    MSM_CHECK_LAUNCH("msm_mask_logits_bf16_fwd");
    return MSM_OK;
}
