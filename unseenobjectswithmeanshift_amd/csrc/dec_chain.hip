// Fused row-local tails of a decoder layer (see include/msm_hip.h: msm_dec_post_cross / msm_dec_post_self /
// msm_dec_heads).
//
// Reference: between two attention calls every op of a decoder layer is row-local on the (B*Q, 256) query
// matrix -- out_proj, residual + LayerNorm (forward_post, DEC:245-260 / DEC:171-181), the in-projections of
// the next attention (AU:134-140), the FFN (DEC:296-300), the block norm (DEC:637-638) and the prediction
// heads (DEC:660-668).  As separate launches that is 13 kernels per layer of ~6 us each on 800 rows: pure
// launch/latency cost (0.1 GFLOP per GEMM).  Here a workgroup owns a 16-row tile, keeps it in LDS across the
// whole chain and streams the (L2-resident) weights past it with v_mfma_f32_16x16x4_f32:
//
//   post_cross : x = LN(res + o Wo^T + bo);  [q|k] = (x + query_pos) Wqk^T + bqk;  v = x Wv^T + bv
//   post_self  : x = LN(res + o Wo^T + bo);  parts[c] = relu(x W1[c]^T + b1[c]) W2[:, c]^T   (hidden split in
//                256-wide chunks over blockIdx.y, so the 4 MB of FFN weights are spread over 8x more CUs)
//   heads      : out = normalize(LN(x + sum_c parts[c] + b2));  d = LN_dec(out);  e = MLP3(d);
//                q_next = (out + query_pos) Wq^T + bq
//
// MFMA operand mapping (K-order freedom): in k-chunk kc, step u, component c lane (lj = l & 15, lq = l >> 4)
// carries k = kc*64 + u*16 + lq*4 + c for BOTH operands, so
//   A: one ds_read_b128 of tile[lj][kc*64 + u*16 + lq*4 ..+3]   (row stride 260 floats: conflict-free)
//   B: one 16-byte global load per lane from the PACKED weight (msm_dec_pack_weight):
//        packed[((t*(K/64) + kc)*4 + u)*256 + lane*4 + c] = W[t*16 + lj][kc*64 + u*16 + lq*4 + c]
//      so a wave load is 1 KiB contiguous.  (Reading torch's (out, in) layout directly makes the 16 lanes of a
//      quarter-wave hit 16 different rows = 64 cache-line lookups per load; measured 9 us per 256x256 stage,
//      L1-address-rate bound, against 1.7 us of MFMA time.)
// Weight fragments are pipelined across the stages of a chain (see gemm256).
//
// Round 4, fp32 chain: 8-ROW tiles on v_mfma_f32_4x4x1_16b_f32.  A 16-row stage is bound by the matrix pipe of the ONE CU that owns
// the tile (256 MFMAs of 32 cycles per SIMD = 3.9 us at the 2.1 GHz the part holds) and the stages of a chain are serial: the
// decoder's 800 rows are 50 such tiles on a 256-CU chip.  The 4x4x1 instruction multiplies sixteen independent 4 x 4 x 1 blocks in 8
// cycles -- the same FLOP rate -- and with the fragment layout ABOVE a lane's block is (k-slice lq, column group lj >> 2): the
// B operand is the packed weight exactly as the 16x16x4 form reads it (same msm_dec_pack_weight format, same loads), the A operand
// of block (lq, .) is x[row i = lj & 3][k of slice lq], and the instruction is issued twice, for rows 0-3 and 4-7.  The four
// k-slices of a column leave partial sums in four lanes 16 apart: three cross-lane adds per accumulator tuple at the end of a stage
// (a reduce-scatter: lane lq ends with row 4 rg + lq).  A stage is then 256 MFMAs of 8 cycles per wave, 2 us per SIMD -- balanced
// against the 64 B/clk at which a CU can stream the 256 KiB of a stage's weights from L2 -- on twice as many tiles (100 x parts).
// The bf16 chain keeps 16-row tiles (its stage is weight-streaming bound already).
#include <stdlib.h>

#include <type_traits>

#include "bf16.h"
#include "common.h"

#ifndef DC_IL
#define DC_IL 4   // MFMAs between two prefetch loads (8*DC_NT loads and 32*DC_NT MFMAs per half stage)
#endif
#ifndef DC_EXP
#define DC_EXP 0   // tuning experiments only: 1 = no MFMAs, 2 = no weight loads (tools/microbench.py tails)
#endif

namespace msm {

constexpr int DC_E = 256;
// Tile kinds: weight type, rows per workgroup, LDS row stride (floats), row groups of 4 per column tile.
//   TileF16  fp32 weights, 16 rows on v_mfma_f32_16x16x4_f32; stride 260 (conflict-free b128 reads of 16 rows)
//   TileF8   fp32 weights,  8 rows on v_mfma_f32_4x4x1_16b_f32 (header); stride 272 floats = 68 slots of 16 bytes = 4 (mod 16): the
//            sixteen distinct float4 a wave's A read touches (row i, k-slice lq) sit in sixteen different slots
//   TileH16  bf16 weights, 16 rows on v_mfma_f32_16x16x16_bf16
// Which fp32 kind a launch takes is decided by the host per kernel and row count: 8-row tiles halve a stage but double the number
// of workgroups that stream the stage's 256 KiB of weights -- they pay while tiles x parts still fit the chip in one round
// (measured at 800 rows: heads 21.7 -> 14.5 us with 200 workgroups; post_cross 14.6 -> 18.4 with 300, post_self 27.7 -> 51 with 800).
//   TileQ32  fp16 weights, 32 rows = TWO 16-row MFMA tiles per workgroup that share every weight fragment (round 6): above ~4000 rows
//            (the second stage of configs[3]: 17 300 rows, configs[4] at batch 4) a launch is bound by the L2 -> CU weight stream --
//            1082 16-row tiles x 2.2 MB = 2.4 GB per post_self launch at 33 TB/s -- and a fragment that feeds two MFMAs halves it
// F8: the 4x4x1 fp32 form (its two "row groups" are the halves of ONE 8-row tile, reduced across lanes); RG otherwise counts 16-row tiles.
struct TileF16 {
    using WT = float;
    static constexpr int R = 16, LD = DC_E + 4, RG = 1;
    static constexpr bool F8 = false;
    static constexpr int KP = 1;
};
struct TileF8 {
    using WT = float;
    static constexpr int R = 8, LD = DC_E + 16, RG = 2;
    static constexpr bool F8 = true;
    static constexpr int KP = 1;
};
struct TileH16 {
    using WT = uint16_t;
    static constexpr int R = 16, LD = DC_E + 4, RG = 1;
    static constexpr bool F8 = false;
    static constexpr int KP = 1;
};
// fp16 weights (precision "f16"): the same bytes as bf16 with 11 instead of 8 significand bits -- Linear weights are O(0.01 .. 1),
// far inside the half range -- on v_mfma_f32_16x16x32_f16, which issues at the bf16 instruction's rate.  The activation fragment
// enters as ONE fp16 term (clamped to the half range): its rounding, 2^-12, is of the order of the weight's, so the hi + lo pair
// the bf16 form needs (to keep the activation's 2^-9 out of the product) would buy nothing.  Half the MFMAs of the bf16 form, an
// eighth of its rounding error: measured over 3200 masks (tools/probes/bf16_pooled_probe.py) the tails alone flip 0.85 % of the
// final mask bits with bf16 weights and 0.2 % with fp16 weights; 28.4 us per layer against 32.2 (B = 8).
struct f16w {
    uint16_t v;
};
struct TileQ16 {
    using WT = f16w;
    static constexpr int R = 16, LD = DC_E + 4, RG = 1;
    static constexpr bool F8 = false;
    static constexpr int KP = 1;
};
// bf16 weights as hi + lo fragments (round 6, the "bf16" plan's tails): the packed matrix is [bf16(W) | bf16(W - bf16(W))] along K (msm_dec_pack_weight_bf16x2),
// a stage walks the hi chunks, then the lo chunks, into the same accumulators -- with the activation's own hi + lo split every product keeps 2^-17:
// the single-bf16 weights were 0.85 % of the plan's 1.08 % of flipped mask bits; twice the weight stream, twice the MFMAs.
struct bf16hl {
    uint16_t v;
};
struct TileH16x2 {
    using WT = bf16hl;
    static constexpr int R = 16, LD = DC_E + 4, RG = 1;
    static constexpr bool F8 = false;
    static constexpr int KP = 2;
};
struct TileQ32 {
    using WT = f16w;
    static constexpr int R = 32, LD = DC_E + 4, RG = 2;
    static constexpr bool F8 = false;
    static constexpr int KP = 1;
};
// (64-row tiles -- RG = 4, one 133-KB workgroup per CU -- were measured for post_self at 17 300 rows: 126 us against 90 for 32-row tiles and
// 136 for 16-row tiles: one resident workgroup cannot keep the weight stream's latency covered.)
#ifndef MSM_DC_NW
#define MSM_DC_NW 8
#endif
constexpr int DC_NW = MSM_DC_NW;    // waves per workgroup
constexpr int DC_NT = 16 / DC_NW;   // 16-column tiles per wave in a 256-column GEMM
constexpr int DC_THREADS = DC_NW * 64;

// B fragments of two 64-wide k-chunks of a 256-column weight block, in MFMA operand order (see the header).
// WT = float: the fp32 chain.  WT = uint16_t: bf16 weights (msm_dec_pack_weight_bf16) for the low-precision mode -- a lane's
// 16 weights of a (column tile, k-chunk) are two 16-byte loads {u = 2 h: c 0..3, u = 2 h + 1: c 0..3}, the activation
// fragment read from LDS is split into hi + lo bf16 operands at the moment it is used (x = hi + lo up to 2^-17 |x|), and
// v_mfma_f32_16x16x16_bf16 takes the 16 k of (k-chunk, u) at once: 32 MFMAs of 8 cycles per half stage instead of 64 of
// 32, half the weight bytes -- a stage is bound by streaming 128 KiB of weights per workgroup, not by the matrix pipe.
template <typename WT>
struct BFrag;
template <>
struct BFrag<float> {
    float4 v[2][DC_NT][4];
};
template <>
struct BFrag<uint16_t> {
    u32x4b v[2][DC_NT][2];
};
template <>
struct BFrag<f16w> : BFrag<uint16_t> {};
template <>
struct BFrag<bf16hl> : BFrag<uint16_t> {};
// W: packed weight, advanced to the first of the 256 output rows wanted (row offset n0 -> + n0*K elements);
// kct = K/64 of the packed matrix; kc0 = first of the two k-chunks to fetch
// (Rotating the k-chunk order per workgroup to de-phase their L2 accesses was measured: -3 % kernel time, and it
// makes a row's rounding depend on its batch position; not kept.)
__device__ __forceinline__ void bload(BFrag<float>& f, const float* __restrict__ W, int kct, int kc_base, int half) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < DC_NT; ++t) {
        const float* wp = W + ((int64_t)(wave * DC_NT + t) * kct + kc_base) * 1024 + lane * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kc = half * 2 + h;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#if DC_EXP == 2
                f.v[h][t][u] = make_float4((float)lane, 1.f, 2.f, (float)kc);
#else
                f.v[h][t][u] = *reinterpret_cast<const float4*>(wp + (kc * 4 + u) * 256);
#endif
            }
        }
    }
}
__device__ __forceinline__ void bload(BFrag<uint16_t>& f, const uint16_t* __restrict__ W, int kct, int kc_base, int half) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < DC_NT; ++t) {
        const uint16_t* wp = W + ((int64_t)(wave * DC_NT + t) * kct + kc_base) * 1024 + lane * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kc = half * 2 + h;
#pragma unroll
            for (int up = 0; up < 2; ++up) {
#if DC_EXP == 2
                f.v[h][t][up] = u32x4b{(unsigned)lane, 1u, 2u, (unsigned)kc};
#else
                f.v[h][t][up] = *reinterpret_cast<const u32x4b*>(wp + (kc * 2 + up) * 512);
#endif
            }
        }
    }
}
// fp32, 16-row tiles: ap = A + (lane & 15) * LD + (lane >> 4) * 4
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][1], const float* __restrict__ ap, const BFrag<float>& f, int half) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 a[4];
        const int kc = half * 2 + h;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(ap + kc * 64 + u * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#if DC_EXP == 1
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0][0] += a[u].x * f.v[h][t][u].x + a[u].y * f.v[h][t][u].y + a[u].z * f.v[h][t][u].z + a[u].w * f.v[h][t][u].w;
            continue;
#endif
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma16(a[u].x, f.v[h][t][u].x, acc[t][0]);
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma16(a[u].y, f.v[h][t][u].y, acc[t][0]);
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma16(a[u].z, f.v[h][t][u].z, acc[t][0]);
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma16(a[u].w, f.v[h][t][u].w, acc[t][0]);
        }
    }
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// fp32: ap = A + (lane & 3) * LD + (lane >> 4) * 4 (row i of the lane's block, k-slice lq); acc[t][rg] = the 4 x 4 block (rows 4 rg ..
// + 3, column t*16 + lj) of the lane's k-slice
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][2], const float* __restrict__ ap, const BFrag<float>& f, int half) {
    constexpr int LD = TileF8::LD;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 a0[4], a1[4];
        const int kc = half * 2 + h;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0[u] = *reinterpret_cast<const float4*>(ap + kc * 64 + u * 16);
            a1[u] = *reinterpret_cast<const float4*>(ap + 4 * LD + kc * 64 + u * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#if DC_EXP == 1
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0][0] += a0[u].x * f.v[h][t][u].x + a1[u].y * f.v[h][t][u].y + a0[u].z * f.v[h][t][u].z + a1[u].w * f.v[h][t][u].w;
            continue;
#endif
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) { acc[t][0] = mfma4(a0[u].x, f.v[h][t][u].x, acc[t][0]); acc[t][1] = mfma4(a1[u].x, f.v[h][t][u].x, acc[t][1]); }
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) { acc[t][0] = mfma4(a0[u].y, f.v[h][t][u].y, acc[t][0]); acc[t][1] = mfma4(a1[u].y, f.v[h][t][u].y, acc[t][1]); }
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) { acc[t][0] = mfma4(a0[u].z, f.v[h][t][u].z, acc[t][0]); acc[t][1] = mfma4(a1[u].z, f.v[h][t][u].z, acc[t][1]); }
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) { acc[t][0] = mfma4(a0[u].w, f.v[h][t][u].w, acc[t][0]); acc[t][1] = mfma4(a1[u].w, f.v[h][t][u].w, acc[t][1]); }
        }
    }
}
// bf16 weights: v_mfma_f32_16x16x32_bf16 on TWO k-steps at once (round 4; the K = 16 instruction issues at half the rate and this loop
// had 64 of them per stage in dependent lo -> hi pairs).  A lane's eight k of a pair (u, u + 1) are its four of u and its four of
// u + 1 on both operands -- the packed fragment f.v[h][t][u >> 1] already holds exactly those eight --, so the pair is one MFMA per
// term: 32 per stage, consecutive ones on different accumulators.
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][1], const float* __restrict__ ap, const BFrag<uint16_t>& f, int half) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 a[4];
        const int kc = half * 2 + h;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(ap + kc * 64 + u * 16);
#pragma unroll
        for (int up = 0; up < 2; ++up) {
            const Split4 x0 = split4(a[2 * up].x, a[2 * up].y, a[2 * up].z, a[2 * up].w);
            const Split4 x1 = split4(a[2 * up + 1].x, a[2 * up + 1].y, a[2 * up + 1].z, a[2 * up + 1].w);
            const bf16x8 xh = cat8(x0.hi, x1.hi), xl = cat8(x0.lo, x1.lo);
#if DC_EXP == 1
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) {
                const u32x4b xa = __builtin_bit_cast(u32x4b, xh), xb = __builtin_bit_cast(u32x4b, xl);
                acc[t][0][0] += __uint_as_float(((xa.x ^ xb.y ^ f.v[h][t][up].x) & 0x3fffffffu) | (xa.z ^ xb.w ^ f.v[h][t][up].z) >> 8);
            }
            continue;
#endif
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma_bf16k32(xl, __builtin_bit_cast(bf16x8, f.v[h][t][up]), acc[t][0]);
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma_bf16k32(xh, __builtin_bit_cast(bf16x8, f.v[h][t][up]), acc[t][0]);
        }
    }
}
// bf16 hi + lo weights: the bf16 form's loads and MFMAs on either half of the packed K
__device__ __forceinline__ void bload(BFrag<bf16hl>& f, const bf16hl* __restrict__ W, int kct, int kc_base, int half) {
    bload(static_cast<BFrag<uint16_t>&>(f), reinterpret_cast<const uint16_t*>(W), kct, kc_base, half);
}
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][1], const float* __restrict__ ap, const BFrag<bf16hl>& f, int half) {
    mfma_half(acc, ap, static_cast<const BFrag<uint16_t>&>(f), half);
}
// fp16 weights: the same fragment layout and loads as bf16; one MFMA per pair of k-steps and column tile
__device__ __forceinline__ void bload(BFrag<f16w>& f, const f16w* __restrict__ W, int kct, int kc_base, int half) {
    bload(static_cast<BFrag<uint16_t>&>(f), reinterpret_cast<const uint16_t*>(W), kct, kc_base, half);
}
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][1], const float* __restrict__ ap, const BFrag<f16w>& f, int half) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 a[4];
        const int kc = half * 2 + h;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(ap + kc * 64 + u * 16);
#pragma unroll
        for (int up = 0; up < 2; ++up) {
            const float4 p = a[2 * up], q = a[2 * up + 1];
            const f16x8 x = cvt8h(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w);
#pragma unroll
            for (int t = 0; t < DC_NT; ++t) acc[t][0] = mfma_f16k32(x, __builtin_bit_cast(f16x8, f.v[h][t][up]), acc[t][0]);
        }
    }
}
// fp16 weights, RT 16-row tiles per workgroup (TileQ32: RT = 2): every weight fragment feeds all tiles' MFMAs
template <int RT>
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[DC_NT][RT], const float* __restrict__ ap, const BFrag<f16w>& f, int half) {
    constexpr int LD = TileQ32::LD;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kc = half * 2 + h;
#pragma unroll
        for (int up = 0; up < 2; ++up) {
            f16x8 x[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 p = *reinterpret_cast<const float4*>(ap + rt * 16 * LD + kc * 64 + (2 * up) * 16);
                const float4 q = *reinterpret_cast<const float4*>(ap + rt * 16 * LD + kc * 64 + (2 * up + 1) * 16);
                x[rt] = cvt8h(p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w);
            }
#pragma unroll
            for (int t = 0; t < DC_NT; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[t][rt] = mfma_f16k32(x[rt], __builtin_bit_cast(f16x8, f.v[h][t][up]), acc[t][rt]);
        }
    }
}
// prefetch loads per half stage (bload) and MFMAs between two of them
template <typename TK>
struct Pipe {
    static constexpr int LOADS = std::is_same<typename TK::WT, float>::value ? 8 * DC_NT : 4 * DC_NT;
    // MFMAs between two prefetch loads: 8-row fp32 has two 8-cycle MFMAs where the 16-row form has one of 32; bf16 (hi + lo activation) has
    // 8 * DC_NT per half stage, fp16 (one term) 4 * DC_NT
    static constexpr int IL = std::is_same<typename TK::WT, f16w>::value ? TK::RG
                              : !std::is_same<typename TK::WT, float>::value ? DC_IL / 2 : (TK::F8 ? 2 * DC_IL : DC_IL);      // (bf16hl: as bf16)
};

// D[16][256] = act(A[16][256] . W[n][k]^T + bias).  A in LDS.  TO_GLOBAL: D is row-major global with row stride
// ldd, rows >= rows_valid are not written; otherwise D is an LDS tile (stride TK::LD).
// Weight pipeline across the stages of a chain: on entry `lo` already holds k-chunks 0,1 of W (loaded during the
// previous stage or at kernel start); chunks 2,3 are fetched while 0,1 are consumed, and the next stage's chunks
// 0,1 (Wn, may be null) while 2,3 are consumed, so L2 latency hides behind 64 MFMAs (~2000 cycles) each time.
// acc += A[16][256] . W-block^T for this wave's DC_NT column tiles (fragment pipeline as described above)
template <bool NEXT, typename TK>
__device__ __forceinline__ void gemm_core(f32x4 (&acc)[DC_NT][TK::RG], const float* __restrict__ A, const typename TK::WT* __restrict__ W, int kct,
                                          int kc_base, BFrag<typename TK::WT>& lo, const typename TK::WT* __restrict__ Wn, int kctn, int kcn) {
    using WT = typename TK::WT;
    const int lane = threadIdx.x & 63;
    const float* ap = A + (TK::F8 ? (lane & 3) : (lane & 15)) * TK::LD + (lane >> 4) * 4;
    BFrag<WT> hi;
    // The prefetch loads are spread evenly between the MFMAs of the half they hide behind (1 load : DC_IL MFMAs,
    // sched_group_barrier), not issued as a burst in front of them: measured 16.2 -> 14.1 us (post_cross), 34.1 -> 29.1
    // (post_self), 23.1 -> 21.2 (heads).
    __builtin_amdgcn_sched_barrier(0);
    bload(hi, W, kct, kc_base, 1);
    mfma_half(acc, ap, lo, 0);
#pragma unroll
    for (int i = 0; i < Pipe<TK>::LOADS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // one VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, Pipe<TK>::IL, 0);    // DC_IL MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    // NEXT is a compile-time flag: a run-time branch here makes the waitcnt pass assume the shorter queue and
    // wait for the prefetch itself before the last MFMAs
    if constexpr (NEXT) bload(lo, Wn, kctn, kcn, 0);
    mfma_half(acc, ap, hi, 1);
#pragma unroll
    for (int i = 0; i < Pipe<TK>::LOADS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, Pipe<TK>::IL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// gemm_core over all of a stage's k: one pass, or (TK::KP == 2: hi + lo weight fragments, packed K' = 2 K) the hi chunks [kc_base, + 4) and then
// the lo chunks [kct + kc_base, + 4) of the same rows.  kct / kctn count the 64-wide chunks of the ORIGINAL K.
template <bool NEXT, typename TK>
__device__ __forceinline__ void gemm_core_k(f32x4 (&acc)[DC_NT][TK::RG], const float* __restrict__ A, const typename TK::WT* __restrict__ W, int kct,
                                            int kc_base, BFrag<typename TK::WT>& lo, const typename TK::WT* __restrict__ Wn, int kctn, int kcn) {
    if constexpr (TK::KP == 1) {
        gemm_core<NEXT, TK>(acc, A, W, kct, kc_base, lo, Wn, kctn, kcn);
    } else {
        gemm_core<true, TK>(acc, A, W, 2 * kct, kc_base, lo, W, 2 * kct, kct + kc_base);
        gemm_core<NEXT, TK>(acc, A, W, 2 * kct, kct + kc_base, lo, Wn, 2 * kctn, kcn);
    }
}

// D = act(acc + bias).  TO_GLOBAL: D is row-major global with row stride ldd, rows >= rows_valid are not written;
// otherwise D is an LDS tile (stride TK::LD).
template <bool TO_GLOBAL, int LD, int RG>
__device__ __forceinline__ void gemm_store16(const f32x4 (&acc)[DC_NT][RG], const float (&bv)[DC_NT], bool relu, float* __restrict__ D,
                                             int64_t ldd, int rows_valid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int t = 0; t < DC_NT; ++t) {
        const int n = (wave * DC_NT + t) * 16 + lj;
#pragma unroll
        for (int rt = 0; rt < RG; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][rt][r] + bv[t];
                if (relu) v = fmaxf(v, 0.f);
                const int row = rt * 16 + lq * 4 + r;
                if constexpr (TO_GLOBAL) {
                    if (row < rows_valid) D[(int64_t)row * ldd + n] = v;
                } else {
                    D[row * LD + n] = v;
                }
            }
    }
}
// fp32 (4x4x1 blocks): the four k-slices of a column sit in the lanes lq = 0..3 of that column; a reduce-scatter over them leaves lane
// lq with the total of row 4 rg + lq (three cross-lane adds per tuple), which it stores
template <bool TO_GLOBAL, int LD>
__device__ __forceinline__ void gemm_store8(const f32x4 (&acc)[DC_NT][2], const float (&bv)[DC_NT], bool relu, float* __restrict__ D,
                                            int64_t ldd, int rows_valid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    const bool up2 = lq & 2, up1 = lq & 1;
#pragma unroll
    for (int t = 0; t < DC_NT; ++t) {
        const int n = (wave * DC_NT + t) * 16 + lj;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const f32x4 d = acc[t][rg];
            float k0 = up2 ? d[2] : d[0], k1 = up2 ? d[3] : d[1];
            k0 += __shfl_xor(up2 ? d[0] : d[2], 32, 64);
            k1 += __shfl_xor(up2 ? d[1] : d[3], 32, 64);
            float v = (up1 ? k1 : k0) + __shfl_xor(up1 ? k0 : k1, 16, 64) + bv[t];
            if (relu) v = fmaxf(v, 0.f);
            const int row = rg * 4 + lq;
            if constexpr (TO_GLOBAL) {
                if (row < rows_valid) D[(int64_t)row * ldd + n] = v;
            } else {
                D[row * LD + n] = v;
            }
        }
    }
}

template <bool TO_GLOBAL, typename TK>
__device__ __forceinline__ void gemm_store(const f32x4 (&acc)[DC_NT][TK::RG], const float (&bv)[DC_NT], bool relu, float* __restrict__ D,
                                           int64_t ldd, int rows_valid) {
    if constexpr (TK::F8) gemm_store8<TO_GLOBAL, TK::LD>(acc, bv, relu, D, ldd, rows_valid);
    else gemm_store16<TO_GLOBAL, TK::LD, TK::RG>(acc, bv, relu, D, ldd, rows_valid);
}

__device__ __forceinline__ void load_bias(float (&bv)[DC_NT], const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < DC_NT; ++t) bv[t] = bias ? bias[(wave * DC_NT + t) * 16 + (lane & 15)] : 0.f;
}

// D[16][256] = act(A[16][256] . W-block^T + bias): one stage of a chain.  On entry `lo` holds k-chunks 0,1 of W
// (fetched during the previous stage or at kernel start); chunks 2,3 are fetched while 0,1 are consumed and the next
// stage's first two chunks (Wn) while 2,3 are consumed, so L2 latency hides behind 64 MFMAs per wave each time.
template <bool TO_GLOBAL, bool NEXT, typename TK>
__device__ __forceinline__ void gemm256(const float* __restrict__ A, const typename TK::WT* __restrict__ W, int kct, int kc_base,
                                        const float* __restrict__ bias, bool relu, float* __restrict__ D, int64_t ldd,
                                        int rows_valid, BFrag<typename TK::WT>& lo, const typename TK::WT* __restrict__ Wn, int kctn, int kcn) {
    f32x4 acc[DC_NT][TK::RG];
#pragma unroll
    for (int t = 0; t < DC_NT; ++t)
#pragma unroll
        for (int rg = 0; rg < TK::RG; ++rg) acc[t][rg] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bv[DC_NT];
    load_bias(bv, bias);            // requested before the MFMAs, consumed after them
    gemm_core_k<NEXT, TK>(acc, A, W, kct, kc_base, lo, Wn, kctn, kcn);
    gemm_store<TO_GLOBAL, TK>(acc, bv, relu, D, ldd, rows_valid);
}

// tile[16][256] <- src rows row0.. (clamped to the last valid row), 16-byte coalesced
template <typename TK>
__device__ __forceinline__ void load_tile(float* __restrict__ tile, const float* __restrict__ src, int64_t ld, int row0,
                                          int rows) {
    for (int i = threadIdx.x; i < TK::R * (DC_E / 4); i += DC_THREADS) {
        const int r = i / (DC_E / 4), c4 = i - r * (DC_E / 4);
        const int gr = min(row0 + r, rows - 1);
        *reinterpret_cast<float4*>(tile + r * TK::LD + c4 * 4) = *reinterpret_cast<const float4*>(src + (int64_t)gr * ld + c4 * 4);
    }
}

// ---- L2 prefetch rows (round 6) -------------------------------------------------------------------------------------------------
// A layer's tail weights are 3.2 MB of fp16 fragments (6.4 MB fp32) that nothing keeps in the 4-MiB per-XCD L2s between two passes: every
// tail starts on HBM latency (tools/probes/tails_cold_time.py, 800 rows: post_self 17.5 us cold, 13.4 with its weights resident; heads
// 15.2 / 13.1; post_cross 10.7 / 9.6).  A tail launch can therefore carry ONE EXTRA ROW of workgroups (the last blockIdx.y) that do no
// tail work (two rows when the ranges are long): they touch the byte ranges the NEXT launches of the chain will stream
// (msm_dec_set_prefetch) -- one 4-byte load per 128-byte line -- and exit.  They are dispatched behind the launch's working workgroups onto CUs the 50-200 of them leave idle, and their loads
// sit in their own waves' queues (a wave's loads return in order: the same loads issued by a working wave would stall its first wait).
// Each XCD's L2 needs its own copy: the row's workgroups on XCD x (block id % 8 == x: an observed placement, relied on for speed only)
// share every range between them.  A forked side stream for the same job costs ~10 us per fork / join inside a HIP graph (measured:
// 1.41 -> 1.68 ms per pass).
constexpr int DC_PF_MAX = 6;
struct PfRanges {
    const unsigned char* p[DC_PF_MAX];
    int64_t bytes[DC_PF_MAX];
    int n, rows;                                                       // rows: prefetch rows appended to the grid (0 = none)
};
// rows of gx workgroups so that a thread has about eight lines to touch (every XCD's share of the rows reads ALL the bytes), at most 2
// (1, 2 and 4 rows measured alike at 800 rows: 1.375 - 1.381 ms per f16 pass against 1.405 without)
static int prefetch_rows(const PfRanges& pf, int gx) {
    if (pf.n <= 0) return 0;
    int64_t lines = 0;
    for (int j = 0; j < pf.n; ++j) lines += (pf.bytes[j] + 127) >> 7;
    const int64_t want = (lines * 8 + (int64_t)gx * DC_THREADS * 8 - 1) / ((int64_t)gx * DC_THREADS * 8);
    return (int)max((int64_t)1, min((int64_t)2, want));
}
__device__ __forceinline__ bool prefetch_part(const PfRanges& pf) {
    if (pf.rows <= 0 || (int)blockIdx.y < (int)gridDim.y - pf.rows) return false;
    const int first = gridDim.x * (gridDim.y - pf.rows);               // linear id of the first prefetch workgroup
    const int total = gridDim.x * pf.rows;
    const int j = ((int)blockIdx.y - ((int)gridDim.y - pf.rows)) * gridDim.x + blockIdx.x, xcd = (first + j) & 7;
    const int j0 = (xcd - first) & 7;                                  // the first prefetch workgroup on this XCD
    const int k = (j - j0) >> 3;                                       // this workgroup's index among the prefetch workgroups on its XCD
    const int cnt = (total - j0 + 7) >> 3;                             // ... and their number
    unsigned acc = 0;
    for (int r = 0; r < pf.n; ++r) {
        const int64_t lines = (pf.bytes[r] + 127) >> 7;
        const int64_t per = (lines + cnt - 1) / cnt;
        const int64_t l0 = (int64_t)k * per, l1 = min(lines, l0 + per);
        const unsigned char* base = pf.p[r];
        // four loads in flight per thread (a trip's loads are independent; lines past the end re-read the last one)
        for (int64_t l = l0 + threadIdx.x; l < l1; l += 4 * DC_THREADS) {
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const unsigned*>(base + (min(l + u * DC_THREADS, l1 - 1) << 7));
            acc ^= (v[0] ^ v[1]) ^ (v[2] ^ v[3]);
        }
    }
    if (acc == 0x9e3779b9u) asm volatile("s_nop 0" ::: "memory");     // (keeps the loads alive)
    return true;
}

struct RowStats {
    float mean, rstd;
};
__device__ __forceinline__ RowStats row_stats(float4 v, float eps) {
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / DC_E);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / DC_E);
    return RowStats{mean, 1.0f / sqrtf(var + eps)};
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 affine4(float4 v, RowStats s, float4 g, float4 b) {
    return make_float4((v.x - s.mean) * s.rstd * g.x + b.x, (v.y - s.mean) * s.rstd * g.y + b.y,
                       (v.z - s.mean) * s.rstd * g.z + b.z, (v.w - s.mean) * s.rstd * g.w + b.w);
}
// rows per wave in the row-wise phases: TK::R / DC_NW; a lane owns columns 4*lane .. 4*lane+3

// ---- x = LN(res + o Wo^T + bo), shared head of post_cross / post_self -------------------------------------------
// On return (after the trailing barrier) X holds x and, if XP, XP holds x + query_pos; x rows are written to
// x_out by the workgroups with store_x.  Every global operand of the row phase is requested before the GEMM so
// its latency hides behind the MFMAs.
template <typename TK, typename WT = typename TK::WT>
__device__ __forceinline__ void attn_out_ln(const float* __restrict__ o, const float* __restrict__ res,
                                            const WT* __restrict__ wo, const float* __restrict__ bo,
                                            const float* __restrict__ g, const float* __restrict__ b,
                                            const float* __restrict__ qpos, int Q, float* __restrict__ x_out, bool store_x,
                                            float* __restrict__ T0, float* __restrict__ X, float* __restrict__ XP, int row0,
                                            int rows, float eps, BFrag<WT>& f, const WT* __restrict__ w_next, int kct_next,
                                            int kc_next) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bload(f, wo, 4 * TK::KP, 0, 0);
    load_tile<TK>(T0, o, DC_E, row0, rows);
    float4 rv[(TK::R / DC_NW)], pv[(TK::R / DC_NW)];
#pragma unroll
    for (int i = 0; i < (TK::R / DC_NW); ++i) {
        const int gr = min(row0 + wave * (TK::R / DC_NW) + i, rows - 1);
        rv[i] = ld4(res + (int64_t)gr * DC_E + lane * 4);
        pv[i] = XP ? ld4(qpos + (int64_t)(gr % Q) * DC_E + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 gv = ld4(g + lane * 4), bv = ld4(b + lane * 4);
    __syncthreads();
    gemm256<false, true, TK>(T0, wo, 4, 0, bo, false, X, 0, 0, f, w_next, kct_next, kc_next);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (TK::R / DC_NW); ++i) {
        const int r = wave * (TK::R / DC_NW) + i;
        const float4 v = add4(ld4(X + r * TK::LD + lane * 4), rv[i]);
        const float4 y = affine4(v, row_stats(v, eps), gv, bv);
        st4(X + r * TK::LD + lane * 4, y);
        if (XP) st4(XP + r * TK::LD + lane * 4, add4(y, pv[i]));
        if (store_x && row0 + r < rows) st4(x_out + (int64_t)(row0 + r) * DC_E + lane * 4, y);
    }
    __syncthreads();
}

template <typename TK, typename WT = typename TK::WT>
__global__ __launch_bounds__(DC_THREADS) void dec_post_cross_kernel(
    const float* __restrict__ o, const float* __restrict__ res, const float* __restrict__ qpos, const WT* __restrict__ wo,
    const float* __restrict__ bo, const float* __restrict__ g, const float* __restrict__ b, const WT* __restrict__ w_in,
    const float* __restrict__ b_in, float* __restrict__ x_out, float* __restrict__ qk_out, float* __restrict__ v_out,
    int rows, int Q, float eps, PfRanges pf) {
    __shared__ __attribute__((aligned(16))) float lds[3 * TK::R * TK::LD];
    float *T0 = lds, *X = lds + TK::R * TK::LD, *XP = lds + 2 * TK::R * TK::LD;
    if (prefetch_part(pf)) return;                                     // the prefetch rows (see PfRanges)
    const int row0 = blockIdx.x * TK::R;
    if (row0 >= rows) return;                                          // (grid.x is padded to a multiple of 8: see tile_grid_x)
    const int valid = min(TK::R, rows - row0);
    // blockIdx.y picks the projection (0: q, 1: k, 2: v): the 16-row tile is MFMA-bound on ONE CU at 3.4 us per
    // 256x256 stage, so the three independent projections go to three CUs; each repeats the Wo + LN stage.
    // gridDim.y == 2 (8-row tiles: 100 tiles x 3 parts would not fit the chip in one round): part 0 runs q THEN k, part 1 runs v.
    const int part = blockIdx.y;
    const bool two = (int)gridDim.y - pf.rows == 2;
    const int first = two ? (part == 0 ? 0 : 2) : part;        // the projection this workgroup starts with
    const WT* wp = w_in + (int64_t)first * DC_E * DC_E * TK::KP;
    BFrag<WT> f;
    attn_out_ln<TK>(o, res, wo, bo, g, b, qpos, Q, x_out, part == 0, T0, X, XP, row0, rows, eps, f, wp, 4, 0);
    // q and k share tgt + query_pos (DEC:171-175); v = tgt
    if (first == 2) {
        gemm256<true, false, TK>(X, wp, 4, 0, b_in + 2 * DC_E, false, v_out + (int64_t)row0 * DC_E, DC_E, valid, f, nullptr, 0, 0);
    } else if (two) {
        gemm256<true, true, TK>(XP, wp, 4, 0, b_in, false, qk_out + (int64_t)row0 * 2 * DC_E, 2 * DC_E, valid, f, wp + DC_E * DC_E * TK::KP, 4, 0);
        gemm256<true, false, TK>(XP, wp + DC_E * DC_E * TK::KP, 4, 0, b_in + DC_E, false, qk_out + (int64_t)row0 * 2 * DC_E + DC_E, 2 * DC_E, valid, f,
                                 nullptr, 0, 0);
    } else {
        gemm256<true, false, TK>(XP, wp, 4, 0, b_in + first * DC_E, false, qk_out + (int64_t)row0 * 2 * DC_E + first * DC_E, 2 * DC_E,
                                 valid, f, nullptr, 0, 0);
    }
}

template <typename TK, typename WT = typename TK::WT>
__global__ __launch_bounds__(DC_THREADS) void dec_post_self_kernel(
    const float* __restrict__ o, const float* __restrict__ res, const WT* __restrict__ wo, const float* __restrict__ bo,
    const float* __restrict__ g, const float* __restrict__ b, const WT* __restrict__ w1, const float* __restrict__ b1,
    const WT* __restrict__ w2, int F, float* __restrict__ x_out, float* __restrict__ parts, int rows, float eps, PfRanges pf) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TK::R * TK::LD];
    float *T0 = lds, *X = lds + TK::R * TK::LD;
    if (prefetch_part(pf)) return;                                     // the prefetch rows (see PfRanges)
    const int n_chunks_y = (int)gridDim.y - pf.rows;
    const int row0 = blockIdx.x * TK::R, chunk = blockIdx.y;
    if (row0 >= rows) return;                                          // (padding tile)
    const int valid = min(TK::R, rows - row0);
    // blockIdx.y owns F/256/gridDim.y consecutive 256-wide hidden chunks; their W2 products accumulate in registers.
    // linear1: rows [c*256, +256) of the packed (F, 256) matrix; linear2 (256, F): k-chunks 4c..4c+3 of every row tile.
    const int per = (F / DC_E) / n_chunks_y, c0 = chunk * per;
    const int kct2 = F / 64;
    BFrag<WT> f;
    attn_out_ln<TK>(o, res, wo, bo, g, b, nullptr, 1, x_out, chunk == 0, T0, X, nullptr, row0, rows, eps, f,
                    w1 + (int64_t)c0 * DC_E * DC_E * TK::KP, 4, 0);
    f32x4 acc2[DC_NT][TK::RG];
#pragma unroll
    for (int t = 0; t < DC_NT; ++t)
#pragma unroll
        for (int rg = 0; rg < TK::RG; ++rg) acc2[t][rg] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float zero_bias[DC_NT] = {};
    for (int c = c0; c < c0 + per; ++c) {
        // h = relu(x W1[c]^T + b1[c]) (DEC:297) -> T0;  acc2 += h W2[:, c]^T
        gemm256<false, true, TK>(X, w1 + (int64_t)c * DC_E * DC_E * TK::KP, 4, 0, b1 + c * DC_E, true, T0, 0, 0, f, w2, kct2, 4 * c);
        __syncthreads();
        const int cn = min(c + 1, c0 + per - 1);    // the last prefetch re-reads the current chunk: no branch in the pipeline
        gemm_core_k<true, TK>(acc2, T0, w2, kct2, 4 * c, f, w1 + (int64_t)cn * DC_E * DC_E * TK::KP, 4, 0);
        __syncthreads();
    }
    gemm_store<true, TK>(acc2, zero_bias, false, parts + ((int64_t)chunk * rows + row0) * DC_E, DC_E, valid);
}

// The next layer's attention mask as the epilogue of the heads kernel (round 6, 16-bit plans with attention masks at key resolution,
// csrc/attn_mask.hip): mask[b][q][t] = (sum_c e[b][q][c] pooled[b][t][c] + e[b][q][qcol]) < 0 over the T pooled keys of the level, and
// row_any[b][q] = 1 where a row keeps an unmasked key (DEC:618, 677-680) -- the arithmetic of attn_mask_pooled_kernel bit for bit, in
// either of its operand forms (sixteen fp32 MFMAs per key block, or two v_mfma_f32_16x16x32_f16; the query bias as the accumulator's
// initial value), without the launch: e never leaves the workgroup's LDS tile before it is contracted.  Tiles are IMAGE-ALIGNED then (a tile's queries share
// one pooled map): blockIdx.x = image * tiles_per_image + tile.  The key blocks of an image are shared out over `parts` workgroups per
// tile (blockIdx.y: 0 and, behind the optional query-projection part, 2 ..): every part repeats the row phase and the three MLP
// stages (weights from L2, 384 KiB) on its own CU -- the chain is latency, not throughput -- and only part 0 stores out / d / e.
// row_any must arrive ZEROED (the decoder clears the flags of all its predictions in the pooling launch).
struct HeadsMask {
    const float* pooled;        // (B, T, 64) fp32 (msm_pool_mask_taps)
    uint8_t* attn;              // (B, Q, T) bytes, or the bit-packed blocked form of msm_attn_pack_mask_bits when bits
    int32_t* row_any;           // (B, Q), zeroed by the caller
    int T, bits, parts, tiles_per_image, qcol, f16ops;
};

template <typename TK, bool MASK = false, typename WT = typename TK::WT>
__global__ __launch_bounds__(DC_THREADS) void dec_heads_kernel(
    const float* __restrict__ x, const float* __restrict__ parts, int n_parts, const float* __restrict__ bias,
    const float* __restrict__ g1, const float* __restrict__ b1, int l2norm, const float* __restrict__ g2,
    const float* __restrict__ b2, const WT* __restrict__ m0w, const float* __restrict__ m0b, const WT* __restrict__ m1w,
    const float* __restrict__ m1b, const WT* __restrict__ m2w, const float* __restrict__ m2b, const WT* __restrict__ wq,
    const float* __restrict__ bq, const float* __restrict__ qpos, float* __restrict__ out, float* __restrict__ d_out,
    float* __restrict__ e_out, float* __restrict__ q_out, int32_t* __restrict__ row_any_zero, int rows, int Q, float eps, HeadsMask hm, PfRanges pf) {
    __shared__ __attribute__((aligned(16))) float lds[3 * TK::R * TK::LD];
    float *XP = lds, *Dn = lds + TK::R * TK::LD, *T0 = lds + 2 * TK::R * TK::LD;
    if (prefetch_part(pf)) return;                                     // the prefetch rows (see PfRanges)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int row0 = blockIdx.x * TK::R, valid = min(TK::R, rows - row0);
    int img = 0, tile = 0;
    if constexpr (MASK) {
        img = blockIdx.x / hm.tiles_per_image, tile = blockIdx.x - img * hm.tiles_per_image;
        row0 = img * Q + tile * TK::R;
        valid = min(TK::R, Q - tile * TK::R);
    }
    if (row0 >= rows) return;                                          // (padding tile)
    // blockIdx.y == 1 (only launched when wq is given): the next layer's query projection; y == 0: the MLP chain; MASK: y = 0 and
    // y >= (wq ? 2 : 1) are the mask parts
    const bool qpart = wq != nullptr && blockIdx.y == 1;
    const int mpart = MASK ? (blockIdx.y == 0 ? 0 : (int)blockIdx.y - (wq ? 1 : 0)) : 0;
    const bool primary = mpart == 0;                                             // stores out / d / e (uniform)
    // the mask step that consumes e_out needs its row_any flags cleared: done here instead of a separate fill launch
    if (row_any_zero && !qpart && primary && (int)threadIdx.x < valid) row_any_zero[row0 + threadIdx.x] = 0;
    BFrag<WT> f;
    bload(f, qpart ? wq : m0w, 4 * TK::KP, 0, 0);
    // row phase: lane owns 4 consecutive columns; all global operands of the wave's rows are requested up front
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[(TK::R / DC_NW)], pv[(TK::R / DC_NW)];
#pragma unroll
    for (int i = 0; i < (TK::R / DC_NW); ++i) {
        const int gr = min(row0 + wave * (TK::R / DC_NW) + i, rows - 1);
        v[i] = ld4(x + (int64_t)gr * DC_E + lane * 4);
        pv[i] = qpart ? ld4(qpos + (int64_t)(gr % Q) * DC_E + lane * 4) : zero4;
    }
    const float4 biasv = bias ? ld4(bias + lane * 4) : zero4;
    const float4 g1v = g1 ? ld4(g1 + lane * 4) : zero4, b1v = g1 ? ld4(b1 + lane * 4) : zero4;
    const float4 g2v = ld4(g2 + lane * 4), b2v = ld4(b2 + lane * 4);
    constexpr int PS = (TK::R / DC_NW) > 2 ? 4 : 8;                 // partial sums in flight per row (register budget)
    for (int s0 = 0; s0 < n_parts; s0 += PS) {
        float4 p[(TK::R / DC_NW)][PS];
#pragma unroll
        for (int i = 0; i < (TK::R / DC_NW); ++i) {
            const int gr = min(row0 + wave * (TK::R / DC_NW) + i, rows - 1);
#pragma unroll
            for (int s = 0; s < PS; ++s)
                p[i][s] = ld4(parts + ((int64_t)min(s0 + s, n_parts - 1) * rows + gr) * DC_E + lane * 4);
        }
#pragma unroll
        for (int i = 0; i < (TK::R / DC_NW); ++i)
#pragma unroll
            for (int s = 0; s < PS; ++s)
                if (s0 + s < n_parts) v[i] = add4(v[i], p[i][s]);
    }
#pragma unroll
    for (int i = 0; i < (TK::R / DC_NW); ++i) {
        const int r = wave * (TK::R / DC_NW) + i;
        const bool live = r < valid && primary;
        float4 t = add4(v[i], biasv);
        if (g1) t = affine4(t, row_stats(t, eps), g1v, b1v);                         // FFN norm (DEC:300)
        if (l2norm) {                                                                // block norm (DEC:637-638)
            const float nrm = fmaxf(sqrtf(wave_sum((t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w))), 1e-12f);
            t = make_float4(t.x / nrm, t.y / nrm, t.z / nrm, t.w / nrm);
        }
        if (qpart) {
            st4(XP + r * TK::LD + lane * 4, add4(t, pv[i]));
            continue;
        }
        if (out && live) st4(out + (int64_t)(row0 + r) * DC_E + lane * 4, t);
        const float4 y = affine4(t, row_stats(t, eps), g2v, b2v);                    // decoder_norm (DEC:661)
        st4(Dn + r * TK::LD + lane * 4, y);
        if (d_out && live) st4(d_out + (int64_t)(row0 + r) * DC_E + lane * 4, y);
    }
    __syncthreads();
    if (qpart) {                                                                     // next layer's query (uniform branch)
        gemm256<true, false, TK>(XP, wq, 4, 0, bq, false, q_out + (int64_t)row0 * DC_E, DC_E, valid, f, nullptr, 0, 0);
        return;
    }
    gemm256<false, true, TK>(Dn, m0w, 4, 0, m0b, true, T0, 0, 0, f, m1w, 4, 0);                                          // mask_embed MLP (DEC:665)
    __syncthreads();
    gemm256<false, true, TK>(T0, m1w, 4, 0, m1b, true, Dn, 0, 0, f, m2w, 4, 0);
    __syncthreads();
    if constexpr (!MASK) {
        gemm256<true, false, TK>(Dn, m2w, 4, 0, m2b, false, e_out + (int64_t)row0 * DC_E, DC_E, valid, f, nullptr, 0, 0);
    } else {
        static_assert(TK::R == 16, "the mask epilogue walks one 16-query block per workgroup");
        gemm256<false, false, TK>(Dn, m2w, 4, 0, m2b, false, T0, 0, 0, f, nullptr, 0, 0);                                // e -> LDS
        __syncthreads();
        if (primary)                                                                   // e_out: the final mask step / aux consumers read it
            for (int i = threadIdx.x; i < valid * (DC_E / 4); i += DC_THREADS) {
                const int r = i / (DC_E / 4), c4 = i - r * (DC_E / 4);
                st4(e_out + (int64_t)(row0 + r) * DC_E + c4 * 4, ld4(T0 + r * TK::LD + c4 * 4));
            }
        // ---- the mask of this tile's 16 queries against key blocks mpart*DC_NW + wave, + parts*DC_NW, ... (attn_mask_pooled_kernel, F16 form) ----
        const int lj = lane & 15, lq = lane >> 4;
        const int T = hm.T, nkb = (T + 15) / 16, stride = hm.parts * DC_NW;
        // operand forms of attn_mask_pooled_kernel: f16ops -- a lane's channels 8 lq .. + 7 and 32 + 8 lq .. + 7 (the two K = 32 steps of
        // v_mfma_f32_16x16x32_f16); else fp32 -- channels 16 lq .. + 15, sixteen v_mfma_f32_16x16x4_f32 (the plans' default: the mask bits
        // feed back into the attention, so their operands stay fp32 unless lp_pooled_masks asks otherwise)
        const bool h16 = hm.f16ops != 0;                                               // (uniform)
        const int lqw = h16 ? 8 : 16, o1 = 4, o2 = h16 ? 32 : 8, o3 = h16 ? 36 : 12;
        const float* er = T0 + lj * TK::LD + lq * lqw;
        const float4 w0 = ld4(er), w1 = ld4(er + o1), w2 = ld4(er + o2), w3 = ld4(er + o3);
        const f16x8 wh0 = cvt8h(w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w), wh1 = cvt8h(w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w);
        const float qb = T0[lj * TK::LD + hm.qcol];
        const bool qlive = lj < valid;
        const int q = tile * 16 + lj;
        const float* pb = hm.pooled + (int64_t)img * T * 64 + lq * lqw;
        uint8_t* ab = hm.attn + (int64_t)img * Q * T;
        unsigned anyu = 0;
        int kb = mpart * DC_NW + wave;
        float4 a[4], an[4];
        if (kb < nkb) {
            const float* ap = pb + (int64_t)min(kb * 16 + lj, T - 1) * 64;
            a[0] = ld4(ap), a[1] = ld4(ap + o1), a[2] = ld4(ap + o2), a[3] = ld4(ap + o3);
        }
        for (; kb < nkb; kb += stride) {
            const int kn = min(kb + stride, nkb - 1);                                  // the next block's keys are requested before this block's MFMAs
            const float* apn = pb + (int64_t)min(kn * 16 + lj, T - 1) * 64;
            an[0] = ld4(apn), an[1] = ld4(apn + o1), an[2] = ld4(apn + o2), an[3] = ld4(apn + o3);
            f32x4 acc = f32x4{qb, qb, qb, qb};
            if (h16) {
                acc = mfma_f16k32(cvt8h(a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w), wh0, acc);
                acc = mfma_f16k32(cvt8h(a[2].x, a[2].y, a[2].z, a[2].w, a[3].x, a[3].y, a[3].z, a[3].w), wh1, acc);
            } else {
                const float4 wv[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = mfma16(a[u].x, wv[u].x, acc);
                    acc = mfma16(a[u].y, wv[u].y, acc);
                    acc = mfma16(a[u].z, wv[u].z, acc);
                    acc = mfma16(a[u].w, wv[u].w, acc);
                }
            }
            const int key0 = kb * 16 + lq * 4;
            const unsigned m0 = acc[0] < 0.f, m1 = acc[1] < 0.f, m2 = acc[2] < 0.f, m3 = acc[3] < 0.f;   // sigmoid(x) < 0.5 <=> x < 0 (DEC:677)
            if (hm.bits) {
                unsigned nib = (m0 | (m1 << 1) | (m2 << 2) | (m3 << 3)) << (4 * lq);
                nib = or_lane_rows(nib);
                if ((m0 & m1 & m2 & m3) == 0) anyu = 1u;
                const int qc = tile / 7, mb = tile - qc * 7, qchunks = (Q + 111) / 112;  // 112-query chunk and block within it (attention.hip: AQB = 7)
                if (lq == 0 && qlive)
                    reinterpret_cast<unsigned short*>(hm.attn)[((((int64_t)img * qchunks + qc) * nkb + kb) * 16 + lj) * 8 + mb] = (unsigned short)nib;
            } else if (qlive) {
                if ((T & 3) == 0 && key0 + 3 < T) {
                    *reinterpret_cast<uint32_t*>(ab + (int64_t)q * T + key0) = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
                    if ((m0 & m1 & m2 & m3) == 0) anyu = 1u;
                } else {
                    const unsigned mm[4] = {m0, m1, m2, m3};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + r < T) {
                            ab[(int64_t)q * T + key0 + r] = (uint8_t)mm[r];
                            if (!mm[r]) anyu = 1u;
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = an[u];
        }
        anyu = or_lane_rows(anyu);                                                     // (same flag value from every writer)
        if (lq == 0 && anyu && qlive) hm.row_any[(int64_t)img * Q + q] = 1;
    }
}

// packed[((t*(K/64) + kc)*4 + u)*256 + (lq*16 + lj)*4 + c] = W[t*16 + lj][kc*64 + u*16 + lq*4 + c]
__global__ __launch_bounds__(256) void dec_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int N,
                                                              int K) {
    const int64_t total4 = (int64_t)N * K / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int u = (int)(r & 3);
        r >>= 2;
        const int kct = K / 64;
        const int kc = (int)(r % kct), t = (int)(r / kct);
        const int lj = lane & 15, lq = lane >> 4;
        *reinterpret_cast<float4*>(packed + i * 4) =
            *reinterpret_cast<const float4*>(w + (int64_t)(t * 16 + lj) * K + kc * 64 + u * 16 + lq * 4);
    }
}

// bf16: packed[(((t*(K/64) + kc)*2 + up)*64 + lane)*8 + h*4 + c] = bf16(W[t*16 + lj][kc*64 + (2 up + h)*16 + lq*4 + c])
// HL: the packed matrix has K' = 2 K columns: chunks kc < K/64 hold bf16(W), chunks K/64 + kc hold bf16(W - bf16(W)) of the same columns
template <bool F16, bool HL = false>
__global__ __launch_bounds__(256) void dec_pack_weight_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ packed, int N,
                                                                   int K) {
    const int kct = K / 64, kct_p = HL ? 2 * kct : kct;
    const int64_t total8 = (int64_t)N * kct_p * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int up = (int)(r & 1);
        r >>= 1;
        const int kcp = (int)(r % kct_p), t = (int)(r / kct_p);
        const bool low = HL && kcp >= kct;
        const int kc = low ? kcp - kct : kcp;
        const int lj = lane & 15, lq = lane >> 4;
        const float* src = w + (int64_t)(t * 16 + lj) * K + kc * 64 + (2 * up) * 16 + lq * 4;
        float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 16);
        u32x2b ua, ub;
        if constexpr (F16) {
            ua = pack4h(a.x, a.y, a.z, a.w), ub = pack4h(b.x, b.y, b.z, b.w);
        } else {
            if (low) {                                       // the residual of the hi term, itself rounded to bf16
                const Split4 sa = split4(a.x, a.y, a.z, a.w), sb = split4(b.x, b.y, b.z, b.w);
                ua = __builtin_bit_cast(u32x2b, sa.lo), ub = __builtin_bit_cast(u32x2b, sb.lo);
            } else {
                ua = __builtin_bit_cast(u32x2b, pack4(a.x, a.y, a.z, a.w)), ub = __builtin_bit_cast(u32x2b, pack4(b.x, b.y, b.z, b.w));
            }
        }
        *reinterpret_cast<u32x4b*>(packed + i * 8) = u32x4b{ua.x, ua.y, ub.x, ub.y};
    }
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// the ranges the NEXT tail launch of this thread carries in its prefetch row (msm_dec_set_prefetch); taken (and cleared) by that launch
static thread_local PfRanges g_next_prefetch = {};
static PfRanges take_prefetch() {
    const PfRanges pf = g_next_prefetch;
    g_next_prefetch.n = 0;
    return pf;
}

}  // namespace msm

using namespace msm;

extern "C" int msm_dec_pack_weight(const float* w, float* packed, int N, int K, void* stream) {
    MSM_REQUIRE(w && packed && w != packed, "msm_dec_pack_weight: null or aliased pointer");
    MSM_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "msm_dec_pack_weight: N=%d must be a multiple of 16, K=%d of 64", N, K);
    MSM_REQUIRE(aligned16(w) && aligned16(packed), "msm_dec_pack_weight: pointers must be 16-byte aligned");
    const int64_t total4 = (int64_t)N * K / 4;
    hipLaunchKernelGGL(dec_pack_weight_kernel, dim3((unsigned)min((int64_t)2048, (total4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, packed, N, K);
    MSM_CHECK_LAUNCH("msm_dec_pack_weight");
    return MSM_OK;
}

extern "C" int msm_dec_pack_weight_bf16(const float* w, uint16_t* packed, int N, int K, void* stream) {
    MSM_REQUIRE(w && packed, "msm_dec_pack_weight_bf16: null pointer");
    MSM_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "msm_dec_pack_weight_bf16: N=%d must be a multiple of 16, K=%d of 64", N, K);
    MSM_REQUIRE(aligned16(w) && aligned16(packed), "msm_dec_pack_weight_bf16: pointers must be 16-byte aligned");
    const int64_t total8 = (int64_t)N * K / 8;
    hipLaunchKernelGGL(dec_pack_weight_bf16_kernel<false>, dim3((unsigned)min((int64_t)2048, (total8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, packed, N, K);
    MSM_CHECK_LAUNCH("msm_dec_pack_weight_bf16");
    return MSM_OK;
}

extern "C" int msm_dec_pack_weight_bf16x2(const float* w, uint16_t* packed, int N, int K, void* stream) {
    MSM_REQUIRE(w && packed, "msm_dec_pack_weight_bf16x2: null pointer");
    MSM_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "msm_dec_pack_weight_bf16x2: N=%d must be a multiple of 16, K=%d of 64", N, K);
    MSM_REQUIRE(aligned16(w) && aligned16(packed), "msm_dec_pack_weight_bf16x2: pointers must be 16-byte aligned");
    const int64_t total8 = (int64_t)N * K / 4;               // packed holds N x 2 K elements
    hipLaunchKernelGGL((dec_pack_weight_bf16_kernel<false, true>), dim3((unsigned)min((int64_t)2048, (total8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, packed, N, K);
    MSM_CHECK_LAUNCH("msm_dec_pack_weight_bf16x2");
    return MSM_OK;
}

extern "C" int msm_dec_pack_weight_f16(const float* w, uint16_t* packed, int N, int K, void* stream) {
    MSM_REQUIRE(w && packed, "msm_dec_pack_weight_f16: null pointer");
    MSM_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "msm_dec_pack_weight_f16: N=%d must be a multiple of 16, K=%d of 64", N, K);
    MSM_REQUIRE(aligned16(w) && aligned16(packed), "msm_dec_pack_weight_f16: pointers must be 16-byte aligned");
    const int64_t total8 = (int64_t)N * K / 8;
    hipLaunchKernelGGL(dec_pack_weight_bf16_kernel<true>, dim3((unsigned)min((int64_t)2048, (total8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, packed, N, K);
    MSM_CHECK_LAUNCH("msm_dec_pack_weight_f16");
    return MSM_OK;
}

// 8-row fp32 tiles while tiles x parts fit the chip in one round (see the tile kinds above)
static bool use_tile8(int rows, int parts) { return cdiv(rows, 8) * parts <= 256; }
// XCD-aware grids (round 6): workgroups are dealt to the 8 XCDs by linear id % 8 = (x + gridDim.x * y) % 8; with gridDim.x a multiple of 8 a
// row tile x runs on XCD x % 8 in EVERY part (y) of EVERY tail launch, so what one tail wrote for tile x (x_out, q / k / v, the FFN partial sums:
// plain stores stay in the writing XCD's L2) is read back by the next tail's workgroups of the same tile on the same XCD.  The padding tiles
// (at most 7) exit at once.
static int tile_grid_x(int tiles) { return (tiles + 7) & ~7; }
// fp16 weights: 32-row tiles (TileQ32) once the 16-row tiles alone oversubscribe the chip -- the launch is then bound by the aggregate
// L2 -> CU weight stream, which a fragment shared by two tiles halves (MSM_OPT_DEC_TILE32: 1 always, 0 never)
static bool use_tile32(int rows) {
    const int o = opt(MSM_OPT_DEC_TILE32);
    return o == 1 || (o != 0 && rows >= 4096);
}

template <typename TK, typename WT = typename TK::WT>
static int dec_post_cross_impl(const char* who, const float* attn_out, const float* res, const float* query_pos, const WT* wo, const float* bo,
                               const float* ln_g, const float* ln_b, const WT* w_in, const float* b_in, float* x_out, float* qk_out,
                               float* v_out, int rows, int Q, int E, float eps, void* stream) {
    MSM_REQUIRE(attn_out && res && query_pos && wo && bo && ln_g && ln_b && w_in && b_in && x_out && qk_out && v_out, "%s: null pointer", who);
    MSM_REQUIRE(E == DC_E, "%s: E=%d, only 256 is supported", who, E);
    MSM_REQUIRE(rows > 0 && Q > 0, "%s: bad sizes", who);
    MSM_REQUIRE(aligned16(attn_out) && aligned16(wo) && aligned16(w_in), "%s: pointers must be 16-byte aligned", who);
    // 8-row tiles take two parts (q then k | v): three would oversubscribe the chip at 800 rows (see use_tile8)
    PfRanges pf = take_prefetch();
    const int gx = tile_grid_x(cdiv(rows, TK::R));
    pf.rows = prefetch_rows(pf, gx);
    hipLaunchKernelGGL(dec_post_cross_kernel<TK>, dim3(gx, (TK::F8 ? 2 : 3) + pf.rows), dim3(DC_THREADS), 0, (hipStream_t)stream,
                       attn_out, res, query_pos, wo, bo, ln_g, ln_b, w_in, b_in, x_out, qk_out, v_out, rows, Q, eps, pf);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_dec_post_cross(const float* attn_out, const float* res, const float* query_pos, const float* wo,
                                  const float* bo, const float* ln_g, const float* ln_b, const float* w_in, const float* b_in,
                                  float* x_out, float* qk_out, float* v_out, int rows, int Q, int E, float eps, void* stream) {
    if (use_tile8(rows, 2))
        return dec_post_cross_impl<TileF8>("msm_dec_post_cross", attn_out, res, query_pos, wo, bo, ln_g, ln_b, w_in, b_in, x_out, qk_out, v_out, rows,
                                           Q, E, eps, stream);
    return dec_post_cross_impl<TileF16>("msm_dec_post_cross", attn_out, res, query_pos, wo, bo, ln_g, ln_b, w_in, b_in, x_out, qk_out, v_out, rows,
                                        Q, E, eps, stream);
}
extern "C" int msm_dec_post_cross_bf16(const float* attn_out, const float* res, const float* query_pos, const uint16_t* wo,
                                       const float* bo, const float* ln_g, const float* ln_b, const uint16_t* w_in, const float* b_in,
                                       float* x_out, float* qk_out, float* v_out, int rows, int Q, int E, float eps, void* stream) {
    return dec_post_cross_impl<TileH16>("msm_dec_post_cross_bf16", attn_out, res, query_pos, wo, bo, ln_g, ln_b, w_in, b_in, x_out, qk_out,
                                         v_out, rows, Q, E, eps, stream);
}

extern "C" int msm_dec_post_cross_bf16x2(const float* attn_out, const float* res, const float* query_pos, const uint16_t* wo,
                                         const float* bo, const float* ln_g, const float* ln_b, const uint16_t* w_in, const float* b_in,
                                         float* x_out, float* qk_out, float* v_out, int rows, int Q, int E, float eps, void* stream) {
    return dec_post_cross_impl<TileH16x2>("msm_dec_post_cross_bf16x2", attn_out, res, query_pos, (const bf16hl*)wo, bo, ln_g, ln_b, (const bf16hl*)w_in, b_in,
                                          x_out, qk_out, v_out, rows, Q, E, eps, stream);
}

extern "C" int msm_dec_post_cross_f16(const float* attn_out, const float* res, const float* query_pos, const uint16_t* wo,
                                      const float* bo, const float* ln_g, const float* ln_b, const uint16_t* w_in, const float* b_in,
                                      float* x_out, float* qk_out, float* v_out, int rows, int Q, int E, float eps, void* stream) {
    // (measured at 17 300 rows: 55 us with 16-row tiles, 66 with 32 -- three resident workgroups per CU hide more than one does)
    if (opt(MSM_OPT_DEC_TILE32) == 1)
        return dec_post_cross_impl<TileQ32>("msm_dec_post_cross_f16", attn_out, res, query_pos, (const f16w*)wo, bo, ln_g, ln_b, (const f16w*)w_in, b_in,
                                            x_out, qk_out, v_out, rows, Q, E, eps, stream);
    return dec_post_cross_impl<TileQ16>("msm_dec_post_cross_f16", attn_out, res, query_pos, (const f16w*)wo, bo, ln_g, ln_b, (const f16w*)w_in, b_in,
                                        x_out, qk_out, v_out, rows, Q, E, eps, stream);
}

template <typename TK, typename WT = typename TK::WT>
static int dec_post_self_impl(const char* who, const float* attn_out, const float* res, const WT* wo, const float* bo, const float* ln_g,
                              const float* ln_b, const WT* w1, const float* b1, const WT* w2, int F, float* x_out, float* parts, int n_parts,
                              int rows, int E, float eps, void* stream) {
    MSM_REQUIRE(attn_out && res && wo && bo && ln_g && ln_b && w1 && b1 && w2 && x_out && parts, "%s: null pointer", who);
    MSM_REQUIRE(E == DC_E, "%s: E=%d, only 256 is supported", who, E);
    MSM_REQUIRE(rows > 0 && F > 0 && F % DC_E == 0, "%s: F=%d must be a positive multiple of 256", who, F);
    MSM_REQUIRE(n_parts > 0 && (F / DC_E) % n_parts == 0, "%s: n_parts=%d must divide F/256=%d", who, n_parts, F / DC_E);
    MSM_REQUIRE(aligned16(attn_out) && aligned16(wo) && aligned16(w1) && aligned16(w2), "%s: pointers must be 16-byte aligned", who);
    PfRanges pf = take_prefetch();
    const int gx = tile_grid_x(cdiv(rows, TK::R));
    pf.rows = prefetch_rows(pf, gx);
    hipLaunchKernelGGL(dec_post_self_kernel<TK>, dim3(gx, n_parts + pf.rows), dim3(DC_THREADS), 0, (hipStream_t)stream, attn_out,
                       res, wo, bo, ln_g, ln_b, w1, b1, w2, F, x_out, parts, rows, eps, pf);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_dec_post_self(const float* attn_out, const float* res, const float* wo, const float* bo, const float* ln_g,
                                 const float* ln_b, const float* w1, const float* b1, const float* w2, int F, float* x_out,
                                 float* parts, int n_parts, int rows, int E, float eps, void* stream) {
    if (n_parts > 0 && use_tile8(rows, n_parts))
        return dec_post_self_impl<TileF8>("msm_dec_post_self", attn_out, res, wo, bo, ln_g, ln_b, w1, b1, w2, F, x_out, parts, n_parts, rows, E, eps,
                                          stream);
    return dec_post_self_impl<TileF16>("msm_dec_post_self", attn_out, res, wo, bo, ln_g, ln_b, w1, b1, w2, F, x_out, parts, n_parts, rows, E, eps,
                                       stream);
}
extern "C" int msm_dec_post_self_bf16(const float* attn_out, const float* res, const uint16_t* wo, const float* bo, const float* ln_g,
                                      const float* ln_b, const uint16_t* w1, const float* b1, const uint16_t* w2, int F, float* x_out,
                                      float* parts, int n_parts, int rows, int E, float eps, void* stream) {
    return dec_post_self_impl<TileH16>("msm_dec_post_self_bf16", attn_out, res, wo, bo, ln_g, ln_b, w1, b1, w2, F, x_out, parts, n_parts, rows, E,
                                        eps, stream);
}

extern "C" int msm_dec_post_self_bf16x2(const float* attn_out, const float* res, const uint16_t* wo, const float* bo, const float* ln_g,
                                        const float* ln_b, const uint16_t* w1, const float* b1, const uint16_t* w2, int F, float* x_out,
                                        float* parts, int n_parts, int rows, int E, float eps, void* stream) {
    return dec_post_self_impl<TileH16x2>("msm_dec_post_self_bf16x2", attn_out, res, (const bf16hl*)wo, bo, ln_g, ln_b, (const bf16hl*)w1, b1, (const bf16hl*)w2, F,
                                         x_out, parts, n_parts, rows, E, eps, stream);
}

extern "C" int msm_dec_post_self_f16(const float* attn_out, const float* res, const uint16_t* wo, const float* bo, const float* ln_g,
                                     const float* ln_b, const uint16_t* w1, const float* b1, const uint16_t* w2, int F, float* x_out,
                                     float* parts, int n_parts, int rows, int E, float eps, void* stream) {
    if (use_tile32(rows))
        return dec_post_self_impl<TileQ32>("msm_dec_post_self_f16", attn_out, res, (const f16w*)wo, bo, ln_g, ln_b, (const f16w*)w1, b1, (const f16w*)w2, F,
                                           x_out, parts, n_parts, rows, E, eps, stream);
    return dec_post_self_impl<TileQ16>("msm_dec_post_self_f16", attn_out, res, (const f16w*)wo, bo, ln_g, ln_b, (const f16w*)w1, b1, (const f16w*)w2, F,
                                       x_out, parts, n_parts, rows, E, eps, stream);
}

template <typename TK, bool MASK = false, typename WT = typename TK::WT>
static int dec_heads_impl(const char* who, const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g,
                          const float* ln_b, int l2norm, const float* dec_g, const float* dec_b, const WT* m0w, const float* m0b, const WT* m1w,
                          const float* m1b, const WT* m2w, const float* m2b, const WT* wq, const float* bq, const float* query_pos, float* out,
                          float* d_out, float* e_out, float* q_out, int32_t* row_any_zero, int rows, int Q, int E, float eps, void* stream,
                          HeadsMask hm = HeadsMask{}) {
    MSM_REQUIRE(x && dec_g && dec_b && m0w && m0b && m1w && m1b && m2w && m2b && e_out, "%s: null pointer", who);
    MSM_REQUIRE(E == DC_E, "%s: E=%d, only 256 is supported", who, E);
    MSM_REQUIRE(rows > 0 && Q > 0 && n_parts >= 0, "%s: bad sizes", who);
    MSM_REQUIRE(n_parts == 0 || parts, "%s: parts missing", who);
    MSM_REQUIRE((ln_g == nullptr) == (ln_b == nullptr), "%s: ln_g/ln_b must both be given or both be null", who);
    MSM_REQUIRE(!wq || (bq && query_pos && q_out), "%s: the next-query projection needs bq, query_pos and q_out", who);
    MSM_REQUIRE(aligned16(m0w) && aligned16(m1w) && aligned16(m2w) && aligned16(wq), "%s: weights must be 16-byte aligned", who);
    dim3 grid(tile_grid_x(cdiv(rows, TK::R)), wq ? 2 : 1);
    if constexpr (MASK) {
        MSM_REQUIRE(hm.pooled && hm.attn && hm.row_any && hm.T > 0 && rows % Q == 0, "%s: the mask epilogue needs pooled / attn / row_any, T > 0 and rows = B * Q", who);
        MSM_REQUIRE(hm.qcol >= 64 && hm.qcol < DC_E && aligned16(hm.pooled), "%s: qcol=%d must name a column of e behind the 64 embedding columns", who, hm.qcol);
        MSM_REQUIRE(!hm.bits || (hm.T % 16 == 0 && aligned16(hm.attn)), "%s: the bit-packed mask needs T %% 16 == 0 and a 16-byte aligned buffer", who);
        hm.tiles_per_image = cdiv(Q, TK::R);
        const int tiles = (rows / Q) * hm.tiles_per_image, nkb = cdiv(hm.T, 16);
        // parts: about one chip of workgroups in all, at least ~3 key blocks per wave (every part repeats the MLP chain)
        hm.parts = max(1, min(min(cdiv(nkb, 3 * DC_NW), 8), max(1, 256 / tiles - (wq ? 1 : 0))));
        grid = dim3(tiles, hm.parts + (wq ? 1 : 0));
    }
    PfRanges pf = take_prefetch();
    pf.rows = prefetch_rows(pf, (int)grid.x);
    grid.y += pf.rows;
    hipLaunchKernelGGL((dec_heads_kernel<TK, MASK>), grid, dim3(DC_THREADS), 0, (hipStream_t)stream, x, parts, n_parts, bias,
                       ln_g, ln_b, l2norm, dec_g, dec_b, m0w, m0b, m1w, m1b, m2w, m2b, wq, bq, query_pos, out, d_out, e_out, q_out,
                       row_any_zero, rows, Q, eps, hm, pf);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_dec_heads(const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g,
                             const float* ln_b, int l2norm, const float* dec_g, const float* dec_b, const float* m0w,
                             const float* m0b, const float* m1w, const float* m1b, const float* m2w, const float* m2b,
                             const float* wq, const float* bq, const float* query_pos, float* out, float* d_out, float* e_out,
                             float* q_out, int32_t* row_any_zero, int rows, int Q, int E, float eps, void* stream) {
    if (use_tile8(rows, wq ? 2 : 1))
        return dec_heads_impl<TileF8>("msm_dec_heads", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, m0w, m0b, m1w, m1b, m2w, m2b, wq, bq,
                                      query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
    return dec_heads_impl<TileF16>("msm_dec_heads", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, m0w, m0b, m1w, m1b, m2w, m2b, wq, bq,
                                   query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
}
extern "C" int msm_dec_heads_bf16(const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g,
                                  const float* ln_b, int l2norm, const float* dec_g, const float* dec_b, const uint16_t* m0w,
                                  const float* m0b, const uint16_t* m1w, const float* m1b, const uint16_t* m2w, const float* m2b,
                                  const uint16_t* wq, const float* bq, const float* query_pos, float* out, float* d_out, float* e_out,
                                  float* q_out, int32_t* row_any_zero, int rows, int Q, int E, float eps, void* stream) {
    return dec_heads_impl<TileH16>("msm_dec_heads_bf16", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, m0w, m0b, m1w, m1b, m2w, m2b,
                                    wq, bq, query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
}
extern "C" int msm_dec_heads_bf16x2(const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g,
                                    const float* ln_b, int l2norm, const float* dec_g, const float* dec_b, const uint16_t* m0w,
                                    const float* m0b, const uint16_t* m1w, const float* m1b, const uint16_t* m2w, const float* m2b,
                                    const uint16_t* wq, const float* bq, const float* query_pos, float* out, float* d_out, float* e_out,
                                    float* q_out, int32_t* row_any_zero, int rows, int Q, int E, float eps, void* stream) {
    return dec_heads_impl<TileH16x2>("msm_dec_heads_bf16x2", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, (const bf16hl*)m0w, m0b, (const bf16hl*)m1w,
                                     m1b, (const bf16hl*)m2w, m2b, (const bf16hl*)wq, bq, query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
}

extern "C" int msm_dec_heads_f16(const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g,
                                 const float* ln_b, int l2norm, const float* dec_g, const float* dec_b, const uint16_t* m0w,
                                 const float* m0b, const uint16_t* m1w, const float* m1b, const uint16_t* m2w, const float* m2b,
                                 const uint16_t* wq, const float* bq, const float* query_pos, float* out, float* d_out, float* e_out,
                                 float* q_out, int32_t* row_any_zero, int rows, int Q, int E, float eps, void* stream) {
    if (use_tile32(rows))
        return dec_heads_impl<TileQ32>("msm_dec_heads_f16", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, (const f16w*)m0w, m0b, (const f16w*)m1w,
                                       m1b, (const f16w*)m2w, m2b, (const f16w*)wq, bq, query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
    return dec_heads_impl<TileQ16>("msm_dec_heads_f16", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, (const f16w*)m0w, m0b, (const f16w*)m1w,
                                   m1b, (const f16w*)m2w, m2b, (const f16w*)wq, bq, query_pos, out, d_out, e_out, q_out, row_any_zero, rows, Q, E, eps, stream);
}

// heads + the next layer's attention mask at key resolution in one launch (see HeadsMask above).  flags: 1 = bit-packed blocked mask
// (msm_attn_pack_mask_bits' layout), 2 = fp16 weight fragments (msm_dec_pack_weight_f16; else bf16 fragments), 4 = the mask contraction on
// IEEE-half operands (msm_attn_mask_pooled's flag 2; else its fp32 MFMA chain).
extern "C" int msm_dec_heads_mask(const float* x, const float* parts, int n_parts, const float* bias, const float* ln_g, const float* ln_b, int l2norm,
                                  const float* dec_g, const float* dec_b, const uint16_t* m0w, const float* m0b, const uint16_t* m1w,
                                  const float* m1b, const uint16_t* m2w, const float* m2b, const uint16_t* wq, const float* bq,
                                  const float* query_pos, float* out, float* d_out, float* e_out, float* q_out, const float* pooled, int T, int qcol,
                                  uint8_t* attn, int32_t* row_any, int flags, int rows, int Q, int E, float eps, void* stream) {
    MSM_REQUIRE((flags & ~7) == 0, "msm_dec_heads_mask: flags=%d (1 = bit-packed mask, 2 = fp16 weight fragments, 4 = IEEE-half mask operands)", flags);
    HeadsMask hm{};
    hm.pooled = pooled, hm.attn = attn, hm.row_any = row_any, hm.T = T, hm.bits = flags & 1, hm.qcol = qcol, hm.f16ops = (flags >> 2) & 1;
    if (flags & 2)
        return dec_heads_impl<TileQ16, true>("msm_dec_heads_mask", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, (const f16w*)m0w, m0b,
                                             (const f16w*)m1w, m1b, (const f16w*)m2w, m2b, (const f16w*)wq, bq, query_pos, out, d_out, e_out, q_out, nullptr,
                                             rows, Q, E, eps, stream, hm);
    return dec_heads_impl<TileH16, true>("msm_dec_heads_mask", x, parts, n_parts, bias, ln_g, ln_b, l2norm, dec_g, dec_b, m0w, m0b, m1w, m1b, m2w, m2b,
                                         wq, bq, query_pos, out, d_out, e_out, q_out, nullptr, rows, Q, E, eps, stream, hm);
}

// The byte ranges (16-byte aligned device pointers; HOST arrays of n <= 6 entries) that the NEXT msm_dec_post_cross* / msm_dec_post_self* /
// msm_dec_heads* launch issued by this thread touches from an extra row of workgroups (see PfRanges above): the weights of the launches
// that follow it in the chain.  n = 0 clears a pending request.  Affects speed only.
extern "C" int msm_dec_set_prefetch(const void* const* ptrs, const int64_t* bytes, int n) {
    MSM_REQUIRE(n >= 0 && n <= DC_PF_MAX && (n == 0 || (ptrs && bytes)), "msm_dec_set_prefetch: 0..%d ranges", DC_PF_MAX);
    PfRanges pf{};
    for (int j = 0; j < n; ++j) {
        MSM_REQUIRE(ptrs[j] && bytes[j] > 0 && aligned16(ptrs[j]), "msm_dec_set_prefetch: range %d must be a 16-byte aligned device pointer with a positive size", j);
        pf.p[j] = (const unsigned char*)ptrs[j];
        pf.bytes[j] = bytes[j];
    }
    pf.n = n;
    g_next_prefetch = pf;
    return MSM_OK;
}
