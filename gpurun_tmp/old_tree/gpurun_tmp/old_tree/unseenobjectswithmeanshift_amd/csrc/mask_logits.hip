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

#include <type_traits>

#include "bf16.h"
#include "common.h"

namespace msm {

constexpr int QB = 7;           // 16-row MFMA blocks per query chunk
constexpr int QCH = QB * 16;    // 112 queries per chunk
#ifndef MSM_MASK_KU
#define MSM_MASK_KU 4
#endif
#ifndef MSM_WRITE_AUX
#define MSM_WRITE_AUX 0   // cache policy bits of the logit stores (tuning builds: 2 = nt)
#endif
#ifndef MSM_MASK_MW
#define MSM_MASK_MW 8
#endif
constexpr int KU = MSM_MASK_KU;  // k-steps (of 4) per prefetch group
constexpr int MW = MSM_MASK_MW;  // waves per workgroup: the 116 KB mask_embed chunk allows ONE workgroup per CU, so 8 waves
                                // give every SIMD two instruction streams (one wave alone cannot hide its own ds_read /
                                // buffer-load issue and waitcnt bubbles behind its MFMAs)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef MSM_MASK_TS   // probe build only (tools/probes/mask_ts.py): per-wave phase timestamps, 100 MHz wall clock
__device__ unsigned long long g_mask_ts[256 * 8 * 16];
#define MASK_TS(slot)                                                                                          \
    if (lane == 0 && blockIdx.y == 0 && (int)(blockIdx.z * gridDim.x + blockIdx.x) < 256 && (slot) < 16)       \
        g_mask_ts[((blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 16 + (slot)] = wall_clock64();
// shader-clock counter (s_memtime) into a slot: with the 100 MHz stamps it gives the clock the SIMD actually ran at
#define MASK_TSC(slot)                                                                                         \
    if (lane == 0 && blockIdx.y == 0 && (int)(blockIdx.z * gridDim.x + blockIdx.x) < 256 && (slot) < 16)       \
        g_mask_ts[((blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 16 + (slot)] = clock64();
#else
#define MASK_TS(slot)
#define MASK_TSC(slot)
#endif

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

// ---- accumulator layout -----------------------------------------------------------------------------------------
// The product is issued TRANSPOSED: the MFMA A operand is the feature fragment (16 pixels x 4 channels), the B operand
// the mask_embed fragment (4 channels x 16 queries), so D[row = pixel][col = query] and lane (lq = l >> 4, lj = l & 15)
// holds, for query 16 m + lj, the pixels of A rows 4 lq .. 4 lq + 3.  A row i is whatever pixel lane i loaded:
//   NC == 2: lane i loads the float2 at columns c0 + 2 i, + 1 (even / odd column blocks)  -> lane lq owns the 8
//            CONSECUTIVE pixels c0 + 8 lq .. + 7 of both image rows of the tile;
//   NC == 1: lane i loads column c0 + i -> 4 consecutive pixels c0 + 4 lq .. + 3.
// Every 2x2 tap of the bilinear downsample by 2, 4 (columns 4i+1, 4i+2) or 8 (8i+3, 8i+4) then lies inside ONE lane --
// no cross-lane traffic -- and the attention-mask bytes of a lane are consecutive keys of one query: one dword (or
// short / byte) store per 16-query block instead of one byte store per query.  (Queries-as-rows, the layout of round 1,
// needed 28 byte-store instructions and 56 DPP moves per tile: 5-8 of the 30 us of a launch.)
// The one case where a tap would straddle lanes (NC == 1, downsample by 8: columns 3|4 and 11|12) permutes the pixels
// the lanes load: A rows 0..15 <- columns {3,4,2,5, 0,1,6,7, 11,12,10,13, 8,9,14,15}.
template <int POOL, int NC>
struct PixMap {
    static constexpr bool PERM = (POOL == 8 && NC == 1);
    // column (relative to the tile's first column) loaded by lane i of a 16-lane row
    __device__ static __forceinline__ int load_col(int i) {
        if constexpr (PERM) {
            const int g = i >> 2, r = i & 3;
            return (g >> 1) * 8 + (int)((((g & 1) ? 0x7610u : 0x5243u) >> (4 * r)) & 15u);
        } else {
            return NC * i;
        }
    }
    // column of the lane's j-th value (j = 0 .. 4 NC - 1, in register order: NC == 2 -> reg r = j >> 1, block cc = j & 1)
    __device__ static __forceinline__ int out_col(int lq, int j) {
        if constexpr (PERM) return load_col(lq * 4 + j);
        else return 4 * NC * lq + j;
    }
};

template <int NC>
__device__ __forceinline__ float acc_val(const f32x4 (&acc)[QB][2 * NC], int m, int row, int j) {
    if constexpr (NC == 2) return acc[m][row * 2 + (j & 1)][j >> 1];
    else return acc[m][row][j];
}

// Generic per-tile epilogue shared by the fp32 and bf16 kernels (see the layout note above): any tile (partly outside the
// map, unaligned rows, permuted pixels).  c0: first column of the tile; DO_WRITE / DO_ATTN select the two halves so that a
// kernel can take the fast path (mask_tile_epilogue_fast below) for one and this one for the other.
template <int POOL, bool WRITE, int NC, bool DO_WRITE = true, bool DO_ATTN = true, bool R4 = false>
__device__ __forceinline__ void mask_tile_epilogue(const f32x4 (&acc)[QB][2 * NC], float* __restrict__ mask_out,
                                                   uint8_t* __restrict__ attn_out, int* __restrict__ any_flags, int b, int Q,
                                                   int q0, int H, int W, int th, int tw, int ytop, int ybot, int c0, int lj, int lq) {
    using PM = PixMap<POOL, NC>;
    constexpr int NP = 4 * NC;                       // values per lane and image row
    // The query offset is made opaque: the per-query output base addresses depend only on the lane, so LICM would
    // otherwise hoist them out of the tile loop and hold dozens of VGPRs across the K loop.
    int ql = lj;
    asm volatile("" : "+v"(ql));
    const int xb = c0 + NP * lq;                     // first column of this lane (identity mapping)
    if constexpr (WRITE && DO_WRITE) {
#pragma unroll
        for (int m = 0; m < QB; ++m) {
            // (R4: in the 4-query block lane lj of the lq == 0 quarter holds query 96 + (lj & 3) and pixel group lj >> 2)
            const bool rem = R4 && m == QB - 1;
            const int pg = rem ? (ql >> 2) : lq;
            const int xb = c0 + NP * pg;
            const bool vec = !PM::PERM && xb + NP <= W && (W & 3) == 0;   // 16-byte aligned rows, whole lane inside the map
            const int q = rem ? q0 + m * 16 + (ql & 3) : q0 + m * 16 + ql;
            if (q >= Q || (rem && lq != 0)) continue;
            float* o = mask_out + ((int64_t)b * Q + q) * ((int64_t)H * W);
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const int y = row ? ybot : ytop;
                if (y < 0 || y >= H) continue;
                float* orow = o + (int64_t)y * W;
                if (vec) {
#pragma unroll
                    for (int j = 0; j < NP; j += 4)
                        *reinterpret_cast<float4*>(orow + xb + j) = make_float4(acc_val<NC>(acc, m, row, j), acc_val<NC>(acc, m, row, j + 1),
                                                                               acc_val<NC>(acc, m, row, j + 2), acc_val<NC>(acc, m, row, j + 3));
                } else {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const int x = c0 + PM::out_col(pg, j);
                        if (x < W) orow[x] = acc_val<NC>(acc, m, row, j);
                    }
                }
            }
        }
    }
    if constexpr (!DO_ATTN) {
        return;
    } else if constexpr (POOL == 1) {
        // mask at full resolution: one byte per logit, NP consecutive keys per lane and row
        const int HW = H * W;
        const bool vec = xb + NP <= W && (W & 3) == 0;
#pragma unroll
        for (int m = 0; m < QB; ++m) {
            const int q = q0 + m * 16 + ql;
            if (q >= Q) continue;
            uint8_t* o = attn_out + ((int64_t)b * Q + q) * HW;
            bool any = false;
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const int y = row ? ybot : ytop;
                if (y < 0 || y >= H) continue;
#pragma unroll
                for (int j0 = 0; j0 < NP; j0 += 4) {
                    uint32_t w = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w |= (acc_val<NC>(acc, m, row, j0 + j) < 0.f ? 1u : 0u) << (8 * j);
                    if (vec) {
                        *reinterpret_cast<uint32_t*>(o + (int64_t)y * W + xb + j0) = w;
                        any |= w != 0x01010101u;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int x = xb + j0 + j;
                            if (x < W) { o[(int64_t)y * W + x] = (uint8_t)((w >> (8 * j)) & 1u); any |= ((w >> (8 * j)) & 1u) == 0u; }
                        }
                    }
                }
            }
            if (any) any_flags[q - q0] = 1;
        }
    } else if constexpr (POOL != 0) {
        // tap rows are (POOL*i + POOL/2 - 1, +1): the pair (ytop, ybot) is a tap pair iff ytop % POOL == POOL/2 - 1
        // (always true for POOL == 2 with even pairing).  Wave-uniform: every other pair at POOL 4, 3 of 4 at 8 leave here.
        if (!((ytop >= 0) && (ybot < H) && ((ytop % POOL) == POOL / 2 - 1))) return;
        const int ty = ytop / POOL;
        if (ty >= th) return;
        // taps of this lane: value indices (jl, jl + 1) of both rows; their target columns are consecutive
        constexpr int NT = PM::PERM ? 1 : (NP / POOL > 0 ? NP / POOL : 1);            // taps per lane
        constexpr int J0 = PM::PERM ? 0 : POOL / 2 - 1;                                 // first left tap (value index)
        constexpr int JS = PM::PERM ? 0 : POOL;                                         // value-index step between taps
        // lanes that hold a tap: all (NP >= POOL), or -- one tap per two lanes -- the permuted NC == 1 / POOL == 8 case
        const bool lane_tap = PM::PERM ? ((lq & 1) == 0) : true;
        const int tx0 = PM::PERM ? (c0 / 8 + (lq >> 1)) : (xb / POOL);
        static_assert(PM::PERM || NP >= POOL, "a lane must hold whole taps");
        const bool vec = NT == 4 ? ((tw & 3) == 0 && tx0 + 4 <= tw) : (NT == 2 ? ((tw & 1) == 0 && tx0 + 2 <= tw) : (tx0 < tw));
#pragma unroll
        for (int m = 0; m < QB; ++m) {
            uint32_t w = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jl = J0 + t * JS;
                const float s = (acc_val<NC>(acc, m, 0, jl) + acc_val<NC>(acc, m, 0, jl + 1)) +
                                (acc_val<NC>(acc, m, 1, jl) + acc_val<NC>(acc, m, 1, jl + 1));
                w |= (s < 0.f ? 1u : 0u) << (8 * t);
            }
            const int q = q0 + m * 16 + ql;
            if (q < Q && lane_tap) {
                uint8_t* o = attn_out + ((int64_t)b * Q + q) * (th * tw) + ty * tw + tx0;
                if (vec) {
                    if constexpr (NT == 4) *reinterpret_cast<uint32_t*>(o) = w;
                    else if constexpr (NT == 2) *reinterpret_cast<uint16_t*>(o) = (uint16_t)w;
                    else *o = (uint8_t)w;
                    constexpr uint32_t ALL = NT == 4 ? 0x01010101u : (NT == 2 ? 0x0101u : 0x01u);
                    if (w != ALL) any_flags[q - q0] = 1;      // LDS: flushed to row_any once per workgroup
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (tx0 + t < tw) {
                            o[t] = (uint8_t)((w >> (8 * t)) & 1u);
                            if (((w >> (8 * t)) & 1u) == 0u) any_flags[q - q0] = 1;
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one query block at a time: keeps the epilogue's live set small
        }
    }
}

// Per-lane constants of the fast epilogue, computed once per kernel: byte offsets of the lane's query rows (one per
// 16-query block) in the attention-mask and mask outputs of image b, relative to buffer descriptors over that image's
// outputs.  Rows q >= Q get offsets beyond the descriptor's size: the hardware drops those stores, so the tile loop has no
// per-lane bounds logic at all; the per-tile part of the address (row, first column) is wave-uniform and travels in SGPRs.
template <int POOL, bool WRITE, int NC>
struct MaskEpiConst {
    unsigned aoff[QB];     // attention mask: (q * TT + lane's first key) bytes, TT = keys per query
    unsigned moff[QB];     // mask logits: (q * H*W + lane's first pixel) * 4 bytes
    unsigned anyv[QB];     // != 0 once a key of the row was seen attendable
    __amdgpu_buffer_rsrc_t arsrc, mrsrc;
    bool attn_fast, write_fast;      // kernel-uniform: the alignment conditions of the vector stores hold
};

// a + b as ONE v_add_f32: left to itself hipcc pairs the tap sums into v_pk_add_f32 and pays two v_movs per pair to line
// the operands up; every VALU instruction of the epilogue runs beside the sibling wave's MFMAs and costs ~40 cycles there
__device__ __forceinline__ float add1(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 1 iff x < 0 (the reference's sigmoid(x) < 0.5), as an integer: the sign bit.  x is a sum (a+b)+(c+d) of four logits:
// -0.0 -- sign bit set, not < 0 -- would need all four logits to be exactly -0.0.
__device__ __forceinline__ unsigned neg_bit(float x) { return __float_as_uint(x) >> 31; }

__device__ __forceinline__ void* uniform_ptr64(const void* p) {
    const uint64_t u = (uint64_t)p;
    return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                   (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
}

// r4: the chunk is 6 full query blocks + 4 queries and block 6 is multiplied on v_mfma_f32_4x4x1_16b_f32 (see mask_logits_kernel):
// there lane lj of the lq == 0 quarter holds query 96 + (lj & 3) and the pixels of group lj >> 2 (the role lq plays in a full block)
template <int POOL, bool WRITE, int NC>
__device__ __forceinline__ void mask_epi_init(MaskEpiConst<POOL, WRITE, NC>& k, float* mask_out, uint8_t* attn_out, int b, int Q, int q0,
                                              int H, int W, int th, int tw, int lj, int lq, bool r4 = false) {
    using PM = PixMap<POOL, NC>;
    constexpr int NP = 4 * NC;
    const int HW = H * W;
    const int TT = POOL == 1 ? HW : th * tw;
    constexpr int NT = PM::PERM ? 1 : (POOL == 1 ? NP : (NP / (POOL > 0 ? POOL : 1) > 0 ? NP / (POOL > 0 ? POOL : 1) : 1));
#pragma unroll
    for (int m = 0; m < QB; ++m) {
        const bool rem = r4 && m == QB - 1;
        const int pgl = rem ? (lj >> 2) : lq;                  // pixel group of the lane: lq, or lj >> 2 in the 4-query block
        const int q = rem ? q0 + m * 16 + (lj & 3) : q0 + m * 16 + lj;
        // first key of the lane inside a tile row
        const int lane_key = POOL == 1 ? NP * pgl : (PM::PERM ? (pgl >> 1) : (NP * pgl) / (POOL > 0 ? POOL : 1));
        const bool lane_on = (PM::PERM ? ((pgl & 1) == 0) : true) && (!rem || lq == 0);
        k.aoff[m] = (lane_on && q < Q) ? (unsigned)(q * TT + lane_key) : 0xF0000000u;
        k.moff[m] = (q < Q && (!rem || lq == 0)) ? (unsigned)((q * HW + NP * pgl) * 4) : 0xF0000000u;
        k.anyv[m] = 0u;
    }
    if constexpr (POOL != 0)
        k.arsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(attn_out + (int64_t)b * Q * TT), 0, Q * TT, 0x00020000);
    if constexpr (WRITE)
        k.mrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(mask_out + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000);
    const int keys_row = POOL == 1 ? W : tw;
    k.attn_fast = POOL != 0 && (NT == 4 || NT == 8 ? (keys_row & 3) == 0 : (NT == 2 ? (keys_row & 1) == 0 : true)) && (int64_t)Q * TT < 0xF0000000ll;
    k.write_fast = WRITE && !PM::PERM && (W & 3) == 0 && (int64_t)Q * HW * 4 < 0xF0000000ll;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Fast epilogue of a tile that lies completely inside the map (c0 + 16 NC <= W; the caller checks, wave-uniformly):
// zero address arithmetic -- every store is one buffer instruction with a precomputed per-lane offset and an SGPR tile
// offset --, the row_any flags are OR-ed into registers and reach LDS once per kernel.  VALU instructions next to a busy
// MFMA pipe are expensive (the sibling wave of the SIMD is in its K loop): measured with in-kernel timestamps, the generic
// epilogue cost 1.8 (15x20 / 30x40 targets) to 3.5 us (60x80) per 3 us tile, this one a few hundred ns.
// R4_PARTIAL: block QB - 1 holds per-channel-class PARTIAL sums (4-query block on the 4x4x1 MFMA, see mask_logits_kernel): the tap
// sum is linear, so it is formed on the partials and the four classes (lanes l ^ 16, l ^ 32) are added afterwards -- two values per
// tile and lane at POOL 2, one at 4 / 8, instead of all eight accumulators.
template <int POOL, bool WRITE, int NC, bool DO_WRITE, bool DO_ATTN, bool R4_PARTIAL = false>
__device__ __forceinline__ void mask_tile_epilogue_fast(const f32x4 (&acc)[QB][2 * NC], MaskEpiConst<POOL, WRITE, NC>& k, int H, int W, int tw,
                                                        int ytop, int ybot, int c0) {
    using PM = PixMap<POOL, NC>;
    constexpr int NP = 4 * NC;
    if constexpr (WRITE && DO_WRITE) {
        // NC == 1 only: the 16-byte store data must BE an accumulator tuple.  With 2 x 32 tiles the lane's consecutive pixels
        // alternate between two tuples and hipcc assembles each float4 with v_movs into one scratch tuple, re-used for the
        // next store -- and on gfx950 a buffer_store_dwordx4 WITH an SGPR soffset still reads its data registers when the
        // following v_mov overwrites the first of them (the first dword of the stores of queries 12..15 of a block came out
        // wrong; LLVM's hazard recogniser assumes the SGPR-offset form is exempt and inserts no wait state).  The host never
        // pairs WRITE with NC == 2.
        static_assert(NC == 1, "mask writes take 2 x 16 tiles");
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int y = row ? ybot : ytop;
            if (y < 0 || y >= H) continue;                                  // wave-uniform
            const unsigned soff = (unsigned)(y * W + c0) * 4u;
#pragma unroll
            for (int m = 0; m < QB; ++m)
#pragma unroll
                for (int j = 0; j < NP; j += 4)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc_val<NC>(acc, m, row, j)), __float_as_uint(acc_val<NC>(acc, m, row, j + 1)),
                                                                 __float_as_uint(acc_val<NC>(acc, m, row, j + 2)), __float_as_uint(acc_val<NC>(acc, m, row, j + 3))},
                                                           k.mrsrc, k.moff[m] + 4u * j, soff, MSM_WRITE_AUX);
        }
    }
    if constexpr (!DO_ATTN || POOL == 0) {
        return;
    } else if constexpr (POOL == 1) {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int y = row ? ybot : ytop;
            if (y < 0 || y >= H) continue;
            const unsigned soff = (unsigned)(y * W + c0);
#pragma unroll
            for (int m = 0; m < QB; ++m)
#pragma unroll
                for (int j0 = 0; j0 < NP; j0 += 4) {
                    unsigned w = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w |= acc_val<NC>(acc, m, row, j0 + j) < 0.f ? (1u << (8 * j)) : 0u;
                    __builtin_amdgcn_raw_buffer_store_b32(w, k.arsrc, k.aoff[m] + j0, soff, 0);
                    k.anyv[m] |= w ^ 0x01010101u;
                }
        }
    } else {
        if (!((ytop >= 0) && (ybot < H) && ((ytop % POOL) == POOL / 2 - 1))) return;       // no tap in this row pair (wave-uniform)
        constexpr int NT = PM::PERM ? 1 : NP / POOL;
        constexpr int J0 = PM::PERM ? 0 : POOL / 2 - 1;
        constexpr int JS = PM::PERM ? 0 : POOL;
        static_assert(PM::PERM || NP >= POOL, "a lane must hold whole taps");
        const unsigned soff = (unsigned)((ytop / POOL) * tw + c0 / POOL);
#pragma unroll
        for (int m = 0; m < QB; ++m) {
            unsigned w = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jl = J0 + t * JS;
                float s = add1(add1(acc_val<NC>(acc, m, 0, jl), acc_val<NC>(acc, m, 0, jl + 1)),
                               add1(acc_val<NC>(acc, m, 1, jl), acc_val<NC>(acc, m, 1, jl + 1)));
                if constexpr (R4_PARTIAL) {
                    if (m == QB - 1) {
                        s = sum_lane_rows(s);
                    }
                }
                w = t == 0 ? neg_bit(s) : (w | (neg_bit(s) << (8 * t)));
            }
#ifndef MSM_EPI_AUX
#define MSM_EPI_AUX 0
#endif
            if constexpr (NT == 4) {
#ifndef MSM_EPI_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b32(w, k.arsrc, k.aoff[m], soff, MSM_EPI_AUX);
#endif
                k.anyv[m] |= w ^ 0x01010101u;
            } else if constexpr (NT == 2) {
#ifndef MSM_EPI_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w, k.arsrc, k.aoff[m], soff, MSM_EPI_AUX);
#endif
                k.anyv[m] |= w ^ 0x0101u;
            } else {
#ifndef MSM_EPI_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)w, k.arsrc, k.aoff[m], soff, MSM_EPI_AUX);
#endif
                k.anyv[m] |= w ^ 0x01u;
            }
        }
    }
}

// flags collected by the fast path -> LDS (the generic path writes LDS directly); rows / lanes that never stored keep 0
template <int POOL, bool WRITE, int NC>
__device__ __forceinline__ void mask_epi_flush(const MaskEpiConst<POOL, WRITE, NC>& k, int* __restrict__ any_flags, int lj, bool r4 = false) {
    if constexpr (POOL != 0) {
#pragma unroll
        for (int m = 0; m < QB; ++m)
            if (k.anyv[m] != 0u && k.aoff[m] != 0xF0000000u) any_flags[m * 16 + ((r4 && m == QB - 1) ? (lj & 3) : lj)] = 1;
    }
}

// POOL: 0 = no attention mask; 1 = mask at the resolution of the logits (single-level decoder,
// meanshiftformer_transformer_decoder.py:1012-1035 with target size == mask size: interpolate is the
// identity); 2/4/8 = 2x2-tap average of a bilinear downsample by that factor.
// D: depth of the feature prefetch ring in groups of KU k-steps (G = C / (4 KU) must be a multiple of D).
template <int POOL, bool WRITE, int NC, int D, bool R4 = false>
__global__ __launch_bounds__(MW * 64) void mask_logits_kernel(const float* __restrict__ emb, const float* __restrict__ feat,
                                                          float* __restrict__ mask_out, uint8_t* __restrict__ attn_out,
                                                          int32_t* __restrict__ row_any, int Q, int C, int H, int W,
                                                          int th, int tw, int ypar, int n_rowpairs, int rp_step,
                                                          int rp_first, int feat_bytes, int64_t emb_ld,
                                                          const float* __restrict__ qbias, int64_t qbias_ld) {
    extern __shared__ __attribute__((aligned(16))) float Es[];   // [QCH][C + 2] embeddings, then [QCH] per-query biases
    constexpr int TW = 16 * NC;      // tile width in columns
    constexpr int NA = 2 * NC;       // accumulator pixel blocks: [row (top,bottom)][cc]
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
    MASK_TS(0)

    const int ctiles = (W + TW - 1) / TW;
    const int ntiles = n_rowpairs * ctiles;
    // buffer descriptor over this image's feature map, held in SGPRs (feat_bytes = C*H*W*4 from the host: stays scalar)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(feat + (int64_t)b * C * HW), 0, feat_bytes, 0x00020000);

    // Tile schedule: full rounds go to all MW waves of every workgroup; the leftover tiles go first to waves 0..3
    // (one per SIMD) of every workgroup, then to waves 4..7, and so on, so no SIMD gets two leftover tiles
    // while another gets none (waves w, w+4, w+8, ... share a SIMD).
    const int slots = gridDim.x * MW;
    const int full_rounds = ntiles / slots;
    const int left = ntiles - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int my_tiles = full_rounds + (left_slot < left ? 1 : 0);
    auto tile_of = [&](int it) {
        return (it < full_rounds) ? it * slots + (int)blockIdx.x * MW + wave : full_rounds * slots + left_slot;
    };
    // per-lane byte offsets of this lane's pixels in k-row `lq` for a tile; the k-group part of the address is
    // wave-uniform and travels in the buffer instruction's SGPR soffset, so the loads need no per-lane 64-bit address
    // arithmetic at all
    auto tile_voffs = [&](int t, unsigned& vtop, unsigned& vbot) {
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);  // may be -1 (odd pairing): clamp loads
        const int ybot = ytop + 1;                               // may be H
        const int c = ct * TW + PixMap<POOL, NC>::load_col(lj);  // the column(s) this lane feeds into A row lj
        const int cl = c < W ? c : 0;                            // W is a multiple of NC; lanes beyond the map re-read column 0
        vtop = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 4);
        vbot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 4);
    };
    // K loop: groups of KU k-steps through a ring of D register buffers that runs ACROSS tiles: as soon as the MFMAs of a
    // group are issued its buffer is refilled with the group D ahead -- of the next tile once this one runs out --, pinned
    // there with sched_barrier (left alone, hipcc sinks the loads behind the MFMAs and waits at once); no buffer copies, so
    // the only vmcnt waits are the counted ones at each buffer's first use.  D - 1 groups of MFMAs (D = 4: 2.2 us at the
    // full MFMA rate) cover a load's latency: with D = 2 a wave that had the MFMA pipe to itself -- its sibling on the SIMD
    // in its epilogue -- ran at half rate (in-kernel timestamps), so the two could not take turns.
    Cols<NC> tR[D][KU], bR[D][KU];
    auto load_group = [&](Cols<NC>(&t)[KU], Cols<NC>(&bt)[KU], int kbase, unsigned vtop, unsigned vbot) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const unsigned soff = (unsigned)(kbase + u * 4) * (unsigned)HW * 4u;
            t[u] = ld_cols<NC>(rsrc, vtop, soff);
            bt[u] = ld_cols<NC>(rsrc, vbot, soff);
        }
    };
    // stage this chunk of mask_embed (rows >= Q are zero): its loads are issued FIRST, then the first tile's feature loads
    // (memory returns in order: the LDS writes below wait for the embedding rows only, the feature loads stay in flight
    // behind them while the chunk is staged)
    const float* eb = emb + ((int64_t)b * Q + q0) * emb_ld;
    const int c4n = C >> 2;
    const bool colwise = (MW * 64) % c4n == 0 && 4 * ((MW * 64) / c4n) >= QCH;   // C = 64, 128, 256: a thread keeps its column, <= 4 rows
    const int rpp = colwise ? (MW * 64) / c4n : 1, r0 = colwise ? tid / c4n : 0, c4 = colwise ? (tid - r0 * c4n) * 4 : 0;
    float4 ev[4];
    if (colwise) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + i * rpp;
            ev[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < QCH && q0 + r < Q) ev[i] = *reinterpret_cast<const float4*>(eb + (int64_t)r * emb_ld + c4);
        }
    }
    unsigned voff_top = 0, voff_bot = 0;
    tile_voffs(tile_of(0), voff_top, voff_bot);       // my_tiles == 0: tile index beyond the schedule, clamped addresses
    if (my_tiles > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_group(tR[d], bR[d], d * (4 * KU), voff_top, voff_bot);
    }
    if (colwise) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + i * rpp;
            if (r < QCH) {
                float2* d = reinterpret_cast<float2*>(&Es[r * SE + c4]);
                d[0] = make_float2(ev[i].x, ev[i].y);
                d[1] = make_float2(ev[i].z, ev[i].w);
            }
        }
    } else {
        for (int idx = tid; idx < QCH * c4n; idx += MW * 64) {
            const int r = idx / c4n, cc4 = (idx - r * c4n) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)r * emb_ld + cc4);
            float2* d = reinterpret_cast<float2*>(&Es[r * SE + cc4]);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    }
    float* qb = Es + QCH * SE;
    int* any_flags = reinterpret_cast<int*>(qb + QCH);
    for (int r = tid; r < QCH; r += MW * 64) {
        qb[r] = (qbias && q0 + r < Q) ? qbias[((int64_t)b * Q + q0 + r) * qbias_ld] : 0.f;
        any_flags[r] = 0;
    }
    __syncthreads();
    MASK_TS(1)

    // per-lane constants: the bias of the lane's query of every block as a ready-made accumulator tuple (the first MFMA of a
    // tile takes it as its C operand: no accumulator initialisation instructions), the store offsets of the fast epilogue
    // (NC == 1 only: with 2 x 32 tiles the 28 registers do not fit next to 112 accumulators, the tuples are rebuilt per tile)
    constexpr int NB4 = NC == 1 ? QB : 1;
    f32x4 bias4[NB4];
    if constexpr (NC == 1) {
#pragma unroll
        for (int m = 0; m < QB; ++m) {
            float q_b = qb[m * 16 + lj];
            if (R4 && m == QB - 1) q_b = lq == 0 ? qb[m * 16 + (lj & 3)] : 0.f;   // 4-query block: the four channel classes lq are
                                                                                  // summed at the end, the bias enters once
            bias4[m] = f32x4{q_b, q_b, q_b, q_b};
        }
    }
    static_assert(!R4 || NC == 1, "the 4-query block exists for 2 x 16 tiles only");
    MaskEpiConst<POOL, WRITE, NC> epi;
    mask_epi_init<POOL, WRITE, NC>(epi, mask_out, attn_out, b, Q, q0, H, W, th, tw, lj, lq, R4);

    const int G = C / (4 * KU);          // even and >= 2: C is a multiple of 32
    // (Reading the mask_embed fragments of a group one group ahead -- two register sets -- measured no gain: the sibling
    // wave of the SIMD covers that latency.)
    // Two waves share a SIMD (w and w + 4).  Left alone they run in lockstep -- both in their K loop (each at half the MFMA
    // rate), then both in their epilogue with the MFMA pipe idle (in-kernel timestamps: 1.6 - 3.5 us of epilogue per 6 us
    // K loop).  Priority to waves 0..3 breaks the symmetry: the favoured wave's K loop runs at the full rate, its sibling
    // fills the pipe while it is in its epilogue, and from then on the two alternate.
#ifndef MSM_MASK_NOPRIO
    if (wave < 4) __builtin_amdgcn_s_setprio(3);
#endif
    for (int it = 0; it < my_tiles; ++it) {
        const int t = tile_of(it);
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);
        const int ybot = ytop + 1;
        const int c0 = ct * TW;
        unsigned nvoff_top, nvoff_bot;
        tile_voffs(tile_of(min(it + 1, my_tiles - 1)), nvoff_top, nvoff_bot);   // the last tile re-reads its own first group

        f32x4 acc[QB][NA];
        auto compute_group = [&](auto first, const Cols<NC>(&t_)[KU], const Cols<NC>(&bt)[KU], int kbase) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float* er = &Es[lj * SE + kbase + u * 4 + lq];
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    if constexpr (R4) {
                        if (m == QB - 1) {
                            // Q = 96 + 4: the last block holds four queries.  v_mfma_f32_4x4x1_16b_f32 = sixteen 4 x 4 x 1 products
                            // (block = lane / 4; 8 cycles instead of 32): with this A operand block (lq, lj >> 2) is the four
                            // pixels 4 (lj >> 2) .. + 3 at channel 4 u + lq, so B = mask_embed[96 + (lj & 3)][4 u + lq] gives
                            // lane (lq, lj) the partial sum of channel class lq for query 96 + (lj & 3), pixels 4 (lj >> 2) + reg
                            const float a4 = Es[(m * 16 + (lj & 3)) * SE + kbase + u * 4 + lq];
                            if (decltype(first)::value && u == 0) {
                                acc[m][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(t_[u].v[0], a4, bias4[m], 0, 0, 0);
                                acc[m][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bt[u].v[0], a4, bias4[m], 0, 0, 0);
                            } else {
                                acc[m][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(t_[u].v[0], a4, acc[m][0], 0, 0, 0);
                                acc[m][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bt[u].v[0], a4, acc[m][1], 0, 0, 0);
                            }
                            continue;
                        }
                    }
                    const float a = er[m * 16 * SE];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        if (decltype(first)::value && u == 0) {                          // D[pixel][query], C operand = the query's bias
                            f32x4 c4;
                            if constexpr (NC == 1) {
                                c4 = bias4[m];
                            } else {
                                const float q_b = qb[m * 16 + lj];
                                c4 = f32x4{q_b, q_b, q_b, q_b};
                            }
                            acc[m][cc] = mfma16(t_[u].v[cc], a, c4);
                            acc[m][NC + cc] = mfma16(bt[u].v[cc], a, c4);
                        } else {
                            acc[m][cc] = mfma16(t_[u].v[cc], a, acc[m][cc]);
                            acc[m][NC + cc] = mfma16(bt[u].v[cc], a, acc[m][NC + cc]);
                        }
                    }
                }
            }
        };
        // straight-line bodies: nothing for LLVM to sink the loads into; the first D groups are peeled (bias as C operand)
        auto ring_pass = [&](auto first, int gb) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int g = gb + d;
                if (d == 0) compute_group(first, tR[d], bR[d], g * (4 * KU));
                else compute_group(std::false_type{}, tR[d], bR[d], g * (4 * KU));
                __builtin_amdgcn_sched_barrier(0);
                const bool more = g + D < G;                                          // wave-uniform
                load_group(tR[d], bR[d], (more ? g + D : g + D - G) * (4 * KU), more ? voff_top : nvoff_top, more ? voff_bot : nvoff_bot);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        ring_pass(std::true_type{}, 0);
        for (int gb = D; gb < G; gb += D) ring_pass(std::false_type{}, gb);
        voff_top = nvoff_top;
        voff_bot = nvoff_bot;
        MASK_TS(2 + 3 * it)
#ifdef MSM_MASK_TS
        {
            float dep = acc[QB - 1][NA - 1][3];                   // result of the tile's last MFMA: the move issues once it is complete
            asm volatile("v_mov_b32 %0, %0" : "+v"(dep));
            MASK_TS(3 + 3 * it)
        }
#endif
        const bool inside = c0 + TW <= W;                                             // wave-uniform
        if constexpr (R4) {
            // 4-query block: the attention-mask bits ALWAYS come from tap sums formed on the per-class partials and reduced
            // afterwards -- the same arithmetic whether the launch also writes the logits (aux outputs) or not, so the two
            // modes stay bit-identical; the written logits are the fully reduced accumulators.  (Host: every tile inside.)
            mask_tile_epilogue_fast<POOL, WRITE, NC, false, true, true>(acc, epi, H, W, tw, ytop, ybot, c0);
            if constexpr (WRITE) {
#pragma unroll
                for (int row = 0; row < 2; ++row)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[QB - 1][row][r];
                        v = sum_lane_rows(v);
                        acc[QB - 1][row][r] = v;
                    }
            }
        }
        if constexpr (WRITE) {
            static_assert(NC == 1, "mask writes take 2 x 16 tiles (see mask_tile_epilogue_fast)");
            if (inside && epi.write_fast) mask_tile_epilogue_fast<POOL, WRITE, NC, true, false>(acc, epi, H, W, tw, ytop, ybot, c0);
            else mask_tile_epilogue<POOL, WRITE, NC, true, false, R4>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
        }
        if constexpr (!R4) {
            if (inside && epi.attn_fast) mask_tile_epilogue_fast<POOL, WRITE, NC, false, true>(acc, epi, H, W, tw, ytop, ybot, c0);
            else mask_tile_epilogue<POOL, WRITE, NC, false, true>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
        }
        MASK_TS(4 + 3 * it)
    }
    if constexpr (POOL != 0) {
        mask_epi_flush<POOL, WRITE, NC>(epi, any_flags, lj, R4);
        __syncthreads();
        for (int r = tid; r < QCH; r += MW * 64)
            if (any_flags[r] && q0 + r < Q) row_any[(int64_t)b * Q + q0 + r] = 1;
    }
}

// Same product with bf16 operands and fp32 accumulation (v_mfma_f32_16x16x16_bf16): at 2.5 PFLOP/s the 7.9 GFLOP of a
// launch are ~4 us of MFMA, so the step becomes a stream over the feature map -- HBM-bound (AI 71.6 FLOP/B
// against a bf16 ridge of ~312).  The features are kept in a channel-quad packed layout [B][C/4][HW][4] bf16
// (msm_pack_mask_features_bf16): the MFMA B operand of lane (pixel lj, k-group lq) is then ONE 8-byte load and the 16
// pixels of a k-group are a 128-byte line.  A whole tile's operands (32 loads) are requested while the previous tile
// is being multiplied.  Tile shape, schedule and the fused attention-mask epilogue are those of the fp32 kernel.
constexpr int BKS = 16;            // 16-channel k-steps held per tile: C <= 256

__device__ __forceinline__ unsigned short f2bf(float x) {   // round to nearest even
    const unsigned int u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// NKS: 16-channel k-steps held per tile -- 4 for the folded step (C = 64: 8 loads and 56 MFMAs per tile; the generic 16 would
// request every clamped slot again, 32 loads per tile of which 8 are needed), BKS otherwise (C <= 256, runtime count)
// F16 (flag MSM_MASK_F16, precision "f16"): both operands hold IEEE halves (msm_pack_mask_features_f16; mask_embed converted with a
// clamp when it is staged) and the product runs on v_mfma_f32_16x16x16_f16 -- same layouts, same rate, 2^-12 instead of 2^-9 roundings
// on the operands of the one step whose sign IS the output.
__device__ __forceinline__ unsigned short f2h(float x) {     // round to nearest even, clamped to the half range
    return __builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f));
}
typedef _Float16 f16x4m __attribute__((ext_vector_type(4)));
template <int POOL, bool WRITE, int NKS, bool F16 = false>
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
        if constexpr (F16) {
            pk.x = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16);
            pk.y = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
        } else {
            pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
            pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
        }
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
        u32x2 t[NKS], bt[NKS];
    };
    auto load_tile = [&](int t, TileRegs& r) {
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step), ybot = ytop + 1;
        const int c = ct * 16 + PixMap<POOL, 1>::load_col(lj);
        const int cl = c < W ? c : 0;
        // packed element (k-quad, pixel): 8 bytes at ((kq * HW) + pixel) * 8; the k-step part travels in soffset
        const unsigned voff_top = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 8);
        const unsigned voff_bot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 8);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned soff = (unsigned)min(ks, nks - 1) * 4u * (unsigned)HW * 8u;   // clamped: C < 256 re-reads, never faults
            r.t[ks] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_top, soff, 0);
            r.bt[ks] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_bot, soff, 0);
        }
    };
    MaskEpiConst<POOL, WRITE, 1> epi;
    mask_epi_init<POOL, WRITE, 1>(epi, mask_out, attn_out, b, Q, q0, H, W, th, tw, lj, lq);
    TileRegs cur, nxt;
    if (my_tiles > 0) load_tile(tile_of(0), cur);
    for (int it = 0; it < my_tiles; ++it) {
        const int t = tile_of(it);
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);
        const int ybot = ytop + 1;
        const int c0 = ct * 16;
        load_tile(tile_of(min(it + 1, my_tiles - 1)), nxt);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[QB][2];
#pragma unroll
        for (int m = 0; m < QB; ++m) { const float q_b = qb[m * 16 + lj]; acc[m][0] = acc[m][1] = f32x4{q_b, q_b, q_b, q_b}; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nks) {                                              // wave-uniform
                const unsigned short* er = &Eb[lj * SEb + ks * 16 + lq * 4];
                const bf16x4 bt_ = __builtin_bit_cast(bf16x4, cur.t[ks]);
                const bf16x4 bb_ = __builtin_bit_cast(bf16x4, cur.bt[ks]);
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    const bf16x4 a = __builtin_bit_cast(bf16x4, *reinterpret_cast<const u32x2*>(er + m * 16 * SEb));
                    if constexpr (F16) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4m, bt_), __builtin_bit_cast(f16x4m, a), acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4m, bb_), __builtin_bit_cast(f16x4m, a), acc[m][1], 0, 0, 0);
                    } else {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bt_, a, acc[m][0], 0, 0, 0);      // D[pixel][query]
                        acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bb_, a, acc[m][1], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the fast epilogue of the fp32 kernel (precomputed per-lane offsets, one store per query block) for tiles inside the map
        const bool inside = c0 + 16 <= W;                                             // wave-uniform
        if constexpr (WRITE) {
            if (inside && epi.write_fast) mask_tile_epilogue_fast<POOL, WRITE, 1, true, false>(acc, epi, H, W, tw, ytop, ybot, c0);
            else mask_tile_epilogue<POOL, WRITE, 1, true, false>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
        }
        if (inside && epi.attn_fast) mask_tile_epilogue_fast<POOL, WRITE, 1, false, true>(acc, epi, H, W, tw, ytop, ybot, c0);
        else mask_tile_epilogue<POOL, WRITE, 1, false, true>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            cur.t[ks] = nxt.t[ks];
            cur.bt[ks] = nxt.bt[ks];
        }
    }
    if constexpr (POOL != 0) {
        mask_epi_flush<POOL, WRITE, 1>(epi, any_flags, lj);
        __syncthreads();
        for (int r = tid; r < QCH; r += MW * 64)
            if (any_flags[r] && q0 + r < Q) row_any[(int64_t)b * Q + q0 + r] = 1;
    }
}

// fp32 NCHW [B][C][HW] -> bf16 channel-quad packed [B][C/4][HW][4]
template <bool F16>
__global__ __launch_bounds__(256) void pack_mask_features_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                                      int64_t total, int C4, int HW) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;                 // b * C4 + c4
        const float* src = in + (r * 4) * HW + p;
        u32x2 pk;
        if constexpr (F16) {
            pk.x = (unsigned)f2h(src[0]) | ((unsigned)f2h(src[HW]) << 16);
            pk.y = (unsigned)f2h(src[2 * (int64_t)HW]) | ((unsigned)f2h(src[3 * (int64_t)HW]) << 16);
        } else {
            pk.x = (unsigned)f2bf(src[0]) | ((unsigned)f2bf(src[HW]) << 16);
            pk.y = (unsigned)f2bf(src[2 * (int64_t)HW]) | ((unsigned)f2bf(src[3 * (int64_t)HW]) << 16);
        }
        *reinterpret_cast<u32x2*>(out + i * 4) = pk;
    }
}

// ---- fp32-accurate mask step on the bf16 matrix pipe (precision mode f32_split; folded form, C = 64) -------------------------
// The construction of enc_block_split.hip applied to the mask step: both operands as exact three-term bf16
// splits x = h + m + l, a product as the six cross terms of weight >= 2^-18 -- l.h, h.l, m.m, m.h, h.m, h.h, every
// bf16 x bf16 product exact in the fp32 accumulator -- on v_mfma_f32_16x16x32_bf16: 12 MFMAs of 16 cycles per 16-pixel x
// 16-query block and image row instead of 16 fp32 MFMAs of 32 cycles.  The 64-channel activation is split ONCE per forward
// (msm_pack_mask_features_split: [B][3 terms][C/8][HW][8] bf16, a lane's A operand = one 16-byte load), mask_embed when a
// workgroup stages its query chunk (three bf16 copies in LDS).  Tile shape, schedule, per-query bias as the accumulators'
// initial value and the fused attention-mask epilogue are those of the kernels above.
constexpr int SPK = 2;             // K = 32 k-steps: C = 64
template <int POOL, bool WRITE>
__global__ __launch_bounds__(MW * 64) void mask_logits_split_kernel(const float* __restrict__ emb, const unsigned short* __restrict__ featp,
                                                                float* __restrict__ mask_out, uint8_t* __restrict__ attn_out,
                                                                int32_t* __restrict__ row_any, int Q, int C, int H, int W, int th,
                                                                int tw, int ypar, int n_rowpairs, int rp_step, int rp_first,
                                                                int feat_bytes, int64_t emb_ld, const float* __restrict__ qbias,
                                                                int64_t qbias_ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned short Esp[];   // [3 terms][QCH][C + 8] bf16, then [QCH] fp32 biases, [QCH] flags
    constexpr int SE = 64 + 8;         // 144-byte rows: 16-byte aligned b128 reads
    const int b = blockIdx.z, qc = blockIdx.y;
    const int q0 = qc * QCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int HW = H * W;

    const float* eb = emb + ((int64_t)b * Q + q0) * emb_ld;
    float* qb = reinterpret_cast<float*>(Esp + 3 * QCH * SE);
    int* any_flags = reinterpret_cast<int*>(qb + QCH);
    for (int r = tid; r < QCH; r += MW * 64) {
        qb[r] = (qbias && q0 + r < Q) ? qbias[((int64_t)b * Q + q0 + r) * qbias_ld] : 0.f;
        any_flags[r] = 0;
    }
    for (int idx = tid; idx < QCH * (64 / 4); idx += MW * 64) {
        const int r = idx / 16, c4 = (idx - r * 16) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)r * emb_ld + c4);
        const Split3 sp = split3(v.x, v.y, v.z, v.w);
        *reinterpret_cast<u32x2*>(&Esp[(0 * QCH + r) * SE + c4]) = __builtin_bit_cast(u32x2, sp.h);
        *reinterpret_cast<u32x2*>(&Esp[(1 * QCH + r) * SE + c4]) = __builtin_bit_cast(u32x2, sp.m);
        *reinterpret_cast<u32x2*>(&Esp[(2 * QCH + r) * SE + c4]) = __builtin_bit_cast(u32x2, sp.l);
    }
    __syncthreads();

    const int ctiles = (W + 15) / 16;
    const int ntiles = n_rowpairs * ctiles;
    const uint64_t fbu = (uint64_t)(featp + (int64_t)b * 3 * C * HW);
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
        u32x4b t[SPK][3], bt[SPK][3];          // [k-step][term]: top / bottom image row
    };
    auto load_tile = [&](int t, TileRegs& r) {
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step), ybot = ytop + 1;
        const int c = ct * 16 + PixMap<POOL, 1>::load_col(lj);
        const int cl = c < W ? c : 0;
        // packed element (term, k-octet kg = ks*4 + lq, pixel): 16 bytes at ((term * C/8 + kg) * HW + pixel) * 16
        const unsigned voff_top = (unsigned)(((int64_t)lq * HW + (int64_t)max(ytop, 0) * W + cl) * 16);
        const unsigned voff_bot = (unsigned)(((int64_t)lq * HW + (int64_t)min(ybot, H - 1) * W + cl) * 16);
#pragma unroll
        for (int ks = 0; ks < SPK; ++ks)
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                const unsigned soff = (unsigned)(tm * (64 / 8) + ks * 4) * (unsigned)HW * 16u;
                r.t[ks][tm] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_top, soff, 0);
                r.bt[ks][tm] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bot, soff, 0);
            }
    };
    MaskEpiConst<POOL, WRITE, 1> epi;
    mask_epi_init<POOL, WRITE, 1>(epi, mask_out, attn_out, b, Q, q0, H, W, th, tw, lj, lq);
    TileRegs cur, nxt;
    if (my_tiles > 0) load_tile(tile_of(0), cur);
    for (int it = 0; it < my_tiles; ++it) {
        const int t = tile_of(it);
        const int rp = t / ctiles, ct = t - rp * ctiles;
        const int ytop = ypar + 2 * (rp_first + rp * rp_step);
        const int ybot = ytop + 1;
        const int c0 = ct * 16;
        load_tile(tile_of(min(it + 1, my_tiles - 1)), nxt);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[QB][2];
#pragma unroll
        for (int m = 0; m < QB; ++m) { const float q_b = qb[m * 16 + lj]; acc[m][0] = acc[m][1] = f32x4{q_b, q_b, q_b, q_b}; }
#pragma unroll
        for (int ks = 0; ks < SPK; ++ks) {
            bf16x8 ft[3], fb[3];
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                ft[tm] = __builtin_bit_cast(bf16x8, cur.t[ks][tm]);
                fb[tm] = __builtin_bit_cast(bf16x8, cur.bt[ks][tm]);
            }
            // query blocks in pairs: the six terms of a pair's four accumulators interleave (no back-to-back dependent MFMAs)
#pragma unroll
            for (int m0 = 0; m0 < QB; m0 += 2) {
                bf16x8 e[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm)
                        if (m0 + j < QB)
                            e[j][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(&Esp[(tm * QCH + (m0 + j) * 16 + lj) * SE + ks * 32 + lq * 8]));
                // (feature term, embedding term) of the six products, small ones first: l.h, h.l, m.m, m.h, h.m, h.h
                constexpr int FT[6] = {2, 0, 1, 1, 0, 0}, ET[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (m0 + j < QB) {
                            acc[m0 + j][0] = mfma_bf16k32(ft[FT[p]], e[j][ET[p]], acc[m0 + j][0]);      // D[pixel][query]
                            acc[m0 + j][1] = mfma_bf16k32(fb[FT[p]], e[j][ET[p]], acc[m0 + j][1]);
                        }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool inside = c0 + 16 <= W;                                             // wave-uniform
        if constexpr (WRITE) {
            if (inside && epi.write_fast) mask_tile_epilogue_fast<POOL, WRITE, 1, true, false>(acc, epi, H, W, tw, ytop, ybot, c0);
            else mask_tile_epilogue<POOL, WRITE, 1, true, false>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
        }
        if (inside && epi.attn_fast) mask_tile_epilogue_fast<POOL, WRITE, 1, false, true>(acc, epi, H, W, tw, ytop, ybot, c0);
        else mask_tile_epilogue<POOL, WRITE, 1, false, true>(acc, mask_out, attn_out, any_flags, b, Q, q0, H, W, th, tw, ytop, ybot, c0, lj, lq);
        cur = nxt;
    }
    if constexpr (POOL != 0) {
        mask_epi_flush<POOL, WRITE, 1>(epi, any_flags, lj);
        __syncthreads();
        for (int r = tid; r < QCH; r += MW * 64)
            if (any_flags[r] && q0 + r < Q) row_any[(int64_t)b * Q + q0 + r] = 1;
    }
}

// fp32 NCHW [B][C][HW] -> the exact three-term bf16 split [B][3][C/8][HW][8]: x = h + m + l (bf16.h split3)
__global__ __launch_bounds__(256) void pack_mask_features_split_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                                       int64_t total, int C8, int HW) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;                 // b * C8 + kg
        const int64_t bimg = r / C8;
        const int kg = (int)(r - bimg * C8);
        const float* src = in + (r * 8) * HW + p;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * HW];
        const Split3 a = split3(v[0], v[1], v[2], v[3]), c = split3(v[4], v[5], v[6], v[7]);
        const Split3x8 s8 = join(a, c);
        unsigned short* dst = out + ((bimg * 3 * C8 + kg) * (int64_t)HW + p) * 8;
        *reinterpret_cast<u32x4b*>(dst) = __builtin_bit_cast(u32x4b, s8.h);
        *reinterpret_cast<u32x4b*>(dst + (int64_t)C8 * HW * 8) = __builtin_bit_cast(u32x4b, s8.m);
        *reinterpret_cast<u32x4b*>(dst + 2 * (int64_t)C8 * HW * 8) = __builtin_bit_cast(u32x4b, s8.l);
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
    // (ties: with few rounds the narrow tile wins -- B = 2: 12.8 against 19.3 us, its tiles spread over twice the SIMDs --,
    // with many the wide one does -- B = 16: 40.6 against 43.5 us, 1280x960 with 300 queries: 48.8 against 57.4)
    int nc = (cost16 * 1.04 < cost32 || (cost16 <= cost32 && cost32 < 3.0)) ? 1 : 2;
    if (const int o = opt(MSM_OPT_MASK_NC); o != MSM_OPT_AUTO) nc = o == 1 ? 1 : 2;
    if (mask_out) nc = 1;      // launches that write the logits take 2 x 16 tiles: their float4 stores are accumulator tuples
    const int ctiles = cdiv(W, 16 * nc);
    const int ntiles = n_rowpairs * ctiles;
    // persistent-ish grid: enough workgroups per (image, chunk) to cover the chip once
    int wg_per = cdiv(ntiles, MW);
    const size_t lds = sizeof(float) * ((size_t)QCH * (C + 2) + 2 * QCH);
    const int target = cdiv(256, B * qchunks);
    if (wg_per > target) wg_per = max(target, 1);
    dim3 grid(wg_per, qchunks, B), block(MW * 64);
    typedef void (*kern_t)(const float*, const float*, float*, uint8_t*, int32_t*, int, int, int, int, int, int, int, int, int, int, int, int64_t,
                           const float*, int64_t);
    kern_t kern;
    const bool wr = mask_out != nullptr;
    // Q = 96 + 4 (the 100 queries of every shipped configuration): the last query block on the 4x4x1 MFMA (8 cycles per k-step
    // and image row instead of 32): 10.7 % less matrix time per tile.  Needs 2 x 16 tiles that all take the fast epilogue.
    // MSM_OPT_MASK_KERNEL = 5 selects the kernel without that block (the one tested fallback; round-2 experiments --
    // mask_embed in registers with one wave per SIMD, a four-group prefetch ring, a software-pipelined block-major epilogue --
    // were all measured slower, DESIGN.md section 5, and left the library in round 3).
    const bool r4 = nc == 1 && Q == 100 && W % 16 == 0 && (int64_t)Q * H * W * 4 < 0xF0000000ll && opt(MSM_OPT_MASK_KERNEL) != 5 &&
                    (pool == 0 || pool == 2 || pool == 4 || pool == 8);
#define MASK_PICK(P)                                                                                                   \
    (wr ? (r4 ? (kern_t)mask_logits_kernel<P, true, 1, 2, true> : (kern_t)mask_logits_kernel<P, true, 1, 2>)             \
        : (nc == 2 ? (kern_t)mask_logits_kernel<P, false, 2, 2>                                                          \
                   : (r4 ? (kern_t)mask_logits_kernel<P, false, 1, 2, true> : (kern_t)mask_logits_kernel<P, false, 1, 2>)))
    switch (pool) {
        case 0: kern = r4 ? (kern_t)mask_logits_kernel<0, true, 1, 2, true> : (kern_t)mask_logits_kernel<0, true, 1, 2>; break;
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

#ifdef MSM_MASK_TS
extern "C" int msm_debug_mask_ts(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(msm::g_mask_ts), sizeof(unsigned long long) * 256 * 8 * 16);
}
#endif

static int pack_mask_features_16(const char* who, const float* mask_feat, uint16_t* packed, int B, int C, int HW, bool f16, void* stream) {
    MSM_REQUIRE(mask_feat && packed, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "%s: C=%d must be a multiple of 4", who, C);
    MSM_REQUIRE((((uintptr_t)packed) & 7) == 0, "%s: packed must be 8-byte aligned", who);
    const int64_t total = (int64_t)B * (C / 4) * HW;
    const dim3 grid((unsigned)min((int64_t)4096, (total + 255) / 256));
    if (f16) hipLaunchKernelGGL(pack_mask_features_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, mask_feat, packed, total, C / 4, HW);
    else hipLaunchKernelGGL(pack_mask_features_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, mask_feat, packed, total, C / 4, HW);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
extern "C" int msm_pack_mask_features_bf16(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream) {
    return pack_mask_features_16("msm_pack_mask_features_bf16", mask_feat, packed, B, C, HW, false, stream);
}
extern "C" int msm_pack_mask_features_f16(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream) {
    return pack_mask_features_16("msm_pack_mask_features_f16", mask_feat, packed, B, C, HW, true, stream);
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
    const bool f16 = (flags & MSM_MASK_F16) != 0;
#define MASKB_PICK_T(P, WR, NK) (f16 ? (kern_t)mask_logits_bf16_kernel<P, WR, NK, true> : (kern_t)mask_logits_bf16_kernel<P, WR, NK, false>)
#define MASKB_PICK(P) (C <= 64 ? (wr ? MASKB_PICK_T(P, true, 4) : MASKB_PICK_T(P, false, 4)) : (wr ? MASKB_PICK_T(P, true, BKS) : MASKB_PICK_T(P, false, BKS)))
    switch (pool) {
        case 0: kern = C <= 64 ? MASKB_PICK_T(0, true, 4) : MASKB_PICK_T(0, true, BKS); break;
        case 1: kern = MASKB_PICK(1); break;
        case 2: kern = MASKB_PICK(2); break;
        case 4: kern = MASKB_PICK(4); break;
        default: kern = MASKB_PICK(8); break;
    }
#undef MASKB_PICK
#undef MASKB_PICK_T
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, mask_embed, mask_feat_packed, mask_out, attn_out, row_any, Q, C, H, W, th, tw, ypar,
                       n_rowpairs, rp_step, rp_first, (int)((int64_t)C * H * W * 2), embed_ld, qbias, qbias_ld);
    MSM_CHECK_LAUNCH("msm_mask_logits_bf16_fwd");
    return MSM_OK;
}

extern "C" int msm_pack_mask_features_split(const float* mask_feat, uint16_t* packed, int B, int C, int HW, void* stream) {
    MSM_REQUIRE(mask_feat && packed, "msm_pack_mask_features_split: null pointer");
    MSM_REQUIRE(B > 0 && C == 64 && HW > 0, "msm_pack_mask_features_split: C=%d, the split mask step takes the 64-channel folded form", C);
    MSM_REQUIRE((((uintptr_t)packed) & 15) == 0, "msm_pack_mask_features_split: output must be 16-byte aligned");
    const int64_t total = (int64_t)B * (C / 8) * HW;
    hipLaunchKernelGGL(pack_mask_features_split_kernel, dim3((unsigned)min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mask_feat, packed, total, C / 8, HW);
    MSM_CHECK_LAUNCH("msm_pack_mask_features_split");
    return MSM_OK;
}

extern "C" int msm_mask_logits_split_fwd(const float* mask_embed, const uint16_t* mask_feat_split, float* mask_out,
                                         uint8_t* attn_out, int32_t* row_any, int B, int Q, int C, int H, int W, int th, int tw,
                                         int flags, int64_t embed_ld, const float* qbias, int64_t qbias_ld, void* stream) {
    const int sparse = flags & MSM_MASK_SPARSE;
    MSM_REQUIRE(mask_embed && mask_feat_split, "msm_mask_logits_split_fwd: null input");
    MSM_REQUIRE(mask_out || attn_out, "msm_mask_logits_split_fwd: nothing to produce");
    MSM_REQUIRE(B > 0 && Q > 0 && H > 1 && W > 1, "msm_mask_logits_split_fwd: bad sizes");
    MSM_REQUIRE(C == 64, "msm_mask_logits_split_fwd: C=%d, only the 64-channel folded form", C);
    if (int rc = mask_embed_check("msm_mask_logits_split_fwd", mask_embed, embed_ld, qbias, qbias_ld, C)) return rc;
    MSM_REQUIRE(W % 2 == 0 && H % 2 == 0, "msm_mask_logits_split_fwd: H=%d W=%d must be even", H, W);
    MSM_REQUIRE((int64_t)3 * C * H * W * 2 < (int64_t)1 << 31, "msm_mask_logits_split_fwd: one image of the split features must be < 2 GiB");
    MSM_REQUIRE((((uintptr_t)mask_embed) & 15) == 0 && (((uintptr_t)mask_feat_split) & 15) == 0, "msm_mask_logits_split_fwd: misaligned pointer");
    int pool = 0;
    if (attn_out) {
        MSM_REQUIRE(row_any, "msm_mask_logits_split_fwd: row_any required with attn_out");
        MSM_REQUIRE(th > 0 && tw > 0 && H % th == 0 && W % tw == 0 && H / th == W / tw,
                    "msm_mask_logits_split_fwd: target %dx%d incompatible with %dx%d", th, tw, H, W);
        pool = H / th;
        MSM_REQUIRE(pool == 1 || pool == 2 || pool == 4 || pool == 8, "msm_mask_logits_split_fwd: pool factor %d not in {1,2,4,8}", pool);
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
    int wg_per = cdiv(ntiles, MW);
    const int target = cdiv(256, B * qchunks);
    if (wg_per > target) wg_per = max(target, 1);
    dim3 grid(wg_per, qchunks, B), block(MW * 64);
    const size_t lds = sizeof(unsigned short) * (size_t)3 * QCH * (64 + 8) + sizeof(float) * 2 * QCH;
    typedef void (*kern_t)(const float*, const unsigned short*, float*, uint8_t*, int32_t*, int, int, int, int, int, int, int, int, int, int, int,
                           int64_t, const float*, int64_t);
    const bool wr = mask_out != nullptr;
    kern_t kern;
#define MASKS_PICK(P) (wr ? (kern_t)mask_logits_split_kernel<P, true> : (kern_t)mask_logits_split_kernel<P, false>)
    switch (pool) {
        case 0: kern = (kern_t)mask_logits_split_kernel<0, true>; break;
        case 1: kern = MASKS_PICK(1); break;
        case 2: kern = MASKS_PICK(2); break;
        case 4: kern = MASKS_PICK(4); break;
        default: kern = MASKS_PICK(8); break;
    }
#undef MASKS_PICK
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, mask_embed, mask_feat_split, mask_out, attn_out, row_any, Q, C, H, W, th, tw, ypar,
                       n_rowpairs, rp_step, rp_first, (int)((int64_t)3 * C * H * W * 2), embed_ld, qbias, qbias_ld);
    MSM_CHECK_LAUNCH("msm_mask_logits_split_fwd");
    return MSM_OK;
}
