// fp32 encoder-layer tail on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 terms and a product is
// six bf16 MFMAs with fp32 accumulation (see include/msm_hip.h: msm_encoder_block_split_fwd).
//
// Same function as enc_block_kernel (enc_block.hip; reference msdeformattn.py:116-131, ops/modules/ms_deform_attn.py:95-104,
// 123).  Why: on gfx950 v_mfma_f32_16x16x4_f32 peaks at 157 TFLOP/s and the bf16 MFMAs at 2.5 PFLOP/s -- sixteen times the
// rate -- and bf16 has fp32's exponent range, so an fp32 value is x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): each residual is exact in fp32 and shrinks by 2^-9, the third term carries what is left to
// 2^-27 |x|.  A product of two such triples needs the six terms of weight >= 2^-18:
//     x w  =  x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1)  +  O(2^-27 |x w|),
// every bf16 x bf16 product is exact in the fp32 accumulator, and the dropped terms are below fp32's own rounding (2^-24).
// Six MFMAs at 1/16 of the fp32 MFMA's cost: 0.375 of the matrix time for results that deviate from float64 exactly as much
// as the fp32-MFMA kernel's do (tests/test_gpu_ops.py::test_encoder_block_split_is_fp32_accurate measures both).  Weights are
// split once per checkpoint on the host (pack_encoder_block_split), activations with five VALU instructions per pair when
// they become operands.  This is NOT the low-precision mode (MODE 1 below rounds operands to one or two bf16 terms).
//
// Register layout L -- lane (token lj, quarter lq) holds features fb*16 + lq*4 + r --,
// weights as the MFMA A operand in fragment-ordered 2-KiB blocks streamed through two LDS stages by LDS-DMA, one 16-token
// tile per wave, 4 waves per workgroup.  The MFMA is v_mfma_f32_16x16x32_bf16 (K = 32: the full-rate gfx950 instruction;
// the K = 16 form of the low-precision kernels runs at half the rate): the k index a lane feeds is free as long as both
// operands agree, so 32-wide k-group G is the lane's fragments of the 16-wide feature blocks 2G and 2G + 1 side by side --
// a block is [2 k-groups][64 lanes][8 bf16], one ds_read_b128 per operand.  A logical block is three physical ones
// (h, m, l); a stage holds 12 (24 KiB): stage 0 = output_proj (4 row blocks); an FFN stage = two hidden blocks of 16:
// [W1(q0) h, m, l | W1(q1) h, m, l | W2 h | W2 m | W2 l] where a W2 copy is 4 KiB = [4 output row blocks][64 lanes][8 bf16]
// over the 32 hidden units of the stage; then value_proj (4 row blocks) and the sampling projection, four row blocks per stage.
#include "bf16.h"
#include "common.h"

#ifndef ES_EXP
#define ES_EXP 0   // tuning builds only: 1 = weight fragments are register constants (no LDS reads), 2 = no MFMAs, 3 = no stage barriers / DMA waits
#endif

namespace msm {

constexpr int ES_C = 64;                     // d_model
constexpr int ES_BLOCK = 2048;               // bytes per physical weight block
constexpr int ES_STAGE = 12 * ES_BLOCK;      // bytes per LDS stage
constexpr int ES_NBUF = 3;                   // LDS stage buffers: a stage is requested two stages before it is used

struct EncSmallS {                           // offsets (floats) into the packed small-parameter vector (as enc_block.hip)
    int bo, g1, be1, b1, b2, g2, be2, bv, bp;
};

// the lane's operand of 32-wide k-group (or output row block) i of a physical block
__device__ __forceinline__ bf16x8 sfrag(const char* __restrict__ blk, int i, int lane) {
#if ES_EXP == 1
    return __builtin_bit_cast(bf16x8, u32x4b{(unsigned)lane * 0x10001u, (unsigned)i, 0x3f803f80u, (unsigned)(uintptr_t)blk});
#else
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(blk + (i * 64 + lane) * 16));
#endif
}
// the six significant products of one k-group, smallest first, into two accumulators (low-order terms / leading term): the
// chains of different row blocks interleave, and the small terms are summed among themselves before they meet the large one
__device__ __forceinline__ void mac6(f32x4& lo, f32x4& hi, bf16x8 wh, bf16x8 wm, bf16x8 wl, const Split3x8& x) {
    lo = mfma_bf16k32(wl, x.h, lo);
    lo = mfma_bf16k32(wh, x.l, lo);
    lo = mfma_bf16k32(wm, x.m, lo);
    lo = mfma_bf16k32(wm, x.h, lo);
    lo = mfma_bf16k32(wh, x.m, lo);
    hi = mfma_bf16k32(wh, x.h, hi);
}
// MODE 0: fp32 accuracy -- three weight copies everywhere, all six product terms.  MODE 1: the low-precision ("bf16") mode on
// the same structure -- operand roundings of the low-precision mode: the 64-wide projections w(h + m) x(h + m) without the
// m x m term, linear1 w(h) x(h + m), linear2 single operands -- with only the copies it reads in the stream, so a stage holds
// three hidden pairs or six projection row blocks and a layer is 16 stages instead of 38.
template <int MODE>
struct ESMode;
template <>
struct ESMode<0> {
    static constexpr int WC = 3, W1C = 3, W2C = 3;      // weight copies per logical block: projections, linear1, linear2
    static constexpr int HP = 1, PB = 4;                // hidden pairs (of 2 x 16) per FFN stage, projection row blocks per stage
    static constexpr unsigned PROJ = 0x3f, L1 = 0x3f, L2 = 0x3f;      // product terms of mac_term used (bit = term)
};
template <>
struct ESMode<1> {
    static constexpr int WC = 2, W1C = 1, W2C = 1;
    static constexpr int HP = 3, PB = 6;
    static constexpr unsigned PROJ = 0x38, L1 = 0x30, L2 = 0x20;      // {wm xh, wh xm, wh xh}, {wh xm, wh xh}, {wh xh}
};
// fragment i of a logical block whose h / m / l copies lie `step` bytes apart
template <int COPIES>
__device__ __forceinline__ Frag3 ld3(const char* __restrict__ blk, int step, int i, int lane) {
    Frag3 f;
    f.h = sfrag(blk, i, lane);
    f.m = COPIES > 1 ? sfrag(blk + step, i, lane) : f.h;          // copies a mode does not stream are never multiplied with
    f.l = COPIES > 2 ? sfrag(blk + 2 * step, i, lane) : f.h;
    return f;
}
// one [16 rows][64 k] logical block (physical blocks blk, blk + 1, blk + 2 = h, m, l) applied to the two k-groups of NT token
// tiles: D (layout L) = bias + W x.  Every weight fragment is read from LDS ONCE for the NT tiles of the wave: with one tile
// per wave the kernel sat on the LDS read port (a CU's waves read 288 KiB of fragments per stage round: 2304 cycles, as
// many as the MFMAs of that round take).
template <int NT, int COPIES, unsigned TERMS>
__device__ __forceinline__ void rowblock6(const char* __restrict__ blk, int lane, const Split3x8 (&xb)[NT][2], const float* __restrict__ bias, int lq,
                                          f32x4 (&out)[NT]) {
    const float4 b = *reinterpret_cast<const float4*>(bias + lq * 4);
    f32x4 hi[NT], lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hi[t] = f32x4{b.x, b.y, b.z, b.w};
        lo[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const Frag3 w = ld3<COPIES>(blk, ES_BLOCK, g, lane);
#pragma unroll
        for (int term = 0; term < 6; ++term)
            if ((TERMS >> term) & 1u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) mac_term(term, lo[t], hi[t], w, xb[t][g]);
            }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = hi[t] + lo[t];
}
__device__ __forceinline__ float relu1s(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.0e38f); }

__device__ __forceinline__ void layer_norm_Ls(float (&v)[4][4], const float* __restrict__ g, const float* __restrict__ b, int lq, float eps) {
    float s = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[fb][r];
    s = sum_lane_rows(s);
    const float mean = s * (1.0f / ES_C);
    float q = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = v[fb][r] - mean;
            q += d * d;
        }
    q = sum_lane_rows(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / ES_C) + eps);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 gg = *reinterpret_cast<const float4*>(g + fb * 16 + lq * 4);
        const float4 bb = *reinterpret_cast<const float4*>(b + fb * 16 + lq * 4);
        v[fb][0] = (v[fb][0] - mean) * rstd * gg.x + bb.x;
        v[fb][1] = (v[fb][1] - mean) * rstd * gg.y + bb.y;
        v[fb][2] = (v[fb][2] - mean) * rstd * gg.z + bb.z;
        v[fb][3] = (v[fb][3] - mean) * rstd * gg.w + bb.w;
    }
}

// LDS-DMA of 16 bytes per lane (see enc_block.hip: glds16)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// small: the fp32 vector of enc_block.hip (biases, LayerNorm parameters); everything but linear1's bias (d_ffn floats, read
// where it is) is copied to LDS.  NT = 16-token tiles per wave; tile0 = this workgroup's first tile (wave w: tiles
// tile0 + w * NT ...).
template <int NT, int MODE>
__device__ __forceinline__ void enc_block_split_body(const float* __restrict__ attn, const float* __restrict__ src,
                                                     const char* __restrict__ wstream, const float* __restrict__ small, EncSmallS so,
                                                     const float* __restrict__ pos, float* __restrict__ src_out,
                                                     float* __restrict__ value_out, float* __restrict__ proj_out, int M, int S, int nffn_stages,
                                                     int nproj_blocks, int proj_ld, float eps, int n_small, int value_heads, int tile0) {
    extern __shared__ __attribute__((aligned(16))) char wls[];      // [ES_NBUF][ES_STAGE] weight stages, then the small parameters
    float* sm = reinterpret_cast<float*>(wls + ES_NBUF * ES_STAGE);       // indexed like `small`, linear1's bias left out (gap closed)
    const int d_ffn = so.b2 - so.b1;
    for (int i = threadIdx.x; i < n_small - d_ffn; i += 256) sm[i] = small[i < so.b1 ? i : i + d_ffn];
    auto smo = [&](int off) { return sm + (off < so.b1 ? off : off - d_ffn); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    int tok[NT], tk[NT];
    bool tok_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        tok[t] = (tile0 + wave * NT + t) * 16 + lj;
        tok_ok[t] = tok[t] < M;
        tk[t] = tok_ok[t] ? tok[t] : M - 1;
    }
    const bool next = value_out != nullptr;
    using MD = ESMode<MODE>;
    const int ntail = next ? 1 + (nproj_blocks + MD::PB - 1) / MD::PB : 0;
    const int nsteps = 1 + nffn_stages + ntail;

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wls;
    auto stage_issue = [&](int s, int bufi) {                       // 24 KiB = 6 x (256 lanes x 16 B)
        const char* sb = wstream + (int64_t)s * ES_STAGE;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            glds16s(sb + i * 4096 + wave * 1024, (unsigned)lane * 16u, lds_base + (unsigned)bufi * ES_STAGE + i * 4096u + (unsigned)wave * 1024u);
    };
    // Stage s lives in buffer s % 3 and is requested at the start of stage s - 2: a stage's 96 - 192 MFMAs (0.4 - 0.75 us
    // per wave) do not cover an LDS-DMA round trip under load, two of them do (with two buffers a workgroup took ~1.8 us per
    // stage whatever it computed).  At the end of stage s the data of s + 1 must be there: everything but this stage's own
    // six DMA instructions (vmcnt is in order; later loads / stores only make the wait more conservative).
#if ES_EXP == 3
#define ES_WAIT_PREV() ;
#define ES_WAIT_ALL() ;
#define ES_SYNC()
#else
#define ES_WAIT_PREV() asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#define ES_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define ES_SYNC() __syncthreads();
#endif

    float x[NT][4][4];
    Split3x8 xb[NT][2];
    auto split_tile = [&](const float (&v)[4][4], Split3x8 (&o)[2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
            o[g] = join(split3(v[2 * g][0], v[2 * g][1], v[2 * g][2], v[2 * g][3]), split3(v[2 * g + 1][0], v[2 * g + 1][1], v[2 * g + 1][2], v[2 * g + 1][3]));
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float act[4][4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)tk[t] * ES_C + fb * 16 + lq * 4);
            const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk[t] * ES_C + fb * 16 + lq * 4);
            act[fb][0] = a.x; act[fb][1] = a.y; act[fb][2] = a.z; act[fb][3] = a.w;
            x[t][fb][0] = r.x; x[t][fb][1] = r.y; x[t][fb][2] = r.z; x[t][fb][3] = r.w;
        }
        split_tile(act, xb[t]);
    }
    stage_issue(0, 0);
    if (nsteps > 1) stage_issue(1, 1);
    ES_WAIT_ALL()
    __syncthreads();

    // ---- stage 0: output_proj + residual + LayerNorm1 (msdeformattn.py:124-126) ----
    {
        const bool more = 2 < nsteps;
        if (more) stage_issue(2, 2);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 d[NT];
            rowblock6<NT, MD::WC, MD::PROJ>(wls + (MD::WC * ob) * ES_BLOCK, lane, xb, smo(so.bo) + ob * 16, lq, d);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                x[t][ob][0] += d[t][0]; x[t][ob][1] += d[t][1]; x[t][ob][2] += d[t][2]; x[t][ob][3] += d[t][3];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            layer_norm_Ls(x[t], smo(so.g1), smo(so.be1), lq, eps);
            split_tile(x[t], xb[t]);
        }
        if (more) ES_WAIT_PREV() else ES_WAIT_ALL()
        ES_SYNC()
    }
    // ---- FFN: two hidden blocks of 16 per stage; the hidden activation never leaves registers ----
    f32x4 acc[NT][4];      // linear2's running sums: one chain per (tile, output block); low-order terms first within a k-group
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) acc[t][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 1; s <= nffn_stages; ++s) {
        const char* buf = wls + (s % ES_NBUF) * ES_STAGE;
        const bool more = s + 2 < nsteps;
        if (more) stage_issue(s + 2, (s + 2) % ES_NBUF);
        // Explicit order (sched_barrier pins it): all of linear1's fragments up front, linear2's requested while linear1's
        // MFMAs run, and consecutive MFMAs always on different accumulators -- term by term across the (hidden block, tile)
        // and (output block, tile) chains: as nested rowblock calls every block was load -> wait -> six dependent MFMAs.
#pragma unroll
        for (int p = 0; p < MD::HP; ++p) {
            const char* pb = buf + p * (2 * MD::W1C + 2 * MD::W2C) * ES_BLOCK;
            const int hb0 = ((s - 1) * MD::HP + p) * 2;             // first of the pair's two 16-wide hidden blocks
            Frag3 w1[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < 2; ++g) w1[q][g] = ld3<MD::W1C>(pb + (MD::W1C * q) * ES_BLOCK, ES_BLOCK, g, lane);
            f32x4 hh[2][NT], hl[2][NT];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // (MODE 1 pads the hidden dimension to whole stages with zero weights: their bias is zero too)
                const float4 b = (hb0 + q) * 16 < d_ffn ? *reinterpret_cast<const float4*>(small + so.b1 + (hb0 + q) * 16 + lq * 4)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    hh[q][t] = f32x4{b.x, b.y, b.z, b.w};
                    hl[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int term = 0; term < 6; ++term)
                    if ((MD::L1 >> term) & 1u) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int t = 0; t < NT; ++t) mac_term(term, hl[q][t], hh[q][t], w1[q][g], xb[t][g]);
                    }
            __builtin_amdgcn_sched_barrier(0);
            const char* w2 = pb + 2 * MD::W1C * ES_BLOCK;           // copies of 4 KiB: [4 output row blocks][64 lanes][8 k]
            Frag3 w2f[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) w2f[ob] = ld3<MD::W2C>(w2, 2 * ES_BLOCK, ob, lane);
            Split3x8 hb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 h0 = hh[0][t] + hl[0][t], h1 = hh[1][t] + hl[1][t];
                hb[t] = join(split3(relu1s(h0[0]), relu1s(h0[1]), relu1s(h0[2]), relu1s(h0[3])),
                             split3(relu1s(h1[0]), relu1s(h1[1]), relu1s(h1[2]), relu1s(h1[3])));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 6; ++term)
                if ((MD::L2 >> term) & 1u) {
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                        for (int t = 0; t < NT; ++t) mac_term(term, acc[t][ob], acc[t][ob], w2f[ob], hb[t]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) ES_WAIT_PREV() else ES_WAIT_ALL()
        ES_SYNC()
    }
    // ---- residual + LayerNorm2 (msdeformattn.py:116-118), write the layer output ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const float4 b2 = *reinterpret_cast<const float4*>(smo(so.b2) + ob * 16 + lq * 4);
            x[t][ob][0] += acc[t][ob][0] + b2.x;
            x[t][ob][1] += acc[t][ob][1] + b2.y;
            x[t][ob][2] += acc[t][ob][2] + b2.z;
            x[t][ob][3] += acc[t][ob][3] + b2.w;
        }
        layer_norm_Ls(x[t], smo(so.g2), smo(so.be2), lq, eps);
        if (tok_ok[t]) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
                *reinterpret_cast<float4*>(src_out + (int64_t)tok[t] * ES_C + ob * 16 + lq * 4) =
                    make_float4(x[t][ob][0], x[t][ob][1], x[t][ob][2], x[t][ob][3]);
        }
    }
    if (!next) return;                                               // (uniform)
    // ---- tail: next layer's value_proj (4 row blocks), then [sampling_offsets | attention_weights] ----
    const int dh = value_heads ? ES_C / value_heads : ES_C;
    int t_img[NT], t_pos[NT];
    Split3x8 xq[NT][2];                                               // src + pos (query of the next layer, msdeformattn.py:124)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        t_img[t] = tk[t] / S;
        t_pos[t] = tk[t] - t_img[t] * S;
        float xp[4][4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float4 pp = *reinterpret_cast<const float4*>(pos + (int64_t)t_pos[t] * ES_C + fb * 16 + lq * 4);
            xp[fb][0] = x[t][fb][0] + pp.x; xp[fb][1] = x[t][fb][1] + pp.y; xp[fb][2] = x[t][fb][2] + pp.z; xp[fb][3] = x[t][fb][3] + pp.w;
        }
        split_tile(x[t], xb[t]);
        split_tile(xp, xq[t]);
    }
    for (int s = 1 + nffn_stages; s < nsteps; ++s) {
        const char* buf = wls + (s % ES_NBUF) * ES_STAGE;
        const bool more = s + 2 < nsteps;
        if (more) stage_issue(s + 2, (s + 2) % ES_NBUF);
        const int ts = s - 1 - nffn_stages;                           // tail stage: 0 = value_proj, 1.. = four proj row blocks each
        if (ts == 0) {
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                f32x4 d[NT];
                rowblock6<NT, MD::WC, MD::PROJ>(buf + (MD::WC * tb) * ES_BLOCK, lane, xb, smo(so.bv) + tb * 16, lq, d);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (!tok_ok[t]) continue;
                    // token-major [tok][64], or head-major [b][head][t][64/heads] for msm_msdeform_attn_enc_hm_fwd
                    const int f = tb * 16 + lq * 4;
                    float* o = value_heads ? value_out + (((int64_t)t_img[t] * value_heads + f / dh) * S + t_pos[t]) * dh + f % dh
                                           : value_out + (int64_t)tk[t] * ES_C + f;
                    if (dh >= 4) {
                        *reinterpret_cast<float4*>(o) = make_float4(d[t][0], d[t][1], d[t][2], d[t][3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int fr = f + r;
                            value_out[(((int64_t)t_img[t] * value_heads + fr / dh) * S + t_pos[t]) * dh + fr % dh] = d[t][r];
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < MD::PB; ++j) {
                const int ob = (ts - 1) * MD::PB + j;
                if (ob < nproj_blocks) {
                    f32x4 d[NT];
                    rowblock6<NT, MD::WC, MD::PROJ>(buf + (MD::WC * j) * ES_BLOCK, lane, xq, smo(so.bp) + ob * 16, lq, d);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (tok_ok[t])
                            *reinterpret_cast<float4*>(proj_out + (int64_t)tk[t] * proj_ld + ob * 16 + lq * 4) = make_float4(d[t][0], d[t][1], d[t][2], d[t][3]);
                }
            }
        }
        if (more) ES_WAIT_PREV() else ES_WAIT_ALL()
        ES_SYNC()
    }
}
#undef ES_WAIT_PREV
#undef ES_WAIT_ALL
#undef ES_SYNC

// Workgroups 0 .. n_heavy - 1 take two tiles per wave, the others one: 3150 tiles over the 2048 waves of two workgroups per
// CU are one each and a second for 1102 of them, and the dispatcher's round-robin puts a heavy and a light workgroup on
// every CU (3 tiles per SIMD; 20 CUs get two heavy ones).  Two straight-line instantiations behind a workgroup-uniform
// branch: per-wave tile counts decided by branches inside the stage loop cut it into 12-MFMA basic blocks whose LDS reads
// nothing overlapped (116 us against 105).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void enc_block_split_kernel(
    const float* __restrict__ attn, const float* __restrict__ src, const char* __restrict__ wstream, const float* __restrict__ small, EncSmallS so,
    const float* __restrict__ pos, float* __restrict__ src_out, float* __restrict__ value_out, float* __restrict__ proj_out, int M, int S,
    int nffn_stages, int nproj_blocks, int proj_ld, float eps, int n_small, int value_heads, int n_heavy) {
    const int b = blockIdx.x;
    if (b < n_heavy)
        enc_block_split_body<2, MODE>(attn, src, wstream, small, so, pos, src_out, value_out, proj_out, M, S, nffn_stages, nproj_blocks, proj_ld, eps,
                                      n_small, value_heads, b * 8);
    else
        enc_block_split_body<1, MODE>(attn, src, wstream, small, so, pos, src_out, value_out, proj_out, M, S, nffn_stages, nproj_blocks, proj_ld, eps,
                                      n_small, value_heads, n_heavy * 8 + (b - n_heavy) * 4);
}

}  // namespace msm

using namespace msm;

template <int MODE>
static int64_t split_stream_bytes(int d_ffn, int proj_width) {
    using MD = ESMode<MODE>;
    const int ntail = proj_width > 0 ? 1 + cdiv(proj_width / 16, MD::PB) : 0;
    return (int64_t)(1 + cdiv(d_ffn, 32 * MD::HP) + ntail) * ES_STAGE;
}

template <int MODE>
static int split_launch(const char* who, const float* attn, const float* src, const void* wstream, const float* small, const float* pos, float* src_out,
                        float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn, int proj_width, int value_heads, float eps,
                        void* stream) {
    using MD = ESMode<MODE>;
    MSM_REQUIRE(attn && src && wstream && small && src_out, "%s: null pointer", who);
    MSM_REQUIRE(M > 0 && tokens_per_image > 0 && d_ffn > 0 && d_ffn % 32 == 0, "%s: d_ffn=%d must be a positive multiple of 32", who, d_ffn);
    MSM_REQUIRE((value_out == nullptr) == (proj_out == nullptr), "%s: value_out and proj_out go together", who);
    MSM_REQUIRE(!value_out || (pos && proj_width > 0 && proj_width % 16 == 0), "%s: the next layer's projections need pos and a proj width that is a multiple of 16", who);
    MSM_REQUIRE(value_heads == 0 || (ES_C % value_heads == 0), "%s: value_heads=%d must divide 64", who, value_heads);
    MSM_REQUIRE(((((uintptr_t)attn) | ((uintptr_t)src) | ((uintptr_t)wstream) | ((uintptr_t)src_out) | ((uintptr_t)value_out) | ((uintptr_t)proj_out) | ((uintptr_t)pos) | ((uintptr_t)small)) & 15) == 0,
                "%s: pointers must be 16-byte aligned", who);
    const int pw = value_out ? proj_width : 0;
    EncSmallS so;
    so.bo = 0; so.g1 = 64; so.be1 = 128; so.b1 = 192; so.b2 = 192 + d_ffn; so.g2 = so.b2 + 64; so.be2 = so.g2 + 64; so.bv = so.be2 + 64; so.bp = so.bv + 64;
    const int n_small = so.bp + proj_width;
    const size_t lds = ES_NBUF * ES_STAGE + sizeof(float) * (size_t)(n_small - d_ffn);
    // Two workgroups (2 x 4 waves) per CU -- a lone wave per SIMD exposes every LDS-read and barrier latency (measured: 4 tiles
    // per wave, one wave per SIMD: 123 us against 105 for 2 x 2).  Up to 512 workgroups: light ones (4 tiles) until the chip is
    // full, then heavy ones (8 tiles) replace them two for one; beyond 4096 tiles all are heavy and the grid grows.
    const int tiles = cdiv(M, 16);
    int n_heavy, n_light;
    if (tiles <= 2048) {
        n_heavy = 0;
        n_light = cdiv(tiles, 4);
    } else if (tiles <= 4096) {
        n_heavy = cdiv(tiles - 2048, 4);                    // every heavy workgroup carries 4 tiles more than a light one
        n_light = 512 - n_heavy;
    } else {
        n_heavy = cdiv(tiles, 8);
        n_light = 0;
    }
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_split_kernel<MODE>, lds));
    hipLaunchKernelGGL(enc_block_split_kernel<MODE>, dim3(n_heavy + n_light), dim3(256), lds, (hipStream_t)stream, attn, src, (const char*)wstream, small,
                       so, pos, src_out, value_out, proj_out, M, tokens_per_image, cdiv(d_ffn, 32 * MD::HP), pw / 16, proj_width, eps, n_small,
                       value_heads, n_heavy);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int64_t msm_encoder_block_split_stream_bytes(int d_ffn, int proj_width) { return split_stream_bytes<0>(d_ffn, proj_width); }
extern "C" int msm_encoder_block_split_fwd(const float* attn, const float* src, const void* wstream, const float* small, const float* pos,
                                           float* src_out, float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn,
                                           int proj_width, int value_heads, float eps, void* stream) {
    return split_launch<0>("msm_encoder_block_split_fwd", attn, src, wstream, small, pos, src_out, value_out, proj_out, M, tokens_per_image, d_ffn,
                           proj_width, value_heads, eps, stream);
}
extern "C" int64_t msm_encoder_block_lp_stream_bytes(int d_ffn, int proj_width) { return split_stream_bytes<1>(d_ffn, proj_width); }
extern "C" int msm_encoder_block_lp_fwd(const float* attn, const float* src, const void* wstream, const float* small, const float* pos,
                                        float* src_out, float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn,
                                        int proj_width, int value_heads, float eps, void* stream) {
    return split_launch<1>("msm_encoder_block_lp_fwd", attn, src, wstream, small, pos, src_out, value_out, proj_out, M, tokens_per_image, d_ffn,
                           proj_width, value_heads, eps, stream);
}
