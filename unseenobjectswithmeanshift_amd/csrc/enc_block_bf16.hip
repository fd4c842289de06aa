// bf16 form of the fused encoder-layer tail (see include/msm_hip.h: msm_encoder_block_bf16_fwd) -- BASELINE configs 3 / 5.
//
// Same function as enc_block_kernel (enc_block.hip; reference msdeformattn.py:116-131, ops/modules/ms_deform_attn.py:95-104,
// 123): src = LN1(src + output_proj(msda_out)); src = LN2(src + linear2(relu(linear1(src)))); value = value_proj(src);
// proj = [sampling_offsets | attention_weights](src + pos) -- with bf16 MFMA operands and fp32 accumulation
// (v_mfma_f32_16x16x16_bf16).  The residual stream, both LayerNorms, every bias and every tensor that leaves the kernel stay
// fp32: what is rounded to bf16 are the weights (once per checkpoint) and the activations at the moment they become an
// MFMA operand (v_cvt_pk_bf16_f32, one instruction per pair).
//
// The fp32 kernel is MFMA-bound (72 % of the fp32 peak, 150 us per layer at B = 8); bf16 MFMAs are 16x faster per FLOP, so
// this one is bound by what is left -- the HBM streams (26 MB in, 84 MB out per layer), LDS fragment reads and the
// LayerNorm / ReLU / conversion VALU work -- and needs none of the fp32 kernel's machinery for keeping the matrix pipe fed
// (LDS-DMA between MFMA groups, cooperative tiles):
//   * register layout L of the fp32 kernel: lane (token lj = l & 15, quarter lq = l >> 4) holds features fb*16 + lq*4 + r.
//     That is the C/D layout of a transposed MFMA tile (rows = output features, columns = tokens) AND, packed to bf16x4,
//     the B operand of a 16x16x16 MFMA over feature block fb -- activations never leave registers between the GEMMs;
//   * weights are the A operand, pre-packed on the host in FRAGMENT order (pack_encoder_block_bf16): a block is 2 KiB =
//     [4 k-groups][64 lanes][4 bf16], lane l's operand of k-group g is the 8 bytes at (g*64 + l)*8 -- conflict-free
//     ds_read_b64, and a stage of 8 blocks is a plain 16 KiB copy (LDS-DMA, no swizzle);
//   * one 16-token tile per wave, 4 waves per workgroup, 2 x 16 KiB LDS stages + parameters: 4 workgroups per CU.
#include "bf16.h"
#include "common.h"

namespace msm {

constexpr int EB_C = 64;                    // d_model
constexpr int EB_BLOCK = 2048;              // bytes per weight block
constexpr int EB_STAGE = 8 * EB_BLOCK;      // bytes per LDS stage

struct EncSmallB {                          // offsets (floats) into the packed small-parameter vector (as enc_block.hip)
    int bo, g1, be1, b1, b2, g2, be2, bv, bp;
};

// An activation block as TWO bf16 operands, x = hi + lo up to 2^-17 |x|: the 64-wide contractions (output_proj, linear1, value
// and sampling projections) have little averaging over k, and rounding their inputs to 8 mantissa bits moved the attention-mask
// bits of the decoder behind them about five times as often as rounding the weights alone does (random-init worst case:
// final-mask mismatch against the fp32 reference 1.7 % -> see DESIGN.md).  One more MFMA per k-group; the 1024-wide linear2
// keeps a single bf16 operand.
__device__ __forceinline__ bf16x4 frag(const char* __restrict__ blk, int g, int lane) {
    return __builtin_bit_cast(bf16x4, *reinterpret_cast<const u32x2b*>(blk + (g * 64 + lane) * 8));
}
// one [16 rows][64 k] block applied to the four packed feature blocks of the tile: D (layout L) = bias + W x
__device__ __forceinline__ f32x4 rowblock(const char* __restrict__ blk, int lane, const Split4 (&xb)[4], const float* __restrict__ bias, int lq) {
    const float4 b = *reinterpret_cast<const float4*>(bias + lq * 4);
    f32x4 d = f32x4{b.x, b.y, b.z, b.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bf16x4 w = frag(blk, g, lane);
        d = mfma_bf16(w, xb[g].lo, d);
        d = mfma_bf16(w, xb[g].hi, d);
    }
    return d;
}
// the same with the weight block given as hi + lo bf16 blocks (w = hi + lo up to 2^-17 |w|): three MFMAs per k-group,
// w_lo x_hi + w_hi x_lo + w_hi x_hi.  Used for output_proj, value_proj and above all the sampling-offset / attention-weight
// projection: rounding THAT matrix to 8 mantissa bits alone moved 1.9 % of the decoder's final mask bits on random-init
// weights (an emulation of each rounding on the CPU oracle: FFN weights 0.8 %, this projection 1.9 %, all of them 1.8 %).
__device__ __forceinline__ f32x4 rowblock3(const char* __restrict__ hi, const char* __restrict__ lo, int lane, const Split4 (&xb)[4],
                                           const float* __restrict__ bias, int lq) {
    const float4 b = *reinterpret_cast<const float4*>(bias + lq * 4);
    f32x4 d = f32x4{b.x, b.y, b.z, b.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bf16x4 wh = frag(hi, g, lane), wl = frag(lo, g, lane);
        d = mfma_bf16(wl, xb[g].hi, d);
        d = mfma_bf16(wh, xb[g].lo, d);
        d = mfma_bf16(wh, xb[g].hi, d);
    }
    return d;
}
__device__ __forceinline__ float relu1b(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.0e38f); }

__device__ __forceinline__ void layer_norm_Lb(float (&v)[4][4], const float* __restrict__ g, const float* __restrict__ b, int lq, float eps) {
    float s = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[fb][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / EB_C);
    float q = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = v[fb][r] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / EB_C) + eps);
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
__device__ __forceinline__ void glds16b(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Stream = stages of 8 blocks: stage 0 = output_proj (4 hi row blocks, 4 lo row blocks); stages 1 .. d_ffn/64: four hidden
// blocks each as [linear1 block, linear2 block] pairs; then value_proj (4 hi + 4 lo row blocks) and the proj row blocks as
// [hi, lo] pairs, four pairs per stage.
__global__ __launch_bounds__(256) void enc_block_bf16_kernel(const float* __restrict__ attn, const float* __restrict__ src,
                                                             const char* __restrict__ wstream, const float* __restrict__ small, EncSmallB so,
                                                             const float* __restrict__ pos, float* __restrict__ src_out,
                                                             float* __restrict__ value_out, float* __restrict__ proj_out, int M, int S, int nffn_stages,
                                                             int nproj_blocks, int proj_ld, float eps, int n_small, int value_heads) {
    extern __shared__ __attribute__((aligned(16))) char wlb[];      // [2][EB_STAGE] weight stages, then the small parameters
    float* sm = reinterpret_cast<float*>(wlb + 2 * EB_STAGE);
    for (int i = threadIdx.x; i < n_small; i += 256) sm[i] = small[i];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int tile = (int)blockIdx.x * 4 + wave;
    const int tok = tile * 16 + lj;
    const bool tok_ok = tok < M;
    const int tk = tok_ok ? tok : M - 1;
    const bool next = value_out != nullptr;
    const int ntail = next ? 1 + (nproj_blocks + 3) / 4 : 0;
    const int nsteps = 1 + nffn_stages + ntail;

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wlb;
    auto stage_issue = [&](int s, int bufi) {                       // 16 KiB = 4 x (256 lanes x 16 B)
        const char* sb = wstream + (int64_t)s * EB_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16b(sb + i * 4096 + wave * 1024, (unsigned)lane * 16u, lds_base + (unsigned)bufi * EB_STAGE + i * 4096u + (unsigned)wave * 1024u);
    };
#define EB_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float act[4][4], x[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)tk * EB_C + fb * 16 + lq * 4);
        const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk * EB_C + fb * 16 + lq * 4);
        act[fb][0] = a.x; act[fb][1] = a.y; act[fb][2] = a.z; act[fb][3] = a.w;
        x[fb][0] = r.x; x[fb][1] = r.y; x[fb][2] = r.z; x[fb][3] = r.w;
    }
    stage_issue(0, 0);
    EB_WAIT()
    __syncthreads();

    Split4 xb[4];
    // ---- stage 0: output_proj + residual + LayerNorm1 (msdeformattn.py:124-126) ----
    {
        stage_issue(1, 1);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) xb[fb] = split4(act[fb][0], act[fb][1], act[fb][2], act[fb][3]);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 d = rowblock3(wlb + ob * EB_BLOCK, wlb + (4 + ob) * EB_BLOCK, lane, xb, sm + so.bo + ob * 16, lq);
            x[ob][0] += d[0]; x[ob][1] += d[1]; x[ob][2] += d[2]; x[ob][3] += d[3];
        }
        layer_norm_Lb(x, sm + so.g1, sm + so.be1, lq, eps);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) xb[fb] = split4(x[fb][0], x[fb][1], x[fb][2], x[fb][3]);
        EB_WAIT()
        __syncthreads();
    }
    // ---- FFN: four hidden blocks of 16 per stage; the hidden activation lives in 4 registers (2 packed) ----
    f32x4 acc2[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc2[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 1; s <= nffn_stages; ++s) {
        const char* buf = wlb + (s & 1) * EB_STAGE;
        stage_issue(min(s + 1, nsteps - 1), (s + 1) & 1);           // the last stage of a launch without tail re-fetches itself
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 h = rowblock(buf + (2 * q) * EB_BLOCK, lane, xb, sm + so.b1 + ((s - 1) * 4 + q) * 16, lq);
            // (the 1024-wide contraction keeps ONE bf16 operand per hidden value: splitting it too measured no change in the
            // decoder's outputs -- mean |dmask| 0.106 against 0.108 -- for +9 us per layer)
            const bf16x4 hb = pack4(relu1b(h[0]), relu1b(h[1]), relu1b(h[2]), relu1b(h[3]));
            const char* w2 = buf + (2 * q + 1) * EB_BLOCK;          // [4 output row blocks][64 lanes][4 k]
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma_bf16(frag(w2, ob, lane), hb, acc2[ob]);
        }
        EB_WAIT()
        __syncthreads();
    }
    // ---- residual + LayerNorm2 (msdeformattn.py:116-118), write the layer output ----
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const float4 b2 = *reinterpret_cast<const float4*>(sm + so.b2 + ob * 16 + lq * 4);
        x[ob][0] += acc2[ob][0] + b2.x;
        x[ob][1] += acc2[ob][1] + b2.y;
        x[ob][2] += acc2[ob][2] + b2.z;
        x[ob][3] += acc2[ob][3] + b2.w;
    }
    layer_norm_Lb(x, sm + so.g2, sm + so.be2, lq, eps);
    if (tok_ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            *reinterpret_cast<float4*>(src_out + (int64_t)tok * EB_C + ob * 16 + lq * 4) = make_float4(x[ob][0], x[ob][1], x[ob][2], x[ob][3]);
    }
    if (!next) return;                                               // (uniform)
    // ---- tail: next layer's value_proj (4 row blocks), then [sampling_offsets | attention_weights] ----
    const int t_img = tk / S, t_pos = tk - t_img * S;
    const int dh = value_heads ? EB_C / value_heads : EB_C;
    Split4 xq[4];                                                     // src + pos (query of the next layer, msdeformattn.py:124)
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        xb[fb] = split4(x[fb][0], x[fb][1], x[fb][2], x[fb][3]);
        const float4 pp = *reinterpret_cast<const float4*>(pos + (int64_t)t_pos * EB_C + fb * 16 + lq * 4);
        xq[fb] = split4(x[fb][0] + pp.x, x[fb][1] + pp.y, x[fb][2] + pp.z, x[fb][3] + pp.w);
    }
    for (int s = 1 + nffn_stages; s < nsteps; ++s) {
        const char* buf = wlb + (s & 1) * EB_STAGE;
        if (s + 1 < nsteps) stage_issue(s + 1, (s + 1) & 1);
        const int ts = s - 1 - nffn_stages;                           // tail stage: 0 = value_proj, 1.. = four proj row blocks each
        if (ts == 0) {
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                const f32x4 d = rowblock3(buf + tb * EB_BLOCK, buf + (4 + tb) * EB_BLOCK, lane, xb, sm + so.bv + tb * 16, lq);
                if (tok_ok) {
                    // token-major [tok][64], or head-major [b][head][t][64/heads] for msm_msdeform_attn_enc_hm_fwd
                    const int f = tb * 16 + lq * 4;
                    float* o = value_heads ? value_out + (((int64_t)t_img * value_heads + f / dh) * S + t_pos) * dh + f % dh
                                           : value_out + (int64_t)tk * EB_C + f;
                    if (dh >= 4) {
                        *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int fr = f + r;
                            value_out[(((int64_t)t_img * value_heads + fr / dh) * S + t_pos) * dh + fr % dh] = d[r];
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ob = (ts - 1) * 4 + j;
                if (ob < nproj_blocks) {
                    const f32x4 d = rowblock3(buf + (2 * j) * EB_BLOCK, buf + (2 * j + 1) * EB_BLOCK, lane, xq, sm + so.bp + ob * 16, lq);
                    if (tok_ok) *reinterpret_cast<float4*>(proj_out + (int64_t)tk * proj_ld + ob * 16 + lq * 4) = make_float4(d[0], d[1], d[2], d[3]);
                }
            }
        }
        EB_WAIT()
        __syncthreads();
    }
}
#undef EB_WAIT

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_encoder_block_bf16_stream_bytes(int d_ffn, int proj_width) {
    const int ntail = proj_width > 0 ? 1 + (proj_width / 16 + 3) / 4 : 0;
    return (int64_t)(1 + d_ffn / 64 + ntail) * EB_STAGE;
}

extern "C" int msm_encoder_block_bf16_fwd(const float* attn, const float* src, const void* wstream, const float* small, const float* pos,
                                          float* src_out, float* value_out, float* proj_out, int M, int tokens_per_image, int d_ffn,
                                          int proj_width, int value_heads, float eps, void* stream) {
    MSM_REQUIRE(attn && src && wstream && small && src_out, "msm_encoder_block_bf16_fwd: null pointer");
    MSM_REQUIRE(M > 0 && tokens_per_image > 0 && d_ffn > 0 && d_ffn % 64 == 0, "msm_encoder_block_bf16_fwd: d_ffn=%d must be a positive multiple of 64", d_ffn);
    MSM_REQUIRE((value_out == nullptr) == (proj_out == nullptr), "msm_encoder_block_bf16_fwd: value_out and proj_out go together");
    MSM_REQUIRE(!value_out || (pos && proj_width > 0 && proj_width % 16 == 0), "msm_encoder_block_bf16_fwd: the next layer's projections need pos and a proj width that is a multiple of 16");
    MSM_REQUIRE(value_heads == 0 || (EB_C % value_heads == 0), "msm_encoder_block_bf16_fwd: value_heads=%d must divide 64", value_heads);
    MSM_REQUIRE(((((uintptr_t)attn) | ((uintptr_t)src) | ((uintptr_t)wstream) | ((uintptr_t)src_out) | ((uintptr_t)value_out) | ((uintptr_t)proj_out) | ((uintptr_t)pos)) & 15) == 0,
                "msm_encoder_block_bf16_fwd: pointers must be 16-byte aligned");
    const int pw = value_out ? proj_width : 0;
    EncSmallB so;
    so.bo = 0; so.g1 = 64; so.be1 = 128; so.b1 = 192; so.b2 = 192 + d_ffn; so.g2 = so.b2 + 64; so.be2 = so.g2 + 64; so.bv = so.be2 + 64; so.bp = so.bv + 64;
    const int n_small = so.bp + proj_width;
    const size_t lds = 2 * EB_STAGE + sizeof(float) * (size_t)n_small;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_bf16_kernel, lds));
    const int tiles = cdiv(M, 16);
    hipLaunchKernelGGL(enc_block_bf16_kernel, dim3(cdiv(tiles, 4)), dim3(256), lds, (hipStream_t)stream, attn, src, (const char*)wstream, small, so, pos,
                       src_out, value_out, proj_out, M, tokens_per_image, d_ffn / 64, pw / 16, proj_width, eps, n_small, value_heads);
    MSM_CHECK_LAUNCH("msm_encoder_block_bf16_fwd");
    return MSM_OK;
}
