// The pixel decoder's encoder layers in the bf16 plan (set_precision("bf16"), BASELINE configs[2] / configs[4]) with the
// activations that only this plan's own kernels consume stored as HEAD-MAJOR 16-bit tensors in HBM:
//     value_hm [B][8][S][8]  fp16   (encoder block l -> gather l + 1; a sampling tap = one 16-byte segment)
//     attn_hm  [B][8][S][8]  fp16   (gather l        -> encoder block l; a lane's MFMA operand = one 16-byte load)
//     proj_hm  [B][8][120 S bytes]  (encoder block l -> gather l + 1; per (query, head) 24 sampling offsets as fp32 and 12 attention
//                                    logits as fp16, plane-major, ops/modules/ms_deform_attn.py:99-104; round 5 -- see EH_REC below)
// IEEE half, not bf16: same bytes, three more mantissa bits, and these are O(1) projections of LayerNorm outputs (converted with a
// clamp to the half range).  Measured on the round-3 kernels with the tensor rounded in between (tools/probes/lp_rounding_probe.py,
// batch 8 at 640x480, final-mask mismatch against the fp32 reference; fp32 tensors: 0.93 %): proj as bf16 2.03 % -- the offsets
// decide tap positions -- as fp16 0.99 %; value 0.95 / 0.92 %; attn 0.99 / 0.95 %; all three as fp16 1.00 %.  The matrix pipe still
// multiplies bf16 operands (an fp16 value is a hi + lo bf16 pair exactly).  The residual stream `src` stays fp32 (LayerNorm
// inputs, the FPN level and the decoder's K/V projection read it).  Per layer and batch of 8 at 640x480: 110 MB of HBM
// traffic instead of 202 MB (round 3: fp32 value / proj / attn tensors).
//
// Two kernels (reference: msdeformattn.py:116-131, ops/modules/ms_deform_attn.py:95-125, ms_deform_im2col_cuda.cuh:242-304):
//   enc_block_hm_kernel   output_proj + LN1 + linear1 / ReLU / linear2 + LN2 + the next layer's value_proj
//   msda_enc_lp_kernel    softmax + sampling locations from the stored projection (or the projection itself, FUSED) + bilinear
//                         gather of bf16 value taps
//
// enc_block_hm_kernel (rounds 4-5; since round 6 the default is its half-size form enc_block_hm2_kernel below, same stream, same bits):
// ONE 16-wave workgroup per CU, a 16-token tile per wave, all waves share the weight stream: an FFN stage is
// 32 KiB (four pairs of 16-wide hidden blocks: W1 2 x 2 KiB, W2 4 KiB each) = two 1-KiB LDS-DMA pieces per wave, three stage
// buffers, a stage is requested two stages ahead; output_proj and value_proj (hi + lo copies, 32 KiB) stay resident.  What
// the round-3 kernel (enc_block_split.hip, MODE 1: 4-wave workgroups, 2 per CU, six pieces per wave and stage, 16 stages)
// spent its time on was not the matrix pipe (49 us with, 38 us without its MFMAs): every workgroup streamed the weights for
// 4 - 8 tiles, and a wave issued 96 DMA instructions per launch beside its 540 MFMAs.  Here a CU streams them once for 13 tiles.
// Operand roundings are those of MODE 1: the 64-wide projections w(h + l) x(h + l) without the l x l term, linear1 w(h) x(h + l),
// linear2 single operands.
#include "bf16.h"
#include "common.h"

#ifndef EH_EXP
#define EH_EXP 0   // tuning builds only (tools/probes/enc_hm_parts.sh): 1 = weight fragments are register constants (no LDS reads), 2 = no MFMAs, 3 = no value / projection / src stores
#endif

namespace msm {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int EH_C = 64;                      // d_model
constexpr int EH_STAGE = 32 * 1024;           // bytes per FFN stage (4 pairs x 8 KiB)
constexpr int EH_NBUF = 3;
constexpr int EH_RES = 32 * 1024;             // resident block: output_proj [rb][G][h, l] (16 KiB), value_proj likewise (16 KiB)
constexpr int EH_WAVES = 16;
// fp32 parameter vector (all of it is copied to LDS): bo, g1, be1, b2, g2, be2, bv (row order of the value store), b1 (padded)
constexpr int EH_BO = 0, EH_G1 = 64, EH_BE1 = 128, EH_B2 = 192, EH_G2 = 256, EH_BE2 = 320, EH_BV = 384, EH_BP = 448, EH_B1 = 736;
constexpr int EH_PROJ = 288, EH_PROJ_STAGES = 3;       // [sampling_offsets | attention_weights] rows, offsets first (24 head + c), then logits (192 + 12 head + c): 18 row blocks, 8 per stage

__device__ __forceinline__ const void* uniform_ptr_lp(const void* p) {
    const uint64_t v = (uint64_t)p;
    return (const void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
}
// one 1-KiB LDS-DMA piece (16 bytes per lane): m0 is written in the statement that uses it (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void glds16h(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ bf16x8 ldfrag(const char* blk, int lane) {
#if EH_EXP == 1
    return __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane * 0x10001u, 0x3f803f80u, 0x3f803f80u, (unsigned)(uintptr_t)blk});
#else
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(blk + lane * 16));
#endif
}
#if EH_EXP == 2
#define mfma_f16k32(a, b, c) eh_fake_mfma(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c)
#define mfma_bf16k32 eh_fake_mfma
__device__ __forceinline__ f32x4 eh_fake_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
    c[0] += __uint_as_float(ua.x ^ ub.x);
    return c;
}
#endif
__device__ __forceinline__ float relu1h(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.0e38f); }
__device__ __forceinline__ float relu_h(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 65504.f); }      // ReLU + clamp to the half range
// x (layout L: lane (token lj, quarter lq) holds features fb*16 + lq*4 + r) -> the B operands of its two 32-wide k-groups:
// group G = the lane's values of feature blocks 2G and 2G + 1 side by side (the weights are packed in the same k order)
__device__ __forceinline__ void split_L(const float (&v)[4][4], bf16x8 (&h)[2], bf16x8 (&l)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const Split4 a = split4(v[2 * g][0], v[2 * g][1], v[2 * g][2], v[2 * g][3]);
        const Split4 b = split4(v[2 * g + 1][0], v[2 * g + 1][1], v[2 * g + 1][2], v[2 * g + 1][3]);
        h[g] = cat8(a.hi, b.hi);
        l[g] = cat8(a.lo, b.lo);
    }
}
__device__ __forceinline__ void layer_norm_h(float (&v)[4][4], const float* __restrict__ g, const float* __restrict__ b, int lq, float eps) {
    float s = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[fb][r];
    s = sum_lane_rows(s);
    const float mean = s * (1.0f / EH_C);
    float q = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = v[fb][r] - mean;
            q += d * d;
        }
    q = sum_lane_rows(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / EH_C) + eps);
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
__device__ __forceinline__ u32x4 pack8h(const f32x4& a, const f32x4& b) {                  // eight floats -> eight IEEE halves
    const u32x2b ul = pack4h(a[0], a[1], a[2], a[3]), uh = pack4h(b[0], b[1], b[2], b[3]);
    return u32x4{ul.x, ul.y, uh.x, uh.y};
}

// The sampling projection of a (image, head): EH_REC = 120 bytes per token = the head's 24 sampling offsets as FLOAT32 ((level,
// point, xy) order) and its 12 attention logits as fp16, stored PLANE-major: six planes [S][4 floats] (offsets 4 p .. 4 p + 3 = the
// (x, y) of points 2 p, 2 p + 1) followed by three planes [S][4 halves] (logits 4 p .. 4 p + 3).
//  * fp32 offsets: they are pixel distances of several pixels (measured on the seeded weights: mean 3.8, maximum 36); an fp16 offset is
//    off by up to 2^-11 |o| ~ 2e-3 pixel, and through the bilinear weights that alone put 6e-3 of relative error into the encoder's
//    output -- the three fp16 tensors of round 4 together, with the offsets in fp32, leave 1.5e-3 (tools/probes/bf16_pooled_probe.py).
//  * planes: the 16 tokens of a wave's tile are consecutive, so a store instruction writes 256 contiguous bytes per lane quarter
//    instead of sixteen 16-byte pieces 120 bytes apart (token-major records: 49.8 us per launch; planes: see DESIGN 5b).
// The packed projection rows are ordered offsets first (row 24 head + c), then logits (row 192 + 12 head + c) -- the reference's own
// order --, so a 16-row block of the MFMA output is all offsets (rb < 12) or all logits and a lane's four results are one plane entry.
constexpr int EH_REC = 120;
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
// byte offset of the region of (image, head) in a [B][8][S * EH_REC] projection
__device__ __forceinline__ int64_t proj_region(int img, int head, int S) { return ((int64_t)img * 8 + head) * S * EH_REC; }
__device__ __forceinline__ void store_proj_rb(unsigned char* __restrict__ proj_out, int img, int tpos, int S, int rb, int lq, const f32x4& d) {
    if (rb < 12) {
        const int idx = rb * 16 + lq * 4, head = idx / 24, plane = (idx - head * 24) >> 2;
        *reinterpret_cast<f32x4_a8*>(proj_out + proj_region(img, head, S) + ((int64_t)plane * S + tpos) * 16) = f32x4_a8{d[0], d[1], d[2], d[3]};
    } else {
        const int idx = (rb - 12) * 16 + lq * 4, head = idx / 12, plane = (idx - head * 12) >> 2;
        *reinterpret_cast<u32x2b*>(proj_out + proj_region(img, head, S) + (int64_t)S * 96 + ((int64_t)plane * S + tpos) * 8) = pack4h(d[0], d[1], d[2], d[3]);
    }
}

// residual + linear2 bias + LayerNorm2 (msdeformattn.py:116-118)
__device__ __forceinline__ void finish_ffn(float (&x)[4][4], const f32x4 (&acc)[4], const float* __restrict__ sm, int lq, float eps) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const float4 b2 = *reinterpret_cast<const float4*>(sm + EH_B2 + ob * 16 + lq * 4);
        x[ob][0] += acc[ob][0] + b2.x;
        x[ob][1] += acc[ob][1] + b2.y;
        x[ob][2] += acc[ob][2] + b2.z;
        x[ob][3] += acc[ob][3] + b2.w;
    }
    layer_norm_h(x, sm + EH_G2, sm + EH_BE2, lq, eps);
}
__device__ __forceinline__ void store_src(float* __restrict__ src_out, const float (&x)[4][4], int tok, bool tok_ok, int lq) {
#if EH_EXP == 3
    tok_ok = tok_ok && x[0][0] == 12345.678f;
#endif
    if (tok_ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            *reinterpret_cast<float4*>(src_out + (int64_t)tok * EH_C + ob * 16 + lq * 4) = make_float4(x[ob][0], x[ob][1], x[ob][2], x[ob][3]);
    }
}

// wstream: [resident 32 KiB][nffn FFN stages x 32 KiB][3 projection stages x 32 KiB, not for the last layer]; small: EH_B1 + 128 * nffn
// floats.  attn_hm / value_out: [B][8][S][8] fp16, proj_out: [B][8][S * EH_REC bytes] (plane-major fp32 offsets + fp16 logits, see EH_REC).
// Grid: workgroup g owns tiles g * tpw ... g * tpw + tpw - 1 (tpw <= 16: wave w takes tile w; the other waves only stream).
// F16 (precision "f16"): the FFN stages hold IEEE-half bit patterns, x enters linear1 as ONE fp16 term and the hidden activation as
// one fp16 term (v_mfma_f32_16x16x32_f16: 8 instead of 12 MFMAs per pair of hidden blocks, and 2^-12 roundings where the bf16 form has
// 2^-9 on W1, W2 and the hidden activation).  The 64-wide projections keep their three-term bf16 products (2^-17) in both forms.
template <bool F16>
__global__ __launch_bounds__(EH_WAVES * 64) void enc_block_hm_kernel(const unsigned short* __restrict__ attn_hm, const float* __restrict__ src,
                                                                     const char* __restrict__ wstream, const float* __restrict__ small,
                                                                     const float* __restrict__ pos, float* __restrict__ src_out,
                                                                     unsigned short* __restrict__ value_out, unsigned short* __restrict__ proj_out, int M,
                                                                     int S, int nffn, float eps, int tpw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [resident][EH_NBUF stages][small]
    char* const res = lds;
    char* const ring = lds + EH_RES;
    float* const sm = reinterpret_cast<float*>(lds + EH_RES + EH_NBUF * EH_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int n_small = EH_B1 + 128 * nffn;
    const bool next = value_out != nullptr;                          // (uniform) not the last layer
    const int nstages = nffn + (next ? EH_PROJ_STAGES : 0);
    for (int i = tid; i < n_small; i += EH_WAVES * 64) sm[i] = small[i];

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const void* ws = uniform_ptr_lp(wstream);
    auto issue = [&](int64_t src_off, unsigned dst_off) {            // this wave's two pieces of a 32-KiB block
        glds16h((const char*)ws + src_off + wave * 2048, (unsigned)lane * 16u, lds_base + dst_off + (unsigned)wave * 2048u);
        glds16h((const char*)ws + src_off + wave * 2048 + 1024, (unsigned)lane * 16u, lds_base + dst_off + (unsigned)wave * 2048u + 1024u);
    };
    issue(0, 0);
    if (nstages > 0) issue(EH_RES, EH_RES);
    if (nstages > 1) issue(EH_RES + EH_STAGE, EH_RES + EH_STAGE);

    const int tile = blockIdx.x * tpw + wave;
    const bool active = wave < tpw && tile * 16 < M;                 // wave-uniform
    const int tok = tile * 16 + lj;
    const bool tok_ok = active && tok < M;
    const int tk = tok_ok ? tok : M - 1;
    const int img = tk / S, tpos = tk - img * S;
    float x[4][4];
    bf16x8 xh[2], xl[2];
    f16x8 xf[2];
    if (active) {
        // attn: lane (token, kq = lq) of k-group G reads head 4G + lq, eight dims = 16 bytes of fp16 = a hi + lo bf16 pair exactly
        u32x4 ah[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) ah[g] = *reinterpret_cast<const u32x4*>(attn_hm + (((int64_t)img * 8 + 4 * g + lq) * S + tpos) * 8);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const Split4 a = split4(half_lo(ah[g][0]), half_hi(ah[g][0]), half_lo(ah[g][1]), half_hi(ah[g][1]));
            const Split4 b = split4(half_lo(ah[g][2]), half_hi(ah[g][2]), half_lo(ah[g][3]), half_hi(ah[g][3]));
            xh[g] = cat8(a.hi, b.hi);
            xl[g] = cat8(a.lo, b.lo);
        }
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk * EH_C + fb * 16 + lq * 4);
            x[fb][0] = r.x; x[fb][1] = r.y; x[fb][2] = r.z; x[fb][3] = r.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- output_proj + residual + LayerNorm1 (msdeformattn.py:124-126): w(h + l) x(h + l) without l x l ----
    if (active) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const float4 b = *reinterpret_cast<const float4*>(sm + EH_BO + ob * 16 + lq * 4);
            f32x4 d = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const char* blk = res + ((ob * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                d = mfma_bf16k32(wl, xh[g], d);
                d = mfma_bf16k32(wh, xl[g], d);
                d = mfma_bf16k32(wh, xh[g], d);
            }
            x[ob][0] += d[0]; x[ob][1] += d[1]; x[ob][2] += d[2]; x[ob][3] += d[3];
        }
        layer_norm_h(x, sm + EH_G1, sm + EH_BE1, lq, eps);
        if constexpr (F16) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                xf[g] = cvt8h(x[2 * g][0], x[2 * g][1], x[2 * g][2], x[2 * g][3], x[2 * g + 1][0], x[2 * g + 1][1], x[2 * g + 1][2], x[2 * g + 1][3]);
        } else {
            split_L(x, xh, xl);
        }
    }
    // ---- FFN: four pairs of 16-wide hidden blocks per stage; the hidden activation never leaves registers ----
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    float pq[4][4];
    for (int s = 0; s < nffn; ++s) {
        const bool more = s + 2 < nstages;
        // The next layer's query is src_out + pos (msdeformattn.py:124).  An ordinary load beside LDS-DMA in flight: hipcc waits
        // for it with vmcnt(0) where it is used -- so it is issued ahead of the last FFN stage's pieces and used before the
        // next pieces are requested (below): that wait then retires nothing that has not landed long ago.
        if (next && active && s == nffn - 1) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const float4 r = *reinterpret_cast<const float4*>(pos + (int64_t)tpos * EH_C + fb * 16 + lq * 4);
                pq[fb][0] = r.x; pq[fb][1] = r.y; pq[fb][2] = r.z; pq[fb][3] = r.w;
            }
        }
        // buffer (s + 2) % 3 was read in stage s - 1: every wave is past that stage's closing barrier
        if (more) issue(EH_RES + (int64_t)(s + 2) * EH_STAGE, EH_RES + (unsigned)((s + 2) % EH_NBUF) * EH_STAGE);
        if (active) {
            const char* buf = ring + (s % EH_NBUF) * EH_STAGE;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const char* pb = buf + p * 8192;
                f32x4 hh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 b = *reinterpret_cast<const float4*>(sm + EH_B1 + ((s * 4 + p) * 2 + q) * 16 + lq * 4);
                    hh[q] = f32x4{b.x, b.y, b.z, b.w};
                }
                bf16x8 w1[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < 2; ++g) w1[q][g] = ldfrag(pb + (q * 2 + g) * 1024, lane);
                if constexpr (F16) {
                    // consecutive MFMAs on different accumulators
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_f16k32(__builtin_bit_cast(f16x8, w1[q][g]), xf[g], hh[q]);
                    // ReLU and the clamp to the half range are one v_med3_f32
                    const f16x8 hb = {(_Float16)relu_h(hh[0][0]), (_Float16)relu_h(hh[0][1]), (_Float16)relu_h(hh[0][2]), (_Float16)relu_h(hh[0][3]),
                                      (_Float16)relu_h(hh[1][0]), (_Float16)relu_h(hh[1][1]), (_Float16)relu_h(hh[1][2]), (_Float16)relu_h(hh[1][3])};
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob)
                        acc[ob] = mfma_f16k32(__builtin_bit_cast(f16x8, ldfrag(pb + 4096 + ob * 1024, lane)), hb, acc[ob]);
                } else {
                    // low-order term first; consecutive MFMAs on different accumulators
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_bf16k32(w1[q][g], xl[g], hh[q]);
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_bf16k32(w1[q][g], xh[g], hh[q]);
                    const bf16x8 hb = cat8(pack4(relu1h(hh[0][0]), relu1h(hh[0][1]), relu1h(hh[0][2]), relu1h(hh[0][3])),
                                           pack4(relu1h(hh[1][0]), relu1h(hh[1][1]), relu1h(hh[1][2]), relu1h(hh[1][3])));
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[ob] = mfma_bf16k32(ldfrag(pb + 4096 + ob * 1024, lane), hb, acc[ob]);
                }
            }
        }
        // stage s + 1 (requested at the start of stage s - 1) must have landed: everything but this stage's own two pieces
        if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (s + 1 < nstages) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (active) finish_ffn(x, acc, sm, lq, eps);                     // residual + LayerNorm2: the layer output
    if (!next) {
        if (active) store_src(src_out, x, tok, tok_ok, lq);
        return;
    }
    if (active) {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 4; ++r) pq[fb][r] += x[fb][r];
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) asm volatile("" : "+v"(pq[fb][0]));   // (all four loads are waited for here, ahead of the next request)
    issue(EH_RES + (int64_t)(nffn + 2) * EH_STAGE, EH_RES + (unsigned)((nffn + 2) % EH_NBUF) * EH_STAGE);
    if (active) {
        store_src(src_out, x, tok, tok_ok, lq);                                                              // 4 stores
        // the next layer's value_proj, w(h + l) x(h + l) without l x l: row blocks 2j, 2j + 1 of lane (token, lq) are dims
        // 0-3 / 4-7 of head 4j + lq
        split_L(x, xh, xl);
        f32x4 d[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const float4 b = *reinterpret_cast<const float4*>(sm + EH_BV + rb * 16 + lq * 4);
            d[rb] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const char* blk = res + 16384 + ((rb * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                d[rb] = mfma_bf16k32(wl, xh[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xl[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xh[g], d[rb]);
            }
        if (tok_ok && (EH_EXP != 3 || d[0][0] == 12345.678f)) {                                              // 2 stores
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<u32x4*>(value_out + (((int64_t)img * 8 + 4 * j + lq) * S + tpos) * 8) = pack8h(d[2 * j], d[2 * j + 1]);
        }
        split_L(pq, xh, xl);                                         // the query operand of the projection stages
    }
    // ---- [sampling_offsets | attention_weights](src_out + pos), ms_deform_attn.py:99-101: eight row blocks per stage; row
    // 16 rb + 4 lq + r: an offset (rb < 12) or a logit of the head-major record (store_proj_rb): one store per row block ----
#pragma unroll 1
    for (int t = 0; t < EH_PROJ_STAGES; ++t) {
        const char* buf = ring + ((nffn + t) % EH_NBUF) * EH_STAGE;
        if (active) {
#pragma unroll 2
            for (int j = 0; j < 8; ++j) {
                const int rb = t * 8 + j;
                if (rb < EH_PROJ / 16) {                              // (uniform)
                    const float4 b = *reinterpret_cast<const float4*>(sm + EH_BP + rb * 16 + lq * 4);
                    f32x4 d = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const char* blk = buf + ((j * 2 + g) * 2) * 1024;
                        const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                        d = mfma_bf16k32(wl, xh[g], d);
                        d = mfma_bf16k32(wh, xl[g], d);
                        d = mfma_bf16k32(wh, xh[g], d);
                    }
                    if (tok_ok && (EH_EXP != 3 || d[0] == 12345.678f))
                        store_proj_rb(reinterpret_cast<unsigned char*>(proj_out), img, tpos, S, rb, lq, d);
                }
            }
        }
        // The next stage must have landed: everything but what this wave has issued since it was requested.  t = 0: stage
        // nffn + 1 was requested before the last FFN stage; since then two pieces + 4 + 2 + 8 stores.  t = 1: stage nffn + 2 was
        // requested before the 14 stores of t = 0, then 8 more.  (CDNA4's vmcnt counts stores, in order; a partly valid tile
        // issues every store instruction, exec-masked.)  Waves without a tile issue no stores.
        if (t == 0) {
            if (active) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else if (t == 1) {
            if (active) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (t + 1 < EH_PROJ_STAGES) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---- the same layer tail with HALF the workgroup (round 6): two workgroups per CU --------------------------------------------------
// enc_block_hm_kernel is one 1024-thread workgroup per CU with 133 KB of LDS: its 256 workgroups march in lock step -- matrix pipe busy a
// quarter of the time, then 67 MB of stores in one burst -- and no second launch (another batch in flight) can share a CU with it.  This
// form reads the SAME weight stream in 16-KiB stages (two pairs of hidden blocks; four projection row blocks) with eight waves: resident
// block = output_proj only (16 KiB; the value projection becomes a stage of its own between the FFN and the sampling projection), ring of
// three stages, 71 KB of LDS per workgroup -- two workgroups per CU, each on its own schedule (and either may belong to another batch).
// Per wave and stage two 1-KiB DMA pieces, as before.  Stage sequence: FFN 0 .. nf-1 (nf = d_ffn / 64), [value, projection 0 .. 4].
constexpr int E2_WAVES = 8;
constexpr int E2_STAGE = 16 * 1024;
constexpr int E2_RES = 16 * 1024;
constexpr int E2_PROJ_STAGES = 5;              // 18 row blocks, four per stage

template <bool F16>
__global__ __launch_bounds__(E2_WAVES * 64, 4) void enc_block_hm2_kernel(const unsigned short* __restrict__ attn_hm, const float* __restrict__ src,
                                                                        const char* __restrict__ wstream, const float* __restrict__ small,
                                                                        const float* __restrict__ pos, float* __restrict__ src_out,
                                                                        unsigned short* __restrict__ value_out, unsigned short* __restrict__ proj_out, int M,
                                                                        int S, int nf, float eps, int tpw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [output_proj 16 KiB][3 stages x 16 KiB][small]
    char* const res = lds;
    char* const ring = lds + E2_RES;
    float* const sm = reinterpret_cast<float*>(lds + E2_RES + EH_NBUF * E2_STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int n_small = EH_B1 + 64 * nf;
    const bool next = value_out != nullptr;                          // (uniform) not the last layer
    const int nstages = nf + (next ? 1 + E2_PROJ_STAGES : 0);
    for (int i = tid; i < n_small; i += E2_WAVES * 64) sm[i] = small[i];

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const void* ws = uniform_ptr_lp(wstream);
    // byte offset of stage j in the stream of pack_encoder_block_hm: [output_proj 16K | value_proj 16K | FFN nf x 16K | projection 96K]
    auto stage_off = [&](int j) -> int64_t {
        if (j < nf) return 2 * (int64_t)E2_STAGE + (int64_t)j * E2_STAGE;
        if (j == nf) return E2_STAGE;                                // the value projection
        return 2 * (int64_t)E2_STAGE + (int64_t)nf * E2_STAGE + (int64_t)(j - nf - 1) * E2_STAGE;
    };
    auto issue_at = [&](int64_t src_off, unsigned dst_off) {         // this wave's two pieces of a 16-KiB block
        glds16h((const char*)ws + src_off + wave * 2048, (unsigned)lane * 16u, lds_base + dst_off + (unsigned)wave * 2048u);
        glds16h((const char*)ws + src_off + wave * 2048 + 1024, (unsigned)lane * 16u, lds_base + dst_off + (unsigned)wave * 2048u + 1024u);
    };
    auto issue = [&](int j) { issue_at(stage_off(j), E2_RES + (unsigned)(j % EH_NBUF) * E2_STAGE); };
    issue_at(0, 0);
    if (nstages > 0) issue(0);
    if (nstages > 1) issue(1);

    const int tile = blockIdx.x * tpw + wave;
    const bool active = wave < tpw && tile * 16 < M;                 // wave-uniform
    const int tok = tile * 16 + lj;
    const bool tok_ok = active && tok < M;
    const int tk = tok_ok ? tok : M - 1;
    const int img = tk / S, tpos = tk - img * S;
    float x[4][4];
    bf16x8 xh[2], xl[2];
    f16x8 xf[2];
    if (active) {
        u32x4 ah[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) ah[g] = *reinterpret_cast<const u32x4*>(attn_hm + (((int64_t)img * 8 + 4 * g + lq) * S + tpos) * 8);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const Split4 a = split4(half_lo(ah[g][0]), half_hi(ah[g][0]), half_lo(ah[g][1]), half_hi(ah[g][1]));
            const Split4 b = split4(half_lo(ah[g][2]), half_hi(ah[g][2]), half_lo(ah[g][3]), half_hi(ah[g][3]));
            xh[g] = cat8(a.hi, b.hi);
            xl[g] = cat8(a.lo, b.lo);
        }
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk * EH_C + fb * 16 + lq * 4);
            x[fb][0] = r.x; x[fb][1] = r.y; x[fb][2] = r.z; x[fb][3] = r.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- output_proj + residual + LayerNorm1 (msdeformattn.py:124-126): w(h + l) x(h + l) without l x l ----
    if (active) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const float4 b = *reinterpret_cast<const float4*>(sm + EH_BO + ob * 16 + lq * 4);
            f32x4 d = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const char* blk = res + ((ob * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                d = mfma_bf16k32(wl, xh[g], d);
                d = mfma_bf16k32(wh, xl[g], d);
                d = mfma_bf16k32(wh, xh[g], d);
            }
            x[ob][0] += d[0]; x[ob][1] += d[1]; x[ob][2] += d[2]; x[ob][3] += d[3];
        }
        layer_norm_h(x, sm + EH_G1, sm + EH_BE1, lq, eps);
        if constexpr (F16) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                xf[g] = cvt8h(x[2 * g][0], x[2 * g][1], x[2 * g][2], x[2 * g][3], x[2 * g + 1][0], x[2 * g + 1][1], x[2 * g + 1][2], x[2 * g + 1][3]);
        } else {
            split_L(x, xh, xl);
        }
    }
    // ---- FFN: two pairs of 16-wide hidden blocks per stage; the hidden activation never leaves registers ----
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    float pq[4][4];
    for (int s = 0; s < nf; ++s) {
        const bool more = s + 2 < nstages;
        // the next layer's query is src_out + pos: an ordinary load beside LDS-DMA in flight, issued ahead of the last FFN stage's pieces
        // and used before the next pieces are requested (see enc_block_hm_kernel)
        if (next && active && s == nf - 1) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const float4 r = *reinterpret_cast<const float4*>(pos + (int64_t)tpos * EH_C + fb * 16 + lq * 4);
                pq[fb][0] = r.x; pq[fb][1] = r.y; pq[fb][2] = r.z; pq[fb][3] = r.w;
            }
        }
        if (more) issue(s + 2);                                      // its buffer was read in stage s - 1: every wave is past that stage's barrier
        if (active) {
            const char* buf = ring + (s % EH_NBUF) * E2_STAGE;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const char* pb = buf + p * 8192;
                f32x4 hh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 b = *reinterpret_cast<const float4*>(sm + EH_B1 + ((s * 2 + p) * 2 + q) * 16 + lq * 4);
                    hh[q] = f32x4{b.x, b.y, b.z, b.w};
                }
                bf16x8 w1[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < 2; ++g) w1[q][g] = ldfrag(pb + (q * 2 + g) * 1024, lane);
                if constexpr (F16) {
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_f16k32(__builtin_bit_cast(f16x8, w1[q][g]), xf[g], hh[q]);
                    const f16x8 hb = {(_Float16)relu_h(hh[0][0]), (_Float16)relu_h(hh[0][1]), (_Float16)relu_h(hh[0][2]), (_Float16)relu_h(hh[0][3]),
                                      (_Float16)relu_h(hh[1][0]), (_Float16)relu_h(hh[1][1]), (_Float16)relu_h(hh[1][2]), (_Float16)relu_h(hh[1][3])};
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob)
                        acc[ob] = mfma_f16k32(__builtin_bit_cast(f16x8, ldfrag(pb + 4096 + ob * 1024, lane)), hb, acc[ob]);
                } else {
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_bf16k32(w1[q][g], xl[g], hh[q]);
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) hh[q] = mfma_bf16k32(w1[q][g], xh[g], hh[q]);
                    const bf16x8 hb = cat8(pack4(relu1h(hh[0][0]), relu1h(hh[0][1]), relu1h(hh[0][2]), relu1h(hh[0][3])),
                                           pack4(relu1h(hh[1][0]), relu1h(hh[1][1]), relu1h(hh[1][2]), relu1h(hh[1][3])));
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[ob] = mfma_bf16k32(ldfrag(pb + 4096 + ob * 1024, lane), hb, acc[ob]);
                }
            }
        }
        // stage s + 1 (requested at the start of stage s - 1) must have landed: everything but this stage's own two pieces
        if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (s + 1 < nstages) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (active) finish_ffn(x, acc, sm, lq, eps);                     // residual + LayerNorm2: the layer output
    if (!next) {
        if (active) store_src(src_out, x, tok, tok_ok, lq);
        return;
    }
    if (active) {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 4; ++r) pq[fb][r] += x[fb][r];
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) asm volatile("" : "+v"(pq[fb][0]));   // (the four pos loads are waited for here, ahead of the next request)
    // ---- stage nf: the next layer's value_proj, w(h + l) x(h + l) without l x l; 4 + 2 stores ----
    issue(nf + 2);
    if (active) {
        store_src(src_out, x, tok, tok_ok, lq);
        split_L(x, xh, xl);
        const char* vb = ring + (nf % EH_NBUF) * E2_STAGE;
        f32x4 d[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const float4 b = *reinterpret_cast<const float4*>(sm + EH_BV + rb * 16 + lq * 4);
            d[rb] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const char* blk = vb + ((rb * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                d[rb] = mfma_bf16k32(wl, xh[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xl[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xh[g], d[rb]);
            }
        if (tok_ok) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<u32x4*>(value_out + (((int64_t)img * 8 + 4 * j + lq) * S + tpos) * 8) = pack8h(d[2 * j], d[2 * j + 1]);
        }
        split_L(pq, xh, xl);                                         // the query operand of the projection stages
    }
    // stage nf + 1 (requested before the last FFN stage) must have landed: everything but this stage's two pieces and its 4 + 2 stores
    if (active) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- stages nf + 1 + t: [sampling_offsets | attention_weights](src_out + pos), four row blocks per stage, one store each ----
#pragma unroll 1
    for (int t = 0; t < E2_PROJ_STAGES; ++t) {
        const int j = nf + 1 + t;
        const bool more = j + 2 < nstages;
        if (more) issue(j + 2);
        const char* buf = ring + (j % EH_NBUF) * E2_STAGE;
        if (active) {
#pragma unroll 2
            for (int i = 0; i < 4; ++i) {
                const int rb = t * 4 + i;
                if (rb < EH_PROJ / 16) {                              // (uniform)
                    const float4 b = *reinterpret_cast<const float4*>(sm + EH_BP + rb * 16 + lq * 4);
                    f32x4 d = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const char* blk = buf + ((i * 2 + g) * 2) * 1024;
                        const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                        d = mfma_bf16k32(wl, xh[g], d);
                        d = mfma_bf16k32(wh, xl[g], d);
                        d = mfma_bf16k32(wh, xh[g], d);
                    }
                    if (tok_ok) store_proj_rb(reinterpret_cast<unsigned char*>(proj_out), img, tpos, S, rb, lq, d);
                }
            }
        }
        // Stage j + 1 must have landed: everything but what this wave issued after requesting it (at the start of stage j - 1): the stores of
        // stage j - 1 (6 after the value stage, else 4), this stage's two pieces (when requested) and this stage's 4 stores.  (vmcnt counts
        // stores, in order; a partly valid tile issues every store instruction, exec-masked; waves without a tile issue no stores.)
        if (t + 1 < E2_PROJ_STAGES) {
            if (active) {
                if (t == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (more) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---- gather ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float quad_max_lp(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)));
    return fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)));
}
__device__ __forceinline__ float quad_sum_lp(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ int clamp0_lp(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
__device__ __forceinline__ float rcp_nr_lp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_by_lp(float x, float W, float rW) {
    const float q = x * rW;
    return fmaf(fmaf(-q, W, x), rW, q);
}

// Workgroup = 64 consecutive queries of ONE head (4 waves x 16 queries), as msda_enc_hm8_fused_kernel (msda.hip):
//  A  the head's 36 projection rows (24 offsets, 12 logits; padded to three 16-row blocks, hi + lo bf16 copies, 12 KiB in LDS)
//     times x = src + pos of the wave's 16 tokens: 18 v_mfma_f32_16x16x32_bf16, result through LDS ([token][52 floats]);
//  B  quad = query; lane g OWNS points g, g + 4, g + 8 (one per level): softmax share, tap geometry once for both columns and
//     both rows, left as 8-byte {byte offset, weight} records in LDS (msda_enc_hm8_rec_kernel's scheme);
//  C  lane g = (column cx, row ry) of a tap: per point one ds_read_b64, ONE 16-byte load (the head's eight fp16 dims of that
//     tap), eight v_fma_mix_f32; the quad's four partial sums meet by DPP and lane 0 stores the head's 16 bytes.
// The fp32 kernel needed two loads per lane and point (a lane = a column and a 16-byte HALF of the fp32 dims).
// FUSED = false: the projection was written by enc_block_hm_kernel as head-major records proj[b][m][q] of EH_REC bytes (24 fp32 offsets in
// (level, point, xy) order, 12 logits): phase A is six small loads per owner lane.
template <int LC, bool FUSED>
__global__ __launch_bounds__(256) void msda_enc_lp_kernel(const unsigned short* __restrict__ value, const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lstart, const float* __restrict__ src,
                                                          const float* __restrict__ pos, const u32x4* __restrict__ wpack,
                                                          const float* __restrict__ bpack, const unsigned short* __restrict__ proj,
                                                          unsigned short* __restrict__ out, int B, int S, int M) {
    constexpr int LP = LC * 4, SLOTS = LC;
    static_assert(LP == 12, "three 16-row blocks hold the 36 projection rows of a head");
    constexpr int QSTRIDE = LP * 2 + 2;                      // 16-byte units per quad: LP x 2 records + padding (conflict-free b64 reads)
    constexpr int TSTRIDE = 52;                              // floats per token row of the projection tile
    constexpr int WU = 12 * 64;                              // 16-byte units of a head's weight fragments (12 KiB)
    constexpr int REGION = FUSED && (WU + 4 * 16 * TSTRIDE / 4) > 4 * 16 * QSTRIDE ? (WU + 4 * 16 * TSTRIDE / 4) : 4 * 16 * QSTRIDE;
    __shared__ float4 recs[REGION];
    const int b = blockIdx.x % B;
    const int blk = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blk % M;                                   // uniform: the head of this workgroup
    const int q0 = (blk / M) * 64 + wave * 16;               // first query of this wave
    const int lj = lane & 15, lq = lane >> 4;
    float x[4][4];
    if constexpr (FUSED) {
        const int tk = min(q0 + lj, S - 1);
        float4 a[4], p[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            a[fb] = *reinterpret_cast<const float4*>(src + ((int64_t)b * S + tk) * EH_C + fb * 16 + lq * 4);
            p[fb] = *reinterpret_cast<const float4*>(pos + (int64_t)tk * EH_C + fb * 16 + lq * 4);
        }
#pragma unroll
        for (int i = 0; i < WU / 256; ++i) reinterpret_cast<u32x4*>(recs)[i * 256 + tid] = wpack[m * WU + i * 256 + tid];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            x[fb][0] = a[fb].x + p[fb].x; x[fb][1] = a[fb].y + p[fb].y; x[fb][2] = a[fb].z + p[fb].z; x[fb][3] = a[fb].w + p[fb].w;
        }
    }
    int Hs[LC], Ws[LC], st[LC];                              // level geometry: wave-uniform, stays in SGPRs
    float rws[LC], rhs[LC];
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        Hs[l] = (int)shapes[2 * l];
        Ws[l] = (int)shapes[2 * l + 1];
        st[l] = (int)lstart[l];
        rws[l] = rcp_nr_lp((float)Ws[l]);
        rhs[l] = rcp_nr_lp((float)Hs[l]);
    }
    float* tile = reinterpret_cast<float*>(recs + WU) + wave * 16 * TSTRIDE;
    if constexpr (FUSED) {
        // ---- A: the head's projection of the wave's 16 tokens ----
        bf16x8 xh[2], xl[2];
        split_L(x, xh, xl);
        __syncthreads();
        const char* wl = reinterpret_cast<const char*>(recs);
        f32x4 d[3];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const float4 bb = *reinterpret_cast<const float4*>(bpack + m * 48 + rb * 16 + lq * 4);
            d[rb] = f32x4{bb.x, bb.y, bb.z, bb.w};
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const char* fb = wl + ((rb * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(fb, lane), wlo = ldfrag(fb + 1024, lane);
                d[rb] = mfma_bf16k32(wlo, xh[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xl[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xh[g], d[rb]);
            }
        float* trow = tile + lj * TSTRIDE + lq * 4;
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) *reinterpret_cast<float4*>(trow + rb * 16) = make_float4(d[rb][0], d[rb][1], d[rb][2], d[rb][3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- B: quad = query ----
    const int g = tid & 3, cx = g >> 1, ry = g & 1;
    const int quad = lane >> 2;
    const int q_raw = q0 + quad;
    const bool live = q_raw < S;                             // whole quads live or dead together
    const int qi = live ? q_raw : S - 1;
    int qW = Ws[0], qH = Hs[0], qs = 0;
    float qrw = rws[0], qrh = rhs[0];
#pragma unroll
    for (int l = 1; l < LC; ++l)
        if (qi >= st[l]) { qW = Ws[l]; qH = Hs[l]; qs = st[l]; qrw = rws[l]; qrh = rhs[l]; }
    const int local = qi - qs;
    const int rrow = (int)(((float)local + 0.5f) * qrw), rcol = local - rrow * qW;
    const float ref_x = div_by_lp((float)rcol + 0.5f, (float)qW, qrw);       // msdeformattn.py:141-153
    const float ref_y = div_by_lp((float)rrow + 0.5f, (float)qH, qrh);
    const float* tq = tile + quad * TSTRIDE;
    const unsigned char* pr = reinterpret_cast<const unsigned char*>(proj) + proj_region(b, m, S);       // (M = 8: the head count of the records)
    float px[SLOTS], py[SLOTS], pw[SLOTS];
    float mx = -INFINITY;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int i = slot * 4 + g;
        float2 off;
        float lg;
        if constexpr (FUSED) {
            off = *reinterpret_cast<const float2*>(tq + 2 * i);
            lg = tq[2 * LP + i];
        } else {
            const f32x2_a8 o2 = *reinterpret_cast<const f32x2_a8*>(pr + ((int64_t)(i >> 1) * S + qi) * 16 + (i & 1) * 8);     // fp32 (x, y) of point i
            off = make_float2(o2[0], o2[1]);
            lg = half_lo((unsigned)*reinterpret_cast<const unsigned short*>(pr + (int64_t)S * 96 + ((int64_t)(i >> 2) * S + qi) * 8 + (i & 3) * 2));
        }
        const float lx = ref_x + div_by_lp(off.x, (float)Ws[slot], rws[slot]);      // ms_deform_attn.py:107-109
        const float ly = ref_y + div_by_lp(off.y, (float)Hs[slot], rhs[slot]);
        px[slot] = lx * (float)Ws[slot] - 0.5f;                                  // w_im, cuh:290-291
        py[slot] = ly * (float)Hs[slot] - 0.5f;                                  // h_im
        pw[slot] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = quad_max_lp(mx);                                                        // softmax over the L*P logits (:103)
    float den = 0.f;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        pw[slot] = __builtin_amdgcn_exp2f((pw[slot] - mx) * 1.4426950408889634f);
        den += pw[slot];
    }
    const float rden = rcp_nr_lp(quad_sum_lp(den));
    if constexpr (FUSED) __syncthreads();                    // every wave has read the weights and its tile: records may overwrite them

    float4* qrec = recs + (wave * 16 + quad) * QSTRIDE;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int W = Ws[slot], H = Hs[slot], s0 = st[slot];
        const float w_im = px[slot], h_im = py[slot];
        const float wgt = pw[slot] * rden;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf;
        const float wx0 = (unsigned)w_low < (unsigned)W ? (1.f - lw) * wgt : 0.f;        // per-tap bounds, cuh:247-270
        const float wx1 = (unsigned)(w_low + 1) < (unsigned)W ? lw * wgt : 0.f;
        const bool okt = (unsigned)h_low < (unsigned)H, okb = (unsigned)(h_low + 1) < (unsigned)H;
        const unsigned rowt = __umul24((unsigned)clamp0_lp(h_low, H - 1), (unsigned)(W * 16));
        const unsigned rowb = __umul24((unsigned)clamp0_lp(h_low + 1, H - 1), (unsigned)(W * 16));
        const unsigned c0 = (unsigned)(s0 + clamp0_lp(w_low, W - 1)) * 16u;
        const unsigned c1 = (unsigned)(s0 + clamp0_lp(w_low + 1, W - 1)) * 16u;
        const int i = slot * 4 + g;
        qrec[i * 2 + 0] = make_float4(__uint_as_float(c0 + rowt), okt ? (1.f - lh) * wx0 : 0.f, __uint_as_float(c0 + rowb), okb ? lh * wx0 : 0.f);
        qrec[i * 2 + 1] = make_float4(__uint_as_float(c1 + rowt), okt ? (1.f - lh) * wx1 : 0.f, __uint_as_float(c1 + rowb), okb ? lh * wx1 : 0.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- C: gather; the head's value plane (S x 16 bytes) behind one SGPR descriptor ----
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr_lp(value + ((int64_t)b * M + m) * S * 8), 0, S * 16, 0x00020000);
    const float2* myrec = reinterpret_cast<const float2*>(qrec + cx) + ry;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    constexpr int GP = 6;                                    // loads of GP points in flight together
#pragma unroll
    for (int i0 = 0; i0 < LP; i0 += GP) {
        float2 r[GP];
        u32x4 v[GP];
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            r[j] = myrec[(i0 + j) * 4];                      // record i = 32 bytes = four float2
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, __float_as_uint(r[j].x), 0, 0);
        }
#pragma unroll
        for (int j = 0; j < GP; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[2 * c] = fmaf(r[j].y, half_lo(v[j][c]), acc[2 * c]);                  // (v_fma_mix_f32: the conversion rides in the FMA)
                acc[2 * c + 1] = fmaf(r[j].y, half_hi(v[j][c]), acc[2 * c + 1]);
            }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = quad_sum_lp(acc[c]);
    if (live && g == 0) {
        const u32x2b ul = pack4h(acc[0], acc[1], acc[2], acc[3]), uh = pack4h(acc[4], acc[5], acc[6], acc[7]);
        *reinterpret_cast<u32x4*>(out + (((int64_t)b * M + m) * S + qi) * 8) = u32x4{ul.x, ul.y, uh.x, uh.y};
    }
}

// fp32 -> fp16 (round to nearest even, clamped to the half range), eight values per lane; n8 = elements / 8
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float4* __restrict__ in, u32x4* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = in[2 * i], b = in[2 * i + 1];
        const u32x2b ul = pack4h(a.x, a.y, a.z, a.w), uh = pack4h(b.x, b.y, b.z, b.w);
        out[i] = u32x4{ul.x, ul.y, uh.x, uh.y};
    }
}

// the same for [B][n] rows that are contiguous inside an image but spaced by in_bs floats between images (a level's token range of the
// encoder's concatenated buffer): one launch instead of a copy to contiguous + the conversion
__global__ __launch_bounds__(256) void f32_to_f16_rows_kernel(const float4* __restrict__ in, u32x4* __restrict__ out, int64_t n8, int64_t in_bs4) {
    const float4* src = in + (int64_t)blockIdx.y * in_bs4;
    u32x4* dst = out + (int64_t)blockIdx.y * n8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        const u32x2b ul = pack4h(a.x, a.y, a.z, a.w), uh = pack4h(b.x, b.y, b.z, b.w);
        dst[i] = u32x4{ul.x, ul.y, uh.x, uh.y};
    }
}

// ---- the encoder prologue of this plan (round 4) ----------------------------------------------------------------------------------
// What precedes the first deformable-attention layer (msdeformattn.py:326-329 GroupNorm of the input projections, :60-75 level
// concatenation, layer 0's value_proj / sampling_offsets / attention_weights linears, ops/modules/ms_deform_attn.py:95-104) with the
// two projections as in enc_block_hm_kernel's tail: hi + lo bf16 operands, three K = 32 MFMAs per product, value and sampling record
// written head-major in fp16.  The fp32 prologue (enc_block.hip) spends 352 fp32 MFMAs of 32 cycles on a 16-token tile -- with 3150
// tiles on 1024 SIMDs that is four tiles = 21 us of matrix time per SIMD before anything else; here a tile is 132 MFMAs of 16 cycles.
// Weight-stationary like that kernel: value (16 KiB) and projection (72 KiB) blocks are copied into LDS once per workgroup (all of a
// thread's loads requested before its first store), each of the 16 waves runs one tile, one barrier.
constexpr int PH_W = 16;
constexpr int PH_NIMG = 4;             // images a workgroup's 256 tokens may touch (S >= 86)
constexpr int PH_MAXL = 4;
constexpr int PH_WBYTES = 16384 + 18 * 4096;
struct HmLevels {
    int n;
    int start[PH_MAXL + 1];
};

__global__ __launch_bounds__(PH_W * 64) void enc_prologue_hm_kernel(const float* __restrict__ raw, const double* __restrict__ stats,
                                                                    const float* __restrict__ gnp, HmLevels lv, int groups, float gn_eps,
                                                                    const u32x4* __restrict__ wblocks, const float* __restrict__ small,
                                                                    const float* __restrict__ pos, float* __restrict__ src_out,
                                                                    unsigned short* __restrict__ value_out, unsigned short* __restrict__ proj_out,
                                                                    int M, int S, int B) {
    extern __shared__ __attribute__((aligned(16))) char phl[];    // value + projection blocks | GroupNorm tables | bv, bp
    float* gt = reinterpret_cast<float*>(phl + PH_WBYTES);        // [PH_NIMG images][levels][3][64]: mean, rstd*gamma, beta
    float* sm = gt + PH_NIMG * PH_MAXL * 3 * EH_C;                // bv [64] (value row order), bp [288] (offsets-then-logits row order)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    // this tile's tokens first: their latency hides behind the weight copy and the tables
    const int tile = (int)blockIdx.x * PH_W + wave;
    const int tok = tile * 16 + lj;
    const bool tok_ok = tok < M;
    const int tk = tok_ok ? tok : M - 1;
    float x[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 r = *reinterpret_cast<const float4*>(raw + (int64_t)tk * EH_C + fb * 16 + lq * 4);
        x[fb][0] = r.x; x[fb][1] = r.y; x[fb][2] = r.z; x[fb][3] = r.w;
    }
    {
        constexpr int N16 = PH_WBYTES / 16, PER = (N16 + PH_W * 64 - 1) / (PH_W * 64);      // 5632 pieces, 6 per thread (the last partly)
        u32x4 wv[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) wv[k] = wblocks[min(tid + k * (PH_W * 64), N16 - 1)];
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (tid + k * (PH_W * 64) < N16) reinterpret_cast<u32x4*>(phl)[tid + k * (PH_W * 64)] = wv[k];
    }
    for (int i = tid; i < EH_C + EH_PROJ; i += PH_W * 64) sm[i] = small[i];
    const int b0 = (int)(((int64_t)blockIdx.x * PH_W * 16) / S);
    const int cpg = EH_C / groups;
    for (int i = tid; i < PH_NIMG * lv.n * EH_C; i += PH_W * 64) {
        const int c = i % EH_C, l = (i / EH_C) % lv.n, bi = b0 + i / (EH_C * lv.n);
        float mean = 0.f, a = 0.f, be = 0.f;
        if (bi < B) {
            const int g0 = (c / cpg) * cpg;
            double sum = 0.0, sq = 0.0;
            for (int k = 0; k < cpg; ++k) {
                const double* d = stats + (((int64_t)l * B + bi) * EH_C + g0 + k) * 2;
                sum += d[0];
                sq += d[1];
            }
            const double cnt = (double)cpg * (double)(lv.start[l + 1] - lv.start[l]);
            const double mu = sum / cnt;
            double var = sq / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            a = (float)(1.0 / sqrt(var + (double)gn_eps)) * gnp[(l * 2 + 0) * EH_C + c];
            be = gnp[(l * 2 + 1) * EH_C + c];
        }
        float* t = gt + ((i / (EH_C * lv.n)) * PH_MAXL + l) * 3 * EH_C;
        t[c] = mean;
        t[EH_C + c] = a;
        t[2 * EH_C + c] = be;
    }
    const int img = tk / S, tpos = tk - img * S;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < PH_MAXL; ++l) lvl += (l < lv.n && tpos >= lv.start[l]) ? 1 : 0;
    __syncthreads();               // the only barrier: weights, tables and biases are in LDS
    if (tile * 16 >= M) return;    // wave-uniform
    {
        const float* t = gt + ((img - b0) * PH_MAXL + lvl) * 3 * EH_C;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const int c = fb * 16 + lq * 4;
            const float4 mn = *reinterpret_cast<const float4*>(t + c);
            const float4 sc = *reinterpret_cast<const float4*>(t + EH_C + c);
            const float4 sh = *reinterpret_cast<const float4*>(t + 2 * EH_C + c);
            x[fb][0] = (x[fb][0] - mn.x) * sc.x + sh.x;
            x[fb][1] = (x[fb][1] - mn.y) * sc.y + sh.y;
            x[fb][2] = (x[fb][2] - mn.z) * sc.z + sh.z;
            x[fb][3] = (x[fb][3] - mn.w) * sc.w + sh.w;
        }
    }
    store_src(src_out, x, tok, tok_ok, lq);
    // query = src + pos (msdeformattn.py:124): requested now, added after the value projection
    float4 pp[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) pp[fb] = *reinterpret_cast<const float4*>(pos + (int64_t)tpos * EH_C + fb * 16 + lq * 4);
    bf16x8 xh[2], xl[2];
    split_L(x, xh, xl);
    {
        // value_proj: row blocks 2j, 2j + 1 of lane (token, lq) are dims 0-3 / 4-7 of head 4j + lq (one 16-byte store per j)
        f32x4 d[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const float4 b = *reinterpret_cast<const float4*>(sm + rb * 16 + lq * 4);
            d[rb] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const char* blk = phl + ((rb * 2 + g) * 2) * 1024;
                const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
                d[rb] = mfma_bf16k32(wl, xh[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xl[g], d[rb]);
                d[rb] = mfma_bf16k32(wh, xh[g], d[rb]);
            }
        if (tok_ok) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<u32x4*>(value_out + (((int64_t)img * 8 + 4 * j + lq) * S + tpos) * 8) = pack8h(d[2 * j], d[2 * j + 1]);
        }
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        x[fb][0] += pp[fb].x; x[fb][1] += pp[fb].y; x[fb][2] += pp[fb].z; x[fb][3] += pp[fb].w;
    }
    split_L(x, xh, xl);
    // [sampling_offsets | attention_weights](src + pos): row block rb < 12 = offsets, else logits (store_proj_rb)
#pragma unroll 2
    for (int rb = 0; rb < EH_PROJ / 16; ++rb) {
        const float4 b = *reinterpret_cast<const float4*>(sm + EH_C + rb * 16 + lq * 4);
        f32x4 d = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const char* blk = phl + 16384 + ((rb * 2 + g) * 2) * 1024;
            const bf16x8 wh = ldfrag(blk, lane), wl = ldfrag(blk + 1024, lane);
            d = mfma_bf16k32(wl, xh[g], d);
            d = mfma_bf16k32(wh, xl[g], d);
            d = mfma_bf16k32(wh, xh[g], d);
        }
        if (tok_ok) store_proj_rb(reinterpret_cast<unsigned char*>(proj_out), img, tpos, S, rb, lq, d);
    }
}

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_encoder_block_hm_stream_bytes(int d_ffn, int with_next) {
    return (int64_t)EH_RES + (int64_t)(cdiv(d_ffn, 128) + (with_next ? EH_PROJ_STAGES : 0)) * EH_STAGE;
}
extern "C" int msm_encoder_block_hm_small_floats(int d_ffn) { return EH_B1 + 128 * cdiv(d_ffn, 128); }

extern "C" int msm_encoder_block_hm_fwd(const void* attn_hm, const float* src, const void* wstream, const float* small, const float* pos,
                                        float* src_out, void* value_out, void* proj_out, int M, int tokens_per_image, int d_ffn, float eps,
                                        int ffn_f16, void* stream) {
    const char* who = "msm_encoder_block_hm_fwd";
    MSM_REQUIRE(attn_hm && src && wstream && small && src_out, "%s: null pointer", who);
    MSM_REQUIRE((value_out == nullptr) == (proj_out == nullptr) && (value_out == nullptr || pos != nullptr),
                "%s: value_out, proj_out and pos go together (all null for the last layer)", who);
    MSM_REQUIRE(M > 0 && tokens_per_image > 0 && M % tokens_per_image == 0 && d_ffn > 0 && d_ffn % 32 == 0,
                "%s: M=%d must be a multiple of tokens_per_image=%d, d_ffn=%d a positive multiple of 32", who, M, tokens_per_image, d_ffn);
    MSM_REQUIRE(((((uintptr_t)attn_hm) | ((uintptr_t)src) | ((uintptr_t)wstream) | ((uintptr_t)src_out) | ((uintptr_t)value_out) | ((uintptr_t)proj_out) | ((uintptr_t)small) | ((uintptr_t)pos)) & 15) == 0,
                "%s: pointers must be 16-byte aligned", who);
    const int nffn = cdiv(d_ffn, 128);
    const size_t lds = EH_RES + EH_NBUF * EH_STAGE + sizeof(float) * (size_t)(EH_B1 + 128 * nffn);
    MSM_REQUIRE(lds <= 160 * 1024, "%s: d_ffn=%d needs %zu bytes of LDS", who, d_ffn, lds);
    // one workgroup per CU while the tiles fit (<= 16 per workgroup): 3150 tiles -> 13 per workgroup on 243 CUs
    const int tiles = cdiv(M, 16);
    int tpw = cdiv(tiles, 256);
    if (tpw > EH_WAVES) tpw = EH_WAVES;
    const int grid = cdiv(tiles, tpw);
    // round 6: the half-size form (two workgroups per CU, see enc_block_hm2_kernel) -- MSM_OPT_ENC_NO_COOP = 2 selects the one-workgroup form
    if (opt(MSM_OPT_ENC_NO_COOP) != 2 && d_ffn % 128 == 0) {
        const int nf = d_ffn / 64;
        const size_t lds2 = E2_RES + EH_NBUF * E2_STAGE + sizeof(float) * (size_t)(EH_B1 + 64 * nf);
        int tpw2 = cdiv(tiles, 512);
        if (tpw2 > E2_WAVES) tpw2 = E2_WAVES;
        const int grid2 = cdiv(tiles, tpw2);
        if (ffn_f16) {
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_hm2_kernel<true>, lds2));
            hipLaunchKernelGGL(enc_block_hm2_kernel<true>, dim3(grid2), dim3(E2_WAVES * 64), lds2, (hipStream_t)stream, (const unsigned short*)attn_hm, src,
                               (const char*)wstream, small, pos, src_out, (unsigned short*)value_out, (unsigned short*)proj_out, M, tokens_per_image, nf,
                               eps, tpw2);
        } else {
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_hm2_kernel<false>, lds2));
            hipLaunchKernelGGL(enc_block_hm2_kernel<false>, dim3(grid2), dim3(E2_WAVES * 64), lds2, (hipStream_t)stream, (const unsigned short*)attn_hm, src,
                               (const char*)wstream, small, pos, src_out, (unsigned short*)value_out, (unsigned short*)proj_out, M, tokens_per_image, nf,
                               eps, tpw2);
        }
        MSM_CHECK_LAUNCH(who);
        return MSM_OK;
    }
    if (ffn_f16) {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_hm_kernel<true>, lds));
        hipLaunchKernelGGL(enc_block_hm_kernel<true>, dim3(grid), dim3(EH_WAVES * 64), lds, (hipStream_t)stream, (const unsigned short*)attn_hm, src,
                           (const char*)wstream, small, pos, src_out, (unsigned short*)value_out, (unsigned short*)proj_out, M, tokens_per_image, nffn,
                           eps, tpw);
    } else {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_hm_kernel<false>, lds));
        hipLaunchKernelGGL(enc_block_hm_kernel<false>, dim3(grid), dim3(EH_WAVES * 64), lds, (hipStream_t)stream, (const unsigned short*)attn_hm, src,
                           (const char*)wstream, small, pos, src_out, (unsigned short*)value_out, (unsigned short*)proj_out, M, tokens_per_image, nffn,
                           eps, tpw);
    }
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

static int msda_lp_checks(const char* who, const void* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index, const void* out_hm,
                          int B, int S, int M, int D, int L, int P) {
    MSM_REQUIRE(value_hm && spatial_shapes && level_start_index && out_hm, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && S > 0 && M > 0, "%s: empty problem", who);
    MSM_REQUIRE(M * D == 64 && D == 8 && L == 3 && P == 4, "%s: the shipped geometry only (64 channels, 8 heads, 3 levels x 4 points), got M=%d D=%d L=%d P=%d",
                who, M, D, L, P);
    MSM_REQUIRE((int64_t)S * 16 < (1ll << 31), "%s: S too large for 32-bit tap offsets", who);
    MSM_REQUIRE((int64_t)B * M * cdiv(S, 64) < (1ll << 31), "%s: grid too large", who);
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_enc_lp_fwd(const void* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                            const void* proj_hm, void* out_hm, int B, int S, int M, int D, int L, int P, void* stream) {
    const char* who = "msm_msdeform_attn_enc_lp_fwd";
    if (int rc = msda_lp_checks(who, value_hm, spatial_shapes, level_start_index, out_hm, B, S, M, D, L, P)) return rc;
    MSM_REQUIRE(proj_hm != nullptr, "%s: null pointer", who);
    MSM_REQUIRE(((((uintptr_t)value_hm) | ((uintptr_t)out_hm)) & 15) == 0 && (((uintptr_t)proj_hm) & 7) == 0, "%s: value / out must be 16-byte aligned, proj 8-byte", who);
    hipLaunchKernelGGL((msda_enc_lp_kernel<3, false>), dim3((unsigned)((int64_t)B * M * cdiv(S, 64))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)value_hm, spatial_shapes, level_start_index, (const float*)nullptr, (const float*)nullptr,
                       (const u32x4*)nullptr, (const float*)nullptr, (const unsigned short*)proj_hm, (unsigned short*)out_hm, B, S, M);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_enc_lp_fused_fwd(const void* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                                  const float* src, const float* pos, const void* wpack, const float* bpack, void* out_hm, int B,
                                                  int S, int M, int D, int L, int P, void* stream) {
    const char* who = "msm_msdeform_attn_enc_lp_fused_fwd";
    if (int rc = msda_lp_checks(who, value_hm, spatial_shapes, level_start_index, out_hm, B, S, M, D, L, P)) return rc;
    MSM_REQUIRE(src && pos && wpack && bpack, "%s: null pointer", who);
    MSM_REQUIRE(((((uintptr_t)value_hm) | ((uintptr_t)src) | ((uintptr_t)pos) | ((uintptr_t)wpack) | ((uintptr_t)bpack) | ((uintptr_t)out_hm)) & 15) == 0,
                "%s: pointers must be 16-byte aligned", who);
    hipLaunchKernelGGL((msda_enc_lp_kernel<3, true>), dim3((unsigned)((int64_t)B * M * cdiv(S, 64))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)value_hm, spatial_shapes, level_start_index, src, pos, (const u32x4*)wpack, bpack,
                       (const unsigned short*)nullptr, (unsigned short*)out_hm, B, S, M);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_f32_to_f16(const float* in, void* out, int64_t n, void* stream) {
    const char* who = "msm_f32_to_f16";
    MSM_REQUIRE(in && out && n > 0 && n % 8 == 0, "%s: n=%lld must be a positive multiple of 8", who, (long long)n);
    MSM_REQUIRE(((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    const int64_t n8 = n / 8;
    const int grid = (int)(n8 / 256 + 1 > 4096 ? 4096 : n8 / 256 + 1);
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (u32x4*)out, n8);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_f32_to_f16_rows(const float* in, void* out, int B, int64_t n, int64_t in_batch_stride, void* stream) {
    const char* who = "msm_f32_to_f16_rows";
    MSM_REQUIRE(in && out && B > 0 && B <= 65535 && n > 0 && n % 8 == 0 && in_batch_stride >= n && in_batch_stride % 4 == 0,
                "%s: n=%lld must be a positive multiple of 8, the batch stride %lld a multiple of 4 floats >= n", who, (long long)n, (long long)in_batch_stride);
    MSM_REQUIRE(((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    const int64_t n8 = n / 8;
    const int grid = (int)(n8 / 256 + 1 > 1024 ? 1024 : n8 / 256 + 1);
    hipLaunchKernelGGL(f32_to_f16_rows_kernel, dim3(grid, B), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (u32x4*)out, n8, in_batch_stride / 4);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int64_t msm_encoder_prologue_hm_weight_bytes(void) { return PH_WBYTES; }

extern "C" int msm_encoder_prologue_hm_fwd(const float* raw, const double* stats, const float* gn_params, const int32_t* level_starts,
                                           int n_levels, int groups, float gn_eps, const void* wblocks, const float* small, const float* pos,
                                           float* src_out, void* value_out, void* proj_out, int B, int S, void* stream) {
    const char* who = "msm_encoder_prologue_hm_fwd";
    MSM_REQUIRE(raw && stats && gn_params && level_starts && wblocks && small && pos && src_out && value_out && proj_out, "%s: null pointer", who);
    MSM_REQUIRE(n_levels >= 1 && n_levels <= PH_MAXL, "%s: n_levels=%d outside [1, %d]", who, n_levels, PH_MAXL);
    MSM_REQUIRE(B > 0 && S >= 86 && (int64_t)B * S < ((int64_t)1 << 31), "%s: need B > 0 and at least 86 tokens per image (S=%d)", who, S);
    MSM_REQUIRE(groups > 0 && EH_C % groups == 0, "%s: groups=%d must divide 64", who, groups);
    MSM_REQUIRE(((((uintptr_t)raw) | ((uintptr_t)wblocks) | ((uintptr_t)small) | ((uintptr_t)src_out) | ((uintptr_t)value_out) |
                  ((uintptr_t)proj_out) | ((uintptr_t)pos) | ((uintptr_t)gn_params)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "%s: pointers must be 16-byte aligned", who);
    HmLevels lv;
    lv.n = n_levels;
    for (int l = 0; l <= PH_MAXL; ++l) lv.start[l] = level_starts[l < n_levels ? l : n_levels];
    MSM_REQUIRE(lv.start[0] == 0 && lv.start[n_levels] == S, "%s: level_starts must run from 0 to S", who);
    for (int l = 0; l < n_levels; ++l) MSM_REQUIRE(lv.start[l + 1] > lv.start[l], "%s: level_starts must increase", who);
    const int M = B * S;
    const size_t lds = (size_t)PH_WBYTES + sizeof(float) * (size_t)(PH_NIMG * PH_MAXL * 3 * EH_C + EH_C + EH_PROJ);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_prologue_hm_kernel, lds));
    hipLaunchKernelGGL(enc_prologue_hm_kernel, dim3(cdiv(cdiv(M, 16), PH_W)), dim3(PH_W * 64), lds, (hipStream_t)stream, raw, stats, gn_params, lv,
                       groups, gn_eps, reinterpret_cast<const u32x4*>(wblocks), small, pos, src_out, (unsigned short*)value_out,
                       (unsigned short*)proj_out, M, S, B);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
