// Multi-head hypersphere (von Mises-Fisher) attention core (see include/msm_hip.h).
//
// Reference: hypersphere_attention, attention_util.py:64-82 -- q^ = q/|q|, k^ = k/|k| per head,
// A = softmax(kappa * q^ k^T + mask), out = normalize(A v); head split/merge attention_util.py:364-375,
// 424; bool mask -> -inf conversion attention_util.py:411-414; the decoder's all-masked-row reset,
// meanshiftformer_transformer_decoder.py:618.
//
// The reference materialises (B*h, Lq, S) score / float-mask / softmax tensors.  Here nothing of
// size Lq x S touches memory except the 1-byte mask:
//   * logits are bounded (|kappa q^.k^| <= kappa), so softmax needs no running max: p = exp(s - kappa)
//     in [e^-2kappa, 1]; partial sums over key ranges combine by plain addition;
//   * S^T = K^ Q^T is computed with v_mfma_f32_16x16x4_f32 so that the probabilities land in
//     registers already in A-operand layout for P V (lane = (query l&15, key slot l>>4));
//   * the head dimension (32) is walked in the permuted order d = 8*(l>>4) + t so that every lane
//     reads its q/k fragment as two contiguous float4;
//   * each wave keeps all <=112 queries of a chunk (Q^ as 56 VGPRs, O as 56) and streams its share
//     of the 16-key blocks; 4 waves + key splits across workgroups are reduced through LDS and a
//     small combine kernel that also applies 1/l and the output L2 normalisation.
#include <stdlib.h>

#include <type_traits>

#include "bf16.h"
#include "common.h"

namespace msm {

constexpr int AQB = 7;             // 16-query blocks per chunk
constexpr int AQCH = AQB * 16;     // 112
constexpr int HD = 32;             // head dim
constexpr int PSTRIDE = HD + 1;    // partial row: 32 outputs + softmax denominator

static int attn_nsplit(int B, int qchunks, int heads, int S) {
    const int base = B * qchunks * heads;
    const int target = opt(MSM_OPT_ATTN_TARGET) > 0 ? opt(MSM_OPT_ATTN_TARGET) : 512;
    int ns = cdiv(target, base);
    const int maxs = max(1, S / 128);  // >= 2 key blocks per wave
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    return ns;
}

// ---- low-precision mode (BASELINE configs 3 / 5) --------------------------------------------------------------------------
// Template parameters of the two default kernels below: KVT = storage type of K and V (float, or uint16_t = bf16 as written by
// msm_kv_project_multi_bf16), BF = multiply on v_mfma_f32_16x16x16_bf16 (q^, k^, the probabilities and V rounded to bf16 at
// the moment they become operands; fp32 accumulation, fp32 exp / row sums / normalisation).  A key block is then 2 + 2 MFMAs
// of 8 cycles per query block instead of 8 + 8 of 32.  Same lane mapping: the k index a lane feeds is free as long as both
// operands agree, so chunk c (0, 1) of the head dimension is dims lq*8 + 4c .. + 3, exactly the two halves of the 8 values
// a lane already holds.
template <typename KVT>
struct KVRaw;
template <>
struct KVRaw<float> {
    float4 ka, kc;
    float v[4][2];
};
template <>
struct KVRaw<uint16_t> {
    u32x4b k;
    unsigned short v[4][2];
};
// kvh16 (precision "f16"): K stored as IEEE half (raw projection, as written by msm_kv_project_multi_bf16 with half_format = 1), V as
// bf16.  BF = 2: q^ and k^ enter the score MFMA as fp16 (v_mfma_f32_16x16x32_f16) -- with kappa = 30 in front of the cosine the bf16
// form's 2^-9 roundings of q^, k^ and the stored K are 2 % of a softmax weight; fp16 makes that 0.25 %.  P V stays on bf16 MFMAs: a
// softmax weight exp(kappa (cos - 1)) spans e^-60 .. 1, which bf16 carries and fp16 does not.
struct kvh16 {
    uint16_t v;
};
template <>
struct KVRaw<kvh16> : KVRaw<uint16_t> {};
typedef unsigned u32x2a __attribute__((ext_vector_type(2)));
// K / V through buffer descriptors: kr / vr cover K and V of one (image, head) from its first element, ko = byte offset of this
// lane's 8 dims of its key row, vo[r] = byte offset of dim lj of the row of key 4 lq + r, ks / vs = wave-uniform byte offsets
// (the key block) that ride in the scalar offset of the load
__device__ __forceinline__ void kv_fetch(KVRaw<float>& f, __amdgpu_buffer_rsrc_t kr, unsigned ko, unsigned ks, __amdgpu_buffer_rsrc_t vr,
                                         const unsigned (&vo)[4], unsigned vs) {
    const u32x4b a = __builtin_amdgcn_raw_buffer_load_b128(kr, ko, ks, 0), c = __builtin_amdgcn_raw_buffer_load_b128(kr, ko + 16u, ks, 0);
    f.ka = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    f.kc = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f.v[r][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vr, vo[r], vs, 0));
        f.v[r][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vr, vo[r] + 64u, vs, 0));
    }
}
__device__ __forceinline__ void kv_fetch(KVRaw<uint16_t>& f, __amdgpu_buffer_rsrc_t kr, unsigned ko, unsigned ks, __amdgpu_buffer_rsrc_t vr,
                                         const unsigned (&vo)[4], unsigned vs) {
    f.k = __builtin_amdgcn_raw_buffer_load_b128(kr, ko, ks, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f.v[r][0] = __builtin_amdgcn_raw_buffer_load_b16(vr, vo[r], vs, 0);
        f.v[r][1] = __builtin_amdgcn_raw_buffer_load_b16(vr, vo[r] + 32u, vs, 0);
    }
}
__device__ __forceinline__ void kv_fetch(KVRaw<kvh16>& f, __amdgpu_buffer_rsrc_t kr, unsigned ko, unsigned ks, __amdgpu_buffer_rsrc_t vr,
                                         const unsigned (&vo)[4], unsigned vs) {
    kv_fetch(static_cast<KVRaw<uint16_t>&>(f), kr, ko, ks, vr, vo, vs);
}
__device__ __forceinline__ void k_floats(const KVRaw<kvh16>& f, float (&kf)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kf[2 * i] = half_lo(f.k[i]);
        kf[2 * i + 1] = half_hi(f.k[i]);
    }
}
__device__ __forceinline__ void v_operands(const KVRaw<kvh16>& f, bf16x4 (&vb)[2]);
__device__ __forceinline__ float v_float(const KVRaw<kvh16>& f, int r, int hh) { return __uint_as_float((unsigned)f.v[r][hh] << 16); }
__device__ __forceinline__ void k_floats(const KVRaw<float>& f, float (&kf)[8]) {
    kf[0] = f.ka.x; kf[1] = f.ka.y; kf[2] = f.ka.z; kf[3] = f.ka.w;
    kf[4] = f.kc.x; kf[5] = f.kc.y; kf[6] = f.kc.z; kf[7] = f.kc.w;
}
__device__ __forceinline__ void k_floats(const KVRaw<uint16_t>& f, float (&kf)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kf[2 * i] = __uint_as_float(f.k[i] << 16);
        kf[2 * i + 1] = __uint_as_float(f.k[i] & 0xffff0000u);
    }
}
// the V fragments of a key block: fp32 MFMA operands (vf) or two bf16x4 B operands (dims lj and 16 + lj of keys 4 lq .. + 3)
__device__ __forceinline__ void v_operands(const KVRaw<float>& f, bf16x4 (&vb)[2]) {
    vb[0] = pack4(f.v[0][0], f.v[1][0], f.v[2][0], f.v[3][0]);
    vb[1] = pack4(f.v[0][1], f.v[1][1], f.v[2][1], f.v[3][1]);
}
__device__ __forceinline__ void v_operands(const KVRaw<uint16_t>& f, bf16x4 (&vb)[2]) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        vb[hh] = __builtin_bit_cast(bf16x4, u32x2b{(unsigned)f.v[0][hh] | ((unsigned)f.v[1][hh] << 16), (unsigned)f.v[2][hh] | ((unsigned)f.v[3][hh] << 16)});
}
__device__ __forceinline__ void v_operands(const KVRaw<kvh16>& f, bf16x4 (&vb)[2]) { v_operands(static_cast<const KVRaw<uint16_t>&>(f), vb); }
__device__ __forceinline__ float v_float(const KVRaw<float>& f, int r, int hh) { return f.v[r][hh]; }
__device__ __forceinline__ float v_float(const KVRaw<uint16_t>& f, int r, int hh) { return __uint_as_float((unsigned)f.v[r][hh] << 16); }

// 1 / max(sqrt(ss), 1e-12) (F.normalize, AU:70-71): v_rsq_f32 and one Newton step instead of sqrt + IEEE division
__device__ __forceinline__ float rnorm(float ss) {
    const float x = fmaxf(ss, 1e-24f);
    const float y = __builtin_amdgcn_rsqf(x);
    return fmaf(y, fmaf(-0.5f * x * y, y, 0.5f), y);
}

// ---- the key stream shared by the two kernels ------------------------------------------------------------------------------
// NQ: 16-query blocks a wave holds.  MM: 0 no mask; 1 mask read as one 4-byte word per (query, 4 keys) -- needs S % 4 == 0;
// 2 mask read bytewise (any S).
//
// In a wave's loop over 16-key blocks every VALU instruction costs matrix-pipe issue slots (the two waves of a SIMD take
// turns on it), so the per-block overhead is kept off the vector ALU:
//   * addresses: K, V and the mask are read through buffer descriptors (SGPRs) with loop-invariant 32-bit lane offsets; the
//     key block rides in the scalar offset of the load, so a FULL key block (all 16 keys < S) costs no vector arithmetic.
//     Only the ragged last block of a sequence (and every block of the bytewise mask mode) takes the clamped per-lane
//     offsets (keys_fetch_any);
//   * masking: the mask byte enters as the INITIAL VALUE of the score accumulator, s0 = byte * -1e5 (v_cvt_f32_ubyteN +
//     one multiply), so a masked pair leaves the MFMA chain at -1e5 + cos and exp2 returns exactly 0 -- no compare /
//     select per score; keys beyond S are handled the same way (tail block only);
//   * the key norm as v_rsq_f32 + one Newton step.
template <typename KVT, int NQ, int MM>
struct KeyFrag {
    KVRaw<KVT> kv;
    uint32_t mw[MM ? NQ : 1];
};
__device__ __forceinline__ void* uniform_ptr64(const void* p) {
    const uint64_t u = (uint64_t)p;
    return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                   (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
}
template <typename KVT, int NQ, int MM>
struct KeyCursor {
    __amdgpu_buffer_rsrc_t kr, vr, mr;   // K, V of this (image, head); the mask of this image
    unsigned ldk_b, ldv_b;               // row strides in bytes
    int S, lj, lq;
    unsigned ko;                         // lane: byte offset of dims lq*8.. of key lj
    unsigned vo[4];                      // lane: byte offset of dim lj of key 4 lq + r
    unsigned mo[MM ? NQ : 1];            // lane: byte offset of (this lane's clamped query row of block m, key 4 lq)

    // qrow0: first query row of query block 0 of this wave (rows qrow0 + 16 m + lj).  The launcher checks that S * ld * sizeof(KVT)
    // and Lq * S fit the 32-bit offsets.
    __device__ __forceinline__ void init(const KVT* k, const KVT* v, const uint8_t* masked_b, int qrow0, int Lq, int S_, int64_t ldk, int64_t ldv,
                                         int lj_, int lq_) {
        S = S_; lj = lj_; lq = lq_;
        ldk_b = (unsigned)ldk * (unsigned)sizeof(KVT);
        ldv_b = (unsigned)ldv * (unsigned)sizeof(KVT);
        kr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(k), 0, (unsigned)(S - 1) * ldk_b + HD * (unsigned)sizeof(KVT), 0x00020000);
        vr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(v), 0, (unsigned)(S - 1) * ldv_b + HD * (unsigned)sizeof(KVT), 0x00020000);
        ko = (unsigned)lj * ldk_b + (unsigned)lq * 8u * (unsigned)sizeof(KVT);
#pragma unroll
        for (int r = 0; r < 4; ++r) vo[r] = (unsigned)(lq * 4 + r) * ldv_b + (unsigned)lj * (unsigned)sizeof(KVT);
        if constexpr (MM != 0) {
            mr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(masked_b), 0, (unsigned)Lq * (unsigned)S, 0x00020000);
#pragma unroll
            for (int m_ = 0; m_ < NQ; ++m_) mo[m_] = (unsigned)min(qrow0 + m_ * 16 + lj, Lq - 1) * (unsigned)S + (unsigned)lq * 4u;
        }
    }
};

// a full key block (16 kb + 15 < S); kb is wave-uniform: no vector arithmetic
template <typename KVT, int NQ, int MM>
__device__ __forceinline__ void keys_fetch_full(const KeyCursor<KVT, NQ, MM>& c, int kb, KeyFrag<KVT, NQ, MM>& f) {
    kv_fetch(f.kv, c.kr, c.ko, (unsigned)kb * 16u * c.ldk_b, c.vr, c.vo, (unsigned)kb * 16u * c.ldv_b);
    if constexpr (MM != 0) {
#pragma unroll
        for (int m_ = 0; m_ < NQ; ++m_) f.mw[m_] = __builtin_amdgcn_raw_buffer_load_b32(c.mr, c.mo[m_], (unsigned)kb * 16u, 0);
    }
}
// any key block: offsets clamped into the sequence (the out-of-range keys are switched off in keys_consume<true, ..>)
template <typename KVT, int NQ, int MM>
__device__ __forceinline__ void keys_fetch_any(const KeyCursor<KVT, NQ, MM>& c, int kb, KeyFrag<KVT, NQ, MM>& f) {
    const int key_c0 = kb * 16 + c.lq * 4;
    const unsigned ko = (unsigned)min(kb * 16 + c.lj, c.S - 1) * c.ldk_b + (unsigned)c.lq * 8u * (unsigned)sizeof(KVT);
    unsigned vo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vo[r] = (unsigned)min(key_c0 + r, c.S - 1) * c.ldv_b + (unsigned)c.lj * (unsigned)sizeof(KVT);
    kv_fetch(f.kv, c.kr, ko, 0u, c.vr, vo, 0u);
    if constexpr (MM != 0) {
#pragma unroll
        for (int m_ = 0; m_ < NQ; ++m_) {
            const unsigned row = c.mo[m_] - (unsigned)c.lq * 4u;
            uint32_t w = 0;
            if constexpr (MM == 1) {
                w = __builtin_amdgcn_raw_buffer_load_b32(c.mr, row + (unsigned)min(key_c0, c.S - 4), 0u, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) w |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(c.mr, row + (unsigned)min(key_c0 + r, c.S - 1), 0u, 0) << (8 * r);
            }
            f.mw[m_] = w;
        }
    }
}

constexpr float MASK_BIAS = -1.0e5f;      // initial score of a masked pair: exp2(k2 * (-1e5 + cos) - k2) = 0

typedef float f32x2l __attribute__((ext_vector_type(2)));

// One key block against the NQ query blocks of a wave: S^T = K^ Q^^T, p = exp(kappa s - kappa), O += P V, l += row sums.
// exp(kappa*s - kappa) is one fma + v_exp_f32 on the finished cosine: the argument is in [-2*kappa, 0], so the absolute error of
// the fp32 argument (< 6e-6 at kappa = 30) bounds the relative error of p at ~4e-6.  (Carrying kappa log2(e) in the K fragment
// and -kappa log2(e) in the accumulator's initial value saves that fma -- and loses two bits: the chain then rounds at the
// magnitude of kappa log2(e) = 43 instead of 1; measured 3x the error against float64, not kept.)
// TAIL: keys >= S of this block are switched off.
template <bool TAIL, typename KVT, int BF, int NQ, int MM>
__device__ __forceinline__ void keys_consume(const KeyFrag<KVT, NQ, MM>& f, int kb, int S, int lq, float k2, const float (&qf)[NQ][8],
                                             const bf16x4 (&qh)[BF ? NQ : 1][2], const bool (&use_mask)[NQ], f32x4 (&o)[NQ][2],
                                             f32x2l (&lacc)[NQ]) {
    float kr[8];
    k_floats(f.kv, kr);
    float ss = (kr[0] * kr[0] + kr[1] * kr[1] + kr[2] * kr[2] + kr[3] * kr[3]) + (kr[4] * kr[4] + kr[5] * kr[5] + kr[6] * kr[6] + kr[7] * kr[7]);
    ss = sum_lane_rows(ss);
    const float rn = rnorm(ss);
    const float kf[8] = {kr[0] * rn, kr[1] * rn, kr[2] * rn, kr[3] * rn, kr[4] * rn, kr[5] * rn, kr[6] * rn, kr[7] * rn};
    bf16x4 kh[2], vb[2];
    if constexpr (BF) {
        if constexpr (BF == 2) {                   // fp16 score operands (unit vectors: no clamp needed)
            kh[0] = __builtin_bit_cast(bf16x4, pack4h_nc(kf[0], kf[1], kf[2], kf[3]));
            kh[1] = __builtin_bit_cast(bf16x4, pack4h_nc(kf[4], kf[5], kf[6], kf[7]));
        } else {
            kh[0] = pack4(kf[0], kf[1], kf[2], kf[3]);
            kh[1] = pack4(kf[4], kf[5], kf[6], kf[7]);
        }
        v_operands(f.kv, vb);
    }
    uint32_t oob = 0;
    if constexpr (TAIL) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (kb * 16 + lq * 4 + r >= S) oob |= 0xffu << (8 * r);
    }
    auto scores = [&](int m) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MM != 0 || TAIL) {
            uint32_t mw = 0;
            if constexpr (MM != 0) mw = use_mask[m] ? f.mw[m] : 0u;
            if constexpr (TAIL) mw |= oob;
            s = f32x4{(float)(mw & 0xffu) * MASK_BIAS, (float)((mw >> 8) & 0xffu) * MASK_BIAS, (float)((mw >> 16) & 0xffu) * MASK_BIAS,
                      (float)(mw >> 24) * MASK_BIAS};
        }
        if constexpr (BF) {
            // one K = 32 MFMA over the head's 32 dims (a lane's eight dims 8 lq .. + 7 on both operands) instead of two dependent
            // K = 16 ones, which also issue at half the rate
            if constexpr (BF == 2) s = mfma_f16k32(__builtin_bit_cast(f16x8, cat8(kh[0], kh[1])), __builtin_bit_cast(f16x8, cat8(qh[m][0], qh[m][1])), s);
            else s = mfma_bf16k32(cat8(kh[0], kh[1]), cat8(qh[m][0], qh[m][1]), s);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) s = mfma16(kf[t], qf[m][t], s);
        }
        return s;                            // s[r]: key 16 kb + 4 lq + r, query lj of block m
    };
    // the score MFMAs of block m + 1 are issued before the exponentials of block m: their 8 x 32 cycles cover the result
    // latency of the chain and the vector work in between
    f32x4 s = scores(0);
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        f32x4 sn = s;
        if (m + 1 < NQ) sn = scores(m + 1);
        // the four exponents as two v_pk_fma_f32 (same values as fmaf per score; the loop is vector-issue bound)
        const f32x2l k2v = f32x2l{k2, k2};
        const f32x2l e01 = __builtin_elementwise_fma(f32x2l{s[0], s[1]}, k2v, -k2v), e23 = __builtin_elementwise_fma(f32x2l{s[2], s[3]}, k2v, -k2v);
        const float p[4] = {__builtin_amdgcn_exp2f(e01[0]), __builtin_amdgcn_exp2f(e01[1]), __builtin_amdgcn_exp2f(e23[0]), __builtin_amdgcn_exp2f(e23[1])};
        lacc[m] += f32x2l{p[0], p[1]} + f32x2l{p[2], p[3]};
        if constexpr (BF) {
            const bf16x4 pp = pack4(p[0], p[1], p[2], p[3]);
            o[m][0] = mfma_bf16(pp, vb[0], o[m][0]);
            o[m][1] = mfma_bf16(pp, vb[1], o[m][1]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[m][0] = mfma16(p[r], v_float(f.kv, r, 0), o[m][0]);
                o[m][1] = mfma16(p[r], v_float(f.kv, r, 1), o[m][1]);
            }
        }
        s = sn;
    }
}

// The key blocks first, first + stride, ... < end of one wave: full blocks two per trip through a ping-pong pair of fragments
// (block i + 2 strides is requested before the MFMAs of block i, pinned with sched_barrier), then the ragged last block of the
// sequence, if it is this wave's, through the clamped path.
template <typename KVT, int BF, int NQ, int MM>
__device__ __forceinline__ void keys_stream(const KeyCursor<KVT, NQ, MM>& cur, int first, int stride, int end, float k2, const float (&qf)[NQ][8],
                                            const bf16x4 (&qh)[BF ? NQ : 1][2], const bool (&use_mask)[NQ], f32x4 (&o)[NQ][2],
                                            float (&lsum)[NQ]) {
    const int S = cur.S;
    const int n_full = S / 16;
    int kb = first;
    f32x2l lacc[NQ];                          // row sums as two partial sums per query block (v_pk_add_f32)
#pragma unroll
    for (int m = 0; m < NQ; ++m) lacc[m] = f32x2l{0.f, 0.f};
    if constexpr (MM != 2) {
        const int full_end = min(end, n_full);
        KeyFrag<KVT, NQ, MM> fa, fb;
        if (kb < full_end) keys_fetch_full(cur, kb, fa);
        for (; kb + stride < full_end; kb += 2 * stride) {
            keys_fetch_full(cur, kb + stride, fb);
            __builtin_amdgcn_sched_barrier(0);
            keys_consume<false, KVT, BF, NQ, MM>(fa, kb, S, cur.lq, k2, qf, qh, use_mask, o, lacc);
            __builtin_amdgcn_sched_barrier(0);
            keys_fetch_full(cur, min(kb + 2 * stride, n_full - 1), fa);
            __builtin_amdgcn_sched_barrier(0);
            keys_consume<false, KVT, BF, NQ, MM>(fb, kb + stride, S, cur.lq, k2, qf, qh, use_mask, o, lacc);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb < full_end) {
            keys_consume<false, KVT, BF, NQ, MM>(fa, kb, S, cur.lq, k2, qf, qh, use_mask, o, lacc);
            kb += stride;
        }
    }
    for (; kb < end; kb += stride) {
        KeyFrag<KVT, NQ, MM> f;
        keys_fetch_any(cur, kb, f);
        keys_consume<true, KVT, BF, NQ, MM>(f, kb, S, cur.lq, k2, qf, qh, use_mask, o, lacc);
    }
#pragma unroll
    for (int m = 0; m < NQ; ++m) lsum[m] += lacc[m][0] + lacc[m][1];
}

// Q^ fragments (B operand) of NQ query blocks starting at row qrow0: lane (query lj of block m, dims lq*8 + t); the per-row
// mask enable: rows whose keys are all masked attend everywhere (DEC:618)
template <int BF, int NQ>
__device__ __forceinline__ void load_queries(const float* __restrict__ qb, int64_t ldq, int qrow0, int Lq, int lj, bool masked,
                                             const int32_t* __restrict__ row_any_b, float (&qf)[NQ][8], bf16x4 (&qh)[BF ? NQ : 1][2],
                                             bool (&use_mask)[NQ]) {
#pragma unroll
    for (int m = 0; m < NQ; ++m) {
        const int qi = qrow0 + m * 16 + lj;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        if (qi < Lq) {
            const float* p = qb + (int64_t)qi * ldq;
            a = *reinterpret_cast<const float4*>(p);
            c = *reinterpret_cast<const float4*>(p + 4);
        }
        float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        ss = sum_lane_rows(ss);
        const float rn = rnorm(ss);
        qf[m][0] = a.x * rn; qf[m][1] = a.y * rn; qf[m][2] = a.z * rn; qf[m][3] = a.w * rn;
        qf[m][4] = c.x * rn; qf[m][5] = c.y * rn; qf[m][6] = c.z * rn; qf[m][7] = c.w * rn;
        if constexpr (BF == 2) {
            qh[m][0] = __builtin_bit_cast(bf16x4, pack4h_nc(qf[m][0], qf[m][1], qf[m][2], qf[m][3]));
            qh[m][1] = __builtin_bit_cast(bf16x4, pack4h_nc(qf[m][4], qf[m][5], qf[m][6], qf[m][7]));
        } else if constexpr (BF == 1) {
            qh[m][0] = pack4(qf[m][0], qf[m][1], qf[m][2], qf[m][3]);
            qh[m][1] = pack4(qf[m][4], qf[m][5], qf[m][6], qf[m][7]);
        }
        use_mask[m] = masked && qi < Lq && (row_any_b == nullptr || row_any_b[qi] != 0);
    }
}

template <typename KVT, int BF, int MM>
__global__ __launch_bounds__(256, 2) void hs_attn_kernel(const float* __restrict__ q, const KVT* __restrict__ k,
                                                      const KVT* __restrict__ v, const uint8_t* __restrict__ masked,
                                                      const int32_t* __restrict__ row_any, float* __restrict__ part,
                                                      float* __restrict__ out, int Lq, int S, int heads, int qchunks, int nsplit, int64_t ldq,
                                                      int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv,
                                                      int64_t v_sb, float kappa) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][AQCH][PSTRIDE]
    const int split = blockIdx.x, h = blockIdx.y;
    const int b = blockIdx.z / qchunks, qc = blockIdx.z - b * qchunks;
    const int q0 = qc * AQCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: scalar key-block loop
    const int lj = lane & 15, lq = lane >> 4;

    float qf[AQB][8];
    bf16x4 qh[BF ? AQB : 1][2];             // BF: the same fragments as bf16 B operands (dims lq*8 + 0..3 | + 4..7)
    bool use_mask[AQB];
    load_queries<BF, AQB>(q + (int64_t)b * q_sb + h * HD + lq * 8, ldq, q0, Lq, lj, MM != 0, row_any ? row_any + (int64_t)b * Lq : nullptr, qf, qh,
                          use_mask);

    f32x4 o[AQB][2];
    float lsum[AQB];
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        o[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        o[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        lsum[m] = 0.f;
    }

    // key blocks of 16: contiguous range per split, round-robin over the 4 waves
    const int nkb = (S + 15) / 16;
    const int kb_per = (nkb + nsplit - 1) / nsplit;
    const int kb_beg = split * kb_per, kb_end = min(nkb, kb_beg + kb_per);
    KeyCursor<KVT, AQB, MM> cur;
    cur.init(k + (int64_t)b * k_sb + h * HD, v + (int64_t)b * v_sb + h * HD, MM != 0 ? masked + (int64_t)b * Lq * S : nullptr, q0, Lq, S, ldk, ldv, lj,
             lq);
    keys_stream<KVT, BF, AQB, MM>(cur, kb_beg + wave, 4, kb_end, kappa * 1.4426950408889634f /* kappa * log2(e) */, qf, qh, use_mask, o, lsum);

    // ---- reduce the 4 waves through LDS, then one partial per workgroup ----
    float* mine = red + wave * (AQCH * PSTRIDE);
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        float l = lsum[m];
        l = sum_lane_rows(l);
        if (lq == 0) mine[(m * 16 + lj) * PSTRIDE + HD] = l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * 16 + lq * 4 + r;
            mine[row * PSTRIDE + lj] = o[m][0][r];
            mine[row * PSTRIDE + 16 + lj] = o[m][1][r];
        }
    }
    __syncthreads();
    if (nsplit == 1) {
        // single key range: finish here (1/l, output L2 normalisation, head merge) -- no partials, no 2nd launch
        const int ql = tid;
        const int qi = q0 + ql;
        if (ql < AQCH && qi < Lq) {
            float acc[HD + 1];
#pragma unroll
            for (int d = 0; d <= HD; ++d)
                acc[d] = (red[ql * PSTRIDE + d] + red[AQCH * PSTRIDE + ql * PSTRIDE + d]) +
                         (red[2 * AQCH * PSTRIDE + ql * PSTRIDE + d] + red[3 * AQCH * PSTRIDE + ql * PSTRIDE + d]);
            const float l = acc[HD];
            float ss = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                acc[d] = acc[d] / l;
                ss += acc[d] * acc[d];
            }
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            float* o_ = out + ((int64_t)b * Lq + qi) * (heads * HD) + h * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) o_[d] = acc[d] / nrm;
        }
        return;
    }
    float* dst = part + ((((int64_t)blockIdx.z * heads + h) * nsplit) + split) * (AQCH * PSTRIDE);
    for (int i = tid; i < AQCH * PSTRIDE; i += 256) {
        dst[i] = (red[i] + red[AQCH * PSTRIDE + i]) + (red[2 * AQCH * PSTRIDE + i] + red[3 * AQCH * PSTRIDE + i]);
    }
}

// ---- long key sequences in the 16-bit plans: the K/V projection INSIDE the attention kernel -------------------------------------
// Reference: the cross-attention of PretrainedMeanShiftTransformerDecoder / the finest level of MeanShiftTransformerDecoder
// (meanshiftformer_transformer_decoder.py:697-1048, :245-260) with ms_in_projection_packed's key / value linears (attention_util.py:134-140)
// and the head split (:364-375).  K and V are affine in the 64-channel level feature x (modeling._folded_kv):
//     [K | V](key) = x(key) W^T + row[y(key)] + col[x(key)]            (separable position constants, msm_kv_project_f32)
// Unfused, the 16-bit plans write them as 1024 bytes per key (2 x 256 bf16) and the attention kernel reads them back -- 3.8 GB per
// decoder pass on the 307 200-key UCN path against the 128 bytes per key of the fp16 feature they are a linear image of, and the
// projection launch (1.8 ms of a 5.9-ms step) does nothing else.  Here a workgroup = (key range, head, query chunk) as hs_attn_kernel;
// per 16-key block a wave
//   * loads its keys' 64 fp16 channels (two 16-byte loads per lane: channels 32 s + 8 lq .. + 7 of key lj -- the B operand of the K
//     products AND the A operand of the V products as they are),
//   * K_h^T [32 dims x 16 keys] = Wk_h x^T and V_h [16 keys x 32 dims] = x Wv_h^T: 4 + 4 v_mfma_f32_16x16x32_f16 with the head's
//     weight fragments resident in 32 VGPRs and the position constants as the accumulators' initial values,
//   * the K accumulators ARE the score MFMA's A operand up to a permutation of the head dimension (lane (key lj, lq) ends with dims
//     4 lq + r and 16 + 4 lq + r: k index 8 lq + j <-> dim 16 (j >> 2) + 4 lq + (j & 3); the query fragments are loaded in the same order),
//     the V accumulators (lane (dim lj, lq): keys 4 lq + r) ARE the B operand of P V: normalise, round, multiply -- no LDS, no transposition,
//   * then the score / exp / P V chain of keys_consume.
// The projection's operands are IEEE halves in both 16-bit plans (x is a unit-norm embedding or a LayerNorm output; one rounding of x,
// one of W); K^ and the probabilities / V enter the score and P V MFMAs in the plan's own formats (BF = 1: bf16, BF = 2: fp16 scores).
// Needs: separable constants, W % 16 == 0 (a key block lies in one image row and starts at a multiple of 16), S % 16 == 0.
#ifndef FK_EXP
#define FK_EXP 0   // tuning builds only (tools/probes/fkv_parts.sh): 1 no constant loads, 2 no mask loads, 3 no x loads, 4 no exponentials, 5 no P V MFMAs
#endif
// Round 5, what bounds it (tools/probes/fkv_parts.sh at 2 x 307 200 keys: 455 us as first written; without its mask loads 272, without its
// constant loads 315, without x 389, without exponentials or P V MFMAs 453 / 451): the memory INSTRUCTIONS -- seven mask words per
// block from a [query][key] byte mask (16 cache lines per instruction, 16 bytes used of each) and eight constant loads.  So:
//   * the mask arrives bit-packed and blocked (msm_attn_pack_mask_bits): per 16-key block 256 bytes = [query lj][query block m]
//     16-bit words (bit k = key 16 kb + k), ONE 16-byte load per lane and block, an eighth of the bytes;
//   * a wave walks DOWN a 16-key column strip (units in column-major order u = strip * H + y): the column constants are loop
//     invariant (16 VGPRs, reloaded when the strip changes), only the row constants (two 64-byte broadcast loads + two dwords) move.
struct FkvFrag {
    u32x4b x[2];            // this lane's 8 + 8 channels (k-steps 0, 1) of key lj
    float4 rk[2];           // K: row[y][16 t + 4 lq ..]
    float rv[2];            // V: row[y][heads * 32 + 16 t + lj]
    u32x4b mw;              // mask words of query lj of the eight query blocks (16 bits per block)
};

template <int BF, int MM>
__global__ __launch_bounds__(256, 2) void hs_attn_fkv_kernel(const float* __restrict__ q, const unsigned short* __restrict__ xh,
                                                           const u32x4b* __restrict__ wfrag, const float* __restrict__ rc,
                                                           const float* __restrict__ cvT, const u32x4b* __restrict__ mask_bits,
                                                           const int32_t* __restrict__ row_any, float* __restrict__ part,
                                                           float* __restrict__ out, int Lq, int S, int Wimg, int heads, int qchunks, int nsplit,
                                                           int64_t ldq, int64_t q_sb, float kappa) {
    static_assert(BF == 1 || BF == 2, "the fused kernel exists for the 16-bit plans");
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][AQCH][PSTRIDE]
    const int split = blockIdx.x, h = blockIdx.y;
    const int b = blockIdx.z / qchunks, qc = blockIdx.z - b * qchunks;
    const int q0 = qc * AQCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int Himg = S / Wimg, strips = Wimg / 16;
    const int N = 2 * heads * HD;                                // columns of the constants: [K | V]

    // Q^ fragments in the head-dimension order of the K accumulators: k index 8 lq + j <-> dim 16 (j >> 2) + 4 lq + (j & 3)
    bf16x4 qh[AQB][2];
    bool use_mask[AQB];
    {
        const float* qb = q + (int64_t)b * q_sb + h * HD + lq * 4;
        const int32_t* ra = row_any ? row_any + (int64_t)b * Lq : nullptr;
#pragma unroll
        for (int m = 0; m < AQB; ++m) {
            const int qi = q0 + m * 16 + lj;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (qi < Lq) {
                const float* p = qb + (int64_t)qi * ldq;
                a = *reinterpret_cast<const float4*>(p);
                c = *reinterpret_cast<const float4*>(p + 16);
            }
            float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
            ss = sum_lane_rows(ss);
            const float rn = rnorm(ss);
            if constexpr (BF == 2) {
                qh[m][0] = __builtin_bit_cast(bf16x4, pack4h_nc(a.x * rn, a.y * rn, a.z * rn, a.w * rn));
                qh[m][1] = __builtin_bit_cast(bf16x4, pack4h_nc(c.x * rn, c.y * rn, c.z * rn, c.w * rn));
            } else {
                qh[m][0] = pack4(a.x * rn, a.y * rn, a.z * rn, a.w * rn);
                qh[m][1] = pack4(c.x * rn, c.y * rn, c.z * rn, c.w * rn);
            }
            use_mask[m] = MM != 0 && qi < Lq && (ra == nullptr || ra[qi] != 0);
        }
    }
    // the head's weight fragments: [kv][tile][k-step] x 16 bytes per lane (msm_attn_pack_kv_weights)
    f16x8 wk[2][2], wv[2][2];
    {
        const u32x4b* wp = wfrag + (int64_t)h * 8 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                wk[t][st] = __builtin_bit_cast(f16x8, wp[((0 * 2 + t) * 2 + st) * 64]);
                wv[t][st] = __builtin_bit_cast(f16x8, wp[((1 * 2 + t) * 2 + st) * 64]);
            }
    }
    f32x4 o[AQB][2];
    f32x2l lacc[AQB];
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        o[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        o[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        lacc[m] = f32x2l{0.f, 0.f};
    }
    // units (16-key blocks) in column-major order, a contiguous range per workgroup, a contiguous quarter of it per wave
    const int nkb = S / 16;
    const int u_per = (nkb + nsplit - 1) / nsplit;
    const int u_beg = split * u_per, u_end = min(nkb, u_beg + u_per);
    const int w_per = (max(u_end - u_beg, 0) + 3) / 4;
    const int u0 = u_beg + wave * w_per, u1 = min(u_end, u0 + w_per);
    const float k2 = kappa * 1.4426950408889634f;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(xh + (int64_t)b * S * 64), 0, (unsigned)S * 128u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(rc), 0, (unsigned)(Himg + Wimg) * (unsigned)N * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t cvr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(cvT), 0, (unsigned)(heads * HD) * (unsigned)Wimg * 4u, 0x00020000);
    __amdgpu_buffer_rsrc_t mr = xr;
    if constexpr (MM != 0)
        mr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr64(mask_bits + ((int64_t)b * qchunks + qc) * nkb * 16), 0, (unsigned)nkb * 256u, 0x00020000);
    const unsigned xo = (unsigned)lj * 128u + (unsigned)lq * 16u;                                  // channels 8 lq .. + 7 of key lj (k-step 1: + 64 bytes)
    const unsigned cko = ((unsigned)(Himg + lj) * (unsigned)N + (unsigned)(h * HD + lq * 4)) * 4u;   // col[lj][K dims 4 lq ..] (+ x0 rows, + 16 t)
    const unsigned rko = (unsigned)(h * HD + lq * 4) * 4u;                                           // row[0][K dims 4 lq ..]
    const unsigned cvo = ((unsigned)(h * HD + lj) * (unsigned)Wimg + (unsigned)lq * 4u) * 4u;        // colT[dim lj][4 lq ..]
    const unsigned rvo = (unsigned)(heads * HD + h * HD + lj) * 4u;                                  // row[0][V dim lj]

    float4 ck[2], cv[2];                                                                             // this strip's column constants
    auto load_cols = [&](int strip) {
        const unsigned x0 = (unsigned)strip * 16u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#if FK_EXP == 1
            ck[t] = cv[t] = make_float4((float)strip, 1.f, 2.f, (float)t);
#else
            const u32x4b a = __builtin_amdgcn_raw_buffer_load_b128(rcr, cko + 64u * t, x0 * (unsigned)N * 4u, 0);
            const u32x4b v_ = __builtin_amdgcn_raw_buffer_load_b128(cvr, cvo + (unsigned)(16 * t) * (unsigned)Wimg * 4u, x0 * 4u, 0);
            ck[t] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
            cv[t] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w));
#endif
        }
    };
    auto fetch = [&](int strip, int y, FkvFrag& f) {
        const int kb = y * strips + strip;                                                           // uniform
        const unsigned ks = (unsigned)kb * 16u * 128u;
#if FK_EXP == 3
        f.x[0] = f.x[1] = u32x4b{(unsigned)kb, 0x3c003c00u, 0x38003800u, (unsigned)lane};
#else
        f.x[0] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, ks, 0);
        f.x[1] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo + 64u, ks, 0);
#endif
        const unsigned rks = (unsigned)y * (unsigned)N * 4u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#if FK_EXP == 1
            f.rk[t] = make_float4((float)kb, 1.f, 2.f, (float)y);
            f.rv[t] = (float)y;
#else
            const u32x4b r_ = __builtin_amdgcn_raw_buffer_load_b128(rcr, rko + 64u * t, rks, 0);
            f.rk[t] = make_float4(__uint_as_float(r_.x), __uint_as_float(r_.y), __uint_as_float(r_.z), __uint_as_float(r_.w));
            f.rv[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcr, rvo + 64u * t, rks, 0));
#endif
        }
        if constexpr (MM != 0) {
#if FK_EXP == 2
            f.mw = u32x4b{(unsigned)kb & 0x01010101u, 0u, (unsigned)y, 0u};
#else
            f.mw = __builtin_amdgcn_raw_buffer_load_b128(mr, (unsigned)lj * 16u, (unsigned)kb * 256u, 0);
#endif
        }
    };
    auto consume = [&](const FkvFrag& f) {
        const f16x8 x0_ = __builtin_bit_cast(f16x8, f.x[0]), x1_ = __builtin_bit_cast(f16x8, f.x[1]);
        // K_h^T (dims x keys) and V_h (keys x dims), constants as the initial values
        f32x4 kt[2], vt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kt[t] = f32x4{ck[t].x + f.rk[t].x, ck[t].y + f.rk[t].y, ck[t].z + f.rk[t].z, ck[t].w + f.rk[t].w};
            vt[t] = f32x4{cv[t].x + f.rv[t], cv[t].y + f.rv[t], cv[t].z + f.rv[t], cv[t].w + f.rv[t]};
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kt[t] = mfma_f16k32(wk[t][0], x0_, kt[t]);
            vt[t] = mfma_f16k32(x0_, wv[t][0], vt[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kt[t] = mfma_f16k32(wk[t][1], x1_, kt[t]);
            vt[t] = mfma_f16k32(x1_, wv[t][1], vt[t]);
        }
        // k^ = K / max(|K|, 1e-12) over the head's 32 dims: this lane's eight + the three other lane quarters of key lj
        float ss = (kt[0][0] * kt[0][0] + kt[0][1] * kt[0][1] + kt[0][2] * kt[0][2] + kt[0][3] * kt[0][3]) +
                   (kt[1][0] * kt[1][0] + kt[1][1] * kt[1][1] + kt[1][2] * kt[1][2] + kt[1][3] * kt[1][3]);
        ss = sum_lane_rows(ss);
        const float rn = rnorm(ss);
        bf16x4 kh[2], vb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (BF == 2) kh[t] = __builtin_bit_cast(bf16x4, pack4h_nc(kt[t][0] * rn, kt[t][1] * rn, kt[t][2] * rn, kt[t][3] * rn));
            else kh[t] = pack4(kt[t][0] * rn, kt[t][1] * rn, kt[t][2] * rn, kt[t][3] * rn);
            vb[t] = pack4(vt[t][0], vt[t][1], vt[t][2], vt[t][3]);
        }
        auto scores = [&](int m) {
            f32x4 s_ = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MM != 0) {
                // this lane's four mask bits of query block m: bit r of the nibble -> all ones -> the bits of MASK_BIAS
                const unsigned nib = use_mask[m] ? (f.mw[m >> 1] >> (16 * (m & 1) + 4 * lq)) : 0u;
                constexpr unsigned BB = 0xc7c35000u;              // -1.0e5f
#pragma unroll
                for (int r = 0; r < 4; ++r) s_[r] = __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)nib, r, 1) & BB);
            }
            if constexpr (BF == 2) return mfma_f16k32(__builtin_bit_cast(f16x8, cat8(kh[0], kh[1])), __builtin_bit_cast(f16x8, cat8(qh[m][0], qh[m][1])), s_);
            else return mfma_bf16k32(cat8(kh[0], kh[1]), cat8(qh[m][0], qh[m][1]), s_);
        };
        f32x4 s_ = scores(0);
#pragma unroll
        for (int m = 0; m < AQB; ++m) {
            f32x4 sn = s_;
            if (m + 1 < AQB) sn = scores(m + 1);
            const f32x2l k2v = f32x2l{k2, k2};
            const f32x2l e01 = __builtin_elementwise_fma(f32x2l{s_[0], s_[1]}, k2v, -k2v), e23 = __builtin_elementwise_fma(f32x2l{s_[2], s_[3]}, k2v, -k2v);
#if FK_EXP == 4
            const float p[4] = {e01[0], e01[1], e23[0], e23[1]};
#else
            const float p[4] = {__builtin_amdgcn_exp2f(e01[0]), __builtin_amdgcn_exp2f(e01[1]), __builtin_amdgcn_exp2f(e23[0]), __builtin_amdgcn_exp2f(e23[1])};
#endif
            lacc[m] += f32x2l{p[0], p[1]} + f32x2l{p[2], p[3]};
            const bf16x4 pp = pack4(p[0], p[1], p[2], p[3]);
#if FK_EXP == 5
            o[m][0][0] += __uint_as_float(__builtin_bit_cast(u32x2b, pp).x ^ __builtin_bit_cast(u32x2b, vb[0]).x);
            o[m][1][0] += __uint_as_float(__builtin_bit_cast(u32x2b, pp).y ^ __builtin_bit_cast(u32x2b, vb[1]).y);
#else
            o[m][0] = mfma_bf16(pp, vb[0], o[m][0]);
            o[m][1] = mfma_bf16(pp, vb[1], o[m][1]);
#endif
            s_ = sn;
        }
    };
    // a wave's range crosses a strip boundary once at most (a strip is H units): the column constants are loaded per strip segment, OUTSIDE
    // the pipelined loop (reloading them under a uniform branch inside it costs 32 more registers than the kernel has)
    for (int u = u0; u < u1;) {
        const int strip = u / Himg, y0 = u - strip * Himg;         // uniform
        const int y1 = min(Himg, y0 + (u1 - u));
        load_cols(strip);
        FkvFrag fa, fb;
        fetch(strip, y0, fa);
        int y = y0;
        for (; y + 1 < y1; y += 2) {
            fetch(strip, y + 1, fb);
            __builtin_amdgcn_sched_barrier(0);
            consume(fa);
            __builtin_amdgcn_sched_barrier(0);
            fetch(strip, min(y + 2, Himg - 1), fa);                 // (one row past the segment at its end: loaded, never used)
            __builtin_amdgcn_sched_barrier(0);
            consume(fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (y < y1) consume(fa);
        u += y1 - y0;
    }

    // ---- reduce the 4 waves through LDS, then one partial per workgroup (as hs_attn_kernel) ----
    float* mine = red + wave * (AQCH * PSTRIDE);
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        float l = lacc[m][0] + lacc[m][1];
        l = sum_lane_rows(l);
        if (lq == 0) mine[(m * 16 + lj) * PSTRIDE + HD] = l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * 16 + lq * 4 + r;
            mine[row * PSTRIDE + lj] = o[m][0][r];
            mine[row * PSTRIDE + 16 + lj] = o[m][1][r];
        }
    }
    __syncthreads();
    if (nsplit == 1) {
        const int ql = tid;
        const int qi = q0 + ql;
        if (ql < AQCH && qi < Lq) {
            float acc[HD + 1];
#pragma unroll
            for (int d = 0; d <= HD; ++d)
                acc[d] = (red[ql * PSTRIDE + d] + red[AQCH * PSTRIDE + ql * PSTRIDE + d]) +
                         (red[2 * AQCH * PSTRIDE + ql * PSTRIDE + d] + red[3 * AQCH * PSTRIDE + ql * PSTRIDE + d]);
            const float l = acc[HD];
            float ss = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                acc[d] = acc[d] / l;
                ss += acc[d] * acc[d];
            }
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            float* o_ = out + ((int64_t)b * Lq + qi) * (heads * HD) + h * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) o_[d] = acc[d] / nrm;
        }
        return;
    }
    float* dst = part + ((((int64_t)blockIdx.z * heads + h) * nsplit) + split) * (AQCH * PSTRIDE);
    for (int i = tid; i < AQCH * PSTRIDE; i += 256) dst[i] = (red[i] + red[AQCH * PSTRIDE + i]) + (red[2 * AQCH * PSTRIDE + i] + red[3 * AQCH * PSTRIDE + i]);
}

// The attention mask of hs_attn_fkv_kernel: bytes [B][Lq][S] (nonzero = masked) -> bit-packed and blocked [B][qchunks][S / 16][16 lj][8 m] uint16:
// word (lj, m) of key block kb holds bit k = masked[q = 112 qc + 16 m + lj][16 kb + k] (m = 7 and queries >= Lq: zero).  A thread packs
// the eight words of one (key block, lj): 16-byte loads from seven query rows, one 16-byte store; a wave covers four consecutive key blocks.
__global__ __launch_bounds__(256) void attn_pack_mask_bits_kernel(const uint8_t* __restrict__ masked, u32x4b* __restrict__ out, int Lq, int S,
                                                                  int qchunks) {
    const int nkb = S / 16;
    const int64_t total = (int64_t)gridDim.y * nkb * 16;                // gridDim.y = B * qchunks
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int bq = blockIdx.y;
    if (idx >= (int64_t)nkb * 16) return;
    (void)total;
    const int kb = (int)(idx >> 4), lj = (int)(idx & 15);
    const int b = bq / qchunks, qc = bq - b * qchunks;
    unsigned wds[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int qi = qc * AQCH + m * 16 + lj;
        unsigned bits = 0;
        if (m < AQB && qi < Lq) {
            const u32x4b v = *reinterpret_cast<const u32x4b*>(masked + ((int64_t)b * Lq + qi) * S + (int64_t)kb * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned w = v[j];
                w = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;       // 1 per nonzero byte
                bits |= ((w * 0x01020408u) >> 24) << (4 * j);                           // byte r -> bit r
            }
        }
        wds[m] = bits;
    }
    out[((int64_t)bq * nkb + kb) * 16 + lj] = u32x4b{wds[0] | (wds[1] << 16), wds[2] | (wds[3] << 16), wds[4] | (wds[5] << 16), wds[6] | (wds[7] << 16)};
}

// W [K 256 | V 256][64] fp32 -> the fp16 fragments of hs_attn_fkv_kernel: [head][kv][tile t][k-step s][lane][8]:
// lane (i = l & 15, kq = l >> 4) holds W[kv * heads * 32 + h * 32 + 16 t + i][32 s + 8 kq .. + 7]
__global__ __launch_bounds__(256) void attn_pack_kv_weights_kernel(const float* __restrict__ w, u32x4b* __restrict__ out, int heads) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= heads * 8 * 64) return;
    const int lane = idx & 63, fs = (idx >> 6) & 1, t = (idx >> 7) & 1, kv = (idx >> 8) & 1, h = idx >> 9;
    const float* src = w + (int64_t)(kv * heads * HD + h * HD + 16 * t + (lane & 15)) * 64 + 32 * fs + 8 * (lane >> 4);
    const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
    const u32x2b lo = pack4h(a.x, a.y, a.z, a.w), hi = pack4h(c.x, c.y, c.z, c.w);
    out[idx] = u32x4b{lo.x, lo.y, hi.x, hi.y};
}

// ---- short and medium key sequences (self-attention: 100 keys; the 15x20 / 30x40 levels) -----------------------------------
// With few keys the kernel above is all fixed cost: every wave holds all 7 query blocks for 1-2 key blocks, the four
// waves are reduced through 59 KB of LDS and 112 threads finish with strided 4-byte stores (14.6 us for 100 keys,
// 4 % MFMA utilisation).  Here the QUERY blocks are split over workgroups: a workgroup owns MQ query blocks, its NW waves
// split the key blocks round-robin, the
// partial sums meet in LDS (lane-contiguous, 9 values per lane and query block) and wave 0 finishes in registers.  No
// partial tensors in memory, no combine launch; K/V of an (image, head) are re-read by the ceil(7/MQ) workgroups of that
// head out of L2.
template <int MQ, int NW, typename KVT, int BF, int MM>
__global__ __launch_bounds__(NW * 64) void hs_attn_qk_kernel(const float* __restrict__ q, const KVT* __restrict__ k,
                                                            const KVT* __restrict__ v, const uint8_t* __restrict__ masked,
                                                            const int32_t* __restrict__ row_any, float* __restrict__ out, int Lq,
                                                            int S, int heads, int64_t ldq, int64_t q_sb, int64_t ldk,
                                                            int64_t k_sb, int64_t ldv, int64_t v_sb, float kappa) {
    // XCD-aware grid (round 6): (head, image, query-block group) -- workgroups are dealt to the 8 XCDs by linear id % 8, so with 8 heads
    // every workgroup of head h runs on XCD h: the ceil(7 / MQ) workgroups that share an (image, head)'s K / V read them out of ONE L2
    // (one HBM fetch instead of up to four) and an XCD's L2 holds an eighth of the level's K / V instead of half of it.  The old order
    // (query-block group fastest) put the sharers on different XCDs.
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW-1][MQ][9][64]: partial O (8) and l (1) per lane
    const int qb0 = blockIdx.z * MQ;                            // first 16-query block of this workgroup (all waves)

    float qf[MQ][8];
    bf16x4 qh[BF ? MQ : 1][2];
    bool use_mask[MQ];
    load_queries<BF, MQ>(q + (int64_t)b * q_sb + h * HD + lq * 8, ldq, qb0 * 16, Lq, lj, MM != 0, row_any ? row_any + (int64_t)b * Lq : nullptr, qf,
                         qh, use_mask);
    f32x4 o[MQ][2];
    float lsum[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        o[m][0] = o[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        lsum[m] = 0.f;
    }
    KeyCursor<KVT, MQ, MM> cur;
    cur.init(k + (int64_t)b * k_sb + h * HD, v + (int64_t)b * v_sb + h * HD, MM != 0 ? masked + (int64_t)b * Lq * S : nullptr, qb0 * 16, Lq, S, ldk,
             ldv, lj, lq);
    keys_stream<KVT, BF, MQ, MM>(cur, wave, NW, (S + 15) / 16, kappa * 1.4426950408889634f, qf, qh, use_mask, o, lsum);   // blocks wave, wave + NW, ...

    // ---- sum the waves' partials: waves 1.. park theirs lane-contiguously, wave 0 adds them to its registers ----
    if (wave > 0) {
        float* mine = red + (size_t)(wave - 1) * MQ * 9 * 64;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mine[(m * 9 + r) * 64 + lane] = o[m][0][r];
                mine[(m * 9 + 4 + r) * 64 + lane] = o[m][1][r];
            }
            mine[(m * 9 + 8) * 64 + lane] = lsum[m];
        }
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < NW; ++w) {
        const float* src = red + (size_t)(w - 1) * MQ * 9 * 64;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[m][0][r] += src[(m * 9 + r) * 64 + lane];
                o[m][1][r] += src[(m * 9 + 4 + r) * 64 + lane];
            }
            lsum[m] += src[(m * 9 + 8) * 64 + lane];
        }
    }
    // ---- finish in registers: o[m][half][r] = query (qb0+m)*16 + lq*4 + r, dim half*16 + lj ----
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        float l = lsum[m];                     // per query lj (any lq) after the two reductions
        l = sum_lane_rows(l);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lr = __shfl(l, lq * 4 + r, 64);          // denominator of this lane's output row
            const float a0 = o[m][0][r] / lr, a1 = o[m][1][r] / lr;
            float ss = a0 * a0 + a1 * a1;
            ss += wave_xor_dpp1(ss), ss += wave_xor_dpp2(ss), ss += wave_xor_dpp4(ss), ss += wave_xor_dpp8(ss);     // (over the 16-lane row, as the shfl_xor loop 1, 2, 4, 8)
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            const int qi = (qb0 + m) * 16 + lq * 4 + r;
            if (qi < Lq) {
                float* o_ = out + ((int64_t)b * Lq + qi) * (heads * HD) + h * HD + lj;
                o_[0] = a0 / nrm;
                o_[16] = a1 / nrm;
            }
        }
    }
}

// out[b][q][h*32 + d] = normalize( (sum_splits O) / (sum_splits l) ).  One workgroup per (head, image-chunk, 16-query block)
// -- 448 workgroups at B = 8 (one per (head, image-chunk) left 192 of the 256 CUs idle for a launch that is all latency): the
// block's 16 x 33 partial values are summed element-wise with 4 splits x 3 elements of independent loads in flight per thread (a
// per-query loop over the splits is a chain of nsplit * 33 dependent L2 round trips: 8.6 us), then two lanes per query
// finish from LDS with 16-byte stores.
// (Folding this into hs_attn_kernel -- last workgroup to arrive combines -- was measured: the device-scope release /
// acquire it needs writes back and invalidates the per-XCD L2s on gfx950, 70 -> 130 us.  Two launches it is.)
__global__ __launch_bounds__(256) void hs_attn_combine_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int Lq, int heads, int qchunks, int nsplit) {
    constexpr int BLK = 16 * PSTRIDE;                    // one query block of a partial tile
    __shared__ float tot[BLK];
    const int h = blockIdx.x, qb = blockIdx.z;
    const int b = blockIdx.y / qchunks, qc = blockIdx.y - b * qchunks;
    const int tid = threadIdx.x;
    if (qc * AQCH + qb * 16 >= Lq) return;              // a block of padding rows only
    const float* base = part + (((int64_t)blockIdx.y * heads + h) * nsplit) * (AQCH * PSTRIDE) + qb * BLK;
    constexpr int NE = (BLK + 255) / 256;
    float t[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) t[j] = 0.f;
    for (int sp0 = 0; sp0 < nsplit; sp0 += 4) {
        float u[4][NE];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool on = sp0 + r < nsplit;                      // uniform
            const float* src = base + (int64_t)(on ? sp0 + r : sp0) * (AQCH * PSTRIDE);
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int i = tid + 256 * j;
                u[r][j] = (on && i < BLK) ? src[i] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NE; ++j) t[j] += u[r][j];            // split order: s = 0, 1, 2, ...
    }
#pragma unroll
    for (int j = 0; j < NE; ++j)
        if (tid + 256 * j < BLK) tot[tid + 256 * j] = t[j];
    __syncthreads();
    const int ql = tid >> 1, half = tid & 1;             // two lanes per query, 16 output dims each
    const int qi = qc * AQCH + qb * 16 + ql;
    if (ql >= 16) return;
    const float l = tot[ql * PSTRIDE + HD];
    float a[16];
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        a[d] = tot[ql * PSTRIDE + half * 16 + d] / l;
        ss += a[d] * a[d];
    }
    ss += wave_xor_dpp1(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    if (qi < Lq) {
        float* o = out + ((int64_t)b * Lq + qi) * (heads * HD) + h * HD + half * 16;
#pragma unroll
        for (int d = 0; d < 16; ++d) o[d] = a[d] / nrm;
    }
}

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_hypersphere_attn_workspace(int B, int Lq, int S, int heads) {
    const int qchunks = cdiv(Lq, AQCH);
    const int ns = attn_nsplit(B, qchunks, heads, S);
    return (int64_t)B * qchunks * heads * ns * AQCH * PSTRIDE;
}

// KVT / BF: see the low-precision note above the kernels.
template <typename KVT, int BF>
static int attn_launch(const char* who, const float* q, const KVT* k, const KVT* v, const uint8_t* masked, const int32_t* row_any, float* out,
                       int B, int Lq, int S, int heads, int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv, int64_t v_sb,
                       float kappa, float* workspace, int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(q && k && v && out && workspace, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && Lq > 0 && S > 0 && heads > 0, "%s: bad sizes", who);
    constexpr int KA = 16 / (int)sizeof(KVT);      // elements per 16 bytes of K
    MSM_REQUIRE(ldq % 4 == 0 && ldk % KA == 0 && q_sb % 4 == 0 && k_sb % KA == 0 && (((uintptr_t)q) & 15) == 0 && (((uintptr_t)k) & 15) == 0,
                "%s: q/k must be 16-byte aligned with row / batch strides that keep them so", who);
    MSM_REQUIRE(!masked || (((uintptr_t)masked) & 3) == 0, "%s: mask must be 4-byte aligned", who);
    MSM_REQUIRE(ldk > 0 && ldv > 0 && (int64_t)S * ldk * (int64_t)sizeof(KVT) < ((int64_t)1 << 32) && (int64_t)S * ldv * (int64_t)sizeof(KVT) < ((int64_t)1 << 32) &&
                    (int64_t)Lq * S < ((int64_t)1 << 32),
                "%s: one image of K / V / mask must stay below 4 GiB (32-bit buffer offsets)", who);
    // mask access of the kernels: 0 none, 1 one 4-byte word per (query, key quad), 2 bytewise (key counts that are not multiples of 4)
    const int mm = !masked ? 0 : (S % 4 == 0 ? 1 : 2);
    const int qchunks = cdiv(Lq, AQCH);
    const int ns = attn_nsplit(B, qchunks, heads, S);
    const int64_t need = (int64_t)B * qchunks * heads * ns * AQCH * PSTRIDE;
    if (workspace_elems < need) {
        set_error("%s: workspace %lld < %lld floats", who, (long long)workspace_elems, (long long)need);
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int force = opt(MSM_OPT_ATTN_KERNEL);      // 3: the split-K kernel + combine at every length (the tested fallback)
    const int qk_max = opt(MSM_OPT_ATTN_QK_MAX) > 0 ? opt(MSM_OPT_ATTN_QK_MAX) : 2048;
    if (S <= qk_max && force != 3) {
        // short and medium sequences: query-split workgroups whose waves split the keys (measured at 1200 keys: 23 us against
        // 25 + 8 us for the split-K kernel + combine; at 4800 keys the split-K kernel, which normalises each key block
        // once for all 7 query blocks, is faster: 60 + 8 against 71 us)
        const int cfg_env = opt(MSM_OPT_ATTN_QKCFG);
        // one query block per workgroup for the shortest sequences (self-attention, 100 keys: 6.9 against 8.0 us), two
        // otherwise (K/V are read by half as many workgroups); other shapes measured slower at every length
        // many images (the second stage of the two-stage harness: 170 crops of 224x224, 49 / 196 / 784 keys): thousands of workgroups of a few
        // key blocks each are all launch and reduction -- four query blocks per workgroup of four waves there (cfg 2)
        const int cfg = cfg_env >= 0 ? cfg_env : (B >= 48 && S <= 1024 ? 2 : (S <= 128 ? 1 : 0));
#define QK_LAUNCH_M(MQ_, NW_, MM_)                                                                                              \
    {                                                                                                                           \
        dim3 grid(heads, B, cdiv(cdiv(Lq, 16), MQ_));                                                                           \
        const size_t lds2 = sizeof(float) * (size_t)(NW_ - 1) * MQ_ * 9 * 64;                                                   \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)hs_attn_qk_kernel<MQ_, NW_, KVT, BF, MM_>, lds2));            \
        hipLaunchKernelGGL((hs_attn_qk_kernel<MQ_, NW_, KVT, BF, MM_>), grid, dim3(NW_ * 64), lds2, st, q, k, v, masked, row_any, out, Lq, \
                           S, heads, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa);                                                   \
    }
#define QK_LAUNCH(MQ_, NW_)                                                                                                     \
    switch (mm) {                                                                                                               \
        case 0: QK_LAUNCH_M(MQ_, NW_, 0) break;                                                                                 \
        case 1: QK_LAUNCH_M(MQ_, NW_, 1) break;                                                                                 \
        default: QK_LAUNCH_M(MQ_, NW_, 2) break;                                                                                \
    }
        switch (cfg) {
            case 1: QK_LAUNCH(1, 8) break;
            case 2: QK_LAUNCH(4, 4) break;
            default: QK_LAUNCH(2, 8) break;
        }
#undef QK_LAUNCH
#undef QK_LAUNCH_M
        MSM_CHECK_LAUNCH(who);
        return MSM_OK;
    }
    const size_t lds = sizeof(float) * 4 * AQCH * PSTRIDE;
    dim3 grid(ns, heads, B * qchunks), block(256);
#define BIG_LAUNCH(MM_)                                                                                                         \
    {                                                                                                                           \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)hs_attn_kernel<KVT, BF, MM_>, lds));                          \
        hipLaunchKernelGGL((hs_attn_kernel<KVT, BF, MM_>), grid, block, lds, st, q, k, v, masked, row_any, workspace, out, Lq, S, heads, \
                           qchunks, ns, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa);                                                \
    }
    switch (mm) {
        case 0: BIG_LAUNCH(0) break;
        case 1: BIG_LAUNCH(1) break;
        default: BIG_LAUNCH(2) break;
    }
#undef BIG_LAUNCH
    MSM_CHECK_LAUNCH(who);
    if (ns == 1) return MSM_OK;
    dim3 g2(heads, B * qchunks, AQB), b2(256);
    hipLaunchKernelGGL(hs_attn_combine_kernel, g2, b2, 0, st, workspace, out, Lq, heads, qchunks, ns);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_hypersphere_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* masked,
                                        const int32_t* row_any, float* out, int B, int Lq, int S, int heads,
                                        int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv,
                                        int64_t v_sb, float kappa, float* workspace, int64_t workspace_elems,
                                        void* stream) {
    return attn_launch<float, 0>("msm_hypersphere_attn_fwd", q, k, v, masked, row_any, out, B, Lq, S, heads, ldq, q_sb, ldk, k_sb, ldv, v_sb,
                                     kappa, workspace, workspace_elems, stream);
}

extern "C" int msm_hypersphere_attn_lp_fwd(const float* q, const void* k, const void* v, int kv_format, const uint8_t* masked,
                                           const int32_t* row_any, float* out, int B, int Lq, int S, int heads,
                                           int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv,
                                           int64_t v_sb, float kappa, float* workspace, int64_t workspace_elems,
                                           void* stream) {
    MSM_REQUIRE(kv_format >= 0 && kv_format <= 3, "msm_hypersphere_attn_lp_fwd: kv_format=%d (0 fp32 K/V, 1 bf16 K/V, 2 fp16 K + bf16 V, 3 fp32 K/V with fp16 scores)", kv_format);
    if (kv_format == 2)
        return attn_launch<kvh16, 2>("msm_hypersphere_attn_lp_fwd", q, (const kvh16*)k, (const kvh16*)v, masked, row_any, out, B, Lq, S, heads, ldq,
                                     q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    if (kv_format == 1)
        return attn_launch<uint16_t, 1>("msm_hypersphere_attn_lp_fwd", q, (const uint16_t*)k, (const uint16_t*)v, masked, row_any, out, B, Lq, S,
                                           heads, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    // very short fp32 sequences (the decoder's self-attention: 100 keys) stay on the fp32 MFMAs: the launch is latency-bound, the
    // operand conversions only add to it (measured 8.5 against 6.6 us)
    if (S <= 128)
        return attn_launch<float, 0>("msm_hypersphere_attn_lp_fwd", q, (const float*)k, (const float*)v, masked, row_any, out, B, Lq, S, heads,
                                         ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    if (kv_format == 3)
        return attn_launch<float, 2>("msm_hypersphere_attn_lp_fwd", q, (const float*)k, (const float*)v, masked, row_any, out, B, Lq, S, heads, ldq,
                                     q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    return attn_launch<float, 1>("msm_hypersphere_attn_lp_fwd", q, (const float*)k, (const float*)v, masked, row_any, out, B, Lq, S, heads, ldq,
                                    q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
}

extern "C" int msm_attn_pack_kv_weights(const float* w, void* packed, int heads, void* stream) {
    MSM_REQUIRE(w && packed && heads > 0 && heads <= 64, "msm_attn_pack_kv_weights: bad arguments");
    MSM_REQUIRE(((((uintptr_t)w) | ((uintptr_t)packed)) & 15) == 0, "msm_attn_pack_kv_weights: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(attn_pack_kv_weights_kernel, dim3(cdiv(heads * 8 * 64, 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4b*)packed, heads);
    MSM_CHECK_LAUNCH("msm_attn_pack_kv_weights");
    return MSM_OK;
}

extern "C" int64_t msm_attn_mask_bits_bytes(int B, int Lq, int S) { return (int64_t)B * cdiv(Lq, AQCH) * (S / 16) * 256; }

extern "C" int msm_attn_pack_mask_bits(const uint8_t* masked, void* bits, int B, int Lq, int S, void* stream) {
    MSM_REQUIRE(masked && bits && B > 0 && Lq > 0 && S > 0 && S % 16 == 0, "msm_attn_pack_mask_bits: bad arguments (S %% 16 == 0)");
    MSM_REQUIRE(((((uintptr_t)masked) | ((uintptr_t)bits)) & 15) == 0, "msm_attn_pack_mask_bits: pointers must be 16-byte aligned");
    const int qchunks = cdiv(Lq, AQCH);
    hipLaunchKernelGGL(attn_pack_mask_bits_kernel, dim3(cdiv((int64_t)(S / 16) * 16, 256), B * qchunks), dim3(256), 0, (hipStream_t)stream, masked,
                       (u32x4b*)bits, Lq, S, qchunks);
    MSM_CHECK_LAUNCH("msm_attn_pack_mask_bits");
    return MSM_OK;
}

extern "C" int msm_hypersphere_attn_fused_kv_fwd(const float* q, const void* x_f16, const void* w_packed, const float* rowcol, const float* col_v_t,
                                                 int score_format, const void* masked, const int32_t* row_any, float* out, int B, int Lq,
                                                 int H, int W, int heads, int64_t ldq, int64_t q_sb, float kappa, float* workspace,
                                                 int64_t workspace_elems, void* stream) {
    const char* who = "msm_hypersphere_attn_fused_kv_fwd";
    MSM_REQUIRE(q && x_f16 && w_packed && rowcol && col_v_t && out && workspace, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && Lq > 0 && H > 0 && W > 0 && heads > 0, "%s: bad sizes", who);
    MSM_REQUIRE(W % 16 == 0, "%s: W=%d must be a multiple of 16 (a 16-key block lies in one image row)", who, W);
    MSM_REQUIRE(score_format == 1 || score_format == 2, "%s: score_format=%d (1 = bf16, 2 = fp16 q^ / k^ operands)", who, score_format);
    const int64_t S64 = (int64_t)H * W;
    MSM_REQUIRE(S64 * 128 < ((int64_t)1 << 32) && (int64_t)Lq * S64 < ((int64_t)1 << 32) && (int64_t)(H + W) * heads * HD * 8 < ((int64_t)1 << 32),
                "%s: one image of x / the mask / the constants must stay below 4 GiB (32-bit buffer offsets)", who);
    MSM_REQUIRE(ldq % 4 == 0 && q_sb % 4 == 0 && ((((uintptr_t)q) | ((uintptr_t)x_f16) | ((uintptr_t)w_packed) | ((uintptr_t)rowcol) | ((uintptr_t)col_v_t)) & 15) == 0,
                "%s: pointers must be 16-byte aligned", who);
    MSM_REQUIRE(!masked || (((uintptr_t)masked) & 15) == 0, "%s: the packed mask must be 16-byte aligned", who);
    const int S = (int)S64;
    const int qchunks = cdiv(Lq, AQCH);
    const int ns = attn_nsplit(B, qchunks, heads, S);
    const int64_t need = (int64_t)B * qchunks * heads * ns * AQCH * PSTRIDE;
    if (workspace_elems < need) {
        set_error("%s: workspace %lld < %lld floats", who, (long long)workspace_elems, (long long)need);
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(float) * 4 * AQCH * PSTRIDE;
    dim3 grid(ns, heads, B * qchunks), block(256);
#define FKV_LAUNCH(BF_, MM_)                                                                                                       \
    {                                                                                                                              \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)hs_attn_fkv_kernel<BF_, MM_>, lds));                             \
        hipLaunchKernelGGL((hs_attn_fkv_kernel<BF_, MM_>), grid, block, lds, st, q, (const unsigned short*)x_f16, (const u32x4b*)w_packed, rowcol, \
                           col_v_t, (const u32x4b*)masked, row_any, workspace, out, Lq, S, W, heads, qchunks, ns, ldq, q_sb, kappa);               \
    }
    if (score_format == 2) {
        if (masked) FKV_LAUNCH(2, 1) else FKV_LAUNCH(2, 0)
    } else {
        if (masked) FKV_LAUNCH(1, 1) else FKV_LAUNCH(1, 0)
    }
#undef FKV_LAUNCH
    MSM_CHECK_LAUNCH(who);
    if (ns == 1) return MSM_OK;
    dim3 g2(heads, B * qchunks, AQB), b2(256);
    hipLaunchKernelGGL(hs_attn_combine_kernel, g2, b2, 0, st, workspace, out, Lq, heads, qchunks, ns);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
