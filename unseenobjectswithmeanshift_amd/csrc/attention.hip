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
// kp: this lane's 8 dims of its key row; vbp + key * ldv: dim lj of a key row
__device__ __forceinline__ void kv_fetch(KVRaw<float>& f, const float* __restrict__ kp, const float* __restrict__ vbp, int64_t ldv, int key_c0, int S) {
    f.ka = *reinterpret_cast<const float4*>(kp);
    f.kc = *reinterpret_cast<const float4*>(kp + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* vp = vbp + (int64_t)min(key_c0 + r, S - 1) * ldv;
        f.v[r][0] = vp[0];
        f.v[r][1] = vp[16];
    }
}
__device__ __forceinline__ void kv_fetch(KVRaw<uint16_t>& f, const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vbp, int64_t ldv, int key_c0,
                                         int S) {
    f.k = *reinterpret_cast<const u32x4b*>(kp);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint16_t* vp = vbp + (int64_t)min(key_c0 + r, S - 1) * ldv;
        f.v[r][0] = vp[0];
        f.v[r][1] = vp[16];
    }
}
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
__device__ __forceinline__ float v_float(const KVRaw<float>& f, int r, int hh) { return f.v[r][hh]; }
__device__ __forceinline__ float v_float(const KVRaw<uint16_t>& f, int r, int hh) { return __uint_as_float((unsigned)f.v[r][hh] << 16); }

template <typename KVT, bool BF>
__global__ __launch_bounds__(256) void hs_attn_kernel(const float* __restrict__ q, const KVT* __restrict__ k,
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

    // ---- Q^ fragments (B operand): lane (query lj of block m, dims lq*8 + t) ----
    float qf[AQB][8];
    bf16x4 qh[BF ? AQB : 1][2];             // BF: the same fragments as bf16 B operands (dims lq*8 + 0..3 | + 4..7)
    const float* qb = q + (int64_t)b * q_sb + h * HD + lq * 8;
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        const int qi = q0 + m * 16 + lj;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        if (qi < Lq) {
            const float* p = qb + (int64_t)qi * ldq;
            a = *reinterpret_cast<const float4*>(p);
            c = *reinterpret_cast<const float4*>(p + 4);
        }
        float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        qf[m][0] = a.x * rn; qf[m][1] = a.y * rn; qf[m][2] = a.z * rn; qf[m][3] = a.w * rn;
        qf[m][4] = c.x * rn; qf[m][5] = c.y * rn; qf[m][6] = c.z * rn; qf[m][7] = c.w * rn;
        if constexpr (BF) {
            qh[m][0] = pack4(qf[m][0], qf[m][1], qf[m][2], qf[m][3]);
            qh[m][1] = pack4(qf[m][4], qf[m][5], qf[m][6], qf[m][7]);
        }
    }
    // per-row mask enable: rows whose keys are all masked attend everywhere (DEC:618)
    bool use_mask[AQB];
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        const int qi = q0 + m * 16 + lj;
        use_mask[m] = masked != nullptr && qi < Lq && (row_any == nullptr || row_any[(int64_t)b * Lq + qi] != 0);
    }

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
    const KVT* kbp = k + (int64_t)b * k_sb + h * HD + lq * 8;
    const KVT* vbp = v + (int64_t)b * v_sb + h * HD + lj;
    const bool mask_vec = (S % 4) == 0;

    // Software prefetch: the K / V fragments and the 7 mask words of key block kb+4 are fetched (from
    // clamped, always-valid addresses) before the 112 MFMAs of block kb, pinned with sched_barrier.
    struct Frag {
        KVRaw<KVT> kv;
        uint32_t mw[AQB];
    };
    auto fetch = [&](int kb, Frag& f) {
        const int key_a = min(kb * 16 + lj, S - 1);
        const int key_c0 = kb * 16 + lq * 4;
        kv_fetch(f.kv, kbp + (int64_t)key_a * ldk, vbp, ldv, key_c0, S);
#pragma unroll
        for (int m = 0; m < AQB; ++m) {
            uint32_t w = 0;
            if (masked != nullptr) {
                const int qi = min(q0 + m * 16 + lj, Lq - 1);
                const uint8_t* mp = masked + ((int64_t)b * Lq + qi) * S;
                if (mask_vec) {
                    w = *reinterpret_cast<const uint32_t*>(mp + min(key_c0, S - 4));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) w |= (uint32_t)mp[min(key_c0 + r, S - 1)] << (8 * r);
                }
            }
            f.mw[m] = w;
        }
    };
    const float k2 = kappa * 1.4426950408889634f;   // kappa * log2(e)
    auto consume = [&](int kb, const Frag& f) {
        float kr[8];
        k_floats(f.kv, kr);
        float ss = (kr[0] * kr[0] + kr[1] * kr[1] + kr[2] * kr[2] + kr[3] * kr[3]) + (kr[4] * kr[4] + kr[5] * kr[5] + kr[6] * kr[6] + kr[7] * kr[7]);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        const float kf[8] = {kr[0] * rn, kr[1] * rn, kr[2] * rn, kr[3] * rn, kr[4] * rn, kr[5] * rn, kr[6] * rn, kr[7] * rn};
        bf16x4 kh[2], vb[2];
        if constexpr (BF) {
            kh[0] = pack4(kf[0], kf[1], kf[2], kf[3]);
            kh[1] = pack4(kf[4], kf[5], kf[6], kf[7]);
            v_operands(f.kv, vb);
        }
        const int key_c0 = kb * 16 + lq * 4;
#pragma unroll
        for (int m = 0; m < AQB; ++m) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                s = mfma_bf16(kh[0], qh[m][0], s);
                s = mfma_bf16(kh[1], qh[m][1], s);
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) s = mfma16(kf[t], qf[m][t], s);
            }
            // s[r]: key key_c0 + r, query q0 + m*16 + lj
            const uint32_t mw = use_mask[m] ? f.mw[m] : 0u;
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool dead = (key_c0 + r >= S) || ((mw >> (8 * r)) & 0xffu);
                // exp(kappa*s - kappa) as one fma + v_exp_f32: the argument is in [-2*kappa, 0], so the absolute
                // error of the fp32 argument (< 6e-6 at kappa = 30) bounds the relative error of p at ~4e-6
                p[r] = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], k2, -k2));
            }
            lsum[m] += (p[0] + p[1]) + (p[2] + p[3]);
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
        }
    };
    {
        Frag fa, fb;
        int kb = kb_beg + wave;
        if (kb < kb_end) fetch(kb, fa);
        for (; kb + 4 < kb_end; kb += 8) {        // two blocks per trip: ping-pong without register copies
            fetch(kb + 4, fb);
            __builtin_amdgcn_sched_barrier(0);
            consume(kb, fa);
            __builtin_amdgcn_sched_barrier(0);
            fetch(min(kb + 8, nkb - 1), fa);
            __builtin_amdgcn_sched_barrier(0);
            consume(kb + 4, fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb < kb_end) consume(kb, fa);
    }

    // ---- reduce the 4 waves through LDS, then one partial per workgroup ----
    float* mine = red + wave * (AQCH * PSTRIDE);
#pragma unroll
    for (int m = 0; m < AQB; ++m) {
        float l = lsum[m];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
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

// ---- short and medium key sequences (self-attention: 100 keys; the 15x20 / 30x40 levels) -----------------------------------
// With few keys the kernel above is all fixed cost: every wave holds all 7 query blocks for 1-2 key blocks, the four
// waves are reduced through 59 KB of LDS and 112 threads finish with strided 4-byte stores (14.6 us for 100 keys,
// 4 % MFMA utilisation).  Here the QUERY blocks are split over workgroups: a workgroup owns MQ query blocks, its NW waves
// split the key blocks round-robin, the
// partial sums meet in LDS (lane-contiguous, 9 values per lane and query block) and wave 0 finishes in registers.  No
// partial tensors in memory, no combine launch; K/V of an (image, head) are re-read by the ceil(7/MQ) workgroups of that
// head out of L2.
template <int MQ, int NW, typename KVT, bool BF>
__global__ __launch_bounds__(NW * 64) void hs_attn_qk_kernel(const float* __restrict__ q, const KVT* __restrict__ k,
                                                            const KVT* __restrict__ v, const uint8_t* __restrict__ masked,
                                                            const int32_t* __restrict__ row_any, float* __restrict__ out, int Lq,
                                                            int S, int heads, int64_t ldq, int64_t q_sb, int64_t ldk,
                                                            int64_t k_sb, int64_t ldv, int64_t v_sb, float kappa) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW-1][MQ][9][64]: partial O (8) and l (1) per lane
    const int qb0 = blockIdx.x * MQ;                            // first 16-query block of this workgroup (all waves)

    float qf[MQ][8];
    bf16x4 qh[BF ? MQ : 1][2];
    bool use_mask[MQ];
    const float* qbp = q + (int64_t)b * q_sb + h * HD + lq * 8;
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        const int qi = (qb0 + m) * 16 + lj;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        if (qi < Lq) {
            const float* p = qbp + (int64_t)qi * ldq;
            a = *reinterpret_cast<const float4*>(p);
            c = *reinterpret_cast<const float4*>(p + 4);
        }
        float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        qf[m][0] = a.x * rn; qf[m][1] = a.y * rn; qf[m][2] = a.z * rn; qf[m][3] = a.w * rn;
        qf[m][4] = c.x * rn; qf[m][5] = c.y * rn; qf[m][6] = c.z * rn; qf[m][7] = c.w * rn;
        if constexpr (BF) {
            qh[m][0] = pack4(qf[m][0], qf[m][1], qf[m][2], qf[m][3]);
            qh[m][1] = pack4(qf[m][4], qf[m][5], qf[m][6], qf[m][7]);
        }
        use_mask[m] = masked != nullptr && qi < Lq && (row_any == nullptr || row_any[(int64_t)b * Lq + qi] != 0);   // DEC:618
    }
    f32x4 o[MQ][2];
    float lsum[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        o[m][0] = o[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        lsum[m] = 0.f;
    }
    const int nkb = (S + 15) / 16;
    const KVT* kbp = k + (int64_t)b * k_sb + h * HD + lq * 8;
    const KVT* vbp = v + (int64_t)b * v_sb + h * HD + lj;
    const bool mask_vec = (S % 4) == 0;
    struct Frag {
        KVRaw<KVT> kv;
        uint32_t mw[MQ];
    };
    auto fetch = [&](int kb, Frag& f) {
        const int key_c0 = kb * 16 + lq * 4;
        kv_fetch(f.kv, kbp + (int64_t)min(kb * 16 + lj, S - 1) * ldk, vbp, ldv, key_c0, S);
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            uint32_t w = 0;
            if (masked != nullptr) {
                const uint8_t* mp = masked + ((int64_t)b * Lq + min((qb0 + m) * 16 + lj, Lq - 1)) * S;
                if (mask_vec) {
                    w = *reinterpret_cast<const uint32_t*>(mp + min(key_c0, S - 4));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) w |= (uint32_t)mp[min(key_c0 + r, S - 1)] << (8 * r);
                }
            }
            f.mw[m] = w;
        }
    };
    const float k2 = kappa * 1.4426950408889634f;
    auto consume = [&](int kb, const Frag& f) {
        float kr[8];
        k_floats(f.kv, kr);
        float ss = (kr[0] * kr[0] + kr[1] * kr[1] + kr[2] * kr[2] + kr[3] * kr[3]) + (kr[4] * kr[4] + kr[5] * kr[5] + kr[6] * kr[6] + kr[7] * kr[7]);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        const float kf[8] = {kr[0] * rn, kr[1] * rn, kr[2] * rn, kr[3] * rn, kr[4] * rn, kr[5] * rn, kr[6] * rn, kr[7] * rn};
        bf16x4 kh[2], vb[2];
        if constexpr (BF) {
            kh[0] = pack4(kf[0], kf[1], kf[2], kf[3]);
            kh[1] = pack4(kf[4], kf[5], kf[6], kf[7]);
            v_operands(f.kv, vb);
        }
        const int key_c0 = kb * 16 + lq * 4;
        f32x4 sc[MQ];
#pragma unroll
        for (int m = 0; m < MQ; ++m) sc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int m = 0; m < MQ; ++m) sc[m] = mfma_bf16(kh[c], qh[m][c], sc[m]);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int m = 0; m < MQ; ++m) sc[m] = mfma16(kf[t], qf[m][t], sc[m]);
        }
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const uint32_t mw = use_mask[m] ? f.mw[m] : 0u;
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool dead = (key_c0 + r >= S) || ((mw >> (8 * r)) & 0xffu);
                p[r] = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(sc[m][r], k2, -k2));
            }
            lsum[m] += (p[0] + p[1]) + (p[2] + p[3]);
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
        }
    };
    {
        Frag fa, fb;
        int kb = wave;                             // key blocks wave, wave + NW, ...
        if (kb < nkb) fetch(kb, fa);
        for (; kb + NW < nkb; kb += 2 * NW) {
            fetch(kb + NW, fb);
            __builtin_amdgcn_sched_barrier(0);
            consume(kb, fa);
            __builtin_amdgcn_sched_barrier(0);
            fetch(min(kb + 2 * NW, nkb - 1), fa);
            __builtin_amdgcn_sched_barrier(0);
            consume(kb + NW, fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kb < nkb) consume(kb, fa);
    }
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
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lr = __shfl(l, lq * 4 + r, 64);          // denominator of this lane's output row
            const float a0 = o[m][0][r] / lr, a1 = o[m][1][r] / lr;
            float ss = a0 * a0 + a1 * a1;
#pragma unroll
            for (int x = 1; x < 16; x <<= 1) ss += __shfl_xor(ss, x, 64);
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

// out[b][q][h*32 + d] = normalize( (sum_splits O) / (sum_splits l) ).  One workgroup per (head, image-chunk): the
// partial tiles are summed element-wise with 4 splits x 15 elements of independent loads in flight per thread (a
// per-query loop over the splits is a chain of nsplit * 33 dependent L2 round trips: 8.6 us), then two lanes per query
// finish from LDS with 16-byte stores.
// (Folding this into hs_attn_kernel -- last workgroup to arrive combines -- was measured: the device-scope release /
// acquire it needs writes back and invalidates the per-XCD L2s on gfx950, 70 -> 130 us.  Two launches it is.)
__global__ __launch_bounds__(256) void hs_attn_combine_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int Lq, int heads, int qchunks, int nsplit) {
    __shared__ float tot[AQCH * PSTRIDE];
    const int h = blockIdx.x;
    const int b = blockIdx.y / qchunks, qc = blockIdx.y - b * qchunks;
    const int tid = threadIdx.x;
    const float* base = part + (((int64_t)blockIdx.y * heads + h) * nsplit) * (AQCH * PSTRIDE);
    constexpr int NE = (AQCH * PSTRIDE + 255) / 256;
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
                u[r][j] = (on && i < AQCH * PSTRIDE) ? src[i] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NE; ++j) t[j] += u[r][j];            // split order: s = 0, 1, 2, ...
    }
#pragma unroll
    for (int j = 0; j < NE; ++j)
        if (tid + 256 * j < AQCH * PSTRIDE) tot[tid + 256 * j] = t[j];
    __syncthreads();
    const int ql = tid >> 1, half = tid & 1;             // two lanes per query, 16 output dims each
    const int qi = qc * AQCH + ql;
    if (ql >= AQCH) return;
    const float l = tot[ql * PSTRIDE + HD];
    float a[16];
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        a[d] = tot[ql * PSTRIDE + half * 16 + d] / l;
        ss += a[d] * a[d];
    }
    ss += __shfl_xor(ss, 1, 64);
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
template <typename KVT, bool BF>
static int attn_launch(const char* who, const float* q, const KVT* k, const KVT* v, const uint8_t* masked, const int32_t* row_any, float* out,
                       int B, int Lq, int S, int heads, int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv, int64_t v_sb,
                       float kappa, float* workspace, int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(q && k && v && out && workspace, "%s: null pointer", who);
    MSM_REQUIRE(B > 0 && Lq > 0 && S > 0 && heads > 0, "%s: bad sizes", who);
    constexpr int KA = 16 / (int)sizeof(KVT);      // elements per 16 bytes of K
    MSM_REQUIRE(ldq % 4 == 0 && ldk % KA == 0 && q_sb % 4 == 0 && k_sb % KA == 0 && (((uintptr_t)q) & 15) == 0 && (((uintptr_t)k) & 15) == 0,
                "%s: q/k must be 16-byte aligned with row / batch strides that keep them so", who);
    MSM_REQUIRE(!masked || (((uintptr_t)masked) & 3) == 0, "%s: mask must be 4-byte aligned", who);
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
        const int cfg = cfg_env >= 0 ? cfg_env : (S <= 128 ? 1 : 0);
#define QK_LAUNCH(MQ_, NW_)                                                                                                     \
    {                                                                                                                           \
        dim3 grid(cdiv(cdiv(Lq, 16), MQ_), heads, B);                                                                           \
        const size_t lds2 = sizeof(float) * (size_t)(NW_ - 1) * MQ_ * 9 * 64;                                                   \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)hs_attn_qk_kernel<MQ_, NW_, KVT, BF>, lds2));                 \
        hipLaunchKernelGGL((hs_attn_qk_kernel<MQ_, NW_, KVT, BF>), grid, dim3(NW_ * 64), lds2, st, q, k, v, masked, row_any, out, Lq, S, \
                           heads, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa);                                                      \
    }
        switch (cfg) {
            case 1: QK_LAUNCH(1, 8) break;
            default: QK_LAUNCH(2, 8) break;
        }
#undef QK_LAUNCH
        MSM_CHECK_LAUNCH(who);
        return MSM_OK;
    }
    const size_t lds = sizeof(float) * 4 * AQCH * PSTRIDE;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)hs_attn_kernel<KVT, BF>, lds));
    dim3 grid(ns, heads, B * qchunks), block(256);
    hipLaunchKernelGGL((hs_attn_kernel<KVT, BF>), grid, block, lds, st, q, k, v, masked, row_any, workspace, out, Lq, S, heads, qchunks,
                       ns, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa);
    MSM_CHECK_LAUNCH(who);
    if (ns == 1) return MSM_OK;
    dim3 g2(heads, B * qchunks), b2(256);
    hipLaunchKernelGGL(hs_attn_combine_kernel, g2, b2, 0, st, workspace, out, Lq, heads, qchunks, ns);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_hypersphere_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* masked,
                                        const int32_t* row_any, float* out, int B, int Lq, int S, int heads,
                                        int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv,
                                        int64_t v_sb, float kappa, float* workspace, int64_t workspace_elems,
                                        void* stream) {
    return attn_launch<float, false>("msm_hypersphere_attn_fwd", q, k, v, masked, row_any, out, B, Lq, S, heads, ldq, q_sb, ldk, k_sb, ldv, v_sb,
                                     kappa, workspace, workspace_elems, stream);
}

extern "C" int msm_hypersphere_attn_lp_fwd(const float* q, const void* k, const void* v, int kv_bf16, const uint8_t* masked,
                                           const int32_t* row_any, float* out, int B, int Lq, int S, int heads,
                                           int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv,
                                           int64_t v_sb, float kappa, float* workspace, int64_t workspace_elems,
                                           void* stream) {
    if (kv_bf16)
        return attn_launch<uint16_t, true>("msm_hypersphere_attn_lp_fwd", q, (const uint16_t*)k, (const uint16_t*)v, masked, row_any, out, B, Lq, S,
                                           heads, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    // very short fp32 sequences (the decoder's self-attention: 100 keys) stay on the fp32 MFMAs: the launch is latency-bound, the
    // operand conversions only add to it (measured 8.5 against 6.6 us)
    if (S <= 128)
        return attn_launch<float, false>("msm_hypersphere_attn_lp_fwd", q, (const float*)k, (const float*)v, masked, row_any, out, B, Lq, S, heads,
                                         ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
    return attn_launch<float, true>("msm_hypersphere_attn_lp_fwd", q, (const float*)k, (const float*)v, masked, row_any, out, B, Lq, S, heads, ldq,
                                    q_sb, ldk, k_sb, ldv, v_sb, kappa, workspace, workspace_elems, stream);
}
