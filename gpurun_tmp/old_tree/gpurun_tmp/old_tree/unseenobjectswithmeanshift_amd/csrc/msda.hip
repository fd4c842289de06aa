// Multi-scale deformable attention forward (see include/msm_hip.h).
//
// Reference: ms_deformable_im2col_gpu_kernel, ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 and its
// bilinear helper :38-89 (one thread per output scalar, 48 dependent 4-byte gathers each); host
// wrapper ops/src/cuda/ms_deform_attn_cuda.cu:25-85; module arithmetic ops/modules/ms_deform_attn.py:
// 101-109 and the encoder's reference points, msdeformattn.py:141-153.
//
// gfx950 mapping: a gather kernel is bound by the number of vector-memory instructions and the L2
// sectors they touch, not by FLOPs.  One lane owns 4 consecutive channels (1 when D % 4 != 0) of one (query, head), so
// every bilinear tap is ONE 16-byte load and the D/4 lanes of a head fetch a contiguous 4*D-byte
// segment of `value` ([B][S][M][D], channel-fastest).  The M*D/4 lanes of a query write one
// contiguous 4*M*D-byte output row.  Level geometry is read once into registers (the reference
// re-reads the int64 shapes inside the point loop).  In the encoder form the softmax over the L*P
// logits and the sampling-location arithmetic are done in registers, so sampling_locations and
// attention_weights (7.3 MB per layer-image at 640x480) never exist in memory.
#include <stdlib.h>

#include "common.h"

namespace msm {

constexpr int MAXL = 8;

template <int V>
struct Vec {
    float e[V];
};
template <int V>
__device__ __forceinline__ Vec<V> ldv(const float* p) {
    Vec<V> r;
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.e[0] = t.x; r.e[1] = t.y; r.e[2] = t.z; r.e[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) r.e[i] = p[i];
    }
    return r;
}

// one sampling point: same arithmetic as cuh:290-300 and cuh:43-88
template <int V>
__device__ __forceinline__ void sample_point(Vec<V>& acc, const float* __restrict__ vl /* level base for (b, m, d) */,
                                             int H, int W, int64_t pix_stride, float loc_x, float loc_y, float wgt) {
    const float h_im = loc_y * (float)H - 0.5f;
    const float w_im = loc_x * (float)W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    Vec<V> v1, v2, v3, v4;
#pragma unroll
    for (int i = 0; i < V; ++i) v1.e[i] = v2.e[i] = v3.e[i] = v4.e[i] = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = ldv<V>(vl + ((int64_t)h_low * W + w_low) * pix_stride);
    if (h_low >= 0 && w_high <= W - 1) v2 = ldv<V>(vl + ((int64_t)h_low * W + w_high) * pix_stride);
    if (h_high <= H - 1 && w_low >= 0) v3 = ldv<V>(vl + ((int64_t)h_high * W + w_low) * pix_stride);
    if (h_high <= H - 1 && w_high <= W - 1) v4 = ldv<V>(vl + ((int64_t)h_high * W + w_high) * pix_stride);
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
    for (int i = 0; i < V; ++i) acc.e[i] += (w1 * v1.e[i] + w2 * v2.e[i] + w3 * v3.e[i] + w4 * v4.e[i]) * wgt;
}

template <bool ENC, int V>
__global__ __launch_bounds__(256) void msda_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                   const int64_t* __restrict__ lstart, const float* __restrict__ loc,
                                                   const float* __restrict__ wgt, const float* __restrict__ proj,
                                                   float* __restrict__ out, int B, int S, int M, int D, int L, int Lq,
                                                   int P) {
    // XCD-aware mapping: workgroup w runs on XCD w % 8, so image b = w % B keeps one image's `value`
    // (1.6 MB at 640x480) inside one XCD's 4 MiB L2 instead of streaming all B images through all of
    // them (measured: 312 MB of fabric reads per launch with the naive mapping).
    const int D4 = D / V;
    const int per_img = Lq * M * D4;
    const int b = blockIdx.x % B;
    const int idx = (blockIdx.x / B) * 256 + threadIdx.x;
    if (idx >= per_img) return;
    const int d4 = idx % D4;
    int t = idx / D4;
    const int m = t % M;
    const int qi = t / M;

    int Hs[MAXL], Ws[MAXL], st[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        if (l < L) {
            Hs[l] = (int)shapes[2 * l];
            Ws[l] = (int)shapes[2 * l + 1];
            st[l] = (int)lstart[l];
        } else {
            Hs[l] = Ws[l] = 1;
            st[l] = 0;
        }
    }
    const int64_t pix_stride = (int64_t)M * D;
    const float* vb = value + (int64_t)b * S * pix_stride + m * D + d4 * V;
    Vec<V> acc;
#pragma unroll
    for (int i = 0; i < V; ++i) acc.e[i] = 0.f;

    if constexpr (ENC) {
        const int LP = L * P;
        const float* pr = proj + ((int64_t)b * S + qi) * (M * LP * 3);
        const float* offp = pr + (int64_t)m * LP * 2;
        const float* lgp = pr + (int64_t)M * LP * 2 + (int64_t)m * LP;
        // reference point = centre of pixel qi in its own level, normalised (msdeformattn.py:141-153)
        int ql = 0;
#pragma unroll
        for (int l = 1; l < MAXL; ++l)
            if (l < L && qi >= st[l]) ql = l;
        const int local = qi - st[ql];
        const int ry = local / Ws[ql], rx = local - ry * Ws[ql];
        const float ref_x = ((float)rx + 0.5f) / (float)Ws[ql];
        const float ref_y = ((float)ry + 0.5f) / (float)Hs[ql];
        // softmax over the L*P logits (ms_deform_attn.py:103)
        float mx = -INFINITY;
        for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lgp[i]);
        float den = 0.f;
        for (int i = 0; i < LP; ++i) den += expf(lgp[i] - mx);
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l >= L) break;
            const float* vl = vb + (int64_t)st[l] * pix_stride;
            for (int p = 0; p < P; ++p) {
                const int i = l * P + p;
                const float lx = ref_x + offp[2 * i] / (float)Ws[l];      // ms_deform_attn.py:107-109
                const float ly = ref_y + offp[2 * i + 1] / (float)Hs[l];
                const float w = expf(lgp[i] - mx);
                sample_point(acc, vl, Hs[l], Ws[l], pix_stride, lx, ly, w / den);
            }
        }
    } else {
        const int64_t base = (((int64_t)b * Lq + qi) * M + m) * L * P;
        const float* lp = loc + base * 2;
        const float* wp = wgt + base;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l >= L) break;
            const float* vl = vb + (int64_t)st[l] * pix_stride;
            for (int p = 0; p < P; ++p) {
                const int i = l * P + p;
                sample_point(acc, vl, Hs[l], Ws[l], pix_stride, lp[2 * i], lp[2 * i + 1], wp[i]);
            }
        }
    }
    float* op = out + (((int64_t)b * Lq + qi) * M + m) * D + d4 * V;
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(op) = make_float4(acc.e[0], acc.e[1], acc.e[2], acc.e[3]);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) op[i] = acc.e[i];
    }
}

// ---- encoder form over a HEAD-MAJOR value tensor [B][M][S][D] ------------------------------------------------------
// Measured on MI355X: the gather is bound by the number of distinct cache lines a wave instruction touches, not by
// bytes -- with the token-major layout a (query, head) tap is a 32-byte segment (D = 8) and an instruction touches
// 32 of them; running the same taps as 64-byte segments took 55 us instead of 79 us.  In head-major order the two
// x-neighbours of a bilinear tap are adjacent in memory, so a group of 2*D/4 lanes owns one (query, head): lane
// (cx, d4) fetches column w_low + cx, channels 4*d4..+3 -- one contiguous 2*D*4-byte segment per tap row -- and
// accumulates its own column's share; the two columns are added with one shuffle at the end.  Per lane that is 2
// loads per sampling point instead of 4.  The producer (msm_encoder_block_fwd, value_head_major = 1) writes this
// layout directly.
template <int V>
__global__ __launch_bounds__(256) void msda_enc_hm_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lstart, const float* __restrict__ proj,
                                                          float* __restrict__ out, int B, int S, int M, int D, int L, int P) {
    const int D4 = D / V;
    const int G = 2 * D4;                       // lanes per (query, head)
    const int per_img = S * M * G;
    const int b = blockIdx.x % B;               // XCD-aware: one image's value map stays in one XCD's L2
    const int idx = (blockIdx.x / B) * 256 + threadIdx.x;
    const bool live = idx < per_img;            // G divides 256: a group is never split by this bound
    const int cidx = live ? idx : 0;
    const int g = cidx % G;
    const int cx = g / D4, d4 = g - cx * D4;
    const int t = cidx / G;
    const int m = t % M;
    const int qi = t / M;

    int Hs[MAXL], Ws[MAXL], st[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        if (l < L) {
            Hs[l] = (int)shapes[2 * l];
            Ws[l] = (int)shapes[2 * l + 1];
            st[l] = (int)lstart[l];
        } else {
            Hs[l] = Ws[l] = 1;
            st[l] = 0;
        }
    }
    const float* vb = value + ((int64_t)b * M + m) * S * D + d4 * V;       // head plane of this image
    Vec<V> acc;
#pragma unroll
    for (int i = 0; i < V; ++i) acc.e[i] = 0.f;

    const int LP = L * P;
    const float* pr = proj + ((int64_t)b * S + qi) * (M * LP * 3);
    const float* offp = pr + (int64_t)m * LP * 2;
    const float* lgp = pr + (int64_t)M * LP * 2 + (int64_t)m * LP;
    // reference point = centre of pixel qi in its own level, normalised (msdeformattn.py:141-153)
    int ql = 0;
#pragma unroll
    for (int l = 1; l < MAXL; ++l)
        if (l < L && qi >= st[l]) ql = l;
    const int local = qi - st[ql];
    const int ry = local / Ws[ql], rx = local - ry * Ws[ql];
    const float ref_x = ((float)rx + 0.5f) / (float)Ws[ql];
    const float ref_y = ((float)ry + 0.5f) / (float)Hs[ql];
    float mx = -INFINITY;                                                    // softmax over the L*P logits
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lgp[i]);
    float den = 0.f;
    for (int i = 0; i < LP; ++i) den += expf(lgp[i] - mx);
    const float rden = 1.0f / den;
    // (Tried and measured slower, 91-99 us against 80 us: v_exp_f32 numerators, float2 offsets, a single predicate
    // instead of the early exits, 32-bit indices -- the early `continue`s skip most of the per-point work of the
    // lane whose column is out of range and keep the body short.)
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        if (l >= L) break;
        const int H = Hs[l], W = Ws[l];
        const float* vl = vb + (int64_t)st[l] * D;
        for (int p = 0; p < P; ++p) {
            const int i = l * P + p;
            const float lx = ref_x + offp[2 * i] / (float)W;                 // ms_deform_attn.py:107-109
            const float ly = ref_y + offp[2 * i + 1] / (float)H;
            const float wgt = expf(lgp[i] - mx) * rden;
            const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
            if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;   // cuh:293
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const int xw = w_low + cx;                                       // this lane's column
            const float wxw = (cx ? lw : 1.f - lw) * wgt;
            if (xw < 0 || xw > W - 1) continue;
            Vec<V> vt, vbm;
#pragma unroll
            for (int c = 0; c < V; ++c) vt.e[c] = vbm.e[c] = 0.f;
            if (h_low >= 0) vt = ldv<V>(vl + ((int64_t)h_low * W + xw) * D);
            if (h_low + 1 <= H - 1) vbm = ldv<V>(vl + ((int64_t)(h_low + 1) * W + xw) * D);
            const float wt = (1.f - lh) * wxw, wb = lh * wxw;
#pragma unroll
            for (int c = 0; c < V; ++c) acc.e[c] += wt * vt.e[c] + wb * vbm.e[c];
        }
    }
#pragma unroll
    for (int c = 0; c < V; ++c) acc.e[c] += __shfl_xor(acc.e[c], D4, 64);    // left + right column
    if (live && cx == 0) {
        float* op = out + (((int64_t)b * S + qi) * M + m) * D + d4 * V;
        if constexpr (V == 4) {
            *reinterpret_cast<float4*>(op) = make_float4(acc.e[0], acc.e[1], acc.e[2], acc.e[3]);
        } else {
#pragma unroll
            for (int c = 0; c < V; ++c) op[c] = acc.e[c];
        }
    }
}

// Specialisation of the head-major form for D = 8 (the pixel decoder: 64 channels / 8 heads), L*P <= 16.
// The 4 lanes of a (query, head) group are an aligned quad.  The generic kernel above makes every lane repeat the
// group's softmax and sampling-location arithmetic (two fp32 divisions and an exp per point) and is VALU-bound
// (80 us).  Here lane g of the quad does that arithmetic only for points g, g+4, g+8, g+12, and the quad exchanges
// (x, y, weight) with DPP quad broadcasts: a third of the divisions/exps per lane plus 3 one-cycle moves per point.
__device__ __forceinline__ float quad_bcast(float v, int src) {   // src is a compile-time constant after unrolling
    const int iv = __float_as_int(v);
    int r;
    switch (src) {
        case 0: r = __builtin_amdgcn_mov_dpp(iv, 0x00, 0xf, 0xf, true); break;
        case 1: r = __builtin_amdgcn_mov_dpp(iv, 0x55, 0xf, 0xf, true); break;
        case 2: r = __builtin_amdgcn_mov_dpp(iv, 0xAA, 0xf, 0xf, true); break;
        default: r = __builtin_amdgcn_mov_dpp(iv, 0xFF, 0xf, 0xf, true); break;
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
    return fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)));   // [2,3,0,1]
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
    return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));
}

// QM (query-major blocks): a workgroup is 64 CONSECUTIVE queries of ONE head instead of 8 queries x 8 heads.  Neighbouring
// queries sample neighbouring locations, and with the head-major value layout a workgroup then works on one 200-KB map
// instead of eight: the lines it gathers are re-used out of the CU's L1 instead of each being fetched from L2 once per
// query that touches it.
typedef unsigned int msda_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldv4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const msda_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
__device__ __forceinline__ int clamp0(int x, int hi) {        // min(max(x, 0), hi) as one v_med3_i32 (one SGPR operand: constant bus)
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// LC / PC > 0: levels and points per level known at compile time (3 x 4: every shipped configuration) -- the per-point level
// geometry is then a fixed SGPR instead of a chain of scalar selects, and the LP < 16 guards fold away.
template <bool QM, int LC, int PC>
__global__ __launch_bounds__(256) void msda_enc_hm8_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart, const float* __restrict__ proj,
                                                           float* __restrict__ out, int B, int S, int M, int L_rt, int P_rt) {
    constexpr int D = 8;
    const int L = LC > 0 ? LC : L_rt, P = PC > 0 ? PC : P_rt;
    const int b = blockIdx.x % B;
    const int blk = blockIdx.x / B;
    const int g = threadIdx.x & 3;
    const int cx = g >> 1, d4 = g & 1;
    int m, qi;
    bool live;                                  // whole quads live or dead together
    if constexpr (QM) {
        m = blk % M;
        const int q_raw = (blk / M) * 64 + ((int)threadIdx.x >> 2);
        live = q_raw < S;
        qi = live ? q_raw : 0;
    } else {
        const int idx = blk * 256 + threadIdx.x;
        live = idx < S * M * 4;
        const int t = (live ? idx : 0) >> 2;
        m = t % M;
        qi = t / M;
    }

    int Hs[MAXL], Ws[MAXL], st[MAXL];           // level geometry: wave-uniform, stays in SGPRs
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        if (l < L) {
            Hs[l] = (int)shapes[2 * l];
            Ws[l] = (int)shapes[2 * l + 1];
            st[l] = (int)lstart[l];
        } else {
            Hs[l] = Ws[l] = 1;
            st[l] = 0x7fffffff;
        }
    }
    const int LP = L * P;
    const float* pr = proj + ((int64_t)b * S + qi) * (M * LP * 3);
    const float* offp = pr + (int64_t)m * LP * 2;
    const float* lgp = pr + (int64_t)M * LP * 2 + (int64_t)m * LP;
    // reference point = centre of pixel qi in its own level, normalised (msdeformattn.py:141-153)
    int qW = Ws[0], qH = Hs[0], qs = 0;
#pragma unroll
    for (int l = 1; l < MAXL; ++l)
        if (qi >= st[l]) { qW = Ws[l]; qH = Hs[l]; qs = st[l]; }
    const int local = qi - qs;
    const int ry = local / qW, rx = local - ry * qW;
    const float ref_x = ((float)rx + 0.5f) / (float)qW;
    const float ref_y = ((float)ry + 0.5f) / (float)qH;

    // ---- this lane's share of the points: i = slot*4 + g ----
    float px[4], py[4], pw[4];
    float mx = -INFINITY;
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
        const int i = slot * 4 + g;
        const bool has = i < LP;
        const int ic = has ? i : 0;
        int W = Ws[0], H = Hs[0];               // level of point i: l = i / P
#pragma unroll
        for (int l = 1; l < MAXL; ++l)
            if (l < L && ic >= l * P) { W = Ws[l]; H = Hs[l]; }
        const float lg = has ? lgp[ic] : -INFINITY;
        const float lx = ref_x + offp[2 * ic] / (float)W;                    // ms_deform_attn.py:107-109
        const float ly = ref_y + offp[2 * ic + 1] / (float)H;
        px[slot] = lx * (float)W - 0.5f;                                     // w_im, cuh:290-291
        py[slot] = ly * (float)H - 0.5f;                                     // h_im
        pw[slot] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = quad_max(mx);                                                       // softmax over the L*P logits (:103)
    float den = 0.f;
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
        pw[slot] = expf(pw[slot] - mx);                                      // exp(-inf) = 0 for the padding slots
        den += pw[slot];
    }
    const float rden = 1.0f / quad_sum(den);

    // ---- gather: every lane walks all points, taking (x, y, w) from the owner lane of each.  Branch-free, in groups of
    // GP points: invalid taps keep a clamped (always readable) address and a zero weight, so the 2*GP loads of a group
    // are issued back to back and the L2 round trip is paid once per group instead of once per point (with the
    // reference's nested validity branches the compiler has to wait for every point's two loads before the next
    // point: 12 serial round trips per wave, measured 57 us; this form: see DESIGN.md).  Same products, same order:
    // a skipped tap adds 0 * v.
    constexpr int GP = 4;
    // buffer loads: the image's value planes behind one SGPR descriptor (M x S x 32 B = 1.6 MB), a 32-bit byte offset per tap
    // instead of 64-bit pointer arithmetic (four VALU instructions per load of a VALU-bound kernel)
    const uint64_t vbase = (uint64_t)(value + (int64_t)b * M * S * D);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(vbase >> 32)) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vbase)),
        0, M * S * D * 4, 0x00020000);
    const unsigned vlane = (unsigned)((m * S) * D + d4 * 4) * 4u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += GP) {
        if (i0 < LP) {
            float4 vt[GP], vbm[GP];
            float wt[GP], wb[GP];
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                const int i = i0 + j;
                wt[j] = wb[j] = 0.f;
                vt[j] = vbm[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < LP) {                       // uniform
                    int W = Ws[0], H = Hs[0], s0 = 0;   // uniform: scalar selects
#pragma unroll
                    for (int l = 1; l < MAXL; ++l)
                        if (l < L && i >= l * P) { W = Ws[l]; H = Hs[l]; s0 = st[l]; }
                    const float w_im = quad_bcast(px[i >> 2], i & 3);
                    const float h_im = quad_bcast(py[i >> 2], i & 3);
                    const float wgt = quad_bcast(pw[i >> 2], i & 3) * rden;
                    // The reference's outer test (-1 < h_im < H, -1 < w_im < W, cuh:293) is implied by its per-tap bounds
                    // (cuh:247-270): outside it no tap index lies in [0, H) x [0, W).  One unsigned compare per tap row /
                    // column, one v_med3 per clamp: the kernel is VALU-bound (~1650 VALU instructions per wave).
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, xw = (int)wf + cx;                // this lane's column
                    const float lh = h_im - hf, lw = w_im - wf;
                    const float wxw = (unsigned)xw < (unsigned)W ? (cx ? lw : 1.f - lw) * wgt : 0.f;
                    wt[j] = (unsigned)h_low < (unsigned)H ? (1.f - lh) * wxw : 0.f;
                    wb[j] = (unsigned)(h_low + 1) < (unsigned)H ? lh * wxw : 0.f;
                    const int xc = clamp0(xw, W - 1);
                    const int yt = clamp0(h_low, H - 1), yb = clamp0(h_low + 1, H - 1);
                    // (24-bit multiplies: full-rate v_mad_u32_u24; a 32-bit integer multiply is a quarter-rate instruction)
                    const unsigned col = vlane + (unsigned)(s0 + xc) * (unsigned)(D * 4);
                    vt[j] = ldv4(vrsrc, col + __umul24((unsigned)yt, (unsigned)(W * D * 4)));
                    vbm[j] = ldv4(vrsrc, col + __umul24((unsigned)yb, (unsigned)(W * D * 4)));
                }
            }
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                acc.x += wt[j] * vt[j].x + wb[j] * vbm[j].x;
                acc.y += wt[j] * vt[j].y + wb[j] * vbm[j].y;
                acc.z += wt[j] * vt[j].z + wb[j] * vbm[j].z;
                acc.w += wt[j] * vt[j].w + wb[j] * vbm[j].w;
            }
        }
    }
    // left + right column: lanes g and g^2
    acc.x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.x), 0x4E, 0xf, 0xf, true));
    acc.y += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.y), 0x4E, 0xf, 0xf, true));
    acc.z += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.z), 0x4E, 0xf, 0xf, true));
    acc.w += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.w), 0x4E, 0xf, 0xf, true));
    if (live && cx == 0) *reinterpret_cast<float4*>(out + (((int64_t)b * S + qi) * M + m) * D + d4 * 4) = acc;
}

// ---- cheap arithmetic for the gather prologues (round 3) ------------------------------------------------------------------
// The rec / fused kernels are VALU-issue bound (72 % VALU-busy at 4.1 cycles per instruction, rocprofv3 SQ counters), and a
// third of their instructions was the prologue's IEEE sequences: nine divisions (~11 instructions each), three expf (~12),
// an integer division (~25).  Divisors here are level widths / heights -- small positive integers, exactly representable,
// their reciprocals (v_rcp_f32 + one Newton step, once per wave and level) are within 1 ulp -- so
//     x / W  ->  q = x * rW;  q += fma(-q, W, x) * rW        (one Newton step on the quotient: within 1 ulp of the IEEE result)
//     exp(x) ->  v_exp_f32(x * log2 e)                      (x <= 0: relative error ~1e-6 from the argument's rounding)
//     n / W  ->  (int)((n + 0.5f) * rW)                      (exact for n < 2^20: the product is >= 0.5 / W away from an integer)
// The sampled locations move by ~1e-7 relative, far below the fp32 tolerances of SURVEY 8c; msda_enc_hm8_kernel (option
// MSDA_GENERIC = 2) keeps the IEEE forms.
struct LevelRcp {
    float rw[4], rh[4];
};
__device__ __forceinline__ float div_by(float x, float W, float rW) {
    const float q = x * rW;
    return fmaf(fmaf(-q, W, x), rW, q);
}
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float rcp_nr(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// ---- owner records (round 3) ------------------------------------------------------------------------------------------------
// What bounds msda_enc_hm8_kernel is VALU issue (723 VALU instructions per wave against 24 loads): after the quad has
// exchanged (x, y, weight) of a point, every one of its four lanes repeats the point's tap geometry -- floor, fractions,
// validity tests, clamps, the four bilinear weights, the address arithmetic: ~40 instructions per point and lane, of which
// only the column choice (cx) differs between the lanes.  Here the lane that OWNS a point (lane g of the quad: points g,
// g + 4, g + 8, as before) does that arithmetic once, for both columns, and leaves two 16-byte records per point in LDS:
//     rec[cx] = { byte offset of the top tap, byte offset of the bottom tap, weight of the top tap, weight of the bottom tap }
// (offsets inside the head's value plane, clamped; invalid taps carry weight 0).  The gather loop of every lane is then, per
// point: one ds_read_b128, two adds (its 16-byte channel half), two buffer loads, eight FMAs.  Products and summation order
// are those of msda_enc_hm8_kernel -- the results are bitwise identical (asserted by the tests).  Compile-time geometry
// (LC levels x PC points, LC * PC = 12 or 16 -> 3 or 4 slots per lane), query-major workgroups (64 consecutive queries of
// one head).  LDS: 4 waves x 16 quads x (LP records x 2 + padding) x 16 B; a quad's block is 16 B longer than its records
// so that neither the owners' ds_write_b128 nor the readers' ds_read_b128 meet a bank conflict.
template <int LC, int PC>
__global__ __launch_bounds__(256) void msda_enc_hm8_rec_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                               const int64_t* __restrict__ lstart, const float* __restrict__ proj,
                                                               float* __restrict__ out, int B, int S, int M) {
    constexpr int D = 8, LP = LC * PC, SLOTS = LP / 4;
    static_assert(PC == 4 && LP % 4 == 0 && LP <= 16, "a lane owns one point of every level: PC == 4");
    constexpr int QSTRIDE = LP * 2 + 1;                      // float4 per quad: LP x 2 records + one of padding
    __shared__ float4 recs[4 * 16 * QSTRIDE];
    const int b = blockIdx.x % B;
    const int blk = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid & 3, cx = g >> 1, d4 = g & 1;
    const int m = blk % M;                                   // uniform: the head of this workgroup
    const int q_raw = (blk / M) * 64 + (tid >> 2);
    const bool live = q_raw < S;                             // whole quads live or dead together
    const int qi = live ? q_raw : 0;

    int Hs[LC], Ws[LC], st[LC];                              // level geometry: wave-uniform, stays in SGPRs
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        Hs[l] = (int)shapes[2 * l];
        Ws[l] = (int)shapes[2 * l + 1];
        st[l] = (int)lstart[l];
    }
    LevelRcp lr;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        lr.rw[l] = rcp_nr((float)Ws[l]);
        lr.rh[l] = rcp_nr((float)Hs[l]);
    }
    const float* pr = proj + ((int64_t)b * S + qi) * (M * LP * 3);
    const float* offp = pr + (int64_t)m * LP * 2;
    const float* lgp = pr + (int64_t)M * LP * 2 + (int64_t)m * LP;
    // reference point = centre of pixel qi in its own level, normalised (msdeformattn.py:141-153)
    int qW = Ws[0], qH = Hs[0], qs = 0;
    float qrw = lr.rw[0], qrh = lr.rh[0];
#pragma unroll
    for (int l = 1; l < LC; ++l)
        if (qi >= st[l]) { qW = Ws[l]; qH = Hs[l]; qs = st[l]; qrw = lr.rw[l]; qrh = lr.rh[l]; }
    const int local = qi - qs;
    const int ry = (int)(((float)local + 0.5f) * qrw), rx = local - ry * qW;
    const float ref_x = div_by((float)rx + 0.5f, (float)qW, qrw);
    const float ref_y = div_by((float)ry + 0.5f, (float)qH, qrh);

    // ---- this lane's points: i = slot * 4 + g, level = slot ----
    float px[SLOTS], py[SLOTS], pw[SLOTS];
    float mx = -INFINITY;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int i = slot * 4 + g;
        const float2 off = *reinterpret_cast<const float2*>(offp + 2 * i);
        const float lg = lgp[i];
        const float lx = ref_x + div_by(off.x, (float)Ws[slot], lr.rw[slot]);   // ms_deform_attn.py:107-109
        const float ly = ref_y + div_by(off.y, (float)Hs[slot], lr.rh[slot]);
        px[slot] = lx * (float)Ws[slot] - 0.5f;                              // w_im, cuh:290-291
        py[slot] = ly * (float)Hs[slot] - 0.5f;                              // h_im
        pw[slot] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = quad_max(mx);                                                       // softmax over the L*P logits (:103)
    float den = 0.f;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        pw[slot] = exp_neg(pw[slot] - mx);
        den += pw[slot];
    }
    const float rden = rcp_nr(quad_sum(den));

    // ---- the owner's tap geometry -> LDS records (same arithmetic as msda_enc_hm8_kernel, per column) ----
    float4* qrec = recs + (wave * 16 + (lane >> 2)) * QSTRIDE;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int W = Ws[slot], H = Hs[slot], s0 = st[slot];
        const float w_im = px[slot], h_im = py[slot];
        const float wgt = pw[slot] * rden;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf;
        const float wx0 = (unsigned)w_low < (unsigned)W ? (1.f - lw) * wgt : 0.f;
        const float wx1 = (unsigned)(w_low + 1) < (unsigned)W ? lw * wgt : 0.f;
        const bool okt = (unsigned)h_low < (unsigned)H, okb = (unsigned)(h_low + 1) < (unsigned)H;
        const unsigned rowt = __umul24((unsigned)clamp0(h_low, H - 1), (unsigned)(W * D * 4));
        const unsigned rowb = __umul24((unsigned)clamp0(h_low + 1, H - 1), (unsigned)(W * D * 4));
        const unsigned c0 = (unsigned)(s0 + clamp0(w_low, W - 1)) * (unsigned)(D * 4);
        const unsigned c1 = (unsigned)(s0 + clamp0(w_low + 1, W - 1)) * (unsigned)(D * 4);
        const int i = slot * 4 + g;
        qrec[i * 2 + 0] = make_float4(__uint_as_float(c0 + rowt), __uint_as_float(c0 + rowb), okt ? (1.f - lh) * wx0 : 0.f, okb ? lh * wx0 : 0.f);
        qrec[i * 2 + 1] = make_float4(__uint_as_float(c1 + rowt), __uint_as_float(c1 + rowb), okt ? (1.f - lh) * wx1 : 0.f, okb ? lh * wx1 : 0.f);
    }
    // a quad's records are written and read by the same wave: the LDS executes a wave's operations in order, so no barrier --
    // only the compiler has to keep the order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- gather: the head's value plane behind one SGPR descriptor (S x 32 B), records from LDS ----
    const uint64_t vbase = (uint64_t)(value + ((int64_t)b * M + m) * S * D);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(vbase >> 32)) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vbase)),
        0, S * D * 4, 0x00020000);
    const unsigned lane_off = (unsigned)d4 * 16u;
    const float4* myrec = qrec + cx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int GP = 4;                                    // loads of GP points in flight together (as msda_enc_hm8_kernel)
#pragma unroll
    for (int i0 = 0; i0 < LP; i0 += GP) {
        float4 vt[GP], vbm[GP], r[GP];
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            r[j] = myrec[(i0 + j) * 2];
            vt[j] = ldv4(vrsrc, __float_as_uint(r[j].x) + lane_off);
            vbm[j] = ldv4(vrsrc, __float_as_uint(r[j].y) + lane_off);
        }
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            acc.x += r[j].z * vt[j].x + r[j].w * vbm[j].x;
            acc.y += r[j].z * vt[j].y + r[j].w * vbm[j].y;
            acc.z += r[j].z * vt[j].z + r[j].w * vbm[j].z;
            acc.w += r[j].z * vt[j].w + r[j].w * vbm[j].w;
        }
    }
    // left + right column: lanes g and g^2
    acc.x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.x), 0x4E, 0xf, 0xf, true));
    acc.y += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.y), 0x4E, 0xf, 0xf, true));
    acc.z += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.z), 0x4E, 0xf, 0xf, true));
    acc.w += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.w), 0x4E, 0xf, 0xf, true));
    if (live && cx == 0) *reinterpret_cast<float4*>(out + (((int64_t)b * S + qi) * M + m) * D + d4 * 4) = acc;
}

// ---- sampling projection fused in (round 3) -------------------------------------------------------------------------------
// The offsets / attention-logit projection of a layer, [sampling_offsets | attention_weights](src + pos)
// (ops/modules/ms_deform_attn.py:99-101, query = src + pos msdeformattn.py:124), used to be written by the previous layer's
// token kernel and read back here: 58 MB each way per layer at B = 8 (2.3x the kernel's other traffic), with an HBM round
// trip at the head of every wave's dependency chain.  Here the workgroup computes it for its own 64 queries and ONE head:
// a wave owns 16 tokens, x = src + pos in MFMA layout L (lane (token lj, quarter lq): features fb*16 + lq*4 + c, the layout
// of enc_block_kernel), the head's 36 weight rows -- 24 offset rows, 12 logit rows, zero-padded to three 16-row blocks -- are
// the A operand (pre-packed per head in fragment order: msm_msda_pack_proj), 48 v_mfma_f32_16x16x4_f32 per wave.  Same
// operands and the same k order as enc_block_kernel's rowblock_mma, so every projected value is bitwise the one that kernel
// wrote.  The result goes through the wave's LDS region ([token][52 floats]; the records of the owner scheme overwrite it
// once the owners have read their three points) and the rest is msda_enc_hm8_rec_kernel.
template <int LC, int PC>
__global__ __launch_bounds__(256) void msda_enc_hm8_fused_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                                 const int64_t* __restrict__ lstart, const float* __restrict__ src,
                                                                 const float* __restrict__ pos, const float4* __restrict__ wpack,
                                                                 const float* __restrict__ bpack, float* __restrict__ out, int B,
                                                                 int S, int M) {
    constexpr int D = 8, LP = LC * PC, SLOTS = LP / 4, EC = 64;
    static_assert(PC == 4 && LP == 12, "three 16-row blocks hold the 36 projection rows of a head");
    constexpr int QSTRIDE = LP * 2 + 1;                      // float4 per quad: LP x 2 records + one of padding
    constexpr int TSTRIDE = 52;                              // floats per token row of the projection tile (48 + 4: conflict-free b128 stores)
    constexpr int WF4 = 3 * 4 * 64;                          // float4 of a head's weight fragments (12 KiB)
    // One LDS region, two uses: phase A = [the head's weight fragments | four waves' projection tiles], afterwards = the four
    // waves' owner records.  The sizes agree to the byte: 768 + 4 * 16 * 13 float4 = 4 * 16 * 25 float4 = 25 600 B.
    static_assert(WF4 + 4 * 16 * TSTRIDE / 4 == 4 * 16 * QSTRIDE, "phase-A layout fills the record region exactly");
    __shared__ float4 recs[4 * 16 * QSTRIDE];
    const int b = blockIdx.x % B;
    const int blk = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blk % M;                                   // uniform: the head of this workgroup
    const int q0 = (blk / M) * 64 + wave * 16;               // first query of this wave

    // the wave's 16 tokens first (x = src + pos, layout L): their latency hides behind the weight copy
    const int lj = lane & 15, lq = lane >> 4;
    float x[4][4];
    {
        const int tk = min(q0 + lj, S - 1);
        float4 a[4], p[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            a[fb] = *reinterpret_cast<const float4*>(src + ((int64_t)b * S + tk) * EC + fb * 16 + lq * 4);
            p[fb] = *reinterpret_cast<const float4*>(pos + (int64_t)tk * EC + fb * 16 + lq * 4);
        }
        // the head's weight fragments: ONE copy per workgroup (as four per-wave reads they were 12 KiB per wave through L1)
#pragma unroll
        for (int i = 0; i < WF4 / 256; ++i) recs[i * 256 + tid] = wpack[m * WF4 + i * 256 + tid];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            x[fb][0] = a[fb].x + p[fb].x; x[fb][1] = a[fb].y + p[fb].y; x[fb][2] = a[fb].z + p[fb].z; x[fb][3] = a[fb].w + p[fb].w;
        }
    }
    int Hs[LC], Ws[LC], st[LC];                              // level geometry: wave-uniform, stays in SGPRs
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        Hs[l] = (int)shapes[2 * l];
        Ws[l] = (int)shapes[2 * l + 1];
        st[l] = (int)lstart[l];
    }
    LevelRcp lr;
#pragma unroll
    for (int l = 0; l < LC; ++l) {
        lr.rw[l] = rcp_nr((float)Ws[l]);
        lr.rh[l] = rcp_nr((float)Hs[l]);
    }
    __syncthreads();
    // ---- phase A: the head's projection of the wave's 16 tokens ----
    float* tile = reinterpret_cast<float*>(recs + WF4) + wave * 16 * TSTRIDE;
    {
        float* trow = tile + lj * TSTRIDE + lq * 4;
        f32x4 d[3];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const float4 bb = *reinterpret_cast<const float4*>(bpack + m * 48 + rb * 16 + lq * 4);
            d[rb] = f32x4{bb.x, bb.y, bb.z, bb.w};
        }
        // per row block one accumulator chain from the bias, k order (c, fb): enc_block_kernel's rowblock_mma; the three
        // chains are independent and interleave
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const float4 w = recs[(rb * 4 + fb) * 64 + lane];
                    d[rb] = mfma16(c == 0 ? w.x : (c == 1 ? w.y : (c == 2 ? w.z : w.w)), x[fb][c], d[rb]);
                }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
            *reinterpret_cast<float4*>(trow + rb * 16) = make_float4(d[rb][0], d[rb][1], d[rb][2], d[rb][3]);   // features rb*16 + lq*4 .. + 3 of token lj
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase B: gather mapping -- quad = query, g = (column cx, channel half d4) ----
    const int g = tid & 3, cx = g >> 1, d4 = g & 1;
    const int quad = lane >> 2;
    const int q_raw = q0 + quad;
    const bool live = q_raw < S;                             // whole quads live or dead together
    const int qi = live ? q_raw : S - 1;                     // (the projection tile clamps the same way)
    // reference point = centre of pixel qi in its own level, normalised (msdeformattn.py:141-153)
    int qW = Ws[0], qH = Hs[0], qs = 0;
    float qrw = lr.rw[0], qrh = lr.rh[0];
#pragma unroll
    for (int l = 1; l < LC; ++l)
        if (qi >= st[l]) { qW = Ws[l]; qH = Hs[l]; qs = st[l]; qrw = lr.rw[l]; qrh = lr.rh[l]; }
    const int local = qi - qs;
    const int ry = (int)(((float)local + 0.5f) * qrw), rx = local - ry * qW;
    const float ref_x = div_by((float)rx + 0.5f, (float)qW, qrw);
    const float ref_y = div_by((float)ry + 0.5f, (float)qH, qrh);

    // ---- this lane's points: i = slot * 4 + g, level = slot; offsets at features 2 i, 2 i + 1, logit at 24 + i ----
    const float* tq = tile + quad * TSTRIDE;
    float px[SLOTS], py[SLOTS], pw[SLOTS];
    float mx = -INFINITY;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int i = slot * 4 + g;
        const float2 off = *reinterpret_cast<const float2*>(tq + 2 * i);
        const float lg = tq[2 * LP + i];
        const float lx = ref_x + div_by(off.x, (float)Ws[slot], lr.rw[slot]);   // ms_deform_attn.py:107-109
        const float ly = ref_y + div_by(off.y, (float)Hs[slot], lr.rh[slot]);
        px[slot] = lx * (float)Ws[slot] - 0.5f;                              // w_im, cuh:290-291
        py[slot] = ly * (float)Hs[slot] - 0.5f;                              // h_im
        pw[slot] = lg;
        mx = fmaxf(mx, lg);
    }
    mx = quad_max(mx);                                                       // softmax over the L*P logits (:103)
    float den = 0.f;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        pw[slot] = exp_neg(pw[slot] - mx);
        den += pw[slot];
    }
    const float rden = rcp_nr(quad_sum(den));
    // every wave has read the weights and its projection tile: the records may overwrite the region
    __syncthreads();

    float4* qrec = recs + (wave * 16 + quad) * QSTRIDE;
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
        const int W = Ws[slot], H = Hs[slot], s0 = st[slot];
        const float w_im = px[slot], h_im = py[slot];
        const float wgt = pw[slot] * rden;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf;
        const float wx0 = (unsigned)w_low < (unsigned)W ? (1.f - lw) * wgt : 0.f;
        const float wx1 = (unsigned)(w_low + 1) < (unsigned)W ? lw * wgt : 0.f;
        const bool okt = (unsigned)h_low < (unsigned)H, okb = (unsigned)(h_low + 1) < (unsigned)H;
        const unsigned rowt = __umul24((unsigned)clamp0(h_low, H - 1), (unsigned)(W * D * 4));
        const unsigned rowb = __umul24((unsigned)clamp0(h_low + 1, H - 1), (unsigned)(W * D * 4));
        const unsigned c0 = (unsigned)(s0 + clamp0(w_low, W - 1)) * (unsigned)(D * 4);
        const unsigned c1 = (unsigned)(s0 + clamp0(w_low + 1, W - 1)) * (unsigned)(D * 4);
        const int i = slot * 4 + g;
        qrec[i * 2 + 0] = make_float4(__uint_as_float(c0 + rowt), __uint_as_float(c0 + rowb), okt ? (1.f - lh) * wx0 : 0.f, okb ? lh * wx0 : 0.f);
        qrec[i * 2 + 1] = make_float4(__uint_as_float(c1 + rowt), __uint_as_float(c1 + rowb), okt ? (1.f - lh) * wx1 : 0.f, okb ? lh * wx1 : 0.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint64_t vbase = (uint64_t)(value + ((int64_t)b * M + m) * S * D);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(vbase >> 32)) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vbase)),
        0, S * D * 4, 0x00020000);
    const unsigned lane_off = (unsigned)d4 * 16u;
    const float4* myrec = qrec + cx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int GP = 4;
#pragma unroll
    for (int i0 = 0; i0 < LP; i0 += GP) {
        float4 vt[GP], vbm[GP], r[GP];
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            r[j] = myrec[(i0 + j) * 2];
            vt[j] = ldv4(vrsrc, __float_as_uint(r[j].x) + lane_off);
            vbm[j] = ldv4(vrsrc, __float_as_uint(r[j].y) + lane_off);
        }
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            acc.x += r[j].z * vt[j].x + r[j].w * vbm[j].x;
            acc.y += r[j].z * vt[j].y + r[j].w * vbm[j].y;
            acc.z += r[j].z * vt[j].z + r[j].w * vbm[j].z;
            acc.w += r[j].z * vt[j].w + r[j].w * vbm[j].w;
        }
    }
    acc.x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.x), 0x4E, 0xf, 0xf, true));
    acc.y += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.y), 0x4E, 0xf, 0xf, true));
    acc.z += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.z), 0x4E, 0xf, 0xf, true));
    acc.w += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc.w), 0x4E, 0xf, 0xf, true));
    if (live && cx == 0) *reinterpret_cast<float4*>(out + (((int64_t)b * S + qi) * M + m) * D + d4 * 4) = acc;
}

// wpack[((m*3 + rb)*4 + fb)*64 + lq*16 + lj] = Wm[rb*16 + lj][fb*16 + lq*4 .. +3], bpack[m*48 + r] = bias of row r, where the 48
// rows of head m are its 2*LP offset rows ((L, P, 2) order), its LP logit rows and zeros (ms_deform_attn.py:73-74 layouts)
__global__ __launch_bounds__(256) void msda_pack_proj_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                             float4* __restrict__ wpack, float* __restrict__ bpack, int M, int LP) {
    const int total = M * 3 * 4 * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, fb = (i >> 6) & 3, rb = (i >> 8) % 3, m = i / 768;
        const int lj = lane & 15, lq = lane >> 4;
        const int r = rb * 16 + lj;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int row = -1;
        if (r < 2 * LP) row = m * 2 * LP + r;
        else if (r < 3 * LP) row = M * 2 * LP + m * LP + (r - 2 * LP);
        if (row >= 0) v = *reinterpret_cast<const float4*>(w + (int64_t)row * 64 + fb * 16 + lq * 4);
        wpack[i] = v;
        if (fb == 0 && lq == 0) bpack[m * 48 + r] = row >= 0 ? bias[row] : 0.f;
    }
}

// value [B][S][M][D] (token-major, what a value_proj GEMM writes) -> [B][M][S][D] (head-major)
__global__ __launch_bounds__(256) void value_to_hm_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total4,
                                                          int S, int M, int D4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int d4 = (int)(i % D4);
        int64_t r = i / D4;
        const int m = (int)(r % M);
        r /= M;
        const int t = (int)(r % S);
        const int64_t b = r / S;
        reinterpret_cast<float4*>(out)[((b * M + m) * S + t) * D4 + d4] = reinterpret_cast<const float4*>(in)[i];
    }
}

// ---- backward (training): reference col2im kernels, cuh:306-925 (bilinear helper cuh:92-239) -------------------
// Same lane mapping as the forward: a lane owns V channels of one (image, query, head), so the 4 corner reads
// are 16-byte loads and the scatter into grad_value is V hardware fp32 atomics per corner
// (global_atomic_add_f32; the reference also accumulates grad_value with atomicAdd, cuh:130-145, so the
// summation order is not fixed there either).  grad_attn_weight / grad_sampling_loc belong to exactly one
// (query, head, level, point): their channel sum is a shuffle butterfly over the head's D/V lanes followed by
// one plain store (the reference's shared-memory reductions, cuh:368-386).  When D/V is not a power of two the
// butterfly is replaced by atomics into zero-initialised outputs.
template <int V, bool POW2>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const float* __restrict__ loc,
                                                       const float* __restrict__ wgt, const float* __restrict__ gout,
                                                       float* __restrict__ gvalue, float* __restrict__ gloc,
                                                       float* __restrict__ gwgt, int B, int S, int M, int D, int L, int Lq,
                                                       int P) {
    const int D4 = D / V;
    const int per_img = Lq * M * D4;
    const int b = blockIdx.x % B;
    const int idx = (blockIdx.x / B) * 256 + threadIdx.x;
    // POW2: the D4 lanes of a head sit in one wave (D4 <= 16 divides 64) and per_img is a multiple of D4, so a
    // head is never split by the bound below; inactive lanes still take part in the shuffles with zeros.
    const bool live = idx < per_img;
    const int cidx = live ? idx : 0;
    const int d4 = cidx % D4;
    const int t = cidx / D4;
    const int m = t % M;
    const int qi = t / M;
    if (!POW2 && !live) return;

    const int64_t pix_stride = (int64_t)M * D;
    const int64_t voff = (int64_t)b * S * pix_stride + m * D + d4 * V;
    const Vec<V> go = ldv<V>(gout + (((int64_t)b * Lq + qi) * M + m) * D + d4 * V);
    const int64_t base = (((int64_t)b * Lq + qi) * M + m) * L * P;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int64_t lvl = voff + (int64_t)lstart[l] * pix_stride;
        for (int p = 0; p < P; ++p) {
            const int i = l * P + p;
            const float loc_x = loc[(base + i) * 2], loc_y = loc[(base + i) * 2 + 1], aw = wgt[base + i];
            const float h_im = loc_y * (float)H - 0.5f;
            const float w_im = loc_x * (float)W - 0.5f;
            float g_w = 0.f, g_x = 0.f, g_y = 0.f;
            if (live && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {   // cuh:352
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1;
                const bool ok3 = h_high <= H - 1 && w_low >= 0, ok4 = h_high <= H - 1 && w_high <= W - 1;
                const int64_t o1 = lvl + ((int64_t)h_low * W + w_low) * pix_stride, o2 = o1 + pix_stride;
                const int64_t o3 = o1 + (int64_t)W * pix_stride, o4 = o3 + pix_stride;
                Vec<V> v1, v2, v3, v4;
#pragma unroll
                for (int c = 0; c < V; ++c) v1.e[c] = v2.e[c] = v3.e[c] = v4.e[c] = 0.f;
                if (ok1) v1 = ldv<V>(value + o1);
                if (ok2) v2 = ldv<V>(value + o2);
                if (ok3) v3 = ldv<V>(value + o3);
                if (ok4) v4 = ldv<V>(value + o4);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    const float tg = go.e[c] * aw;                                           // top_grad_value, cuh:117
                    if (ok1) unsafeAtomicAdd(gvalue + o1 + c, w1 * tg);                       // cuh:128-160
                    if (ok2) unsafeAtomicAdd(gvalue + o2 + c, w2 * tg);
                    if (ok3) unsafeAtomicAdd(gvalue + o3 + c, w3 * tg);
                    if (ok4) unsafeAtomicAdd(gvalue + o4 + c, w4 * tg);
                    g_w += go.e[c] * (w1 * v1.e[c] + w2 * v2.e[c] + w3 * v3.e[c] + w4 * v4.e[c]);   // cuh:164
                    g_x += tg * (-hh * v1.e[c] + hh * v2.e[c] - lh * v3.e[c] + lh * v4.e[c]);      // grad_w_weight
                    g_y += tg * (-hw * v1.e[c] - lw * v2.e[c] + hw * v3.e[c] + lw * v4.e[c]);      // grad_h_weight
                }
                g_x *= (float)W;                                                               // cuh:165-166
                g_y *= (float)H;
            }
            if constexpr (POW2) {
                for (int o = D4 >> 1; o > 0; o >>= 1) {
                    g_w += __shfl_xor(g_w, o, 64);
                    g_x += __shfl_xor(g_x, o, 64);
                    g_y += __shfl_xor(g_y, o, 64);
                }
                if (live && d4 == 0) {
                    gwgt[base + i] = g_w;
                    gloc[(base + i) * 2] = g_x;
                    gloc[(base + i) * 2 + 1] = g_y;
                }
            } else {
                unsafeAtomicAdd(gwgt + base + i, g_w);
                unsafeAtomicAdd(gloc + (base + i) * 2, g_x);
                unsafeAtomicAdd(gloc + (base + i) * 2 + 1, g_y);
            }
        }
    }
}

static int msda_common_checks(const char* name, const void* value, const void* out, int B, int S, int M, int D, int L,
                              int Lq, int P) {
    MSM_REQUIRE(B > 0 && S > 0 && M > 0 && Lq > 0 && P > 0, "%s: bad sizes", name);
    MSM_REQUIRE(D > 0 && D <= 64, "%s: D=%d must be in 1..64", name, D);
    MSM_REQUIRE(L > 0 && L <= MAXL, "%s: L=%d must be <= %d", name, L, MAXL);
    MSM_REQUIRE((((uintptr_t)value) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "%s: value/out must be 16-byte aligned", name);
    return MSM_OK;
}

// shape-generic kernels (msda_generic.hip): any D, any L
int msda_any_fwd_f32(const float* value, const int64_t* shapes, const int64_t* lstart, const float* loc, const float* wgt,
                     float* out, int B, int S, int M, int D, int L, int Lq, int P, void* stream);
int msda_any_bwd_f32(const float* value, const int64_t* shapes, const int64_t* lstart, const float* loc, const float* wgt,
                     const float* gout, float* gvalue, float* gloc, float* gwgt, int B, int S, int M, int D, int L, int Lq,
                     int P, void* stream);

}  // namespace msm

using namespace msm;

extern "C" int msm_msdeform_attn_fwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                     const float* sampling_loc, const float* attn_weight, float* out, int B, int S,
                                     int M, int D, int L, int Lq, int P, void* stream) {
    MSM_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                "msm_msdeform_attn_fwd: null pointer");
    if (D > 64 || L > MAXL)       // outside the tuned kernels' range (the reference's gradcheck sizes, ops/test.py:84)
        return msda_any_fwd_f32(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, B, S, M, D, L, Lq, P, stream);
    int rc = msda_common_checks("msm_msdeform_attn_fwd", value, out, B, S, M, D, L, Lq, P);
    if (rc != MSM_OK) return rc;
    const int V = (D % 4 == 0) ? 4 : 1;
    const int64_t per_img = (int64_t)Lq * M * (D / V);
    dim3 grid((unsigned)(((per_img + 255) / 256) * B)), block(256);
    if (V == 4)
        hipLaunchKernelGGL((msda_kernel<false, 4>), grid, block, 0, (hipStream_t)stream, value, spatial_shapes,
                           level_start_index, sampling_loc, attn_weight, (const float*)nullptr, out, B, S, M, D, L, Lq, P);
    else
        hipLaunchKernelGGL((msda_kernel<false, 1>), grid, block, 0, (hipStream_t)stream, value, spatial_shapes,
                           level_start_index, sampling_loc, attn_weight, (const float*)nullptr, out, B, S, M, D, L, Lq, P);
    MSM_CHECK_LAUNCH("msm_msdeform_attn_fwd");
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_enc_fwd(const float* value, const int64_t* spatial_shapes,
                                         const int64_t* level_start_index, const float* proj, float* out, int B, int S,
                                         int M, int D, int L, int P, void* stream) {
    MSM_REQUIRE(value && spatial_shapes && level_start_index && proj && out, "msm_msdeform_attn_enc_fwd: null pointer");
    int rc = msda_common_checks("msm_msdeform_attn_enc_fwd", value, out, B, S, M, D, L, S, P);
    if (rc != MSM_OK) return rc;
    const int V = (D % 4 == 0) ? 4 : 1;
    const int64_t per_img = (int64_t)S * M * (D / V);
    dim3 grid((unsigned)(((per_img + 255) / 256) * B)), block(256);
    if (V == 4)
        hipLaunchKernelGGL((msda_kernel<true, 4>), grid, block, 0, (hipStream_t)stream, value, spatial_shapes,
                           level_start_index, (const float*)nullptr, (const float*)nullptr, proj, out, B, S, M, D, L, S, P);
    else
        hipLaunchKernelGGL((msda_kernel<true, 1>), grid, block, 0, (hipStream_t)stream, value, spatial_shapes,
                           level_start_index, (const float*)nullptr, (const float*)nullptr, proj, out, B, S, M, D, L, S, P);
    MSM_CHECK_LAUNCH("msm_msdeform_attn_enc_fwd");
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_bwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                     const float* sampling_loc, const float* attn_weight, const float* grad_output,
                                     float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int B, int S,
                                     int M, int D, int L, int Lq, int P, void* stream) {
    MSM_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_output && grad_value &&
                    grad_sampling_loc && grad_attn_weight,
                "msm_msdeform_attn_bwd: null pointer");
    if (D > 64 || L > MAXL)
        return msda_any_bwd_f32(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, grad_value,
                                grad_sampling_loc, grad_attn_weight, B, S, M, D, L, Lq, P, stream);
    int rc = msda_common_checks("msm_msdeform_attn_bwd", value, grad_output, B, S, M, D, L, Lq, P);
    if (rc != MSM_OK) return rc;
    MSM_REQUIRE((((uintptr_t)grad_value) & 15) == 0, "msm_msdeform_attn_bwd: grad_value must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int V = (D % 4 == 0) ? 4 : 1;
    const int D4 = D / V;
    const bool pow2 = (D4 & (D4 - 1)) == 0;
    MSM_CHECK_HIP(hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)B * S * M * D, st));
    if (!pow2) {
        MSM_CHECK_HIP(hipMemsetAsync(grad_sampling_loc, 0, sizeof(float) * (size_t)B * Lq * M * L * P * 2, st));
        MSM_CHECK_HIP(hipMemsetAsync(grad_attn_weight, 0, sizeof(float) * (size_t)B * Lq * M * L * P, st));
    }
    const int64_t per_img = (int64_t)Lq * M * D4;
    dim3 grid((unsigned)(((per_img + 255) / 256) * B)), block(256);
#define MSDA_BWD(VV, PP)                                                                                              \
    hipLaunchKernelGGL((msda_bwd_kernel<VV, PP>), grid, block, 0, st, value, spatial_shapes, level_start_index,        \
                       sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc, grad_attn_weight, B, S, M, \
                       D, L, Lq, P)
    if (V == 4 && pow2) MSDA_BWD(4, true);
    else if (V == 4) MSDA_BWD(4, false);
    else if (pow2) MSDA_BWD(1, true);
    else MSDA_BWD(1, false);
#undef MSDA_BWD
    MSM_CHECK_LAUNCH("msm_msdeform_attn_bwd");
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_enc_hm_fwd(const float* value_hm, const int64_t* spatial_shapes,
                                            const int64_t* level_start_index, const float* proj, float* out, int B, int S,
                                            int M, int D, int L, int P, void* stream) {
    MSM_REQUIRE(value_hm && spatial_shapes && level_start_index && proj && out, "msm_msdeform_attn_enc_hm_fwd: null pointer");
    int rc = msda_common_checks("msm_msdeform_attn_enc_hm_fwd", value_hm, out, B, S, M, D, L, S, P);
    if (rc != MSM_OK) return rc;
    const int V = (D % 4 == 0) ? 4 : 1;
    const int G = 2 * (D / V);
    MSM_REQUIRE(256 % G == 0, "msm_msdeform_attn_enc_hm_fwd: D=%d: 2*D/%d lanes per head must divide 256", D, V);
    MSM_REQUIRE((int64_t)S * D < ((int64_t)1 << 31), "msm_msdeform_attn_enc_hm_fwd: S*D must fit 31 bits");
    const int64_t per_img = (int64_t)S * M * G;
    dim3 grid((unsigned)(((per_img + 255) / 256) * B)), block(256);
    if (D == 8 && L * P <= 16 && (int64_t)M * S * D * 4 < ((int64_t)1 << 31) && opt(MSM_OPT_MSDA_GENERIC) == 3)
        hipLaunchKernelGGL((msda_enc_hm8_kernel<false, 0, 0>), grid, block, 0, (hipStream_t)stream, value_hm, spatial_shapes,
                           level_start_index, proj, out, B, S, M, L, P);
    else if (D == 8 && L == 3 && P == 4 && (int64_t)S * D * 4 < ((int64_t)1 << 31) && opt(MSM_OPT_MSDA_GENERIC) == MSM_OPT_AUTO)
        // default (round 3): tap geometry once per point by its owner lane, records through LDS
        hipLaunchKernelGGL((msda_enc_hm8_rec_kernel<3, 4>), dim3((unsigned)(cdiv(S, 64) * M * B)), block, 0, (hipStream_t)stream, value_hm,
                           spatial_shapes, level_start_index, proj, out, B, S, M);
    else if (D == 8 && L == 3 && P == 4 && (int64_t)M * S * D * 4 < ((int64_t)1 << 31) && opt(MSM_OPT_MSDA_GENERIC) != 1)
        hipLaunchKernelGGL((msda_enc_hm8_kernel<true, 3, 4>), dim3((unsigned)(cdiv(S, 64) * M * B)), block, 0, (hipStream_t)stream, value_hm,
                           spatial_shapes, level_start_index, proj, out, B, S, M, L, P);
    else if (D == 8 && L * P <= 16 && (int64_t)M * S * D * 4 < ((int64_t)1 << 31) && opt(MSM_OPT_MSDA_GENERIC) != 1)
        hipLaunchKernelGGL((msda_enc_hm8_kernel<true, 0, 0>), dim3((unsigned)(cdiv(S, 64) * M * B)), block, 0, (hipStream_t)stream, value_hm,
                           spatial_shapes, level_start_index, proj, out, B, S, M, L, P);
    else if (V == 4)
        hipLaunchKernelGGL((msda_enc_hm_kernel<4>), grid, block, 0, (hipStream_t)stream, value_hm, spatial_shapes,
                           level_start_index, proj, out, B, S, M, D, L, P);
    else
        hipLaunchKernelGGL((msda_enc_hm_kernel<1>), grid, block, 0, (hipStream_t)stream, value_hm, spatial_shapes,
                           level_start_index, proj, out, B, S, M, D, L, P);
    MSM_CHECK_LAUNCH("msm_msdeform_attn_enc_hm_fwd");
    return MSM_OK;
}

extern "C" int msm_msda_pack_proj(const float* w, const float* bias, float* wpack, float* bpack, int M, int L, int P, void* stream) {
    MSM_REQUIRE(w && bias && wpack && bpack, "msm_msda_pack_proj: null pointer");
    MSM_REQUIRE(M > 0 && L * P == 12, "msm_msda_pack_proj: L*P=%d, the fused gather takes 12 sampling points per head", L * P);
    MSM_REQUIRE(((((uintptr_t)w) | ((uintptr_t)wpack)) & 15) == 0, "msm_msda_pack_proj: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(msda_pack_proj_kernel, dim3((unsigned)cdiv(M * 768, 256)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       reinterpret_cast<float4*>(wpack), bpack, M, L * P);
    MSM_CHECK_LAUNCH("msm_msda_pack_proj");
    return MSM_OK;
}

extern "C" int msm_msdeform_attn_enc_fused_fwd(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                               const float* src, const float* pos, const float* wpack, const float* bpack, float* out,
                                               int B, int S, int M, int D, int L, int P, void* stream) {
    MSM_REQUIRE(value_hm && spatial_shapes && level_start_index && src && pos && wpack && bpack && out,
                "msm_msdeform_attn_enc_fused_fwd: null pointer");
    MSM_REQUIRE(B > 0 && S > 0 && M > 0, "msm_msdeform_attn_enc_fused_fwd: bad sizes");
    MSM_REQUIRE(D == 8 && M * D == 64 && L == 3 && P == 4,
                "msm_msdeform_attn_enc_fused_fwd: only the pixel decoder's geometry (64 channels = 8 heads x 8, 3 levels x 4 points); got M=%d D=%d L=%d P=%d",
                M, D, L, P);
    MSM_REQUIRE((int64_t)S * D * 4 < ((int64_t)1 << 31), "msm_msdeform_attn_enc_fused_fwd: S too large");
    MSM_REQUIRE(((((uintptr_t)value_hm) | ((uintptr_t)src) | ((uintptr_t)pos) | ((uintptr_t)wpack) | ((uintptr_t)bpack) | ((uintptr_t)out)) & 15) == 0,
                "msm_msdeform_attn_enc_fused_fwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL((msda_enc_hm8_fused_kernel<3, 4>), dim3((unsigned)(cdiv(S, 64) * M * B)), dim3(256), 0, (hipStream_t)stream, value_hm,
                       spatial_shapes, level_start_index, src, pos, reinterpret_cast<const float4*>(wpack), bpack, out, B, S, M);
    MSM_CHECK_LAUNCH("msm_msdeform_attn_enc_fused_fwd");
    return MSM_OK;
}

extern "C" int msm_value_to_head_major_f32(const float* value, float* value_hm, int B, int S, int M, int D, void* stream) {
    MSM_REQUIRE(value && value_hm && value != value_hm, "msm_value_to_head_major_f32: null or aliased pointer");
    MSM_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && D % 4 == 0, "msm_value_to_head_major_f32: D=%d must be a multiple of 4", D);
    MSM_REQUIRE(((((uintptr_t)value) | ((uintptr_t)value_hm)) & 15) == 0, "msm_value_to_head_major_f32: pointers must be 16-byte aligned");
    const int64_t total4 = (int64_t)B * S * M * (D / 4);
    hipLaunchKernelGGL(value_to_hm_kernel, dim3((unsigned)min((int64_t)2048, (total4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, value, value_hm, total4, S, M, D / 4);
    MSM_CHECK_LAUNCH("msm_value_to_head_major_f32");
    return MSM_OK;
}
