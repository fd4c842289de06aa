// Backward of the multi-head hypersphere attention core (see include/msm_hip.h: msm_hypersphere_attn_bwd).
//
// Reference: hypersphere_attention, attention_util.py:64-82 under torch autograd (training step,
// MSMFormer/tabletop_train_net_pretrained.py:209-246):
//     q^ = q / max(|q|, eps), k^ likewise (per head, 32-d);  s = kappa q^.k^ (+ -inf where masked);  p = softmax_j(s);
//     o = sum_j p_j v_j;  out = o / max(|o|, eps)
// and therefore, with g = d loss / d out:
//     dO  = (g - out (out.g)) / |o|               D = dO.o = sum_j p_j dP_j,  dP_j = dO.v_j
//     dS_j = p_j (dP_j - D)                       dV_j = sum_i p_ij dO_i
//     dq^ = kappa sum_j dS_j k^_j                 dk^_j = kappa sum_i dS_ij q^_i
//     dq = (dq^ - q^ (q^.dq^)) / |q|              dk_j = (dk^_j - k^_j (k^_j.dk^_j)) / |k_j|
// Nothing of size Lq x S is stored: the probabilities are recomputed from the logits (bounded by kappa, so
// p = exp(s - kappa) / l needs no running maximum, as in the forward kernels).  Two launches:
//   * hs_attn_bwd_q_kernel: a workgroup owns BQ query rows of one (image, head), walks all keys twice (l and o, then dS)
//     and leaves dq plus, per row, {l, D, dO[32], q^[32]} in the workspace;
//   * hs_attn_bwd_kv_kernel: a thread owns one key of one (image, head), walks the query rows (their workspace records are
//     wave-uniform loads) and leaves dk, dv -- no atomics, a fixed summation order.
// The training path is a "next" row of the scope table: these kernels are written for correctness and coalescing, not
// for the MFMA roofline (0.5 GFLOP per image and layer at 640x480).
#include "common.h"

namespace msm {

constexpr int BWD_HD = 32;
constexpr int BWD_BQ = 4;          // query rows per workgroup of the q kernel
constexpr int BWD_REC = 2 + 2 * BWD_HD;   // workspace floats per (image, head, query): l, D, dO[32], q^[32]

__device__ __forceinline__ float block_sum_256(float v, float* red) {   // all 256 threads get the sum; red: 4 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ void load32(const float* __restrict__ p, float (&x)[BWD_HD]) {
#pragma unroll
    for (int c = 0; c < BWD_HD; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + c);
        x[c] = t.x; x[c + 1] = t.y; x[c + 2] = t.z; x[c + 3] = t.w;
    }
}
__device__ __forceinline__ float dot32(const float (&a)[BWD_HD], const float (&b)[BWD_HD]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < BWD_HD; ++c) s = fmaf(a[c], b[c], s);
    return s;
}

__global__ __launch_bounds__(256) void hs_attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const uint8_t* __restrict__ masked, const int32_t* __restrict__ row_any,
                                                            const float* __restrict__ gout, float* __restrict__ gq, float* __restrict__ ws,
                                                            int Lq, int S, int heads, int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb,
                                                            int64_t ldv, int64_t v_sb, float kappa) {
    __shared__ float red[4];
    const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * BWD_BQ;
    const int E = heads * BWD_HD;
    float qh[BWD_BQ][BWD_HD], gr[BWD_BQ][BWD_HD], qn[BWD_BQ];
    bool use_mask[BWD_BQ];
#pragma unroll
    for (int r = 0; r < BWD_BQ; ++r) {
        const int i = min(i0 + r, Lq - 1);
        load32(q + (int64_t)b * q_sb + (int64_t)i * ldq + h * BWD_HD, qh[r]);
        load32(gout + ((int64_t)b * Lq + i) * E + h * BWD_HD, gr[r]);
        qn[r] = fmaxf(sqrtf(dot32(qh[r], qh[r])), 1e-12f);
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) qh[r][c] /= qn[r];
        use_mask[r] = masked != nullptr && (row_any == nullptr || row_any[(int64_t)b * Lq + i] != 0);
    }
    const float* kb = k + (int64_t)b * k_sb + h * BWD_HD;
    const float* vb = v + (int64_t)b * v_sb + h * BWD_HD;
    // pass 1: l = sum_j e_j, o = sum_j e_j v_j with e_j = exp(kappa (q^.k^_j - 1)), 0 where masked
    float l[BWD_BQ], o[BWD_BQ][BWD_HD];
#pragma unroll
    for (int r = 0; r < BWD_BQ; ++r) {
        l[r] = 0.f;
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) o[r][c] = 0.f;
    }
    for (int j = threadIdx.x; j < S; j += 256) {
        float kj[BWD_HD], vj[BWD_HD];
        load32(kb + (int64_t)j * ldk, kj);
        load32(vb + (int64_t)j * ldv, vj);
        const float rn = 1.0f / fmaxf(sqrtf(dot32(kj, kj)), 1e-12f);
#pragma unroll
        for (int r = 0; r < BWD_BQ; ++r) {
            const bool m = use_mask[r] && masked[((int64_t)b * Lq + min(i0 + r, Lq - 1)) * S + j] != 0;
            const float e = m ? 0.f : __expf(kappa * (dot32(qh[r], kj) * rn - 1.0f));
            l[r] += e;
#pragma unroll
            for (int c = 0; c < BWD_HD; ++c) o[r][c] = fmaf(e, vj[c], o[r][c]);
        }
    }
    float D[BWD_BQ], dO[BWD_BQ][BWD_HD];
#pragma unroll
    for (int r = 0; r < BWD_BQ; ++r) {
        l[r] = block_sum_256(l[r], red);
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) o[r][c] = block_sum_256(o[r][c], red) / l[r];      // o = A v
        const float on = fmaxf(sqrtf(dot32(o[r], o[r])), 1e-12f);
        float og = 0.f;
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) og = fmaf(o[r][c] / on, gr[r][c], og);              // out.g
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) dO[r][c] = (gr[r][c] - (o[r][c] / on) * og) / on;
        D[r] = dot32(dO[r], o[r]);
    }
    // pass 2: dq^ = kappa sum_j p_j (dO.v_j - D) k^_j
    float dq[BWD_BQ][BWD_HD];
#pragma unroll
    for (int r = 0; r < BWD_BQ; ++r)
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) dq[r][c] = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) {
        float kj[BWD_HD], vj[BWD_HD];
        load32(kb + (int64_t)j * ldk, kj);
        load32(vb + (int64_t)j * ldv, vj);
        const float rn = 1.0f / fmaxf(sqrtf(dot32(kj, kj)), 1e-12f);
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) kj[c] *= rn;
#pragma unroll
        for (int r = 0; r < BWD_BQ; ++r) {
            const bool m = use_mask[r] && masked[((int64_t)b * Lq + min(i0 + r, Lq - 1)) * S + j] != 0;
            const float p = m ? 0.f : __expf(kappa * (dot32(qh[r], kj) - 1.0f)) / l[r];
            const float ds = kappa * p * (dot32(dO[r], vj) - D[r]);
#pragma unroll
            for (int c = 0; c < BWD_HD; ++c) dq[r][c] = fmaf(ds, kj[c], dq[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < BWD_BQ; ++r) {
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) dq[r][c] = block_sum_256(dq[r][c], red);
        const int i = i0 + r;
        if (i < Lq && threadIdx.x == 0) {
            const float qd = dot32(qh[r], dq[r]);
            float* gp = gq + ((int64_t)b * Lq + i) * E + h * BWD_HD;
#pragma unroll
            for (int c = 0; c < BWD_HD; ++c) gp[c] = (dq[r][c] - qh[r][c] * qd) / qn[r];
            float* w = ws + (((int64_t)b * heads + h) * Lq + i) * BWD_REC;
            w[0] = l[r];
            w[1] = D[r];
#pragma unroll
            for (int c = 0; c < BWD_HD; ++c) {
                w[2 + c] = dO[r][c];
                w[2 + BWD_HD + c] = qh[r][c];
            }
        }
    }
}

__global__ __launch_bounds__(256) void hs_attn_bwd_kv_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                             const uint8_t* __restrict__ masked, const int32_t* __restrict__ row_any,
                                                             const float* __restrict__ ws, float* __restrict__ gk, float* __restrict__ gv,
                                                             int Lq, int S, int heads, int64_t ldk, int64_t k_sb, int64_t ldv, int64_t v_sb,
                                                             float kappa) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = min(j, S - 1);
    const int E = heads * BWD_HD;
    float kj[BWD_HD], vj[BWD_HD], dk[BWD_HD], dv[BWD_HD];
    load32(k + (int64_t)b * k_sb + (int64_t)jc * ldk + h * BWD_HD, kj);
    load32(v + (int64_t)b * v_sb + (int64_t)jc * ldv + h * BWD_HD, vj);
    const float kn = fmaxf(sqrtf(dot32(kj, kj)), 1e-12f);
#pragma unroll
    for (int c = 0; c < BWD_HD; ++c) {
        kj[c] /= kn;
        dk[c] = 0.f;
        dv[c] = 0.f;
    }
    const float* rec = ws + ((int64_t)b * heads + h) * Lq * BWD_REC;
    for (int i = 0; i < Lq; ++i) {
        const float* w = rec + (int64_t)i * BWD_REC;            // wave-uniform: scalar loads
        const bool um = masked != nullptr && (row_any == nullptr || row_any[(int64_t)b * Lq + i] != 0);
        const bool m = um && masked[((int64_t)b * Lq + i) * S + jc] != 0;
        float dot = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) {
            dot = fmaf(w[2 + BWD_HD + c], kj[c], dot);
            dp = fmaf(w[2 + c], vj[c], dp);
        }
        const float p = m ? 0.f : __expf(kappa * (dot - 1.0f)) / w[0];
        const float ds = kappa * p * (dp - w[1]);
#pragma unroll
        for (int c = 0; c < BWD_HD; ++c) {
            dk[c] = fmaf(ds, w[2 + BWD_HD + c], dk[c]);
            dv[c] = fmaf(p, w[2 + c], dv[c]);
        }
    }
    if (j < S) {
        const float kd = dot32(kj, dk);
        float* gkp = gk + ((int64_t)b * S + j) * E + h * BWD_HD;
        float* gvp = gv + ((int64_t)b * S + j) * E + h * BWD_HD;
#pragma unroll
        for (int c = 0; c < BWD_HD; c += 4) {
            *reinterpret_cast<float4*>(gkp + c) = make_float4((dk[c] - kj[c] * kd) / kn, (dk[c + 1] - kj[c + 1] * kd) / kn,
                                                              (dk[c + 2] - kj[c + 2] * kd) / kn, (dk[c + 3] - kj[c + 3] * kd) / kn);
            *reinterpret_cast<float4*>(gvp + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
        }
    }
}

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_hypersphere_attn_bwd_workspace(int B, int Lq, int heads) { return (int64_t)B * heads * Lq * BWD_REC; }

extern "C" int msm_hypersphere_attn_bwd(const float* q, const float* k, const float* v, const uint8_t* masked, const int32_t* row_any,
                                        const float* grad_out, float* grad_q, float* grad_k, float* grad_v, int B, int Lq, int S,
                                        int heads, int64_t ldq, int64_t q_sb, int64_t ldk, int64_t k_sb, int64_t ldv, int64_t v_sb,
                                        float kappa, float* workspace, int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(q && k && v && grad_out && grad_q && grad_k && grad_v && workspace, "msm_hypersphere_attn_bwd: null pointer");
    MSM_REQUIRE(B > 0 && Lq > 0 && S > 0 && heads > 0, "msm_hypersphere_attn_bwd: bad sizes");
    MSM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && q_sb % 4 == 0 && k_sb % 4 == 0 && v_sb % 4 == 0,
                "msm_hypersphere_attn_bwd: strides must be multiples of 4 floats");
    MSM_REQUIRE(((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)grad_out) | ((uintptr_t)grad_k) | ((uintptr_t)grad_v)) & 15) == 0,
                "msm_hypersphere_attn_bwd: pointers must be 16-byte aligned");
    if (workspace_elems < msm_hypersphere_attn_bwd_workspace(B, Lq, heads)) {
        set_error("msm_hypersphere_attn_bwd: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hs_attn_bwd_q_kernel, dim3(cdiv(Lq, BWD_BQ), heads, B), dim3(256), 0, st, q, k, v, masked, row_any, grad_out, grad_q,
                       workspace, Lq, S, heads, ldq, q_sb, ldk, k_sb, ldv, v_sb, kappa);
    hipLaunchKernelGGL(hs_attn_bwd_kv_kernel, dim3(cdiv(S, 256), heads, B), dim3(256), 0, st, k, v, masked, row_any, workspace, grad_k, grad_v,
                       Lq, S, heads, ldk, k_sb, ldv, v_sb, kappa);
    MSM_CHECK_LAUNCH("msm_hypersphere_attn_bwd");
    return MSM_OK;
}
