// Fused token-wise block of one MSDeformAttn encoder layer (see include/msm_hip.h:
// msm_encoder_block_fwd).
//
// Reference (per layer, msdeformattn.py:122-131 and ops/modules/ms_deform_attn.py:95-104,123):
//     src = LN1(src + output_proj(msda_out))
//     src = LN2(src + linear2(relu(linear1(src))))                    d_model 64 -> 1024 -> 64
// and, for the NEXT layer's deformable attention,
//     value = value_proj(src) ; proj = [sampling_offsets | attention_weights](src + pos)
// As separate GEMMs these are K=64 / N=64 shapes whose 1024-wide hidden activation (206 MB per
// layer at B=8) round-trips HBM.  Everything above is token-local, so one kernel keeps a tile of 16
// tokens in registers from msda_out to the next layer's value/proj:
//
//   * layout L: lane (token lj = l&15, quarter lq = l>>4) holds features {fb*16 + lq*4 + r}; this is at
//     once the C/D layout of a transposed MFMA tile (rows = output features, cols = tokens) and --
//     walking K in the order (fb, r) -- the B-operand layout of the next GEMM, so activations never
//     leave registers between the five GEMMs of the chain and the two LayerNorms reduce over the 4
//     lanes of a token with two shuffles;
//   * weights are the A operand.  They are pre-packed (host, once per checkpoint) into a stream of
//     4 KiB blocks in consumption order and staged through LDS in 32 KiB chunks (double buffered,
//     one barrier per chunk) with an XOR swizzle that makes every ds_read_b128 conflict-free;
//   * 4 waves x 16 tokens per workgroup, 64 KiB LDS -> 2 workgroups per CU.
#include "common.h"

namespace msm {

constexpr int EC = 64;                 // d_model
constexpr int CHUNK_F4 = 2048;         // float4 per 32 KiB chunk (8 blocks of 256 float4)

struct EncSmall {                      // offsets (floats) into the packed small-parameter vector
    int bo, g1, be1, b1, b2, g2, be2, bv, bp;
};

__device__ __forceinline__ float4 lds4(const float4* base, int idx) { return base[idx]; }

// A-operand fragments of one [16 rows][64 k] weight block: 4 x ds_read_b128 (conflict-free by the XOR swizzle)
__device__ __forceinline__ void rowblock_read(const float4* __restrict__ blk, int lj, int lq, float4 (&w)[4]) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) w[fb] = lds4(blk, lj * 16 + ((fb * 4 + lq) ^ lj));
}
// 16 MFMAs: consecutive instructions alternate between two accumulators, so no MFMA waits for the 40-cycle
// dependent-accumulator latency of its predecessor.  d0 + d1 is the block's output in layout L.
__device__ __forceinline__ void rowblock_mma(const float4 (&w)[4], const float (&act)[4][4], f32x4& d0, f32x4& d1) {
    d0 = mfma16(w[0].x, act[0][0], d0);
    d1 = mfma16(w[1].x, act[1][0], d1);
    d0 = mfma16(w[2].x, act[2][0], d0);
    d1 = mfma16(w[3].x, act[3][0], d1);
    d0 = mfma16(w[0].y, act[0][1], d0);
    d1 = mfma16(w[1].y, act[1][1], d1);
    d0 = mfma16(w[2].y, act[2][1], d0);
    d1 = mfma16(w[3].y, act[3][1], d1);
    d0 = mfma16(w[0].z, act[0][2], d0);
    d1 = mfma16(w[1].z, act[1][2], d1);
    d0 = mfma16(w[2].z, act[2][2], d0);
    d1 = mfma16(w[3].z, act[3][2], d1);
    d0 = mfma16(w[0].w, act[0][3], d0);
    d1 = mfma16(w[1].w, act[1][3], d1);
    d0 = mfma16(w[2].w, act[2][3], d0);
    d1 = mfma16(w[3].w, act[3][3], d1);
}
__device__ __forceinline__ f32x4 rowblock_mm(const float4* __restrict__ blk, int lj, int lq, const float (&act)[4][4]) {
    float4 w[4];
    rowblock_read(blk, lj, lq, w);
    f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
    rowblock_mma(w, act, d0, d1);
    return d0 + d1;
}

__device__ __forceinline__ void layer_norm_L(float (&v)[4][4], const float* __restrict__ g, const float* __restrict__ b,
                                             int lq, float eps) {
    float s = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[fb][r];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / EC);
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
    const float rstd = 1.0f / sqrtf(q * (1.0f / EC) + eps);
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

__global__ __launch_bounds__(256) void enc_block_kernel(const float* __restrict__ attn, const float* __restrict__ src,
                                                        const float4* __restrict__ wstream, const float* __restrict__ small,
                                                        EncSmall so, const float* __restrict__ pos,
                                                        float* __restrict__ src_out, float* __restrict__ value_out,
                                                        float* __restrict__ proj_out, int M, int S, int nffn, int nproj_blocks,
                                                        int proj_ld, float eps, int n_small) {
    extern __shared__ __attribute__((aligned(16))) float4 wl[];   // [2][CHUNK_F4] weight chunks, then the small parameters
    float* sm = reinterpret_cast<float*>(wl + 2 * CHUNK_F4);
    for (int i = threadIdx.x; i < n_small; i += 256) sm[i] = small[i];   // biases / LayerNorm vectors: read from LDS in the loop
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int tok = blockIdx.x * 64 + wave * 16 + lj;
    const bool tok_ok = tok < M;
    const int tk = tok_ok ? tok : M - 1;
    const bool next = value_out != nullptr;
    const int nchunks = 1 + nffn + (next ? 1 + (nproj_blocks - 4 + 7) / 8 : 0);

    // ---- weight chunk staging: 8 float4 per thread per chunk (one per block), swizzled into LDS ----
    // destination float4 index inside a block for this thread (row blocks / linear2 blocks)
    const int dst_row = (tid >> 4) * 16 + ((tid & 15) ^ (tid >> 4));
    const int dst_w2 = (tid >> 2) * 4 + ((tid & 3) ^ ((tid >> 4) & 3));
#define ENC_CHUNK_LOAD(c)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) stage[i] = wstream[(int64_t)(c) * CHUNK_F4 + tid + 256 * i];
#define ENC_CHUNK_STORE(c, buf)                                                               \
    {                                                                                         \
        const bool ffn_ = (c) >= 1 && (c) <= nffn;                                            \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)(buf)[i * 256 + ((ffn_ && (i & 1)) ? dst_w2 : dst_row)] = stage[i]; \
    }
    float4 stage[8];

    // ---- tile inputs in layout L ----
    float act[4][4], res[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)tk * EC + fb * 16 + lq * 4);
        const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk * EC + fb * 16 + lq * 4);
        act[fb][0] = a.x; act[fb][1] = a.y; act[fb][2] = a.z; act[fb][3] = a.w;
        res[fb][0] = r.x; res[fb][1] = r.y; res[fb][2] = r.z; res[fb][3] = r.w;
    }

    ENC_CHUNK_LOAD(0)
    ENC_CHUNK_STORE(0, wl)
    __syncthreads();

    float x[4][4];      // current activations (layout L)
    f32x4 acc2[4];
    for (int c = 0; c < nchunks; ++c) {
        const float4* buf = wl + (c & 1) * CHUNK_F4;
        // prefetch the next chunk into registers; it is written to the other LDS buffer after this chunk's MFMAs
        {
            const int cn = min(c + 1, nchunks - 1);
            ENC_CHUNK_LOAD(cn)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0) {
            // ---- output_proj + residual + LayerNorm1 (msdeformattn.py:124-126) ----
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const f32x4 d = rowblock_mm(buf + ob * 256, lj, lq, act);
                const float4 bo = *reinterpret_cast<const float4*>(sm + so.bo + ob * 16 + lq * 4);
                x[ob][0] = d[0] + bo.x + res[ob][0];
                x[ob][1] = d[1] + bo.y + res[ob][1];
                x[ob][2] = d[2] + bo.z + res[ob][2];
                x[ob][3] = d[3] + bo.w + res[ob][3];
            }
            layer_norm_L(x, sm + so.g1, sm + so.be1, lq, eps);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) acc2[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (c <= nffn) {
            // ---- FFN: 4 hidden blocks of 16 per chunk; the hidden activation lives in 4 registers.
            // Software pipeline over the hidden blocks q: the 16 linear1 MFMAs of block q+1 are issued
            // before block q's result is read back (bias + ReLU) and fed to its 16 linear2 MFMAs, and the
            // LDS fragments are fetched one block ahead, so neither MFMA results nor ds_reads are waited on.
            float4 w1[2][4], w2[4];
            f32x4 dd[2][2];
            rowblock_read(buf + 0 * 256, lj, lq, w1[0]);
            rowblock_read(buf + 2 * 256, lj, lq, w1[1]);
            dd[0][0] = dd[0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            rowblock_mma(w1[0], x, dd[0][0], dd[0][1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cur = q & 1, nxt = cur ^ 1;
                const int hb = (c - 1) * 4 + q;
                // fragments: linear2 block q (used below), linear1 block q+2 (used next iteration)
                const float4* w2p = buf + (2 * q + 1) * 256;
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const int row = ob * 16 + lj;
                    w2[ob] = lds4(w2p, row * 4 + (lq ^ ((row >> 2) & 3)));
                }
                if (q + 1 < 4) {
                    dd[nxt][0] = dd[nxt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    rowblock_mma(w1[nxt], x, dd[nxt][0], dd[nxt][1]);          // linear1 of block q+1
                    if (q + 2 < 4) rowblock_read(buf + (2 * (q + 2)) * 256, lj, lq, w1[cur]);
                }
                const float4 b1 = *reinterpret_cast<const float4*>(sm + so.b1 + hb * 16 + lq * 4);
                f32x4 h = dd[cur][0] + dd[cur][1];
                h[0] = fmaxf(h[0] + b1.x, 0.f);
                h[1] = fmaxf(h[1] + b1.y, 0.f);
                h[2] = fmaxf(h[2] + b1.z, 0.f);
                h[3] = fmaxf(h[3] + b1.w, 0.f);
                // linear2 of block q, order (r, ob): consecutive MFMAs hit different accumulators
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[ob].x, h[0], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[ob].y, h[1], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[ob].z, h[2], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[ob].w, h[3], acc2[ob]);
            }
            if (c == nffn) {
                // ---- residual + LayerNorm2 (msdeformattn.py:116-118), write the layer output ----
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const float4 b2 = *reinterpret_cast<const float4*>(sm + so.b2 + ob * 16 + lq * 4);
                    x[ob][0] += acc2[ob][0] + b2.x;
                    x[ob][1] += acc2[ob][1] + b2.y;
                    x[ob][2] += acc2[ob][2] + b2.z;
                    x[ob][3] += acc2[ob][3] + b2.w;
                }
                layer_norm_L(x, sm + so.g2, sm + so.be2, lq, eps);
                if (tok_ok) {
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob)
                        *reinterpret_cast<float4*>(src_out + (int64_t)tok * EC + ob * 16 + lq * 4) =
                            make_float4(x[ob][0], x[ob][1], x[ob][2], x[ob][3]);
                }
            }
        } else {
            // ---- next layer's value_proj and [sampling_offsets | attention_weights] ----
            const int cc = c - nffn - 1;
            int blk0 = 0;
            if (cc == 0) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const f32x4 d = rowblock_mm(buf + ob * 256, lj, lq, x);
                    const float4 bv = *reinterpret_cast<const float4*>(sm + so.bv + ob * 16 + lq * 4);
                    if (tok_ok)
                        *reinterpret_cast<float4*>(value_out + (int64_t)tok * EC + ob * 16 + lq * 4) =
                            make_float4(d[0] + bv.x, d[1] + bv.y, d[2] + bv.z, d[3] + bv.w);
                }
                // query = src + pos (msdeformattn.py:124): add the level/position code once
                const int sp = tk % S;
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const float4 pp = *reinterpret_cast<const float4*>(pos + (int64_t)sp * EC + fb * 16 + lq * 4);
                    x[fb][0] += pp.x; x[fb][1] += pp.y; x[fb][2] += pp.z; x[fb][3] += pp.w;
                }
                blk0 = 4;
            }
            // proj output blocks held by this chunk: chunk cc=0 has blocks 4..7 -> ob 0..3; cc>=1 has 8 each
            const int ob_base = (cc == 0) ? 0 : 4 + (cc - 1) * 8;
            for (int j = blk0; j < 8; ++j) {
                const int ob = ob_base + (j - blk0);
                if (ob >= nproj_blocks) break;
                const f32x4 d = rowblock_mm(buf + j * 256, lj, lq, x);
                const float4 bp = *reinterpret_cast<const float4*>(sm + so.bp + ob * 16 + lq * 4);
                if (tok_ok)
                    *reinterpret_cast<float4*>(proj_out + (int64_t)tok * proj_ld + ob * 16 + lq * 4) =
                        make_float4(d[0] + bp.x, d[1] + bp.y, d[2] + bp.z, d[3] + bp.w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunks) ENC_CHUNK_STORE(c + 1, wl + ((c + 1) & 1) * CHUNK_F4)
        __syncthreads();
    }
}

#undef ENC_CHUNK_LOAD
#undef ENC_CHUNK_STORE

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_encoder_block_stream_floats(int d_ffn, int proj_width) {
    const int nffn = d_ffn / 64;
    const int npb = cdiv(proj_width, 16);
    const int nchunks = 1 + nffn + 1 + cdiv(max(npb - 4, 0), 8);
    return (int64_t)nchunks * CHUNK_F4 * 4;
}

extern "C" int msm_encoder_block_fwd(const float* attn, const float* src, const float* wstream, const float* small,
                                     const float* pos, float* src_out, float* value_out, float* proj_out, int M, int S,
                                     int d_ffn, int proj_width, float eps, void* stream) {
    MSM_REQUIRE(attn && src && wstream && small && src_out, "msm_encoder_block_fwd: null pointer");
    MSM_REQUIRE((value_out == nullptr) == (proj_out == nullptr), "msm_encoder_block_fwd: value_out and proj_out go together");
    MSM_REQUIRE(!value_out || pos, "msm_encoder_block_fwd: pos required when the next layer's projections are produced");
    MSM_REQUIRE(M > 0 && S > 0 && d_ffn > 0 && d_ffn % 64 == 0, "msm_encoder_block_fwd: bad sizes (d_ffn %% 64 == 0)");
    MSM_REQUIRE(proj_width % 16 == 0 && proj_width >= 64, "msm_encoder_block_fwd: proj_width=%d must be a multiple of 16, >= 64",
                proj_width);
    MSM_REQUIRE(((((uintptr_t)attn) | ((uintptr_t)src) | ((uintptr_t)wstream) | ((uintptr_t)small) | ((uintptr_t)src_out) |
                  ((uintptr_t)value_out) | ((uintptr_t)proj_out) | ((uintptr_t)pos)) & 15) == 0,
                "msm_encoder_block_fwd: pointers must be 16-byte aligned");
    EncSmall so;
    int o = 0;
    so.bo = o; o += 64;
    so.g1 = o; o += 64;
    so.be1 = o; o += 64;
    so.b1 = o; o += d_ffn;
    so.b2 = o; o += 64;
    so.g2 = o; o += 64;
    so.be2 = o; o += 64;
    so.bv = o; o += 64;
    so.bp = o;
    const int n_small = so.bp + proj_width;
    const size_t lds = sizeof(float4) * 2 * CHUNK_F4 + sizeof(float) * (size_t)((n_small + 3) / 4 * 4);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_block_kernel, lds));
    dim3 grid(cdiv(M, 64)), block(256);
    hipLaunchKernelGGL(enc_block_kernel, grid, block, lds, (hipStream_t)stream, attn, src,
                       reinterpret_cast<const float4*>(wstream), small, so, pos, src_out, value_out, proj_out, M, S, d_ffn / 64,
                       proj_width / 16, proj_width, eps, n_small);
    MSM_CHECK_LAUNCH("msm_encoder_block_fwd");
    return MSM_OK;
}
