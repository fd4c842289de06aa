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
//     4 KiB blocks in consumption order and staged through LDS in 16 KiB halves of its 32 KiB chunks
//     (double buffered, one barrier per half) with an XOR swizzle that makes every ds_read_b128 conflict-free;
//   * 4 waves x 16 tokens per workgroup, 38 KiB LDS -> 4 workgroups per CU.
#include <stdlib.h>

#include "bf16.h"
#include "common.h"

namespace msm {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int EC = 64;                 // d_model
constexpr int CHUNK_F4 = 1024;         // float4 per 16 KiB LDS stage = HALF a 32 KiB stream chunk (4 blocks of 256 float4)

struct EncSmall {                      // offsets (floats) into the packed small-parameter vector
    int bo, g1, be1, b1, b2, g2, be2, bv, bp;
};

__device__ __forceinline__ float4 lds4(const float4* base, int idx) { return base[idx]; }

// A-operand fragments of one [16 rows][64 k] weight block: 4 x ds_read_b128 (conflict-free by the XOR swizzle)
__device__ __forceinline__ void rowblock_read(const float4* __restrict__ blk, int lj, int lq, float4 (&w)[4]) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) w[fb] = lds4(blk, lj * 16 + ((fb * 4 + lq) ^ lj));
}
// 16 MFMAs on ONE accumulator chain that starts from `d` (the bias): dependent fp32 MFMAs issue back to back at the full
// rate, while every VALU instruction costs its SIMD about six cycles of MFMA issue (tools/probes/mfma_probe.hip) -- so
// no second accumulator to add up afterwards and no separate bias add.  The result is in layout L.
__device__ __forceinline__ void rowblock_mma(const float4 (&w)[4], const float (&act)[4][4], f32x4& d) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) d = mfma16(w[fb].x, act[fb][0], d);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) d = mfma16(w[fb].y, act[fb][1], d);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) d = mfma16(w[fb].z, act[fb][2], d);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) d = mfma16(w[fb].w, act[fb][3], d);
}
__device__ __forceinline__ f32x4 rowblock_mm(const float4* __restrict__ blk, int lj, int lq, const float (&act)[4][4],
                                             const float* __restrict__ bias) {
    float4 w[4];
    rowblock_read(blk, lj, lq, w);
    const float4 b = *reinterpret_cast<const float4*>(bias + lq * 4);
    f32x4 d = f32x4{b.x, b.y, b.z, b.w};
    rowblock_mma(w, act, d);
    return d;
}

// ReLU in ONE VALU instruction (v_med3_f32 x, 0, 3e38 -- a finite bound, or the compiler folds it back into fmaxf, which
// costs two: it canonicalises its operand first)
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.0e38f); }

__device__ __forceinline__ void layer_norm_L(float (&v)[4][4], const float* __restrict__ g, const float* __restrict__ b,
                                             int lq, float eps) {
    float s = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[fb][r];
    s = sum_lane_rows(s);
    const float mean = s * (1.0f / EC);
    float q = 0.f;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = v[fb][r] - mean;
            q += d * d;
        }
    q = sum_lane_rows(q);
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

// One LDS-DMA piece: lane l's 16 bytes at sbase + voff(l) land at LDS byte address lds_dst + 16 l.  M0 is written in
// the statement that reads it (it is compiler-reserved); both scalar operands come from SALU code (no VALU-to-SGPR hazard).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void enc_block_kernel(const float* __restrict__ attn, const float* __restrict__ src,
                                                        const float4* __restrict__ wstream, const float* __restrict__ small,
                                                        EncSmall so, const float* __restrict__ pos,
                                                        float* __restrict__ src_out, float* __restrict__ value_out,
                                                        float* __restrict__ proj_out, int M, int S, int nffn, int nproj_blocks,
                                                        int proj_ld, float eps, int n_small, int n_normal, int value_heads) {
    extern __shared__ __attribute__((aligned(16))) float4 wl[];   // [2][CHUNK_F4] weight chunks, then the small parameters
    float* sm = reinterpret_cast<float*>(wl + 2 * CHUNK_F4);
    for (int i = threadIdx.x; i < n_small; i += 256) sm[i] = small[i];   // biases / LayerNorm vectors: read from LDS in the loop
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    // Workgroups [0, n_normal): one 16-token tile per wave.  Workgroups >= n_normal are COOPERATIVE: all four waves
    // work on ONE tile and split the FFN (by LDS stage) and the tail row blocks between them, so such a workgroup
    // costs every SIMD about a quarter of a tile.  The host turns the tiles that would otherwise start a nearly
    // empty extra round (B = 8: 3150 tiles = 3 x 1024 SIMDs + 78) into cooperative ones: SIMD makespan 4 -> 3.3 tiles.
    const bool coop = (int)blockIdx.x >= n_normal;
    const int tile = coop ? n_normal * 4 + ((int)blockIdx.x - n_normal) : (int)blockIdx.x * 4 + wave;
    const int tok = tile * 16 + lj;
    const bool tok_ok = tok < M;
    const int tk = tok_ok ? tok : M - 1;
    const bool next = value_out != nullptr;
    // The host stream is organised in 32 KiB chunks of 8 blocks (ops.pack_encoder_block); the kernel walks it in
    // 16 KiB halves so that a workgroup needs 2 x 16 KiB + parameters = 38 KiB of LDS and FOUR workgroups fit a
    // CU (the 3150 16-token tiles of B = 8 then are all resident at once: 3 or 4 waves per SIMD instead of a
    // second, half-empty round).  Step s reads stream half sh(s): 0 = output_proj, (half 1 is the chunk's zero
    // padding: skipped), 2.. = FFN (2 hidden blocks each), then value_proj, then 4 proj row blocks per half.
    const int nhf = 2 * nffn;
    const int ntail = next ? 1 + (nproj_blocks + 3) / 4 : 0;
    const int nsteps = 1 + nhf + ntail;

    // ---- weight staging: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, 4 per wave and stage), issued at
    // the START of the stage before the one that consumes it and awaited (vmcnt(0)) just before that stage's closing
    // barrier, so a stage's weights travel while the previous stage's 64 MFMAs per wave run.  The DMA writes LDS
    // lane-linearly, so the XOR swizzle is applied to the SOURCE index (both swizzles are involutions that stay inside
    // a wave's 64 float4).  Inline asm because hipcc would otherwise wait for an LDS-DMA in flight before ANY LDS read;
    // with compiler-visible loads staged through registers the loads were sunk next to their ds_write (latency exposed
    // on every stage: 164 us per launch).
    const unsigned off_row = (unsigned)((tid >> 4) * 16 + ((tid & 15) ^ (tid >> 4))) * 16u;
    const unsigned off_w2 = (unsigned)((tid >> 2) * 4 + ((tid & 3) ^ ((tid >> 4) & 3))) * 16u;
    const unsigned lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4*)wl + (unsigned)wave * 1024u;
#define ENC_STAGE_PIECE(s_, bufi_, i_)                                                         \
    {                                                                                         \
        const int sh_ = (s_) == 0 ? 0 : (s_) + 1;                                             \
        const bool ffn_ = (s_) >= 1 && (s_) <= nhf;                                           \
        const char* sb_ = reinterpret_cast<const char*>(wstream + (int64_t)sh_ * CHUNK_F4);   \
        const unsigned ld_ = lds_wave + (unsigned)(bufi_) * (CHUNK_F4 * 16u);                 \
        glds16(sb_ + (i_) * 4096, (ffn_ && ((i_) & 1)) ? off_w2 : off_row, ld_ + (i_) * 4096u); \
    }
#define ENC_STAGE_ISSUE(s_, bufi_)                                                             \
    { ENC_STAGE_PIECE(s_, bufi_, 0) ENC_STAGE_PIECE(s_, bufi_, 1) ENC_STAGE_PIECE(s_, bufi_, 2) ENC_STAGE_PIECE(s_, bufi_, 3) }
#define ENC_STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- tile inputs in layout L ----
    float act[4][4], res[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)tk * EC + fb * 16 + lq * 4);
        const float4 r = *reinterpret_cast<const float4*>(src + (int64_t)tk * EC + fb * 16 + lq * 4);
        act[fb][0] = a.x; act[fb][1] = a.y; act[fb][2] = a.z; act[fb][3] = a.w;
        res[fb][0] = r.x; res[fb][1] = r.y; res[fb][2] = r.z; res[fb][3] = r.w;
    }

    ENC_STAGE_ISSUE(0, 0)
    ENC_STAGE_WAIT()
    __syncthreads();

    float x[4][4];      // current activations (layout L)
    f32x4 acc2[4];
    // ---- step 0 (peeled: act/res die here): output_proj + residual + LayerNorm1 (msdeformattn.py:124-126) ----
    {
        ENC_STAGE_ISSUE(1, 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 d = rowblock_mm(wl + ob * 256, lj, lq, act, sm + so.bo + ob * 16);
            x[ob][0] = d[0] + res[ob][0];
            x[ob][1] = d[1] + res[ob][1];
            x[ob][2] = d[2] + res[ob][2];
            x[ob][3] = d[3] + res[ob][3];
        }
        layer_norm_L(x, sm + so.g1, sm + so.be1, lq, eps);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) acc2[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        ENC_STAGE_WAIT()
        __syncthreads();
    }
    // ---- steps 1..nhf: FFN, 2 hidden blocks of 16 per stage; the hidden activation lives in 4 registers.  The 16
    // linear1 MFMAs of block 1 are issued before block 0's result is read back (bias + ReLU) and fed to its 16
    // linear2 MFMAs; both blocks' LDS fragments are requested up front.
    for (int s = 1; s <= nhf; ++s) {
        const float4* buf = wl + (s & 1) * CHUNK_F4;
        // The next stage's weights travel into the other LDS buffer (every wave is past the barrier that followed its
        // last read) while this stage computes; the four DMA pieces are issued between the MFMA groups, where their
        // issue slots are free.
        // (the last stage of a launch without tail stages re-fetches itself into the idle buffer: no branch between the
        // MFMA groups)
        const int sn = min(s + 1, nsteps - 1), bn = (s + 1) & 1;
        if (!coop) {
            float4 w1[2][4], w2[2][4];
            f32x4 dd[2];
            rowblock_read(buf + 0 * 256, lj, lq, w1[0]);
            rowblock_read(buf + 2 * 256, lj, lq, w1[1]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4* w2p = buf + (2 * q + 1) * 256;
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const int row = ob * 16 + lj;
                    w2[q][ob] = lds4(w2p, row * 4 + (lq ^ ((row >> 2) & 3)));
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {        // linear1's bias is the initial accumulator
                const float4 b1 = *reinterpret_cast<const float4*>(sm + so.b1 + ((s - 1) * 2 + q) * 16 + lq * 4);
                dd[q] = f32x4{b1.x, b1.y, b1.z, b1.w};
            }
            rowblock_mma(w1[0], x, dd[0]);
            ENC_STAGE_PIECE(sn, bn, 0)
            rowblock_mma(w1[1], x, dd[1]);
            ENC_STAGE_PIECE(sn, bn, 1)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 h = dd[q];
                h[0] = relu1(h[0]);
                h[1] = relu1(h[1]);
                h[2] = relu1(h[2]);
                h[3] = relu1(h[3]);
                // linear2 of block q, order (r, ob): consecutive MFMAs hit different accumulators
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[q][ob].x, h[0], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[q][ob].y, h[1], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[q][ob].z, h[2], acc2[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = mfma16(w2[q][ob].w, h[3], acc2[ob]);
                if (q == 0) ENC_STAGE_PIECE(sn, bn, 2) else ENC_STAGE_PIECE(sn, bn, 3)
            }
        } else {
            // cooperative tile: wave (cq, ch) = (wave >> 1, wave & 1) runs linear1 of hidden block cq and linear2 of that
            // block into output blocks 2 ch, 2 ch + 1 (accumulated in acc2[0], acc2[1]; sorted out after the loop): 24 MFMAs
            // per wave and stage on EVERY SIMD instead of 64 on one -- a stage-by-stage rotation would slow one wave of
            // each co-resident workgroup in every stage, and their barriers make that the pace of all of them.
            ENC_STAGE_ISSUE(sn, bn)
            const int cq = wave >> 1, ch = wave & 1;
            float4 w1c[4], w2c[2];
            rowblock_read(buf + (2 * cq) * 256, lj, lq, w1c);
            const float4* w2p = buf + (2 * cq + 1) * 256;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (2 * ch + j) * 16 + lj;
                w2c[j] = lds4(w2p, row * 4 + (lq ^ ((row >> 2) & 3)));
            }
            const float4 b1 = *reinterpret_cast<const float4*>(sm + so.b1 + ((s - 1) * 2 + cq) * 16 + lq * 4);
            f32x4 h = f32x4{b1.x, b1.y, b1.z, b1.w};
            rowblock_mma(w1c, x, h);
            h[0] = relu1(h[0]);
            h[1] = relu1(h[1]);
            h[2] = relu1(h[2]);
            h[3] = relu1(h[3]);
            acc2[0] = mfma16(w2c[0].x, h[0], acc2[0]);
            acc2[1] = mfma16(w2c[1].x, h[0], acc2[1]);
            acc2[0] = mfma16(w2c[0].y, h[1], acc2[0]);
            acc2[1] = mfma16(w2c[1].y, h[1], acc2[1]);
            acc2[0] = mfma16(w2c[0].z, h[2], acc2[0]);
            acc2[1] = mfma16(w2c[1].z, h[2], acc2[1]);
            acc2[0] = mfma16(w2c[0].w, h[3], acc2[0]);
            acc2[1] = mfma16(w2c[1].w, h[3], acc2[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ENC_STAGE_WAIT()
        __syncthreads();
    }

    // ---- between the loops no staged registers are live ----
    if (coop) {
        // sum the four waves' partial linear2 outputs through the LDS stage that was consumed last (every wave is
        // past the barrier above, the next stage sits in the other buffer)
        float* red = reinterpret_cast<float*>(wl + (nhf & 1) * CHUNK_F4);
        if (wave & 1) {          // this wave's two accumulators are output blocks 2, 3
            acc2[2] = acc2[0];
            acc2[3] = acc2[1];
            acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            *reinterpret_cast<float4*>(red + ((wave * 4 + ob) * 64 + lane) * 4) =
                make_float4(acc2[ob][0], acc2[ob][1], acc2[ob][2], acc2[ob][3]);
        __syncthreads();
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(red + ((w * 4 + ob) * 64 + lane) * 4);
                t += f32x4{v.x, v.y, v.z, v.w};
            }
            acc2[ob] = t;
        }
        __syncthreads();      // the buffer is a staging target again in the tail loop
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
    layer_norm_L(x, sm + so.g2, sm + so.be2, lq, eps);
    if (tok_ok && (!coop || wave == 0)) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            *reinterpret_cast<float4*>(src_out + (int64_t)tok * EC + ob * 16 + lq * 4) =
                make_float4(x[ob][0], x[ob][1], x[ob][2], x[ob][3]);
    }

    // ---- store addressing of the tail stages, once per tile: byte offsets into buffer descriptors (a scalar offset per
    // row block instead of 64-bit VALU arithmetic per store), image / position of the token by ONE division ----
    const int t_img = tk / S, t_pos = tk - t_img * S;
    const bool v_affine = value_heads == 0 || EC / value_heads <= 16;
    unsigned vo0 = (unsigned)tk * (EC * 4u) + lq * 16u, vstep = 64u;           // token-major
    if (value_heads && v_affine) {
        const int dh = EC / value_heads, f0 = lq * 4;
        vo0 = (unsigned)((((int64_t)t_img * value_heads + f0 / dh) * S + t_pos) * dh + f0 % dh) * 4u;
        vstep = (unsigned)S * 64u;                                             // 16 / dh heads further: 16 S floats
    }
    const unsigned po0 = (unsigned)tk * ((unsigned)proj_ld * 4u) + lq * 16u;
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)value_out, 0, next ? M * EC * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)proj_out, 0, next ? M * proj_ld * 4 : 0, 0x00020000);
    // ---- tail stages: next layer's value_proj, then [sampling_offsets | attention_weights] 4 row blocks per stage ----
    for (int s = nhf + 1; s < nsteps; ++s) {
        const float4* buf = wl + (s & 1) * CHUNK_F4;
        if (s + 1 < nsteps) ENC_STAGE_ISSUE(s + 1, (s + 1) & 1)
        __builtin_amdgcn_sched_barrier(0);
        const int hh = s - nhf - 1;
        // a cooperative tile's four waves take one row block each
        if (hh == 0) {
            {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    if (coop && ob != wave) continue;
                    const f32x4 d = rowblock_mm(buf + ob * 256, lj, lq, x, sm + so.bv + ob * 16);
                    if (tok_ok) {
                        // token-major [tok][64], or head-major [b][head][t][64/heads] for msm_msdeform_attn_enc_hm_fwd
                        unsigned o = vo0 + (unsigned)ob * vstep;
                        if (!v_affine) {
                            const int f = ob * 16 + lq * 4, dh = EC / value_heads;
                            o = (unsigned)((((int64_t)t_img * value_heads + f / dh) * S + t_pos) * dh + f % dh) * 4u;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(d[0]), __float_as_uint(d[1]), __float_as_uint(d[2]), __float_as_uint(d[3])}, vrs, o, 0, 0);
                    }
                }
            }
            // query = src + pos (msdeformattn.py:124): add the level/position code once
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const float4 pp = *reinterpret_cast<const float4*>(pos + (int64_t)t_pos * EC + fb * 16 + lq * 4);
                x[fb][0] += pp.x; x[fb][1] += pp.y; x[fb][2] += pp.z; x[fb][3] += pp.w;
            }
        } else {
            // proj output row blocks (hh-1)*4 .. +3
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ob = (hh - 1) * 4 + j;
                if (ob < nproj_blocks && (!coop || j == wave)) {
                    const f32x4 d = rowblock_mm(buf + j * 256, lj, lq, x, sm + so.bp + ob * 16);
                    if (tok_ok)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(d[0]), __float_as_uint(d[1]), __float_as_uint(d[2]), __float_as_uint(d[3])}, prs, po0, (unsigned)ob * 64u, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ENC_STAGE_WAIT()
        __syncthreads();
    }
}

#undef ENC_STAGE_ISSUE
#undef ENC_STAGE_WAIT

// ---------------------------------------------------------------------------------------------------------------
// Encoder prologue: what precedes the first deformable-attention layer (msdeformattn.py:326-329 GroupNorm of the
// input projections, :60-75 level concatenation, and layer 0's value_proj / sampling_offsets / attention_weights
// linears, ops/modules/ms_deform_attn.py:95-104), in one pass over the token buffer:
//     src   = GroupNorm_l(raw)                      raw = the 1x1 input projections of all levels, already concatenated
//     value = value_proj(src)                       (token- or head-major, as enc_block_kernel writes it)
//     proj  = [sampling_offsets | attention_weights](src + pos)
// Same register layout L and weight-block format as enc_block_kernel's tail; the GroupNorm moments come from
// msm_conv1x1_in_f32.  A workgroup's tokens touch at most two images: their (mean, scale, beta) tables are derived from the
// moments into LDS by the workgroup itself.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PRO_MAXL = 4;
struct ProLevels {
    int n;
    int start[PRO_MAXL + 1];      // token offsets of the levels inside an image; start[n] = S
};

// Weight-stationary: the 22 weight blocks (88 KiB) are copied into LDS once per workgroup and each of its PRO_W waves
// then runs its 16-token tile through all 352 MFMAs without a barrier (the staged form of enc_block_kernel's tail paid
// six barriers and 96 KiB of L2 reads per 64 tokens for 64 MFMAs per wave and stage: 52 us; this form: see DESIGN.md).
constexpr int PRO_W = 16;
constexpr int PRO_NIMG = 4;            // images a workgroup's 256 tokens may touch (S >= 86)
__global__ __launch_bounds__(PRO_W * 64) void enc_prologue_kernel(const float* __restrict__ raw, const double* __restrict__ stats,
                                                           const float* __restrict__ gnp, ProLevels lv, int groups, float gn_eps,
                                                           const float4* __restrict__ wstream, const float* __restrict__ small,
                                                           const float* __restrict__ pos, float* __restrict__ src_out,
                                                           float* __restrict__ value_out, float* __restrict__ proj_out, int M,
                                                           int S, int B, int nproj_blocks, int proj_ld, int value_heads, int out_bf16_hm) {
    extern __shared__ __attribute__((aligned(16))) float4 wl[];   // [4 + nproj_blocks][256] weight blocks, GroupNorm tables, biases
    const int nblocks = 4 + nproj_blocks;
    float* gt = reinterpret_cast<float*>(wl + nblocks * 256);    // [PRO_NIMG images][levels][3][64]: mean, rstd*gamma, beta
    float* sm = gt + PRO_NIMG * PRO_MAXL * 3 * EC;                // bv [64], bp [proj width]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    // this tile's tokens first: their latency hides behind the weight copy and the table
    const int tile = (int)blockIdx.x * PRO_W + wave;
    const int tok = tile * 16 + lj;
    const bool tok_ok = tok < M;
    const int tk = tok_ok ? tok : M - 1;
    float x[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const float4 r = *reinterpret_cast<const float4*>(raw + (int64_t)tk * EC + fb * 16 + lq * 4);
        x[fb][0] = r.x; x[fb][1] = r.y; x[fb][2] = r.z; x[fb][3] = r.w;
    }
    // weight blocks -> LDS with the row-block swizzle (float4 i of a block: row i>>4, column group i&15)
    for (int i = tid; i < nblocks * 256; i += PRO_W * 64) {
        const int blk = i >> 8, e = i & 255;
        wl[blk * 256 + (e >> 4) * 16 + ((e & 15) ^ (e >> 4))] = wstream[i];
    }
    const int n_small = EC + nproj_blocks * 16;
    for (int i = tid; i < n_small; i += PRO_W * 64) sm[i] = small[i];
    const int b0 = (int)(((int64_t)blockIdx.x * PRO_W * 16) / S);
    const int cpg = EC / groups;
    // a workgroup's PRO_W*16 = 256 tokens touch at most PRO_NIMG images (S >= 86, checked by the host)
    for (int i = tid; i < PRO_NIMG * lv.n * EC; i += PRO_W * 64) {
        const int c = i % EC, l = (i / EC) % lv.n, bi = b0 + i / (EC * lv.n);
        float mean = 0.f, a = 0.f, be = 0.f;
        if (bi < B) {
            const int g0 = (c / cpg) * cpg;
            double sum = 0.0, sq = 0.0;
            for (int k = 0; k < cpg; ++k) {
                const double* d = stats + (((int64_t)l * B + bi) * EC + g0 + k) * 2;
                sum += d[0];
                sq += d[1];
            }
            const double cnt = (double)cpg * (double)(lv.start[l + 1] - lv.start[l]);
            const double mu = sum / cnt;
            double var = sq / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            a = (float)(1.0 / sqrt(var + (double)gn_eps)) * gnp[(l * 2 + 0) * EC + c];
            be = gnp[(l * 2 + 1) * EC + c];
        }
        float* t = gt + ((i / (EC * lv.n)) * PRO_MAXL + l) * 3 * EC;
        t[c] = mean;
        t[EC + c] = a;
        t[2 * EC + c] = be;
    }
    const int bi = tk / S, ti = tk - bi * S;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < PRO_MAXL; ++l) lvl += (l < lv.n && ti >= lv.start[l]) ? 1 : 0;
    __syncthreads();               // the only barrier: weights, tables and biases are in LDS
    if (tile * 16 >= M) return;    // wave-uniform
    {
        const float* t = gt + ((bi - b0) * PRO_MAXL + lvl) * 3 * EC;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const int c = fb * 16 + lq * 4;
            const float4 mn = *reinterpret_cast<const float4*>(t + c);
            const float4 sc = *reinterpret_cast<const float4*>(t + EC + c);
            const float4 sh = *reinterpret_cast<const float4*>(t + 2 * EC + c);
            x[fb][0] = (x[fb][0] - mn.x) * sc.x + sh.x;
            x[fb][1] = (x[fb][1] - mn.y) * sc.y + sh.y;
            x[fb][2] = (x[fb][2] - mn.z) * sc.z + sh.z;
            x[fb][3] = (x[fb][3] - mn.w) * sc.w + sh.w;
            if (tok_ok)
                *reinterpret_cast<float4*>(src_out + (int64_t)tok * EC + c) = make_float4(x[fb][0], x[fb][1], x[fb][2], x[fb][3]);
        }
    }
    // query = src + pos (msdeformattn.py:124): requested now, added after the value projection
    float4 pp[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) pp[fb] = *reinterpret_cast<const float4*>(pos + (int64_t)ti * EC + fb * 16 + lq * 4);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const f32x4 d = rowblock_mm(wl + ob * 256, lj, lq, x, sm + ob * 16);
        if (tok_ok && out_bf16_hm) {
            // the bf16 plan's layout (csrc/enc_lp.hip): value [B][8][S][8] fp16 -- this lane's four dims are half a head
            const int f = ob * 16 + lq * 4;
            unsigned short* o = reinterpret_cast<unsigned short*>(value_out) + (((int64_t)bi * 8 + (f >> 3)) * S + ti) * 8 + (f & 7);
            *reinterpret_cast<u32x2b*>(o) = pack4h(d[0], d[1], d[2], d[3]);
        } else if (tok_ok) {
            const int f = ob * 16 + lq * 4;
            int64_t o = (int64_t)tok * EC + f;
            if (value_heads) {
                const int dh = EC / value_heads;
                o = (((int64_t)bi * value_heads + f / dh) * S + ti) * dh + f % dh;
            }
            *reinterpret_cast<float4*>(value_out + o) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        x[fb][0] += pp[fb].x; x[fb][1] += pp[fb].y; x[fb][2] += pp[fb].z; x[fb][3] += pp[fb].w;
    }
    for (int ob = 0; ob < nproj_blocks; ++ob) {
        const f32x4 d = rowblock_mm(wl + (4 + ob) * 256, lj, lq, x, sm + EC + ob * 16);
        if (tok_ok && out_bf16_hm) {
            // the bf16 plan's sampling projection (csrc/enc_lp.hip, EH_REC): per (image, head) 120 S bytes = six planes [S][4 floats] of
            // offsets and three planes [S][4 halves] of logits; row n of [192 offsets | 96 logits] is offset (head, c) = (n / 24, n % 24)
            // or logit (n' / 12, n' % 12), n' = n - 192; a 16-row block is all offsets (ob < 12) or all logits
            const int n = ob * 16 + lq * 4;
            unsigned char* base = reinterpret_cast<unsigned char*>(proj_out);
            if (n < 192) {
                const int head = n / 24, plane = (n - head * 24) >> 2;
                typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
                *reinterpret_cast<f32x4_a8*>(base + ((int64_t)bi * 8 + head) * S * 120 + ((int64_t)plane * S + ti) * 16) = f32x4_a8{d[0], d[1], d[2], d[3]};
            } else {
                const int head = (n - 192) / 12, plane = ((n - 192) - head * 12) >> 2;
                *reinterpret_cast<u32x2b*>(base + ((int64_t)bi * 8 + head) * S * 120 + (int64_t)S * 96 + ((int64_t)plane * S + ti) * 8) = pack4h(d[0], d[1], d[2], d[3]);
            }
        } else if (tok_ok)
            *reinterpret_cast<float4*>(proj_out + (int64_t)tok * proj_ld + ob * 16 + lq * 4) = make_float4(d[0], d[1], d[2], d[3]);
    }
}

}  // namespace msm

using namespace msm;

extern "C" int64_t msm_encoder_block_stream_floats(int d_ffn, int proj_width) {
    const int nffn = d_ffn / 64;
    const int npb = cdiv(proj_width, 16);
    const int nchunks = 1 + nffn + 1 + cdiv(max(npb - 4, 0), 8);
    return (int64_t)nchunks * 2 * CHUNK_F4 * 4;
}

extern "C" int msm_encoder_block_fwd(const float* attn, const float* src, const float* wstream, const float* small,
                                     const float* pos, float* src_out, float* value_out, float* proj_out, int M, int S,
                                     int d_ffn, int proj_width, int value_heads, float eps, void* stream) {
    MSM_REQUIRE(attn && src && wstream && small && src_out, "msm_encoder_block_fwd: null pointer");
    // proj_width == 0 with a value_out: only the next layer's value projection (its sampling projection is computed by
    // msm_msdeform_attn_enc_fused_fwd); otherwise value_out and proj_out go together
    MSM_REQUIRE(proj_width == 0 ? (proj_out == nullptr) : ((value_out == nullptr) == (proj_out == nullptr)),
                "msm_encoder_block_fwd: value_out and proj_out go together (proj_out must be null when proj_width == 0)");
    MSM_REQUIRE(!value_out || pos, "msm_encoder_block_fwd: pos required when the next layer's projections are produced");
    MSM_REQUIRE(M > 0 && S > 0 && d_ffn > 0 && d_ffn % 64 == 0, "msm_encoder_block_fwd: bad sizes (d_ffn %% 64 == 0)");
    MSM_REQUIRE(value_heads == 0 || (value_heads > 0 && EC % value_heads == 0 && (EC / value_heads) % 4 == 0 && M % S == 0),
                "msm_encoder_block_fwd: value_heads=%d needs 64/heads to be a multiple of 4 and M a multiple of S", value_heads);
    MSM_REQUIRE(proj_width == 0 || (proj_width % 16 == 0 && proj_width >= 64), "msm_encoder_block_fwd: proj_width=%d must be 0 or a multiple of 16, >= 64",
                proj_width);
    MSM_REQUIRE((int64_t)M * max(proj_width, EC) * 4 < (int64_t)1 << 31, "msm_encoder_block_fwd: M=%d tokens exceed the 2 GiB the output descriptors address", M);
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
    // workgroup plan: 4 tiles per normal workgroup; the partial last workgroup and, when they would start a sparsely
    // filled extra round over the 256 CUs, the workgroups of that round become one cooperative workgroup per tile
    const int tiles = cdiv(M, 16);
    int n_normal = tiles / 4;
    const int extra = n_normal % 256;
    if (n_normal >= 256 && extra <= 64 && opt(MSM_OPT_ENC_NO_COOP) != 1) n_normal -= extra;
    const int n_coop = tiles - n_normal * 4;
    dim3 grid(n_normal + n_coop), block(256);
    hipLaunchKernelGGL(enc_block_kernel, grid, block, lds, (hipStream_t)stream, attn, src,
                       reinterpret_cast<const float4*>(wstream), small, so, pos, src_out, value_out, proj_out, M, S, d_ffn / 64,
                       proj_width / 16, proj_width, eps, n_small, n_normal, value_heads);
    MSM_CHECK_LAUNCH("msm_encoder_block_fwd");
    return MSM_OK;
}

extern "C" int64_t msm_encoder_prologue_stream_floats(int proj_width) {
    return (int64_t)(1 + cdiv(cdiv(proj_width, 16), 4)) * CHUNK_F4 * 4;
}

extern "C" int msm_encoder_prologue_fwd(const float* raw, const double* stats, const float* gn_params, const int32_t* level_starts,
                                        int n_levels, int groups, float gn_eps, const float* wstream, const float* small,
                                        const float* pos, float* src_out, void* value_out, void* proj_out, int B, int S,
                                        int proj_width, int value_heads, int out_bf16_hm, void* stream) {
    MSM_REQUIRE(raw && stats && gn_params && level_starts && wstream && small && pos && src_out && value_out && (proj_out || proj_width == 0),
                "msm_encoder_prologue_fwd: null pointer");
    MSM_REQUIRE(n_levels >= 1 && n_levels <= PRO_MAXL, "msm_encoder_prologue_fwd: n_levels=%d outside [1, %d]", n_levels, PRO_MAXL);
    MSM_REQUIRE(B > 0 && S >= 86, "msm_encoder_prologue_fwd: need B > 0 and at least 86 tokens per image (S=%d)", S);
    MSM_REQUIRE(groups > 0 && EC % groups == 0, "msm_encoder_prologue_fwd: groups=%d must divide 64", groups);
    MSM_REQUIRE(proj_width % 16 == 0 && proj_width >= 0, "msm_encoder_prologue_fwd: proj_width=%d must be a multiple of 16 (0: value projection only)", proj_width);
    MSM_REQUIRE(value_heads == 0 || (value_heads > 0 && EC % value_heads == 0 && (EC / value_heads) % 4 == 0),
                "msm_encoder_prologue_fwd: value_heads=%d needs 64/heads to be a multiple of 4", value_heads);
    MSM_REQUIRE(((((uintptr_t)raw) | ((uintptr_t)wstream) | ((uintptr_t)small) | ((uintptr_t)src_out) | ((uintptr_t)value_out) |
                  ((uintptr_t)proj_out) | ((uintptr_t)pos) | ((uintptr_t)gn_params)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "msm_encoder_prologue_fwd: pointers must be 16-byte aligned");
    MSM_REQUIRE(!out_bf16_hm || (value_heads == 8 && proj_width == 288), "msm_encoder_prologue_fwd: head-major bf16 outputs need 8 heads and a 288-wide projection");
    ProLevels lv;
    lv.n = n_levels;
    for (int l = 0; l <= PRO_MAXL; ++l) lv.start[l] = level_starts[l < n_levels ? l : n_levels];
    MSM_REQUIRE(lv.start[0] == 0 && lv.start[n_levels] == S, "msm_encoder_prologue_fwd: level_starts must run from 0 to S");
    for (int l = 0; l < n_levels; ++l)
        MSM_REQUIRE(lv.start[l + 1] > lv.start[l], "msm_encoder_prologue_fwd: level_starts must increase");
    const int M = B * S;
    const int npb = proj_width / 16;
    MSM_REQUIRE(proj_width <= 512, "msm_encoder_prologue_fwd: proj_width=%d > 512 (weights are held in LDS)", proj_width);
    const size_t lds = sizeof(float4) * (size_t)(4 + npb) * 256 + sizeof(float) * (size_t)(PRO_NIMG * PRO_MAXL * 3 * EC + (EC + proj_width + 3) / 4 * 4);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)enc_prologue_kernel, lds));
    dim3 grid(cdiv(cdiv(M, 16), PRO_W)), block(PRO_W * 64);
    hipLaunchKernelGGL(enc_prologue_kernel, grid, block, lds, (hipStream_t)stream, raw, stats, gn_params, lv, groups, gn_eps,
                       reinterpret_cast<const float4*>(wstream), small, pos, src_out, (float*)value_out, (float*)proj_out, M, S, B, npb, proj_width,
                       value_heads, out_bf16_hm);
    MSM_CHECK_LAUNCH("msm_encoder_prologue_fwd");
    return MSM_OK;
}
