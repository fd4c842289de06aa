// The mask step of the UCN path (PretrainedMeanShiftTransformerDecoder over SimpleBasePixelDecoder) with the 3x3 mask_features
// convolution folded into the query embedding (16-bit plans).
//
// Reference: mask_features = Conv3x3(64 -> 256, padding 1)(x)          pixel_decoder/fpn.py:238-246,283-290
//            mask[b,q,p]   = sum_o e[b,q,o] mask_features[b,o,p]        DEC:1012-1035 (einsum "bqc,bchw->bqhw")
//            attn_mask     = sigmoid(mask) < 0.5 at key resolution = mask resolution (no interpolation: every pixel is a key)
// Both steps are linear in x, so
//            mask[b,q,(y,x)] = sum_{dy,dx,c} F[b,q,(dy,dx),c] x[b,c,y+dy,x+dx] + e[b,q,:].bias,   F[b,q,(dy,dx),c] = sum_o e[b,q,o] W[o,c,dy,dx]
// -- a 3x3 convolution of the 64-channel embedding with Q per-image filters: K = 576 instead of 256 per (query, pixel), but it reads
// 128 B per pixel (the fp16 tokens the fused K/V attention reads anyway) instead of 512 B of a bf16 copy of mask_features, and the
// (B, 256, H, W) tensor -- 629 MB at batch 2 of 480x640, written by the convolution, re-read by the packing pass and by every one of
// the seven mask steps -- is never made.  F comes from one small GEMM per prediction (e (B Q, 256) x W (256, 576 + 1)).
//
// Kernel: a workgroup (8 waves) holds ONE image's F as fp16 MFMA fragments in LDS (16 q-rows x 576 k = 18 KiB per query block, 7 blocks =
// 126 KiB), converted from the fp32 GEMM output in the prologue.  A wave walks DOWN a 16-pixel column strip, two output rows per step:
// the nine taps of a row are the fp16 tokens of rows y-1 .. y+1 at x-1, x, x+1 -- eighteen 16-byte buffer loads per row
// (3 dx x 2 channel halves x ... per lane: pixel lj, channels 8 lq .. + 7 of the half), out-of-image taps read as zeros through
// an out-of-range buffer offset -- kept in registers in a rolling window of four rows, the next two rows in flight during a step's
// 2 x 126 v_mfma_f32_16x16x32_f16 (every F fragment read from LDS feeds both rows).  Output: the attention mask bit-packed and
// blocked exactly as hs_attn_fkv_kernel reads it (attention.hip: [B][1][S / 16][16 lj][8 m] uint16) + the row_any flags, or fp32
// logits (B, Q, H W) for the final prediction (the K kept queries only).
#include "bf16.h"
#include "common.h"

#ifndef MC_EXP
#define MC_EXP 0        // tuning builds (tools/probes/mask_conv_parts.sh), a bit mask: 1 no MFMAs, 2 no fragment reads in the loop, 4 no x loads in the loop, 8 no prologue conversion
#endif

namespace msm {

constexpr int MC_WAVES = 8, MC_THREADS = MC_WAVES * 64;
constexpr int MC_TAPS = 9, MC_KS = 2 * MC_TAPS;          // k-steps of 32: (tap, channel half)
constexpr int MC_K = 64 * MC_TAPS;                      // 576
constexpr int MC_QB_BYTES = MC_KS * 1024;               // one query block's fragments
constexpr int MC_MAXQB = 7;                             // 112 queries: one query chunk of the attention kernels
constexpr int MC_RING = 6;                              // F fragments in flight per wave (divides MC_KS)
constexpr int MC_SCRATCH = 64 + 512 + 64 + 512;        // wave-private LDS of the bits epilogue (see the kernel)

__device__ __forceinline__ void* mc_uniform_ptr64(const void* p) {
    const uint64_t u = (uint64_t)p;
    return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                   (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
}

typedef unsigned u32x2s __attribute__((ext_vector_type(2)));

struct McRow {                       // one image row's operands of a 16-pixel block: [dx + 1][channel half]
    u32x4b v[3][2];
};

// MODE 0: attention-mask bits + row_any; MODE 1: fp32 logits
// Both modes run the MFMAs as D[pixel 4 lq + r][query lj] (A = x, B = F): a lane holds four consecutive pixels of ONE query.
//   MODE 1 stores them as one float4 per lane.
//   MODE 0: the four sign bits are a nibble of the query's 16-bit word (bit = pixel); the words of the two output rows share one
//   32-bit value that is OR-ed across the four lane rows with v_permlane32_swap / v_permlane16_swap (gfx950), written to a
//   wave-private LDS block in the layout of the output and stored from there once per step (one 16-byte store per lane and row).
//   The whole epilogue of query block m - 1 (about 30 VALU instructions) is issued BETWEEN the MFMAs of block m -- two per MFMA pair --
//   instead of behind a drained matrix pipe.  (A logit of exactly -0.0 counts as masked: the sign bit is read, as in the fast
//   epilogue of the mask step)
template <int MODE>
__global__ __launch_bounds__(MC_THREADS) void mask_conv_fold_kernel(const unsigned short* __restrict__ xh, const float* __restrict__ F, int64_t ldf,
                                                                    int64_t f_sb, u32x4b* __restrict__ bits, int32_t* __restrict__ row_any,
                                                                    float* __restrict__ logits, int Q, int Himg, int Wimg, int nqb, int wgs_per_image,
                                                                    int seg, int nseg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mc_lds[];
    u32x4b* frag = reinterpret_cast<u32x4b*>(mc_lds);
    float* qbias = reinterpret_cast<float*>(mc_lds + (size_t)nqb * MC_QB_BYTES + MC_RING * 1024);     // (behind the ring's read-ahead slack)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    // wave-private scratch (MODE 0): 64 B of padding (block "-1" of the pipelined epilogue writes zeros there), the step's words
    // u16 [2 rows][16 lj][8 m] in the layout of the output, the unmasked-bit accumulators u32 [8 m][16 lj]
    unsigned char* scratch = mc_lds + (size_t)nqb * MC_QB_BYTES + MC_RING * 1024 + 16 * MC_MAXQB * sizeof(float) + (size_t)wave * MC_SCRATCH;
    unsigned short* wl = reinterpret_cast<unsigned short*>(scratch + 64);
    unsigned* anyl = reinterpret_cast<unsigned*>(scratch + 64 + 512 + 64);
    const int b = blockIdx.x / wgs_per_image, wg = blockIdx.x - b * wgs_per_image;
    const int S = Himg * Wimg;
    // ---- prologue: F of this image -> fp16 fragments in LDS.  Slot (qb, ks, lane): row q = 16 qb + (lane & 15), k = 32 ks + 8 (lane >> 4) .. + 7
    {
        const float* Fb = F + (int64_t)b * f_sb;
        // (eight slots per thread at a time: sixteen loads in flight -- one slot after the other exposes an L2 round trip per slot: 10 us)
        const int nslots = (MC_EXP & 8) ? 0 : nqb * MC_KS * 64;
        for (int s0 = tid; s0 < nslots; s0 += 8 * MC_THREADS) {
            float4 a[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * MC_THREADS;
                const int l = s & 63, ks = (s >> 6) % MC_KS, qb = (s >> 6) / MC_KS;
                const int q = 16 * qb + (l & 15), k0 = 32 * ks + 8 * (l >> 4);
                a[u] = c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s < nslots && q < Q) {
                    a[u] = *reinterpret_cast<const float4*>(Fb + (int64_t)q * ldf + k0);
                    c[u] = *reinterpret_cast<const float4*>(Fb + (int64_t)q * ldf + k0 + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * MC_THREADS;
                const u32x2b lo = pack4h(a[u].x, a[u].y, a[u].z, a[u].w), hi = pack4h(c[u].x, c[u].y, c[u].z, c[u].w);
                if (s < nslots) frag[s] = u32x4b{lo.x, lo.y, hi.x, hi.y};
            }
        }
        for (int q = tid; q < 16 * nqb; q += MC_THREADS) qbias[q] = q < Q ? Fb[(int64_t)q * ldf + MC_K] : 0.f;
        for (int i = lane; i < MC_SCRATCH / 4; i += 64) reinterpret_cast<unsigned*>(scratch)[i] = 0u;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(mc_uniform_ptr64(xh + (int64_t)b * S * 64), 0, (unsigned)S * 128u, 0x00020000);
    const int strips = Wimg / 16, tasks = strips * nseg;
    const int nkb = S / 16;
    const unsigned sh0 = 4u * lq, sh1 = 4u * lq + 16u;
    unsigned short* wl_lane = wl + ((lq & 1) * 16 + lj) * 8;            // lane rows 0 / 2 write output row y's word, 1 / 3 row y + 1's
    const unsigned wsel = 16u * (lq & 1);

    // the epilogue of query block m (MODE 0): c0 / c1 = its accumulators of output rows y / y + 1; ok = 0 for the block "-1" of the pipeline
    // (pure VALU: the LDS writes of the result are issued behind the block's ring reads -- a write in front of them orders every read behind it)
    auto sign_words = [&](const f32x4b& c0, const f32x4b& c1, int m, unsigned ok, unsigned& word, unsigned& unmasked) {
        unsigned n0 = __float_as_uint(c0[0]) >> 31, n1 = __float_as_uint(c1[0]) >> 31;
#pragma unroll
        for (int r = 1; r < 4; ++r) n0 |= (__float_as_uint(c0[r]) >> 31) << r, n1 |= (__float_as_uint(c1[r]) >> 31) << r;
        unsigned v = (n0 << sh0) | (n1 << sh1);
        const u32x2s p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        v = p.x | p.y;
        const u32x2s q2 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = q2.x | q2.y;                                                   // every lane row: [row y + 1's word | row y's word] of query 16 m + lj
        const unsigned okq = 16 * m + lj < Q ? ok : 0u;
        word = (v & okq) >> wsel;
        unmasked = ~v & okq;
    };
    auto put_words = [&](int m, unsigned word, unsigned unmasked) {
        wl_lane[m] = (unsigned short)word;
        atomicOr(anyl + m * 16 + lj, unmasked);                            // (result unused: ds_or_b32)
    };

    for (int task = wg * MC_WAVES + wave; task < tasks; task += wgs_per_image * MC_WAVES) {
        const int strip = task / nseg, sg = task - strip * nseg;
        const int x0 = 16 * strip, ya = sg * seg, yb = min(Himg, ya + seg);
        // lane offsets of the three dx taps within a row (bytes; 0xfffffff0: outside the image -> the buffer load returns zeros)
        unsigned xo[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = x0 + lj + d - 1;
            xo[d] = (xx >= 0 && xx < Wimg) ? (unsigned)xx * 128u + (unsigned)lq * 16u : 0xfffffff0u;
        }
        auto load_row = [&](McRow& r, int yy) {
            const bool ok = yy >= 0 && yy < Himg;
            const unsigned ro = ok ? (unsigned)yy * (unsigned)Wimg * 128u : 0u;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned o = (ok && xo[d] != 0xfffffff0u) ? ro + xo[d] : 0xfffffff0u;
                r.v[d][0] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0u, 0);
                r.v[d][1] = __builtin_amdgcn_raw_buffer_load_b128(xr, o == 0xfffffff0u ? o : o + 64u, 0u, 0);
            }
        };
        McRow w0, w1, w2, w3, p0, p1;
        load_row(w0, ya - 1), load_row(w1, ya), load_row(w2, ya + 1), load_row(w3, ya + 2);
        for (int y = ya; y < yb; y += 2) {
            __builtin_amdgcn_sched_barrier(0);
#if MC_EXP & 4
            p0 = w2, p1 = w3;
#else
            load_row(p0, y + 3), load_row(p1, y + 4);          // the next step's new rows: in flight during this step's MFMAs
#endif
            __builtin_amdgcn_sched_barrier(0);
            const bool two = y + 1 < yb;
            const unsigned okrows = two ? 0xffffffffu : 0x0000ffffu;
            // The F fragments of all query blocks are ONE linear stream of nqb * 18 KiB-sized reads: a ring of MC_RING fragments runs
            // MC_RING reads ahead of the MFMAs and straight across the query-block boundaries (the reads past the last block land in
            // the slack behind the fragments).  The query-block loop stays rolled: unrolled, its seven copies cost the registers the ring needs.
            const u32x4b* fq = frag + lane;
            u32x4b ring[MC_RING];
#pragma unroll
            for (int i = 0; i < MC_RING; ++i) ring[i] = fq[i * 64];
            f32x4b pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = pa0;             // the previous block's accumulators (its epilogue runs beside this block's MFMAs)
#pragma unroll 1
            for (int qb = 0; qb < nqb; ++qb) {
                const float qv = qbias[16 * qb + lj];                                                // D columns = queries lj
                f32x4b a0 = {qv, qv, qv, qv}, a1 = a0;
                // the previous block's epilogue in slices of two or three VALU instructions, one slice per k-step (issued in the shadow of that
                // step's MFMAs; a scheduling barrier per k-step keeps the slices and the ring reads where they are written)
                unsigned n0 = 0u, n1 = 0u, ev = 0u, ew = 0u, eu = 0u, okq = 0u;
                const unsigned okb = qb > 0 ? okrows : 0u;
#pragma unroll
                for (int ks = 0; ks < MC_KS; ++ks) {
                    const int tap = ks >> 1, hf = ks & 1, dy = tap / 3, dx = tap - 3 * dy;
                    const f16x8 f = __builtin_bit_cast(f16x8, ring[ks % MC_RING]);
                    const McRow& r0 = dy == 0 ? w0 : dy == 1 ? w1 : w2;        // output row y reads rows y - 1 + dy
                    const McRow& r1 = dy == 0 ? w1 : dy == 1 ? w2 : w3;        // output row y + 1
                    const f16x8 x0v = __builtin_bit_cast(f16x8, r0.v[dx][hf]), x1v = __builtin_bit_cast(f16x8, r1.v[dx][hf]);
#if MC_EXP & 1
                    a0[0] += __uint_as_float(__builtin_bit_cast(u32x4b, f).x ^ __builtin_bit_cast(u32x4b, x0v).x);
                    a1[0] += __uint_as_float(__builtin_bit_cast(u32x4b, f).y ^ __builtin_bit_cast(u32x4b, x1v).y);
#else
                    a0 = mfma_f16k32(x0v, f, a0);
                    a1 = mfma_f16k32(x1v, f, a1);
#endif
#if !(MC_EXP & 2)
                    ring[ks % MC_RING] = fq[(ks + MC_RING) * 64];
#endif
                    if (MODE == 0) {
                        if (ks < 4) n0 |= (__float_as_uint(pa0[ks]) >> 31) << ks;
                        else if (ks < 8) n1 |= (__float_as_uint(pa1[ks - 4]) >> 31) << (ks - 4);
                        else if (ks == 8) ev = (n0 << sh0) | (n1 << sh1);
                        else if (ks == 9) {
                            const u32x2s p = __builtin_amdgcn_permlane32_swap(ev, ev, false, false);
                            ev = p.x | p.y;
                        } else if (ks == 10) {
                            const u32x2s p = __builtin_amdgcn_permlane16_swap(ev, ev, false, false);
                            ev = p.x | p.y;                                // every lane row: [row y + 1's word | row y's word] of query 16 (qb - 1) + lj
                        } else if (ks == 11) okq = 16 * (qb - 1) + lj < Q ? okb : 0u;
                        else if (ks == 12) ew = (ev & okq) >> wsel;
                        else if (ks == 13) eu = ~ev & okq;
                        else if (ks == MC_KS - 1) put_words(qb - 1, ew, eu);       // (behind the block's last ring read: an LDS write orders every read behind it)
                    }
                    __builtin_amdgcn_sched_barrier(0);          // (left alone, the scheduler sinks every read to its use: two reads in flight)
                }
                fq += MC_KS * 64;
                if (MODE == 0) {
                    pa0 = a0, pa1 = a1;
                } else {
                    const int q = 16 * qb + lj;
                    if (q < Q) {
                        float* dst = logits + ((int64_t)b * Q + q) * S + (int64_t)y * Wimg + x0 + 4 * lq;
                        *reinterpret_cast<float4*>(dst) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                        if (two) *reinterpret_cast<float4*>(dst + Wimg) = make_float4(a1[0], a1[1], a1[2], a1[3]);
                    }
                }
            }
            if (MODE == 0) {
                unsigned ew, eu;
                sign_words(pa0, pa1, nqb - 1, okrows, ew, eu);
                put_words(nqb - 1, ew, eu);
                // lanes 0..15 store row y's block, lanes 16..31 row y + 1's (LDS operations of a wave complete in order)
                const u32x4b wv = *reinterpret_cast<const u32x4b*>(wl + ((lq & 1) * 16 + lj) * 8);
                const int row = y + (lq & 1);
                if (lq < 2 && (lq == 0 || two)) bits[((int64_t)b * nkb + ((row * Wimg + x0) >> 4)) * 16 + lj] = wv;
            }
            w0 = w2, w1 = w3, w2 = p0, w3 = p1;
        }
    }
    if (MODE == 0 && lq == 0) {
        for (int m = 0; m < nqb; ++m) {
            const int q = 16 * m + lj;
            if (q < Q && anyl[m * 16 + lj]) row_any[(int64_t)b * Q + q] = 1;
        }
    }
}

}  // namespace msm

using namespace msm;

static int mc_cus() {
    static int cache[64];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < 64) cache[dev] = n;
    return n;
}

extern "C" int msm_mask_conv3x3_folded(const void* x_f16, const float* F, int64_t ldf, int64_t f_sb, void* mask_bits, int32_t* row_any,
                                       int row_any_cleared, float* logits, int B, int Q, int H, int W, void* stream) {
    const char* who = "msm_mask_conv3x3_folded";
    MSM_REQUIRE(x_f16 && F && B > 0 && Q > 0 && H > 0 && W > 0, "%s: bad arguments", who);
    MSM_REQUIRE((mask_bits != nullptr) != (logits != nullptr), "%s: exactly one of mask_bits / logits", who);
    MSM_REQUIRE(!mask_bits || row_any, "%s: row_any is required with mask_bits", who);
    MSM_REQUIRE(Q <= 16 * MC_MAXQB, "%s: Q=%d > %d queries", who, Q, 16 * MC_MAXQB);
    MSM_REQUIRE(W % 16 == 0, "%s: W=%d must be a multiple of 16", who, W);
    MSM_REQUIRE(ldf >= MC_K + 1 && ldf % 4 == 0 && f_sb % 4 == 0, "%s: F rows hold 576 filter taps + the per-query constant, ldf %% 4 == 0", who);
    MSM_REQUIRE((int64_t)H * W * 128 < ((int64_t)1 << 32) - 256, "%s: one image of x must stay below 4 GiB (32-bit buffer offsets)", who);
    MSM_REQUIRE(((((uintptr_t)x_f16) | ((uintptr_t)F) | ((uintptr_t)mask_bits) | ((uintptr_t)logits)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    hipStream_t st = (hipStream_t)stream;
    const int nqb = cdiv(Q, 16);
    const size_t lds = (size_t)nqb * MC_QB_BYTES + MC_RING * 1024 + 16 * MC_MAXQB * sizeof(float) + (size_t)MC_WAVES * MC_SCRATCH;
    const int cus = mc_cus();
    const int wgs_per_image = max(1, cus / B);
    const int strips = W / 16;
    const int64_t units = (int64_t)strips * H, waves = (int64_t)wgs_per_image * MC_WAVES;
    int seg = (int)(2 * ((units + 2 * waves - 1) / (2 * waves)));           // rows per task: even, about one task per wave
    seg = max(2, min(seg, ((H + 1) / 2) * 2));
    const int nseg = cdiv(H, seg);
    if (mask_bits && !row_any_cleared) MSM_CHECK_HIP(hipMemsetAsync(row_any, 0, sizeof(int32_t) * (size_t)B * Q, st));
    dim3 grid(B * wgs_per_image), block(MC_THREADS);
    if (mask_bits) {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)mask_conv_fold_kernel<0>, lds));
        hipLaunchKernelGGL(mask_conv_fold_kernel<0>, grid, block, lds, st, (const unsigned short*)x_f16, F, ldf, f_sb, (u32x4b*)mask_bits, row_any,
                           (float*)nullptr, Q, H, W, nqb, wgs_per_image, seg, nseg);
    } else {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)mask_conv_fold_kernel<1>, lds));
        hipLaunchKernelGGL(mask_conv_fold_kernel<1>, grid, block, lds, st, (const unsigned short*)x_f16, F, ldf, f_sb, (u32x4b*)nullptr,
                           (int32_t*)nullptr, logits, Q, H, W, nqb, wgs_per_image, seg, nseg);
    }
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}
