// Folded K/V projection of one feature level (see include/msm_hip.h: msm_kv_project_f32):
//     out[b][p][n] = sum_k x[b][k][p] * w[n][k] + cmat[p][n],     k < 64, n < N (N = 2E = 512: [K | V])
//
// Reference: the memory/key path of the cross-attention layers -- input_proj (1x1 conv) + level_embed (DEC:575),
// "+ pos" (DEC:251) and the k/v in-projections (AU:134-140); everything affine in the 64-channel level feature is
// folded into w (N, 64) and the per-position constant cmat (HW, N) on the host (modeling._folded_kv).
//
// Why not the generic GEMM: K = 64 is one or two k-steps, so a tiled GEMM workgroup is a load -> wait -> 64 MFMAs
// -> store sequence with nothing to overlap (measured 66 us for 60x80 x 8 images = 1.5 TB/s of the 79 MB it
// writes).  Here the weight is the stationary operand: every workgroup copies the whole w (128 KiB) into LDS once
// and its 16 waves (4 per SIMD) then stream (position tile, image, feature half) units:
//   * MFMA orientation D^T: rows = output features (A = w from LDS, ds_read_b128 with a 68-float row stride),
//     cols = 16 tokens (B = x read straight from NCHW: for a fixed channel 16 consecutive pixels = 64 B);
//     K order k = lq*16 + s for both operands; a lane ends with 4 consecutive features of one token, so cmat is
//     the accumulator's initial value (one 16-byte load) and the result leaves as one 16-byte store;
//   * a wave holds the 16 x-values of its token tile in registers for all 16 feature blocks of the unit;
//   * units are ordered image-fastest, so the 8 images of a position tile read the same cmat rows out of L2.
// Measured (B = 8, 60x80): 36-39 us against 62-66 us for the tiled GEMM; with the MFMAs removed the kernel still
// takes 29 us, i.e. it now sits on the 98 MB it has to move (79 MB of them written).
#include <type_traits>

#include "bf16.h"
#include "common.h"

#ifndef KP_EXP
#define KP_EXP 0   // tuning experiments only (tools/probes/kv_parts.sh): 1 = no stores, 2 = no MFMAs
#endif

namespace msm {

constexpr int KP_K = 64;
constexpr int KP_LD = KP_K + 4;      // LDS row stride of w (floats): 16 rows x b128 conflict-free
constexpr int KP_FB = 16;            // feature blocks (of 16) per unit: 256 features
constexpr int KP_W = 16;             // waves per workgroup (one workgroup per CU: w takes 136 KiB of LDS)

// the projection of ONE (level, layer) job by workgroup `wg` of the `nwg` workgroups assigned to it.
// OT = float, or uint16_t: the result is stored as bf16 (low-precision mode: half the bytes of this write-bound kernel and of
// the attention kernels' K/V reads; the products are still exact fp32 MFMAs, only the stored value is rounded)
// SEP: the constant is SEPARABLE, cmat = [HW / cw row vectors | cw column vectors] x N and the constant of token p = (y, x) is
// row[y] + col[x] -- what the sine position embedding gives (its first half depends on y only, its second on x only,
// position_encoding.py:44-51), so a 307 200-key map reads two tables of 1120 rows instead of 629 MB of constants (as many
// bytes as it writes).  The row vector is the accumulator's initial value, the column vector is added before the store.
template <typename OT, bool SEP>
__device__ __forceinline__ void kv_project_body(const float* __restrict__ x, const float* __restrict__ w,
                                                const float* __restrict__ cmat, int cw, OT* __restrict__ out, int B, int HW, int N,
                                                int tokens, int64_t x_sb, int wg, int nwg, float* wl) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < N * (KP_K / 4); i += KP_W * 64) {
        const int n = i >> 4, c4 = i & 15;
        *reinterpret_cast<float4*>(wl + n * KP_LD + c4 * 4) = *reinterpret_cast<const float4*>(w + (int64_t)n * KP_K + c4 * 4);
    }
    __syncthreads();

    const int tiles = (HW + 15) / 16;
    const int halves = N / (KP_FB * 16);
    const int units = tiles * B * halves;
    // full rounds over all waves; the leftover units go one per SIMD across all workgroups first (waves w, w+4, ...
    // share a SIMD), so no SIMD runs two leftovers while another runs none
    const int slots = nwg * KP_W;
    const int full_rounds = units / slots;
    const int left = units - full_rounds * slots;
    const int left_slot = (wave >> 2) * (nwg * 4) + wg * 4 + (wave & 3);
    const int mine = full_rounds + (left_slot < left ? 1 : 0);
    auto unit_of = [&](int it) {
        return (it < full_rounds) ? it * slots + wg * KP_W + wave : full_rounds * slots + left_slot;
    };
    auto load_x = [&](int u, float (&xv)[16]) {
        const int img = (u / halves) % B, tile = u / (halves * B);
        const int p = min(tile * 16 + lj, HW - 1);
        // B operand: x[img][k = lq*16 + s][p], s = 0..15 (NCHW), or x[img][p][k] (token-major / channels-last:
        // the 16 values are four 16-byte loads)
        if (tokens) {
            const float* xp = x + (int64_t)img * x_sb + (int64_t)p * KP_K + lq * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 t = *reinterpret_cast<const float4*>(xp + s4 * 4);
                xv[s4 * 4 + 0] = t.x; xv[s4 * 4 + 1] = t.y; xv[s4 * 4 + 2] = t.z; xv[s4 * 4 + 3] = t.w;
            }
        } else {
            const float* xp = x + (int64_t)img * x_sb + (int64_t)lq * 16 * HW + p;
#pragma unroll
            for (int s = 0; s < 16; ++s) xv[s] = xp[(int64_t)s * HW];
        }
    };
    float xb[16], xn[16];
    if (mine > 0) load_x(unit_of(0), xb);
    for (int it = 0; it < mine; ++it) {
        const int u = unit_of(it);
        const int half = u % halves;
        const int tile = u / (halves * B), img = (u / halves) % B;
        const int p = min(tile * 16 + lj, HW - 1);             // this lane's token, clamped: lanes past the last token repeat it (see the stores)
        const int n_base = half * KP_FB * 16;
        const int py = SEP ? p / cw : 0;
        const float* cp = cmat + (int64_t)(SEP ? py : p) * N + n_base + lq * 4;
        const float* cq = cmat + (int64_t)(SEP ? HW / cw + (p - py * cw) : 0) * N + n_base + lq * 4;      // (SEP only)
        OT* op = out + ((int64_t)img * HW + p) * N + n_base + lq * 4;
        const float* wp = wl + (n_base + lj) * KP_LD + lq * 16;
        // everything this unit reads from memory is requested before its first MFMA; the next unit's x rides along
        float4 cm[KP_FB];
#pragma unroll
        for (int fb = 0; fb < KP_FB; ++fb) cm[fb] = *reinterpret_cast<const float4*>(cp + fb * 16);
        load_x(unit_of(min(it + 1, mine - 1)), xn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fb = 0; fb < KP_FB; fb += 2) {
            // two feature blocks in flight: consecutive MFMAs alternate accumulators
            f32x4 a0 = f32x4{cm[fb].x, cm[fb].y, cm[fb].z, cm[fb].w};
            f32x4 a1 = f32x4{cm[fb + 1].x, cm[fb + 1].y, cm[fb + 1].z, cm[fb + 1].w};
            float4 q0, q1;                                   // the column vectors of this pair: requested here, added after the MFMAs
            if constexpr (SEP) {
                q0 = *reinterpret_cast<const float4*>(cq + fb * 16);
                q1 = *reinterpret_cast<const float4*>(cq + (fb + 1) * 16);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 w0 = *reinterpret_cast<const float4*>(wp + fb * 16 * KP_LD + s4 * 4);
                const float4 w1 = *reinterpret_cast<const float4*>(wp + (fb + 1) * 16 * KP_LD + s4 * 4);
#if KP_EXP == 2
                a0[0] += w0.x * xb[s4 * 4 + 0] + w0.y * xb[s4 * 4 + 1];
                a1[0] += w1.z * xb[s4 * 4 + 2] + w1.w * xb[s4 * 4 + 3];
                continue;
#endif
                a0 = mfma16(w0.x, xb[s4 * 4 + 0], a0);
                a1 = mfma16(w1.x, xb[s4 * 4 + 0], a1);
                a0 = mfma16(w0.y, xb[s4 * 4 + 1], a0);
                a1 = mfma16(w1.y, xb[s4 * 4 + 1], a1);
                a0 = mfma16(w0.z, xb[s4 * 4 + 2], a0);
                a1 = mfma16(w1.z, xb[s4 * 4 + 2], a1);
                a0 = mfma16(w0.w, xb[s4 * 4 + 3], a0);
                a1 = mfma16(w1.w, xb[s4 * 4 + 3], a1);
            }
            // no `live` test: lanes past the last token hold token HW - 1 again (clamped p, same x, same constant) and store the
            // same values to the same address -- a branch here cuts the unit into eight basic blocks (LDS reads -> wait -> 32
            // MFMAs -> stores, nothing overlapping across them)
            if constexpr (SEP) {
                a0 += f32x4{q0.x, q0.y, q0.z, q0.w};
                a1 += f32x4{q1.x, q1.y, q1.z, q1.w};
            }
#if KP_EXP == 1
            if (a0[0] == 12345.f && a1[1] == 5.f)
#endif
#if KP_EXP == 3     // what would whole-line stores buy?  the same bytes, a wave instruction = 1 KiB contiguous (values land in the wrong places)
            if constexpr (std::is_same<OT, float>::value) {
                float* ob = out + ((int64_t)img * HW + min(tile * 16, HW - 16)) * N + n_base * 16;
                *reinterpret_cast<float4*>(ob + (fb * 64 + lane) * 4) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                *reinterpret_cast<float4*>(ob + ((fb + 1) * 64 + lane) * 4) = make_float4(a1[0], a1[1], a1[2], a1[3]);
            } else
#endif
            if constexpr (std::is_same<OT, float>::value) {
                *reinterpret_cast<float4*>(op + fb * 16) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                *reinterpret_cast<float4*>(op + (fb + 1) * 16) = make_float4(a1[0], a1[1], a1[2], a1[3]);
            } else {
                *reinterpret_cast<bf16x4*>(op + fb * 16) = pack4(a0[0], a0[1], a0[2], a0[3]);
                *reinterpret_cast<bf16x4*>(op + (fb + 1) * 16) = pack4(a1[0], a1[1], a1[2], a1[3]);
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) xb[s] = xn[s];
    }
}

template <bool SEP>
__global__ __launch_bounds__(KP_W * 64) void kv_project_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ cmat, int cw, float* __restrict__ out, int B,
                                                         int HW, int N, int tokens, int64_t x_sb) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [N][KP_LD]
    kv_project_body<float, SEP>(x, w, cmat, cw, out, B, HW, N, tokens, x_sb, blockIdx.x, gridDim.x, wl);
}

// All K/V projections of the decoder (one job per cross-attention layer: its level's features, its folded weight and
// constant) in ONE launch.  They depend only on the pixel decoder's output, and one by one the coarse levels are
// latency bound (15x20: 5 MB in 13 us, 30x40: 20 MB in 19.5 us; nine launches: 211 us per step).  Every job gets a share of
// the chip's workgroups proportional to the bytes it writes; a workgroup copies its job's weight into LDS and streams
// that job's units.
constexpr int KP_MAXJ = 16;
struct KvJobs {
    int n;
    const float* x[KP_MAXJ];
    const float* w[KP_MAXJ];
    const float* cmat[KP_MAXJ];
    void* out[KP_MAXJ];
    int HW[KP_MAXJ], tokens[KP_MAXJ], first[KP_MAXJ + 1];
    int cw[KP_MAXJ];             // width of the map when the constant is separable (see kv_project_body), else 0
    int64_t x_sb[KP_MAXJ];
};
template <typename OT, bool SEP>
__global__ __launch_bounds__(KP_W * 64) void kv_project_multi_kernel(KvJobs jobs, int B, int N) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [N][KP_LD]
    int j = 0;
#pragma unroll
    for (int i = 1; i < KP_MAXJ; ++i) j += (i < jobs.n && (int)blockIdx.x >= jobs.first[i]) ? 1 : 0;
    kv_project_body<OT, SEP>(jobs.x[j], jobs.w[j], jobs.cmat[j], jobs.cw[j], (OT*)jobs.out[j], B, jobs.HW[j], N, jobs.tokens[j], jobs.x_sb[j],
                             (int)blockIdx.x - jobs.first[j], jobs.first[j + 1] - jobs.first[j], wl);
}

// ---- the same projection on the bf16 matrix pipe ---------------------------------------------------------------------------
// MODE 0 (precision "f32_split"): fp32 accuracy -- x and w as exact three-term bf16 splits, six K = 32 MFMAs per product
// (bf16.h); 192 MFMAs of 16 cycles per (16 tokens x 256 features) unit instead of 256 of 32: the launch moves from the
// fp32 MFMA's bound (63 us of matrix time at B = 8) to its 309-MB write.  MODE 1 (precision "bf16"): w rounded to one bf16,
// x as hi + lo, two MFMAs per product, bf16 output.
// The weight lives in LDS as bf16 copies, rows of 64 + 8 (144 B: 16 rows x ds_read_b128 conflict-free); a workgroup keeps ONE
// 256-feature half of its job's weight (three copies: 108 KiB) and takes the units of that half -- workgroup parity picks
// the half.  K order k = lq*16 + G*8 + e on both operands, so a lane's 16 x-values are its two B operands as they are loaded.
constexpr int KS_LD = KP_K + 8;      // LDS row stride of a bf16 weight copy (elements)
constexpr int KS_W = 8;              // waves per workgroup of the bf16-pipe kernel
constexpr int KS_TR = 64 + 8;        // row stride (bf16 elements) of a wave's output transposition tile: 144 B = 36 banks -- a multiple of 16 B, so the ds_read_b128 of the
                                     // store phase is naturally aligned on every row; the 16 token rows x 2 quads of a ds_write_b64 half-wave land on 64 distinct banks

template <typename OT, int MODE, bool SEP>
__device__ __forceinline__ void kv_project_split_body(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ cmat, int cw, OT* __restrict__ out, int B, int HW, int N,
                                                      int tokens, int64_t x_sb, int wg, int nwg, unsigned short* wl) {
    // MODE 2 (precision "f16"): w and x as ONE IEEE-half term each on v_mfma_f32_16x16x32_f16; the K half of [K | V] leaves as half
    constexpr int COPIES = MODE == 0 ? 3 : 1;
    constexpr unsigned TERMS = MODE == 0 ? 0x3fu : 0x30u;      // mac_term bits: all six | {wh xm, wh xh}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int halves = N / (KP_FB * 16);
    // this workgroup's half of the features (two halves: even / odd workgroups; a job given ONE workgroup walks both in turn)
    const int nh = nwg >= halves ? 1 : halves;                  // halves this workgroup covers
    const int h0 = nwg >= halves ? wg % halves : 0;
    const int wg_h = nwg >= halves ? wg / halves : 0, nwg_h = nwg >= halves ? (nwg - h0 + halves - 1) / halves : 1;
    const int tiles = (HW + 15) / 16;
    for (int hh = 0; hh < nh; ++hh) {
        const int half = h0 + hh;
        const int n_base = half * KP_FB * 16;
        if (hh > 0) __syncthreads();
        for (int i = tid; i < KP_FB * 16 * (KP_K / 4); i += KS_W * 64) {
            const int n = i >> 4, c4 = i & 15;
            const float4 v = *reinterpret_cast<const float4*>(w + (int64_t)(n_base + n) * KP_K + c4 * 4);
            unsigned short* dst = wl + n * KS_LD + c4 * 4;
            if constexpr (MODE == 2) {
                *reinterpret_cast<u32x2b*>(dst) = pack4h(v.x, v.y, v.z, v.w);
                continue;
            }
            const Split3 sp = split3(v.x, v.y, v.z, v.w);
            *reinterpret_cast<bf16x4*>(dst) = sp.h;
            if constexpr (COPIES > 1) {
                *reinterpret_cast<bf16x4*>(dst + KP_FB * 16 * KS_LD) = sp.m;
                *reinterpret_cast<bf16x4*>(dst + 2 * KP_FB * 16 * KS_LD) = sp.l;
            }
        }
        __syncthreads();
        const int units = tiles * B;
        const int slots = nwg_h * KS_W;
        const int full_rounds = units / slots;
        const int left = units - full_rounds * slots;
        const int left_slot = (wave >> 2) * (nwg_h * 4) + wg_h * 4 + (wave & 3);
        const int mine = full_rounds + (left_slot < left ? 1 : 0);
        auto unit_of = [&](int it) { return (it < full_rounds) ? it * slots + wg_h * KS_W + wave : full_rounds * slots + left_slot; };
        auto load_x = [&](int u, float (&xv)[16]) {
            const int img = u % B, tile = u / B;
            const int p = min(tile * 16 + lj, HW - 1);
            if (tokens) {
                const float* xp = x + (int64_t)img * x_sb + (int64_t)p * KP_K + lq * 16;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float4 t = *reinterpret_cast<const float4*>(xp + s4 * 4);
                    xv[s4 * 4 + 0] = t.x; xv[s4 * 4 + 1] = t.y; xv[s4 * 4 + 2] = t.z; xv[s4 * 4 + 3] = t.w;
                }
            } else {
                const float* xp = x + (int64_t)img * x_sb + (int64_t)lq * 16 * HW + p;
#pragma unroll
                for (int s = 0; s < 16; ++s) xv[s] = xp[(int64_t)s * HW];
            }
        };
        // (a wave gets a few units: nothing to prefetch across units; 8 waves per workgroup -- the 16 of the fp32 kernel leave
        // 128 registers per lane, which the three-term operands do not fit: 80 - 350 spilled registers, 150 - 470 us)
        for (int it = 0; it < mine; ++it) {
            const int u = unit_of(it);
            const int tile = u / B, img = u % B;
            const int p = min(tile * 16 + lj, HW - 1);
            const int py = SEP ? p / cw : 0;
            const float* cp = cmat + (int64_t)(SEP ? py : p) * N + n_base + lq * 4;
            const float* cq = cmat + (int64_t)(SEP ? HW / cw + (p - py * cw) : 0) * N + n_base + lq * 4;      // (SEP only: kv_project_body)
            OT* op = out + ((int64_t)img * HW + p) * N + n_base + lq * 4;
            float xb[16];
            load_x(u, xb);
            float4 cm[KP_FB];
#pragma unroll
            for (int fb = 0; fb < KP_FB; ++fb) cm[fb] = *reinterpret_cast<const float4*>(cp + fb * 16);
            Split3x8 xs[2];
            f16x8 xf[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if constexpr (MODE == 2)
                    xf[g] = cvt8h(xb[8 * g], xb[8 * g + 1], xb[8 * g + 2], xb[8 * g + 3], xb[8 * g + 4], xb[8 * g + 5], xb[8 * g + 6], xb[8 * g + 7]);
                else
                    xs[g] = join(split3(xb[8 * g], xb[8 * g + 1], xb[8 * g + 2], xb[8 * g + 3]), split3(xb[8 * g + 4], xb[8 * g + 5], xb[8 * g + 6], xb[8 * g + 7]));
            }
            const unsigned short* wp = wl + lj * KS_LD + lq * 16;
            unsigned short* tr = wl + COPIES * KP_FB * 16 * KS_LD + wave * (16 * KS_TR);       // this wave's [16 tokens][KS_TR] transposition tile
            float4 qn[2];                                 // SEP: the column vectors of the NEXT pair of feature blocks (one pair ahead:
            if constexpr (SEP) {                          // with few MFMAs per pair their latency would otherwise be exposed)
                qn[0] = *reinterpret_cast<const float4*>(cq);
                qn[1] = *reinterpret_cast<const float4*>(cq + 16);
            }
#pragma unroll
            for (int fb = 0; fb < KP_FB; fb += 2) {
                // two feature blocks in flight: consecutive MFMAs alternate accumulators; low-order terms apart from the leading one
                f32x4 hi[2], lo[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hi[j] = f32x4{cm[fb + j].x, cm[fb + j].y, cm[fb + j].z, cm[fb + j].w};
                    lo[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (SEP) {                  // the column vector rides in the low-order accumulator
                        lo[j] = f32x4{qn[j].x, qn[j].y, qn[j].z, qn[j].w};
                        qn[j] = *reinterpret_cast<const float4*>(cq + min(fb + 2 + j, KP_FB - 1) * 16);
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    Frag3 wf[2];
                    if constexpr (MODE == 2) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const u32x4b wq = *reinterpret_cast<const u32x4b*>(wp + (fb + j) * 16 * KS_LD + g * 8);
                            hi[j] = mfma_f16k32(__builtin_bit_cast(f16x8, wq), xf[g], hi[j]);
                        }
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned short* q = wp + (fb + j) * 16 * KS_LD + g * 8;
                        wf[j].h = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(q));
                        wf[j].m = COPIES > 1 ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(q + KP_FB * 16 * KS_LD)) : wf[j].h;
                        wf[j].l = COPIES > 1 ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(q + 2 * KP_FB * 16 * KS_LD)) : wf[j].h;
                    }
#pragma unroll
                    for (int term = 0; term < 6; ++term)
                        if ((TERMS >> term) & 1u) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) mac_term(term, lo[j], hi[j], wf[j], xs[g]);
                        }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {          // (no `live` test: see kv_project_body)
                    const f32x4 a = hi[j] + lo[j];
#if KP_EXP == 1
                    if (a[0] == 12345.f)
#endif
                    if constexpr (std::is_same<OT, float>::value) *reinterpret_cast<float4*>(op + (fb + j) * 16) = make_float4(a[0], a[1], a[2], a[3]);
                    else if (MODE == 2 && half == 0) *reinterpret_cast<u32x2b*>(tr + lj * KS_TR + (((fb + j) & 3) * 16 + lq * 4)) = pack4h(a[0], a[1], a[2], a[3]);
                    else *reinterpret_cast<bf16x4*>(tr + lj * KS_TR + (((fb + j) & 3) * 16 + lq * 4)) = pack4(a[0], a[1], a[2], a[3]);
                }
                if constexpr (!std::is_same<OT, float>::value) {
                    // bf16 result: a lane's four values are 8 bytes, a token's run per store instruction 32 bytes -- a quarter of a
                    // line (measured: 76 us as direct stores, 61 with the same bytes as whole lines, 46 without stores).  Four
                    // feature blocks (64 features = 128 B per token) pass through a wave-private LDS tile and leave as whole lines:
                    // lane -> (token l >> 3 (+ 8), 16-byte chunk l & 7).  Rows past the last token hold token HW - 1 again and
                    // store the same values to the same address, as above.
                    if ((fb & 2) != 0) {
#if KP_EXP == 1
                        if (xb[0] == 12345.f)
#endif
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int r = (lane >> 3) + 8 * h;
                            const u32x4b v = *reinterpret_cast<const u32x4b*>(tr + r * KS_TR + (lane & 7) * 8);
                            const int pr = min(tile * 16 + r, HW - 1);
                            *reinterpret_cast<u32x4b*>(out + ((int64_t)img * HW + pr) * N + n_base + (fb - 2) * 16 + (lane & 7) * 8) = v;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one pair of feature blocks at a time: left alone hipcc hoists the fragment
                                                        // reads of all sixteen to the top (96 x 4 registers: 250 spilled)
            }
        }
    }
}

template <typename OT, int MODE, bool SEP>
__global__ __launch_bounds__(KS_W * 64) void kv_project_multi_split_kernel(KvJobs jobs, int B, int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wls[];   // [copies][256][KS_LD] bf16
    int j = 0;
#pragma unroll
    for (int i = 1; i < KP_MAXJ; ++i) j += (i < jobs.n && (int)blockIdx.x >= jobs.first[i]) ? 1 : 0;
    kv_project_split_body<OT, MODE, SEP>(jobs.x[j], jobs.w[j], jobs.cmat[j], jobs.cw[j], (OT*)jobs.out[j], B, jobs.HW[j], N, jobs.tokens[j],
                                         jobs.x_sb[j], (int)blockIdx.x - jobs.first[j], jobs.first[j + 1] - jobs.first[j], wls);
}

// ---- mask_features: GroupNorm + ReLU of the FPN output fused into the 1x1 convolution that follows it --------------
//     out[b][n][p] = bias[n] + sum_k w[n][k] * relu((x[b][p][k] - mean_g) * rstd_g * gamma[k] + beta[k])      (MSD:349-358)
// Same weight-stationary scheme as above with the MFMA operands swapped (rows = tokens, cols = output channels), so a
// lane ends with 4 consecutive TOKENS of one channel and the NCHW result leaves as 16-byte stores; the normalisation
// is applied to the x fragment in registers, which removes the GroupNorm-apply pass (39 MB written and read back).
// stats: per (image, channel) double (sum, sum of squares) over the map, as written by msm_groupnorm_stats_f32.
constexpr int MF_W = 8;              // waves per workgroup (w for N = 256 is 68 KiB: two workgroups per CU)

__global__ __launch_bounds__(MF_W * 64) void tokens_proj_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias,
                                                                    const double* __restrict__ stats,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float* __restrict__ out, int B, int HW, int N, int groups,
                                                                    float eps, int relu) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [N][KP_LD], then per-image (scale, shift) [B][64][2]
    float* aff = wl + N * KP_LD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < N * (KP_K / 4); i += MF_W * 64) {
        const int n = i >> 4, c4 = i & 15;
        *reinterpret_cast<float4*>(wl + n * KP_LD + c4 * 4) = *reinterpret_cast<const float4*>(w + (int64_t)n * KP_K + c4 * 4);
    }
    for (int i = tid; i < B * KP_K; i += MF_W * 64) {
        float sc = 1.f, sh = 0.f;
        if (stats) {
            const int b = i / KP_K, c = i - b * KP_K;
            const int cpg = KP_K / groups, g0 = (c / cpg) * cpg;
            double sm = 0.0, q = 0.0;
            for (int k = 0; k < cpg; ++k) {
                sm += stats[((int64_t)b * KP_K + g0 + k) * 2];
                q += stats[((int64_t)b * KP_K + g0 + k) * 2 + 1];
            }
            const double cnt = (double)cpg * (double)HW;
            const double mean = sm / cnt;
            double var = q / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
            sh = beta[c] - (float)mean * sc;       // y = x*sc + sh  ( = (x - mean)*rstd*gamma + beta up to one rounding)
        }
        aff[i * 2] = sc;
        aff[i * 2 + 1] = sh;
    }
    __syncthreads();

    const int tiles = (HW + 15) / 16;
    const int halves = N / (KP_FB * 16);
    const int units = tiles * B * halves;
    const int slots = gridDim.x * MF_W;
    const int full_rounds = units / slots;
    const int left = units - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int mine = full_rounds + (left_slot < left ? 1 : 0);
    auto unit_of = [&](int it) {
        return (it < full_rounds) ? it * slots + (int)blockIdx.x * MF_W + wave : full_rounds * slots + left_slot;
    };
    auto load_x = [&](int u, float (&xv)[16]) {
        const int img = (u / halves) % B, tile = u / (halves * B);
        const int p = min(tile * 16 + lj, HW - 1);
        const float* xp = x + ((int64_t)img * HW + p) * KP_K + lq * 16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 t = *reinterpret_cast<const float4*>(xp + s4 * 4);
            xv[s4 * 4 + 0] = t.x; xv[s4 * 4 + 1] = t.y; xv[s4 * 4 + 2] = t.z; xv[s4 * 4 + 3] = t.w;
        }
    };
    float xb[16], xn[16];
    if (mine > 0) load_x(unit_of(0), xb);
    for (int it = 0; it < mine; ++it) {
        const int u = unit_of(it);
        const int half = u % halves;
        const int tile = u / (halves * B), img = (u / halves) % B;
        load_x(unit_of(min(it + 1, mine - 1)), xn);
        // A operand: this lane's 16 channels lq*16 .. +15 of token lj, normalised in registers
        const float* af = aff + (img * KP_K + lq * 16) * 2;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float y = fmaf(xb[s], af[2 * s], af[2 * s + 1]);
            xb[s] = relu ? fmaxf(y, 0.f) : y;
        }
        const int n_base = half * KP_FB * 16;
        const int p4 = tile * 16 + lq * 4;                        // first of this lane's 4 output tokens
        const float* wp = wl + (n_base + lj) * KP_LD + lq * 16;
        float* op = out + ((int64_t)img * N + n_base + lj) * HW + p4;
        const bool live = p4 < HW;                                // HW % 4 == 0: whole float4s are in or out
#pragma unroll
        for (int fb = 0; fb < KP_FB; fb += 2) {
            const float b0 = bias ? bias[n_base + fb * 16 + lj] : 0.f, b1 = bias ? bias[n_base + (fb + 1) * 16 + lj] : 0.f;
            f32x4 a0 = f32x4{b0, b0, b0, b0}, a1 = f32x4{b1, b1, b1, b1};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 w0 = *reinterpret_cast<const float4*>(wp + fb * 16 * KP_LD + s4 * 4);
                const float4 w1 = *reinterpret_cast<const float4*>(wp + (fb + 1) * 16 * KP_LD + s4 * 4);
                a0 = mfma16(xb[s4 * 4 + 0], w0.x, a0);
                a1 = mfma16(xb[s4 * 4 + 0], w1.x, a1);
                a0 = mfma16(xb[s4 * 4 + 1], w0.y, a0);
                a1 = mfma16(xb[s4 * 4 + 1], w1.y, a1);
                a0 = mfma16(xb[s4 * 4 + 2], w0.z, a0);
                a1 = mfma16(xb[s4 * 4 + 2], w1.z, a1);
                a0 = mfma16(xb[s4 * 4 + 3], w0.w, a0);
                a1 = mfma16(xb[s4 * 4 + 3], w1.w, a1);
            }
            if (live) {
                *reinterpret_cast<float4*>(op + (int64_t)fb * 16 * HW) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                *reinterpret_cast<float4*>(op + (int64_t)(fb + 1) * 16 * HW) = make_float4(a1[0], a1[1], a1[2], a1[3]);
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) xb[s] = xn[s];
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_kv_project_f32(const float* x, const float* w, const float* cmat, float* out, int B, int C, int HW, int N,
                                  int x_tokens, int64_t x_batch_stride, int cmat_width, void* stream) {
    MSM_REQUIRE(x && w && cmat && out, "msm_kv_project_f32: null pointer");
    MSM_REQUIRE(cmat_width >= 0 && (cmat_width == 0 || HW % cmat_width == 0), "msm_kv_project_f32: cmat_width=%d must divide HW=%d", cmat_width, HW);
    MSM_REQUIRE(C == KP_K, "msm_kv_project_f32: C=%d, only 64 input channels are supported", C);
    MSM_REQUIRE(B > 0 && HW > 0 && N > 0 && N % (KP_FB * 16) == 0 && N <= 512,
                "msm_kv_project_f32: N=%d must be 256 or 512", N);
    MSM_REQUIRE(((((uintptr_t)w) | ((uintptr_t)cmat) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)x) & 3) == 0,
                "msm_kv_project_f32: w/cmat/out must be 16-byte aligned");
    MSM_REQUIRE(x_batch_stride >= (int64_t)C * HW && (!x_tokens || ((((uintptr_t)x) & 15) == 0 && x_batch_stride % 4 == 0)),
                "msm_kv_project_f32: bad x batch stride / alignment");
    const size_t lds = sizeof(float) * (size_t)N * KP_LD;
    const int units = cdiv(HW, 16) * B * (N / (KP_FB * 16));
    const int grid = max(1, min(256, cdiv(units, 4)));
    if (cmat_width > 0) {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kv_project_kernel<true>, lds));
        hipLaunchKernelGGL(kv_project_kernel<true>, dim3(grid), dim3(KP_W * 64), lds, (hipStream_t)stream, x, w, cmat, cmat_width, out, B, HW, N, x_tokens,
                           x_batch_stride);
    } else {
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)kv_project_kernel<false>, lds));
        hipLaunchKernelGGL(kv_project_kernel<false>, dim3(grid), dim3(KP_W * 64), lds, (hipStream_t)stream, x, w, cmat, 0, out, B, HW, N, x_tokens,
                           x_batch_stride);
    }
    MSM_CHECK_LAUNCH("msm_kv_project_f32");
    return MSM_OK;
}

// PIPE: -1 = fp32 MFMAs (kv_project_multi_kernel); 0 / 1 / 2 = MODE of kv_project_multi_split_kernel (bf16 / fp16 matrix pipe)
template <typename OT, int PIPE>
static int kv_project_multi_impl(const char* who, int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                                 OT* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                                 const int32_t* cmat_width, int B, int C, int N, void* stream) {
    MSM_REQUIRE(n_jobs >= 1 && n_jobs <= KP_MAXJ && x && w && cmat && out && HW && x_tokens && x_batch_stride,
                "%s: bad arguments (1..%d jobs)", who, KP_MAXJ);
    MSM_REQUIRE(C == KP_K, "%s: C=%d, only 64 input channels are supported", who, C);
    MSM_REQUIRE(B > 0 && N > 0 && N % (KP_FB * 16) == 0 && N <= 512, "%s: N=%d must be 256 or 512", who, N);
    KvJobs jobs;
    jobs.n = n_jobs;
    int64_t total = 0;
    // separable constants (kv_project_body): all jobs of a launch or none
    const bool sep = cmat_width && cmat_width[0] > 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int cwj = cmat_width ? cmat_width[j] : 0;
        MSM_REQUIRE((cwj > 0) == sep && (cwj == 0 || (HW[j] > 0 && HW[j] % cwj == 0)), "%s: job %d: cmat_width=%d (all jobs separable or none; it must divide HW)", who, j, cwj);
        jobs.cw[j] = cwj;
        MSM_REQUIRE(x[j] && w[j] && cmat[j] && out[j] && HW[j] > 0, "%s: job %d: null pointer or empty level", who, j);
        MSM_REQUIRE(((((uintptr_t)w[j]) | ((uintptr_t)cmat[j]) | ((uintptr_t)out[j])) & 15) == 0 && (((uintptr_t)x[j]) & 3) == 0,
                    "%s: job %d: w/cmat/out must be 16-byte aligned", who, j);
        MSM_REQUIRE(x_batch_stride[j] >= (int64_t)C * HW[j] && (!x_tokens[j] || ((((uintptr_t)x[j]) & 15) == 0 && x_batch_stride[j] % 4 == 0)),
                    "%s: job %d: bad x batch stride / alignment", who, j);
        total += HW[j];
    }
    // workgroups: 256 shared out in proportion to the tokens of a job, at least one each, never more than a job has units / 4
    int wg = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int units = cdiv(HW[j], 16) * B * (N / (KP_FB * 16));
        int n = (int)((256 * (int64_t)HW[j] + total / 2) / total);
        n = max(1, min(n, cdiv(units, 4)));
        jobs.x[j] = x[j]; jobs.w[j] = w[j]; jobs.cmat[j] = cmat[j]; jobs.out[j] = out[j];
        jobs.HW[j] = HW[j]; jobs.tokens[j] = x_tokens[j]; jobs.x_sb[j] = x_batch_stride[j];
        jobs.first[j] = wg;
        wg += n;
    }
    for (int j = n_jobs; j <= KP_MAXJ; ++j) jobs.first[j] = wg;
    for (int j = n_jobs; j < KP_MAXJ; ++j) {
        jobs.x[j] = jobs.w[j] = jobs.cmat[j] = nullptr; jobs.out[j] = nullptr;
        jobs.HW[j] = jobs.tokens[j] = 0; jobs.x_sb[j] = 0;
        jobs.cw[j] = 0;
    }
#define KV_LAUNCH(KERNEL, WAVES)                                                                                      \
    {                                                                                                                 \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)KERNEL, lds));                                      \
        hipLaunchKernelGGL(KERNEL, dim3(wg), dim3(WAVES * 64), lds, (hipStream_t)stream, jobs, B, N);                 \
    }
    if constexpr (PIPE >= 0) {
        const size_t lds = sizeof(unsigned short) * ((size_t)(PIPE == 0 ? 3 : 1) * KP_FB * 16 * KS_LD + (size_t)KS_W * 16 * KS_TR);
        if (sep) KV_LAUNCH((kv_project_multi_split_kernel<OT, PIPE, true>), KS_W)
        else KV_LAUNCH((kv_project_multi_split_kernel<OT, PIPE, false>), KS_W)
    } else {
        const size_t lds = sizeof(float) * (size_t)N * KP_LD;
        if (sep) KV_LAUNCH((kv_project_multi_kernel<OT, true>), KP_W)
        else KV_LAUNCH((kv_project_multi_kernel<OT, false>), KP_W)
    }
#undef KV_LAUNCH
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_kv_project_multi_f32(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                                        float* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                                        const int32_t* cmat_width, int B, int C, int N, void* stream) {
    return kv_project_multi_impl<float, -1>("msm_kv_project_multi_f32", n_jobs, x, w, cmat, out, HW, x_tokens, x_batch_stride, cmat_width, B, C, N, stream);
}
extern "C" int msm_kv_project_multi_bf16(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                                         uint16_t* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                                         const int32_t* cmat_width, int B, int C, int N, int half_format, void* stream) {
    if (half_format) {
        MSM_REQUIRE(half_format == 1 && N == 512, "msm_kv_project_multi_bf16: half_format=%d needs N = 512 ([K | V]), got N=%d", half_format, N);
        return kv_project_multi_impl<uint16_t, 2>("msm_kv_project_multi_bf16", n_jobs, x, w, cmat, out, HW, x_tokens, x_batch_stride, cmat_width, B, C, N, stream);
    }
    // bf16 MFMAs (w rounded to one bf16, x as hi + lo) unless option KV_PIPE says 0: fp32 MFMAs, only the store rounded
    if (opt(MSM_OPT_KV_PIPE) == 0)
        return kv_project_multi_impl<uint16_t, -1>("msm_kv_project_multi_bf16", n_jobs, x, w, cmat, out, HW, x_tokens, x_batch_stride, cmat_width, B, C, N, stream);
    return kv_project_multi_impl<uint16_t, 1>("msm_kv_project_multi_bf16", n_jobs, x, w, cmat, out, HW, x_tokens, x_batch_stride, cmat_width, B, C, N, stream);
}
extern "C" int msm_kv_project_multi_split(int n_jobs, const float* const* x, const float* const* w, const float* const* cmat,
                                          float* const* out, const int32_t* HW, const int32_t* x_tokens, const int64_t* x_batch_stride,
                                          const int32_t* cmat_width, int B, int C, int N, void* stream) {
    return kv_project_multi_impl<float, 0>("msm_kv_project_multi_split", n_jobs, x, w, cmat, out, HW, x_tokens, x_batch_stride, cmat_width, B, C, N, stream);
}

extern "C" int msm_tokens_proj_nchw_f32(const float* x, const float* w, const float* bias, const double* gn_stats,
                                        const float* gn_gamma, const float* gn_beta, int groups, float eps, int relu, float* out,
                                        int B, int C, int HW, int N, void* stream) {
    MSM_REQUIRE(x && w && out, "msm_tokens_proj_nchw_f32: null pointer");
    MSM_REQUIRE(C == KP_K, "msm_tokens_proj_nchw_f32: C=%d, only 64 input channels are supported", C);
    MSM_REQUIRE(B > 0 && B <= 64 && HW > 0 && HW % 4 == 0 && N > 0 && N % (KP_FB * 16) == 0 && N <= 512,
                "msm_tokens_proj_nchw_f32: need B <= 64, HW %% 4 == 0, N in {256, 512}");
    MSM_REQUIRE(!gn_stats || (gn_gamma && gn_beta && groups > 0 && KP_K % groups == 0), "msm_tokens_proj_nchw_f32: bad GroupNorm arguments");
    MSM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0, "msm_tokens_proj_nchw_f32: pointers must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)N * KP_LD + (size_t)B * KP_K * 2);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)tokens_proj_nchw_kernel, lds));
    const int units = cdiv(HW, 16) * B * (N / (KP_FB * 16));
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int grid = max(1, min(256 * per_cu, cdiv(units, 4)));
    hipLaunchKernelGGL(tokens_proj_nchw_kernel, dim3(grid), dim3(MF_W * 64), lds, (hipStream_t)stream, x, w, bias, gn_stats, gn_gamma,
                       gn_beta, out, B, HW, N, groups, eps, relu);
    MSM_CHECK_LAUNCH("msm_tokens_proj_nchw_f32");
    return MSM_OK;
}
