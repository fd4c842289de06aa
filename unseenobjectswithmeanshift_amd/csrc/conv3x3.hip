// 3x3 / pad 1 convolution of a 64-channel token map to 64 channels (see include/msm_hip.h: msm_conv3x3_c64_f32):
//     out[b][y][x][o] = sum_{dy,dx,c} w[o][(dy*3 + dx)*64 + c] * in[b][y+dy-1][x+dx-1][c]      (zero padding)
//     stats[b][o]    += (sum, sum of squares) of out over the map        (the moments of the GroupNorm that follows)
//
// Reference: the FPN output convolution `layer_1 = Conv2d(64, 64, 3, padding=1, bias=False) + GroupNorm + ReLU`
// (msdeformattn.py:264-279, 349-351).  11.3 GFLOP at B = 8, 120 x 160: MFMA work.
//
// As an implicit GEMM through the tiled kernel every 64 x 64 output tile re-reads the whole 147 KB weight out of L2 next
// to its 147 KB of gathered input (133 us = 85 TFLOP/s, and a second pass for the GroupNorm moments).  Here the weight is
// the stationary operand: a workgroup copies all of it into LDS once (row stride 580 floats: conflict-free 16-byte
// reads) and its 16 waves stream 16-pixel tiles of ONE image:
//   * MFMA orientation D^T: rows = output channels (A = w from LDS, one ds_read_b128 = four k-steps), cols = 16
//     consecutive pixels of an image row (B = in: the lane of pixel lj and quarter lq reads channels ks*16 + lq*4 .. +3 of
//     the tap's pixel as one 16-byte load; K order k = tap*64 + ks*16 + lq*4 + c on both operands);
//   * a tile is 9 taps x 64 MFMAs; the 4 loads of the next tap are in flight during the 64 MFMAs of the current one;
//     taps outside the image contribute zeros (clamped address, zeroed operand);
//   * a lane ends with 4 consecutive channels of one pixel: 16-byte token-major stores; the per-channel moments are
//     reduced over the 16 pixels with DPP-free shuffles, kept in registers across the wave's tiles and leave as one double
//     atomic per (workgroup, channel, moment);
//   * tiles are handed out per image (blockIdx.y) in full rounds over the image's workgroups, the leftover tiles one per
//     SIMD first (waves w, w+4, ... share a SIMD).
#include <type_traits>

#include "bf16.h"
#include "common.h"

namespace msm {

constexpr int C3_C = 64;                 // channels in and out
constexpr int C3_K = 9 * C3_C;           // 576
constexpr int C3_LD = C3_K + 8;          // LDS row stride (floats): 146 slots of 16 B.  ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over 16 slots:
                                         // with lane = (row lj, slot offset lq) a stride = 2 (mod 4) slots is conflict-free (an odd stride, 145, is 2-way)
constexpr int C3_W = 16;                 // waves per workgroup (the weight takes 145 KiB: one workgroup per CU)
constexpr int C3_LDB = C3_K + 16;        // LDS row stride of the bf16 weight copies (bf16 elements): 74 slots of 16 B (see C3_LD)

// NCHW = false: token-major output [B][HW][64] (+ moments).  NCHW = true: output [B][Cout][HW] for Cout = 64 * gridDim.z, each
// z slice of workgroups holding its own 64 rows of the weight; the MFMA operands are swapped (rows = pixels) so that a
// lane ends with 4 consecutive PIXELS of one channel and the planes are written with 16-byte stores (W % 4 == 0); bias per
// channel (SimpleBasePixelDecoder.mask_features: Conv2d(64, 256, 3, padding=1) with bias, fpn.py:237-246).
//
// BF (low-precision mode): the weight is rounded to bf16 when it is copied into LDS, a tap's
// activations become hi + lo bf16 operands when they are used (x = hi + lo up to 2^-17 |x|) and v_mfma_f32_16x16x32_bf16 takes
// half a tap's channels at once: 16 MFMAs of 16 cycles per tap instead of 64 of 32 -- the kernel is then a stream over the map.
// K order of a tap: channel kh*32 + lq*8 + i on both operands.
// PB: 16-pixel blocks per wave (a tile is 16 PB consecutive pixels of an image row) -- with PB = 2 a weight fragment read from LDS
// feeds two MFMAs and a tap's loads hide behind twice the matrix work; WV: waves per workgroup (PB = 2 needs the registers of two
// waves per SIMD: WV = 8).
// BF = 2 (precision "f16"): the weight as IEEE halves, a tap's activations as ONE fp16 term (clamped) on v_mfma_f32_16x16x32_f16: half
// the MFMAs and no splitting work, 2^-12 roundings on both operands instead of 2^-9 on the weight.
template <bool NCHW, int BF = 0, int PB = 1, int WV = C3_W>
__global__ __launch_bounds__(WV * 64) void conv3x3_c64_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              double* __restrict__ stats, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [64][C3_LD] (BF: [64][C3_LDB] bf16), then the moment scratch [WV][64][2]
    unsigned short* wlb = reinterpret_cast<unsigned short*>(wl);
    float* msc = BF ? wl + C3_C * C3_LDB / 2 : wl + C3_C * C3_LD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int o0 = NCHW ? (int)blockIdx.z * C3_C : 0;          // first output channel of this workgroup
    {
        // the weight copy: all of a thread's loads are requested before the first is stored (a rolled loop exposes one L2 round trip
        // per iteration -- nine of them in front of a kernel whose tiles take a few microseconds each)
        constexpr int PER = C3_C * (C3_K / 4) / (WV * 64);         // 9 (18) float4 per thread
        static_assert(PER * WV * 64 == C3_C * (C3_K / 4), "the weight copy assumes an exact split");
        float4 wv[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * (WV * 64);
            const int n = i / (C3_K / 4), c4 = i - n * (C3_K / 4);
            wv[k] = *reinterpret_cast<const float4*>(w + (int64_t)(o0 + n) * C3_K + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * (WV * 64);
            const int n = i / (C3_K / 4), c4 = i - n * (C3_K / 4);
            if constexpr (BF == 2) *reinterpret_cast<u32x2b*>(wlb + n * C3_LDB + c4 * 4) = pack4h(wv[k].x, wv[k].y, wv[k].z, wv[k].w);
            else if constexpr (BF == 1) *reinterpret_cast<bf16x4*>(wlb + n * C3_LDB + c4 * 4) = pack4(wv[k].x, wv[k].y, wv[k].z, wv[k].w);
            else *reinterpret_cast<float4*>(wl + n * C3_LD + c4 * 4) = wv[k];
        }
    }
    __syncthreads();

    const int xt = (W + 16 * PB - 1) / (16 * PB);    // tiles per image row
    const int units = xt * H;                        // of this image
    const int slots = gridDim.x * WV;
    const int full_rounds = units / slots;
    const int left = units - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int mine = full_rounds + (left_slot < left ? 1 : 0);
    const float* ib = in + (int64_t)b * H * W * C3_C;
    float* ob = NCHW ? out + ((int64_t)b * gridDim.z * C3_C + o0) * H * W : out + (int64_t)b * H * W * C3_C;
    float bch[4] = {0.f, 0.f, 0.f, 0.f};                         // NCHW: bias of this lane's channel in each 16-channel block
    if (NCHW && bias) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) bch[mt] = bias[o0 + mt * 16 + lj];
    }
    float s[4][4], q[4][4];                          // moments of this lane's 4 x 4 channels over its pixels
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[mt][r] = q[mt][r] = 0.f;
    const float* wp = wl + lj * C3_LD + lq * 4;

    for (int it = 0; it < mine; ++it) {
        const int u = (it < full_rounds) ? it * slots + (int)blockIdx.x * WV + wave : full_rounds * slots + left_slot;
        const int y = u / xt, x0 = (u - y * xt) * (16 * PB);
        // B operand of tap t, pixel block pb: channels ks*16 + lq*4 .. +3 (ks = 0..3) of pixel (y + t/3 - 1, x0 + 16 pb + lj + t%3 - 1),
        // zeros outside
        auto load_tap = [&](int t, float4 (&f)[PB][4]) {
            const int yy = y + t / 3 - 1;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int xx = x0 + pb * 16 + lj + t % 3 - 1;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                // (BF: f[2 kh + h] = channels kh*32 + lq*8 + 4 h .. + 3)
                const float* p = ib + ((int64_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * C3_C + (BF ? lq * 8 : lq * 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (BF ? (ks >> 1) * 32 + (ks & 1) * 4 : ks * 16));
                    f[pb][ks] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        f32x4 acc[PB][4];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto mma_tap = [&](int t, const float4 (&cur)[PB][4]) {
            if constexpr (BF == 2) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    f16x8 xf[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        xf[pb] = cvt8h(cur[pb][2 * kh].x, cur[pb][2 * kh].y, cur[pb][2 * kh].z, cur[pb][2 * kh].w, cur[pb][2 * kh + 1].x,
                                       cur[pb][2 * kh + 1].y, cur[pb][2 * kh + 1].z, cur[pb][2 * kh + 1].w);
                    f16x8 a[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        a[mt] = *reinterpret_cast<const f16x8*>(wlb + (mt * 16 + lj) * C3_LDB + t * C3_C + kh * 32 + lq * 8);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[pb][mt] = NCHW ? mfma_f16k32(xf[pb], a[mt], acc[pb][mt]) : mfma_f16k32(a[mt], xf[pb], acc[pb][mt]);
                }
                return;
            }
            if constexpr (BF == 1) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    bf16x8 xh[PB], xl[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        const Split4 s0 = split4(cur[pb][2 * kh].x, cur[pb][2 * kh].y, cur[pb][2 * kh].z, cur[pb][2 * kh].w);
                        const Split4 s1 = split4(cur[pb][2 * kh + 1].x, cur[pb][2 * kh + 1].y, cur[pb][2 * kh + 1].z, cur[pb][2 * kh + 1].w);
                        xh[pb] = cat8(s0.hi, s1.hi);
                        xl[pb] = cat8(s0.lo, s1.lo);
                    }
                    bf16x8 a[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        a[mt] = *reinterpret_cast<const bf16x8*>(wlb + (mt * 16 + lj) * C3_LDB + t * C3_C + kh * 32 + lq * 8);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[pb][mt] = NCHW ? mfma_bf16k32(xl[pb], a[mt], acc[pb][mt]) : mfma_bf16k32(a[mt], xl[pb], acc[pb][mt]);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[pb][mt] = NCHW ? mfma_bf16k32(xh[pb], a[mt], acc[pb][mt]) : mfma_bf16k32(a[mt], xh[pb], acc[pb][mt]);
                }
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const float4*>(wp + mt * 16 * C3_LD + t * C3_C + ks * 16);
#define C3_STEP(C)                                                                                                                 \
    _Pragma("unroll") for (int pb = 0; pb < PB; ++pb) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                             \
        acc[pb][mt] = NCHW ? mfma16(cur[pb][ks].C, a[mt].C, acc[pb][mt]) : mfma16(a[mt].C, cur[pb][ks].C, acc[pb][mt]);
                C3_STEP(x) C3_STEP(y) C3_STEP(z) C3_STEP(w)
#undef C3_STEP
            }
        };
        float4 fa[PB][4], fb[PB][4];
        load_tap(0, fa);
#pragma unroll 1
        for (int t = 0; t < 8; t += 2) {          // taps in pairs: two register sets, the next tap's loads ride on the current MFMAs
            load_tap(t + 1, fb);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(t, fa);
            __builtin_amdgcn_sched_barrier(0);
            load_tap(t + 2, fa);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(t + 1, fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        mma_tap(8, fa);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            if constexpr (NCHW) {
                // lane: channel mt*16 + lj, pixels x0 + 16 pb + lq*4 .. +3 of row y
                const int xs = x0 + pb * 16 + lq * 4;
                if (xs < W) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        *reinterpret_cast<float4*>(ob + (int64_t)(mt * 16 + lj) * H * W + (int64_t)y * W + xs) =
                            make_float4(acc[pb][mt][0] + bch[mt], acc[pb][mt][1] + bch[mt], acc[pb][mt][2] + bch[mt], acc[pb][mt][3] + bch[mt]);
                }
            } else if (x0 + pb * 16 + lj < W) {
                float* op = ob + ((int64_t)y * W + x0 + pb * 16 + lj) * C3_C + lq * 4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    *reinterpret_cast<float4*>(op + mt * 16) = make_float4(acc[pb][mt][0], acc[pb][mt][1], acc[pb][mt][2], acc[pb][mt][3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[mt][r] += acc[pb][mt][r];
                        q[mt][r] += acc[pb][mt][r] * acc[pb][mt][r];
                    }
                }
            }
        }
    }
    if (!NCHW && stats) {
        // over the 16 pixels of the lane quarter, then over the workgroup's waves (fixed order), then one double add each
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                {          // (over the 16-lane row in the order 1, 2, 4, 8 of the shfl_xor loop it replaces, on DPP: bitwise the same sums)
                    s[mt][r] += wave_xor_dpp1(s[mt][r]);
                    q[mt][r] += wave_xor_dpp1(q[mt][r]);
                    s[mt][r] += wave_xor_dpp2(s[mt][r]);
                    q[mt][r] += wave_xor_dpp2(q[mt][r]);
                    s[mt][r] += wave_xor_dpp4(s[mt][r]);
                    q[mt][r] += wave_xor_dpp4(q[mt][r]);
                    s[mt][r] += wave_xor_dpp8(s[mt][r]);
                    q[mt][r] += wave_xor_dpp8(q[mt][r]);
                }
            }
        if (lj == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = mt * 16 + lq * 4 + r;
                    msc[(wave * C3_C + ch) * 2 + 0] = s[mt][r];
                    msc[(wave * C3_C + ch) * 2 + 1] = q[mt][r];
                }
        }
        __syncthreads();
        if (tid < C3_C * 2) {
            double t = 0.0;
            for (int wv = 0; wv < WV; ++wv) t += (double)msc[wv * C3_C * 2 + tid];
            atomicAdd(stats + (int64_t)b * C3_C * 2 + tid, t);
        }
    }
}

// ---- fp32 results on the bf16 matrix pipe (msm_conv3x3_c64_split) ---------------------------------------------------
// Both operands as exact three-term bf16 splits, six v_mfma_f32_16x16x32_bf16 per product (small terms in their own accumulator).
//   * the activation arrives already split (three bf16 planes written by msm_groupnorm_apply_split): a tap's B operand is three
//     16-byte loads per 32 channels and no vector work -- splitting inside this kernel would repeat it for each of the nine taps
//     and would bound the kernel by the vector pipe instead of the matrix pipe;
//   * the three weight planes of all 64 output channels (3 x 73 KiB) do not fit the LDS: a workgroup holds 32 output channels
//     (blockIdx.z), split once while it copies them in;
//   * per tap and 32-channel half: 2 row blocks x 6 terms = 12 MFMAs on 12 weight fragments (ds_read_b128) and 3 activation
//     fragments; the next tap's six loads ride on the current tap's 24 MFMAs.
constexpr int C3S_N = 32;                // output channels per workgroup
constexpr int C3S_W = 8;                 // waves per workgroup (two per SIMD: a wave keeps two input rows + a shifted copy in registers)

// src shifted by one lane inside each row of 16 lanes (towards higher lanes: SHR, lane lj takes lane lj - 1); the lane without a
// source keeps `old` (bound_ctrl off): that is where the halo pixel goes
template <bool SHR>
__device__ __forceinline__ bf16x8 shift_pixels(const bf16x8& src, const bf16x8& old) {
    const u32x4b s = __builtin_bit_cast(u32x4b, src), o = __builtin_bit_cast(u32x4b, old);
    u32x4b r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp((int)o[i], (int)s[i], SHR ? 0x111 : 0x101, 0xf, 0xf, false);
    return __builtin_bit_cast(bf16x8, r);
}

// The nine taps of a 16-pixel tile read three input rows; the three taps of a row differ by one pixel, i.e. by one lane of the
// B operand.  A row is loaded ONCE (six 16-byte loads per lane + the two halo pixels x0 - 1 / x0 + 16 by the lanes of pixel 0 /
// 15) and the dx = -1 / +1 operands are lane shifts of it (v_mov_b32 row_shr / row_shl, 24 per tap): a third of the L2 -> CU
// traffic of one load per tap -- at 6 bytes per element that traffic, not the matrix pipe, bounded the per-tap form (119 us).
__global__ __launch_bounds__(C3S_W * 64) void conv3x3_c64_split_kernel(const uint16_t* __restrict__ planes, int64_t plane_stride,
                                                                       const float* __restrict__ w, float* __restrict__ out,
                                                                       double* __restrict__ stats, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [3][32][C3_LDB] bf16, then the moment scratch [C3S_W][32][2]
    unsigned short* wlb = reinterpret_cast<unsigned short*>(wl);
    constexpr int PL = C3S_N * C3_LDB;                            // elements per weight plane
    float* msc = wl + 3 * PL / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int o0 = (int)blockIdx.z * C3S_N;
    for (int i = tid; i < C3S_N * (C3_K / 4); i += C3S_W * 64) {
        const int n = i / (C3_K / 4), c4 = i - n * (C3_K / 4);
        const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)(o0 + n) * C3_K + c4 * 4);
        const Split3 t3 = split3(wv.x, wv.y, wv.z, wv.w);
        *reinterpret_cast<bf16x4*>(wlb + n * C3_LDB + c4 * 4) = t3.h;
        *reinterpret_cast<bf16x4*>(wlb + PL + n * C3_LDB + c4 * 4) = t3.m;
        *reinterpret_cast<bf16x4*>(wlb + 2 * PL + n * C3_LDB + c4 * 4) = t3.l;
    }
    __syncthreads();

    const int xt = (W + 15) / 16;
    const int units = xt * H;
    const int slots = gridDim.x * C3S_W;
    const int full_rounds = units / slots;
    const int left = units - full_rounds * slots;
    const int left_slot = (wave >> 2) * ((int)gridDim.x * 4) + (int)blockIdx.x * 4 + (wave & 3);
    const int mine = full_rounds + (left_slot < left ? 1 : 0);
    const uint16_t* ib = planes + (int64_t)b * H * W * C3_C;
    float* ob = out + (int64_t)b * H * W * C3_C + o0;
    float s[2][4], q[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[mt][r] = q[mt][r] = 0.f;
    const unsigned short* wp = wlb + lj * C3_LDB + lq * 8;

    struct Row {
        bf16x8 f[2][3];        // [32-channel half][term h, m, l]
    };
    for (int it = 0; it < mine; ++it) {
        const int u = (it < full_rounds) ? it * slots + (int)blockIdx.x * C3S_W + wave : full_rounds * slots + left_slot;
        const int y = u / xt, x0 = (u - y * xt) * 16;
        const int px = x0 + lj;
        // centre pixels of input row y + dy (zeros outside the map) and, in lanes lj == 0 / lj == 15, the halo pixels x0 - 1 / x0 + 16
        auto load_row = [&](int dy, Row& c, Row& halo) {
            const int yy = y + dy;
            const bool rok = yy >= 0 && yy < H;
            const int yc = min(max(yy, 0), H - 1);
            const u32x4b zero = {0u, 0u, 0u, 0u};
            const bool ok = rok && px < W;
            const uint16_t* p = ib + ((int64_t)yc * W + min(px, W - 1)) * C3_C + lq * 8;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) {
                    const u32x4b v = *reinterpret_cast<const u32x4b*>(p + tm * plane_stride + kh * 32);
                    c.f[kh][tm] = __builtin_bit_cast(bf16x8, ok ? v : zero);
                }
            const int hx = lj == 0 ? x0 - 1 : x0 + 16;
            const bool hok = rok && (lj == 0 || lj == 15) && hx >= 0 && hx < W;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) halo.f[kh][tm] = __builtin_bit_cast(bf16x8, zero);
            if (hok) {
                const uint16_t* ph = ib + ((int64_t)yc * W + hx) * C3_C + lq * 8;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm)
                        halo.f[kh][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(ph + tm * plane_stride + kh * 32));
            }
        };
        f32x4 lo[2], hi[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) lo[mt] = hi[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto mma_tap = [&](int t, const Row& x) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 a[2][3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm)
                        a[mt][tm] = *reinterpret_cast<const bf16x8*>(wp + tm * PL + mt * 16 * C3_LDB + t * C3_C + kh * 32);
                // (weight term, activation term): l.h, h.l, m.m, m.h, h.m into lo; h.h into hi -- four independent chains
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) lo[mt] = mfma_bf16k32(a[mt][2], x.f[kh][0], lo[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) hi[mt] = mfma_bf16k32(a[mt][0], x.f[kh][0], hi[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) lo[mt] = mfma_bf16k32(a[mt][0], x.f[kh][2], lo[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) lo[mt] = mfma_bf16k32(a[mt][1], x.f[kh][1], lo[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) lo[mt] = mfma_bf16k32(a[mt][1], x.f[kh][0], lo[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) lo[mt] = mfma_bf16k32(a[mt][0], x.f[kh][1], lo[mt]);
            }
        };
        // the three taps of input row r (tap index 3 r + dx + 1)
        auto mma_row = [&](int r, const Row& c, const Row& halo) {
            Row sh;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) sh.f[kh][tm] = shift_pixels<true>(c.f[kh][tm], halo.f[kh][tm]);
            mma_tap(3 * r, sh);
            mma_tap(3 * r + 1, c);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) sh.f[kh][tm] = shift_pixels<false>(c.f[kh][tm], halo.f[kh][tm]);
            mma_tap(3 * r + 2, sh);
        };
        Row ca, ha, cb, hb;
        load_row(-1, ca, ha);
        load_row(0, cb, hb);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(0, ca, ha);
        __builtin_amdgcn_sched_barrier(0);
        load_row(1, ca, ha);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(1, cb, hb);
        __builtin_amdgcn_sched_barrier(0);
        mma_row(2, ca, ha);
        if (px < W) {
            float* op = ob + ((int64_t)y * W + px) * C3_C + lq * 4;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 acc = lo[mt] + hi[mt];
                *reinterpret_cast<float4*>(op + mt * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[mt][r] += acc[r];
                    q[mt][r] += acc[r] * acc[r];
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                {          // (over the 16-lane row in the order 1, 2, 4, 8 of the shfl_xor loop it replaces, on DPP: bitwise the same sums)
                    s[mt][r] += wave_xor_dpp1(s[mt][r]);
                    q[mt][r] += wave_xor_dpp1(q[mt][r]);
                    s[mt][r] += wave_xor_dpp2(s[mt][r]);
                    q[mt][r] += wave_xor_dpp2(q[mt][r]);
                    s[mt][r] += wave_xor_dpp4(s[mt][r]);
                    q[mt][r] += wave_xor_dpp4(q[mt][r]);
                    s[mt][r] += wave_xor_dpp8(s[mt][r]);
                    q[mt][r] += wave_xor_dpp8(q[mt][r]);
                }
            }
        if (lj == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = mt * 16 + lq * 4 + r;
                    msc[(wave * C3S_N + ch) * 2 + 0] = s[mt][r];
                    msc[(wave * C3S_N + ch) * 2 + 1] = q[mt][r];
                }
        }
        __syncthreads();
        if (tid < C3S_N * 2) {
            double t = 0.0;
            for (int wv = 0; wv < C3S_W; ++wv) t += (double)msc[wv * C3S_N * 2 + tid];
            atomicAdd(stats + ((int64_t)b * C3_C + o0) * 2 + tid, t);
        }
    }
}


// ---- the "f16" plan's form on an IEEE-half token map (msm_conv3x3_c64_f16h; round 6) ---------------------------------------------
// conv3x3_c64_kernel<false, 2, 2, 8> spends its 49 us (B = 8, 120 x 160) waiting: a wave walks 2.3 tiles of 9 taps, one memory round
// trip per tap with one tap in flight, on two waves per SIMD.  Here the producer (msm_groupnorm_apply_f16) already wrote the map as
// the clamped halves this kernel would round to -- the same operand bits, half the bytes -- and a wave owns a UNIT of up to C3R_R
// consecutive output rows of a 32-pixel strip:
//   * all loads of the unit (its R + 2 input rows: four 16-byte loads per row and lane, + one halo pixel left and right) are issued
//     before anything else -- in front of the workgroup's weight copy, so the copy hides under them: ONE round trip per unit;
//   * an input row is loaded once; the dx = -1 / +1 operands of its three taps are lane shifts (v_mov_b32 row_shr / row_shl, the
//     pixel at the junction of the two 16-pixel blocks through a row rotate, the strip's outer neighbours from the halo registers);
//   * taps in the order 0..8, halves kh = 0, 1, one accumulator chain per (block, channel block): the output bits of
//     conv3x3_c64_kernel<false, 2, 2, 8> on the same halves;
//   * units are numbered strip-fastest; with a grid of a multiple of 8 workgroups per image each XCD (workgroup id mod 8) takes a
//     contiguous range of units, so the rows two bands share are read from HBM by one L2.
constexpr int C3R_R = 3;                 // output rows per unit (R + 2 input rows of 24 registers each stay in flight)
constexpr int C3R_W = 8;                 // waves per workgroup

// lane lj of each 16-lane row takes lane lj - 1 (SHR) / lj + 1 of src; the lane without a source (0 / 15) keeps `old`
template <bool SHR>
__device__ __forceinline__ u32x4b shift_px(const u32x4b& src, const u32x4b& old) {
    u32x4b r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp((int)old[i], (int)src[i], SHR ? 0x111 : 0x101, 0xf, 0xf, false);
    return r;
}
// rotate right by N inside each row of 16 lanes (lane i takes lane (i - N) mod 16)
template <int N>
__device__ __forceinline__ u32x4b rotate_px(const u32x4b& src) {
    u32x4b r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)src[i], 0x120 + N, 0xf, 0xf, false);
    return r;
}

__global__ __launch_bounds__(C3R_W * 64) void conv3x3_c64_rows_kernel(const uint16_t* __restrict__ in, const float* __restrict__ w,
                                                                      float* __restrict__ out, double* __restrict__ stats, int H, int W,
                                                                      int R) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [64][C3_LDB] halves, then the moment scratch [C3R_W][64][2]
    unsigned short* wlb = reinterpret_cast<unsigned short*>(wl);
    float* msc = wl + C3_C * C3_LDB / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int xt = (W + 31) / 32;                    // strips per image row
    const int units = xt * ((H + R - 1) / R);        // of this image
    // this wave's units: first, first + step, ... below last
    int u, step, last;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (units + 7) / 8, xcd = (int)blockIdx.x & 7;
        u = xcd * chunk + ((int)blockIdx.x >> 3) * C3R_W + wave;
        step = ((int)gridDim.x >> 3) * C3R_W;
        last = min(units, (xcd + 1) * chunk);
    } else {
        u = (int)blockIdx.x * C3R_W + wave;
        step = (int)gridDim.x * C3R_W;
        last = units;
    }
    const uint16_t* ib = in + (int64_t)b * H * W * C3_C;
    float* ob = out + (int64_t)b * H * W * C3_C;
    const u32x4b zero = {0u, 0u, 0u, 0u};

    struct Rows {
        u32x4b c[C3R_R + 2][2][2];        // [input row][16-pixel block][32-channel half]: channels kh*32 + lq*8 .. +7 of pixel x0 + 16 pb + lj
        u32x4b h[C3R_R + 2][2];           // lane lj == 0: pixel x0 - 1, lane lj == 15: pixel x0 + 32 (zeros outside the map)
    };
    auto load_unit = [&](int uu, Rows& rw) {
        const int band = uu / xt, x0 = (uu - band * xt) * 32, y0 = band * R;
#pragma unroll
        for (int i = 0; i < C3R_R + 2; ++i) {
            const int yy = y0 - 1 + i;
            const bool rok = yy >= 0 && yy < H && i < R + 2;
            const uint16_t* rp = ib + (int64_t)min(max(yy, 0), H - 1) * W * C3_C + lq * 8;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int px = x0 + pb * 16 + lj;
                const bool ok = rok && px < W;
                const uint16_t* p = rp + (int64_t)min(px, W - 1) * C3_C;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const u32x4b v = *reinterpret_cast<const u32x4b*>(p + kh * 32);
                    rw.c[i][pb][kh] = ok ? v : zero;
                }
            }
            const int hx = lj < 8 ? x0 - 1 : x0 + 32;
            const bool hok = rok && hx >= 0 && hx < W;
            const uint16_t* ph = rp + (int64_t)min(max(hx, 0), W - 1) * C3_C;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const u32x4b v = *reinterpret_cast<const u32x4b*>(ph + kh * 32);
                rw.h[i][kh] = hok ? v : zero;
            }
        }
    };
    Rows rw;
    if (u < last) load_unit(u, rw);
    {
        // the weight copy (fp32 -> halves), behind the first unit's loads
        constexpr int PER = C3_C * (C3_K / 4) / (C3R_W * 64);      // 18 float4 per thread
        static_assert(PER * C3R_W * 64 == C3_C * (C3_K / 4), "the weight copy assumes an exact split");
#pragma unroll
        for (int k0 = 0; k0 < PER; k0 += 6) {
            float4 wv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wv[k] = *reinterpret_cast<const float4*>(w + (int64_t)(tid + (k0 + k) * (C3R_W * 64)) * 4);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = tid + (k0 + k) * (C3R_W * 64);
                const int n = i / (C3_K / 4), c4 = i - n * (C3_K / 4);
                *reinterpret_cast<u32x2b*>(wlb + n * C3_LDB + c4 * 4) = pack4h(wv[k].x, wv[k].y, wv[k].z, wv[k].w);
            }
        }
    }
    __syncthreads();

    float s[4][4], q[4][4];                          // moments of this lane's 4 x 4 channels over its pixels
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[mt][r] = q[mt][r] = 0.f;
    const unsigned short* wp = wlb + lj * C3_LDB + lq * 8;

    for (; u < last; u += step) {
        const int band = u / xt, x0 = (u - band * xt) * 32, y0 = band * R;
        const int nr = min(R, H - y0);
#pragma unroll
        for (int r = 0; r < C3R_R; ++r) {
            if (r >= nr) break;
            f32x4 acc[2][4];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int i = r + dy;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int t = dy * 3 + dx;
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        u32x4b x[2];
                        if (dx == 0) {
                            x[0] = shift_px<true>(rw.c[i][0][kh], rw.h[i][kh]);
                            x[1] = shift_px<true>(rw.c[i][1][kh], rotate_px<1>(rw.c[i][0][kh]));
                        } else if (dx == 1) {
                            x[0] = rw.c[i][0][kh];
                            x[1] = rw.c[i][1][kh];
                        } else {
                            x[0] = shift_px<false>(rw.c[i][0][kh], rotate_px<15>(rw.c[i][1][kh]));
                            x[1] = shift_px<false>(rw.c[i][1][kh], rw.h[i][kh]);
                        }
                        f16x8 a[4];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f16x8*>(wp + mt * 16 * C3_LDB + t * C3_C + kh * 32);
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = mfma_f16k32(a[mt], __builtin_bit_cast(f16x8, x[pb]), acc[pb][mt]);
                    }
                }
            }
            const int y = y0 + r;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                if (x0 + pb * 16 + lj < W) {
                    float* op = ob + ((int64_t)y * W + x0 + pb * 16 + lj) * C3_C + lq * 4;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        *reinterpret_cast<float4*>(op + mt * 16) = make_float4(acc[pb][mt][0], acc[pb][mt][1], acc[pb][mt][2], acc[pb][mt][3]);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            s[mt][rr] += acc[pb][mt][rr];
                            q[mt][rr] += acc[pb][mt][rr] * acc[pb][mt][rr];
                        }
                    }
                }
            }
        }
        if (u + step < last) load_unit(u + step, rw);
    }
    if (stats) {
        // over the 16 pixels of the lane quarter, then over the workgroup's waves (fixed order), then one double add each
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[mt][r] += wave_xor_dpp1(s[mt][r]);
                q[mt][r] += wave_xor_dpp1(q[mt][r]);
                s[mt][r] += wave_xor_dpp2(s[mt][r]);
                q[mt][r] += wave_xor_dpp2(q[mt][r]);
                s[mt][r] += wave_xor_dpp4(s[mt][r]);
                q[mt][r] += wave_xor_dpp4(q[mt][r]);
                s[mt][r] += wave_xor_dpp8(s[mt][r]);
                q[mt][r] += wave_xor_dpp8(q[mt][r]);
            }
        if (lj == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = mt * 16 + lq * 4 + r;
                    msc[(wave * C3_C + ch) * 2 + 0] = s[mt][r];
                    msc[(wave * C3_C + ch) * 2 + 1] = q[mt][r];
                }
        }
        __syncthreads();
        if (tid < C3_C * 2) {
            double t = 0.0;
            for (int wv = 0; wv < C3R_W; ++wv) t += (double)msc[wv * C3_C * 2 + tid];
            atomicAdd(stats + (int64_t)b * C3_C * 2 + tid, t);
        }
    }
}


// ---- the "bf16" plan's form with a unit's loads in flight at once (round 6) --------------------------------------------------------
// conv3x3_c64_kernel<false, 1, 2, 8> has the same chain as the f16 form had: nine taps, one memory round trip each, 2.3 tiles per wave
// (50 us at B = 8, 120 x 160).  Its activation operand is an fp32 value as hi + lo bf16 terms -- no narrower map for a producer to
// write -- so here a unit is ONE output row of a 32-pixel strip: its three input rows (eight 16-byte loads per row and lane + the halo
// pixel left and right) are requested together, split into hi + lo ONCE per row (the per-tap form splits every value nine times), and
// the dx = -1 / +1 operands are lane shifts of the split terms.  Taps 0..8, halves kh = 0, 1, lo then hi into one accumulator chain:
// the output bits of conv3x3_c64_kernel<false, 1, 2, 8>.  Units and XCD ranges as conv3x3_c64_rows_kernel.
__global__ __launch_bounds__(C3R_W * 64) void conv3x3_c64_rows_bf16_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                           float* __restrict__ out, double* __restrict__ stats, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [64][C3_LDB] bf16, then the moment scratch [C3R_W][64][2]
    unsigned short* wlb = reinterpret_cast<unsigned short*>(wl);
    float* msc = wl + C3_C * C3_LDB / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int xt = (W + 31) / 32;
    const int units = xt * H;
    int u, step, last;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (units + 7) / 8, xcd = (int)blockIdx.x & 7;
        u = xcd * chunk + ((int)blockIdx.x >> 3) * C3R_W + wave;
        step = ((int)gridDim.x >> 3) * C3R_W;
        last = min(units, (xcd + 1) * chunk);
    } else {
        u = (int)blockIdx.x * C3R_W + wave;
        step = (int)gridDim.x * C3R_W;
        last = units;
    }
    const float* ib = in + (int64_t)b * H * W * C3_C;
    float* ob = out + (int64_t)b * H * W * C3_C;

    struct Raw {
        float4 c[3][2][2][2];             // [input row][16-pixel block][32-channel half][channels lq*8 + 4 h .. + 3]
        float4 h[3][2][2];                // lanes lj < 8: pixel x0 - 1, lj >= 8: pixel x0 + 32
    };
    struct Row {
        u32x4b ch[2][2], cl[2][2];        // hi / lo bf16 terms of one input row [16-pixel block][32-channel half]
        u32x4b hh[2], hl[2];              // of its halo pixels (lane 0: left, lane 15: right; zeros outside the map)
    };
    auto load_unit = [&](int uu, Raw& rw) {
        const int y = uu / xt, x0 = (uu - y * xt) * 32;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int yy = y - 1 + i;
            const float* rp = ib + (int64_t)min(max(yy, 0), H - 1) * W * C3_C + lq * 8;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const float* p = rp + (int64_t)min(x0 + pb * 16 + lj, W - 1) * C3_C;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int h = 0; h < 2; ++h) rw.c[i][pb][kh][h] = *reinterpret_cast<const float4*>(p + kh * 32 + h * 4);
            }
            const int hx = lj < 8 ? x0 - 1 : x0 + 32;
            const float* ph = rp + (int64_t)min(max(hx, 0), W - 1) * C3_C;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int h = 0; h < 2; ++h) rw.h[i][kh][h] = *reinterpret_cast<const float4*>(ph + kh * 32 + h * 4);
        }
    };
    Raw raw;
    if (u < last) load_unit(u, raw);
    {
        constexpr int PER = C3_C * (C3_K / 4) / (C3R_W * 64);      // 18 float4 per thread
#pragma unroll
        for (int k0 = 0; k0 < PER; k0 += 6) {
            float4 wv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wv[k] = *reinterpret_cast<const float4*>(w + (int64_t)(tid + (k0 + k) * (C3R_W * 64)) * 4);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = tid + (k0 + k) * (C3R_W * 64);
                const int n = i / (C3_K / 4), c4 = i - n * (C3_K / 4);
                *reinterpret_cast<bf16x4*>(wlb + n * C3_LDB + c4 * 4) = pack4(wv[k].x, wv[k].y, wv[k].z, wv[k].w);
            }
        }
    }
    __syncthreads();

    float s[4][4], q[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[mt][r] = q[mt][r] = 0.f;
    const unsigned short* wp = wlb + lj * C3_LDB + lq * 8;
    const u32x4b zero = {0u, 0u, 0u, 0u};
    auto split8 = [&](const float4& a, const float4& c, bool ok, u32x4b& hi, u32x4b& lo) {
        const Split4 s0 = split4(a.x, a.y, a.z, a.w), s1 = split4(c.x, c.y, c.z, c.w);
        hi = ok ? __builtin_bit_cast(u32x4b, cat8(s0.hi, s1.hi)) : zero;
        lo = ok ? __builtin_bit_cast(u32x4b, cat8(s0.lo, s1.lo)) : zero;
    };

    for (; u < last; u += step) {
        const int y = u / xt, x0 = (u - y * xt) * 32;
        f32x4 acc[2][4];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // one input row at a time: split (the row's fp32 registers die here), then its three taps
            Row rw;
            const int yy = y - 1 + i;
            const bool rok = yy >= 0 && yy < H;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    split8(raw.c[i][pb][kh][0], raw.c[i][pb][kh][1], rok && x0 + pb * 16 + lj < W, rw.ch[pb][kh], rw.cl[pb][kh]);
            const int hx = lj < 8 ? x0 - 1 : x0 + 32;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) split8(raw.h[i][kh][0], raw.h[i][kh][1], rok && hx >= 0 && hx < W, rw.hh[kh], rw.hl[kh]);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int t = i * 3 + dx;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    u32x4b xh[2], xl[2];
                    if (dx == 0) {
                        xh[0] = shift_px<true>(rw.ch[0][kh], rw.hh[kh]);
                        xh[1] = shift_px<true>(rw.ch[1][kh], rotate_px<1>(rw.ch[0][kh]));
                        xl[0] = shift_px<true>(rw.cl[0][kh], rw.hl[kh]);
                        xl[1] = shift_px<true>(rw.cl[1][kh], rotate_px<1>(rw.cl[0][kh]));
                    } else if (dx == 1) {
                        xh[0] = rw.ch[0][kh]; xh[1] = rw.ch[1][kh];
                        xl[0] = rw.cl[0][kh]; xl[1] = rw.cl[1][kh];
                    } else {
                        xh[0] = shift_px<false>(rw.ch[0][kh], rotate_px<15>(rw.ch[1][kh]));
                        xh[1] = shift_px<false>(rw.ch[1][kh], rw.hh[kh]);
                        xl[0] = shift_px<false>(rw.cl[0][kh], rotate_px<15>(rw.cl[1][kh]));
                        xl[1] = shift_px<false>(rw.cl[1][kh], rw.hl[kh]);
                    }
                    bf16x8 a[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(wp + mt * 16 * C3_LDB + t * C3_C + kh * 32);
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = mfma_bf16k32(a[mt], __builtin_bit_cast(bf16x8, xl[pb]), acc[pb][mt]);
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[pb][mt] = mfma_bf16k32(a[mt], __builtin_bit_cast(bf16x8, xh[pb]), acc[pb][mt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            if (x0 + pb * 16 + lj < W) {
                float* op = ob + ((int64_t)y * W + x0 + pb * 16 + lj) * C3_C + lq * 4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    *reinterpret_cast<float4*>(op + mt * 16) = make_float4(acc[pb][mt][0], acc[pb][mt][1], acc[pb][mt][2], acc[pb][mt][3]);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        s[mt][rr] += acc[pb][mt][rr];
                        q[mt][rr] += acc[pb][mt][rr] * acc[pb][mt][rr];
                    }
                }
            }
        }
        if (u + step < last) load_unit(u + step, raw);           // (the split terms of a unit and the next unit's fp32 rows do not fit the registers together)
    }
    if (stats) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[mt][r] += wave_xor_dpp1(s[mt][r]);
                q[mt][r] += wave_xor_dpp1(q[mt][r]);
                s[mt][r] += wave_xor_dpp2(s[mt][r]);
                q[mt][r] += wave_xor_dpp2(q[mt][r]);
                s[mt][r] += wave_xor_dpp4(s[mt][r]);
                q[mt][r] += wave_xor_dpp4(q[mt][r]);
                s[mt][r] += wave_xor_dpp8(s[mt][r]);
                q[mt][r] += wave_xor_dpp8(q[mt][r]);
            }
        if (lj == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = mt * 16 + lq * 4 + r;
                    msc[(wave * C3_C + ch) * 2 + 0] = s[mt][r];
                    msc[(wave * C3_C + ch) * 2 + 1] = q[mt][r];
                }
        }
        __syncthreads();
        if (tid < C3_C * 2) {
            double t = 0.0;
            for (int wv = 0; wv < C3R_W; ++wv) t += (double)msc[wv * C3_C * 2 + tid];
            atomicAdd(stats + (int64_t)b * C3_C * 2 + tid, t);
        }
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_conv3x3_c64_split(const uint16_t* planes, const float* w_tap_major, float* out, double* stats, int stats_cleared, int B,
                                     int H, int W, void* stream) {
    MSM_REQUIRE(planes && w_tap_major && out, "msm_conv3x3_c64_split: null pointer");
    MSM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W * C3_C < ((int64_t)1 << 31), "msm_conv3x3_c64_split: bad sizes B=%d H=%d W=%d", B, H, W);
    MSM_REQUIRE(((((uintptr_t)planes) | ((uintptr_t)w_tap_major) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "msm_conv3x3_c64_split: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C3_C * (size_t)B, st));
    const int units = cdiv(W, 16) * H;
    int per_image = max(1, 256 / (2 * B));                  // two channel halves per image tile set, one workgroup per CU
    per_image = min(per_image, cdiv(units, C3S_W));
    const size_t lds = sizeof(unsigned short) * (size_t)3 * C3S_N * C3_LDB + sizeof(float) * (size_t)C3S_W * C3S_N * 2;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_split_kernel, lds));
    hipLaunchKernelGGL(conv3x3_c64_split_kernel, dim3(per_image, B, C3_C / C3S_N), dim3(C3S_W * 64), lds, st, planes,
                       (int64_t)B * H * W * C3_C, w_tap_major, out, stats, H, W);
    MSM_CHECK_LAUNCH("msm_conv3x3_c64_split");
    return MSM_OK;
}

static int conv3x3_c64_launch(const float* in, const float* w_tap_major, float* out, double* stats, int stats_cleared, int B, int H, int W,
                              int bf, void* stream) {
    MSM_REQUIRE(in && w_tap_major && out && in != out, "msm_conv3x3_c64_f32: null or aliased pointer");
    MSM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W * C3_C < ((int64_t)1 << 31), "msm_conv3x3_c64_f32: bad sizes B=%d H=%d W=%d", B, H, W);
    MSM_REQUIRE(((((uintptr_t)in) | ((uintptr_t)w_tap_major) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "msm_conv3x3_c64_f32: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C3_C * (size_t)B, st));
    // workgroups per image: about one round of the chip over the batch, never more than the image has tiles for
    const int units = cdiv(W, 16) * H;
    int per_image = max(1, 256 / B);
    per_image = min(per_image, cdiv(units, C3_W));
    // two 16-pixel blocks per wave (8 waves per workgroup) when the image rows split into 32-pixel tiles without much waste
    // (measured at B = 8, 120 x 160: bf16 55.4 -> 49.9 us, fp32 108.8 -> 108.2: the fp32 form keeps its round-3 shape; same output bits)
    const int wopt = opt(MSM_OPT_CONV3_WIDE);
    const bool wide = (wopt == 1 || (wopt == MSM_OPT_AUTO && bf)) && W >= 32;
    const int units2 = cdiv(W, 32) * H;
    const int per_image2 = min(max(1, 256 / B), cdiv(units2, 8));
    if (bf == 1 && wopt == MSM_OPT_AUTO && W >= 32) {
        // round 6: one output row of a 32-pixel strip per unit, its three input rows requested together (conv3x3_c64_rows_bf16_kernel)
        int pmax = max(1, 256 / B);
        if (pmax >= 8) pmax &= ~7;
        const int runits = cdiv(W, 32) * H;
        const int rper = pmax >= 8 ? min(pmax, 8 * cdiv(cdiv(runits, 8), C3R_W)) : min(pmax, cdiv(runits, C3R_W));
        const size_t lds = sizeof(unsigned short) * (size_t)C3_C * C3_LDB + sizeof(float) * (size_t)C3R_W * C3_C * 2;
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_rows_bf16_kernel, lds));
        hipLaunchKernelGGL(conv3x3_c64_rows_bf16_kernel, dim3(rper, B), dim3(C3R_W * 64), lds, st, in, w_tap_major, out, stats, H, W);
    } else if (bf) {
      auto go = [&](auto tag) -> int {
        constexpr int BFV = decltype(tag)::value;
        // (one load per input ROW with lane shifts for dx = -1 / +1, as the split kernel does, was measured for this form too: bitwise the
        // same result, 55 us either way at B = 8, 120 x 160 -- the per-tile chain load -> split -> MFMA does not overlap with itself at
        // 2.3 tiles per wave, whichever way the operands arrive)
        if (wide) {
            const size_t lds = sizeof(unsigned short) * (size_t)C3_C * C3_LDB + sizeof(float) * (size_t)8 * C3_C * 2;
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<false, BFV, 2, 8>, lds));
            hipLaunchKernelGGL((conv3x3_c64_kernel<false, BFV, 2, 8>), dim3(per_image2, B), dim3(8 * 64), lds, st, in, w_tap_major, nullptr, out, stats,
                               H, W);
        } else {
            const size_t lds = sizeof(unsigned short) * (size_t)C3_C * C3_LDB + sizeof(float) * (size_t)C3_W * C3_C * 2;
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<false, BFV>, lds));
            hipLaunchKernelGGL((conv3x3_c64_kernel<false, BFV>), dim3(per_image, B), dim3(C3_W * 64), lds, st, in, w_tap_major, nullptr, out, stats,
                               H, W);
        }
        return MSM_OK;
      };
      const int rc = bf == 2 ? go(std::integral_constant<int, 2>{}) : go(std::integral_constant<int, 1>{});
      if (rc != MSM_OK) return rc;
    } else if (wide) {
        const size_t lds = sizeof(float) * ((size_t)C3_C * C3_LD + (size_t)8 * C3_C * 2);
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<false, 0, 2, 8>, lds));
        hipLaunchKernelGGL((conv3x3_c64_kernel<false, 0, 2, 8>), dim3(per_image2, B), dim3(8 * 64), lds, st, in, w_tap_major, nullptr, out, stats, H,
                           W);
    } else {
        const size_t lds = sizeof(float) * ((size_t)C3_C * C3_LD + (size_t)C3_W * C3_C * 2);
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<false>, lds));
        hipLaunchKernelGGL(conv3x3_c64_kernel<false>, dim3(per_image, B), dim3(C3_W * 64), lds, st, in, w_tap_major, nullptr, out, stats, H, W);
    }
    MSM_CHECK_LAUNCH("msm_conv3x3_c64_f32");
    return MSM_OK;
}

extern "C" int msm_conv3x3_c64_f32(const float* in, const float* w_tap_major, float* out, double* stats,
                                   int stats_cleared, int B, int H, int W, void* stream) {
    return conv3x3_c64_launch(in, w_tap_major, out, stats, stats_cleared, B, H, W, 0, stream);
}

extern "C" int msm_conv3x3_c64_bf16(const float* in, const float* w_tap_major, float* out, double* stats,
                                    int stats_cleared, int B, int H, int W, void* stream) {
    return conv3x3_c64_launch(in, w_tap_major, out, stats, stats_cleared, B, H, W, 1, stream);
}

static int conv3x3_c64_nchw_launch(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W, int Cout, int bf,
                                   void* stream) {
    MSM_REQUIRE(in && w_tap_major && out && in != out, "msm_conv3x3_c64_nchw_f32: null or aliased pointer");
    MSM_REQUIRE(B > 0 && H > 0 && W > 0 && W % 4 == 0 && Cout > 0 && Cout % C3_C == 0 && Cout <= 1024,
                "msm_conv3x3_c64_nchw_f32: need W %% 4 == 0 and Cout a multiple of 64 (B=%d H=%d W=%d Cout=%d)", B, H, W, Cout);
    MSM_REQUIRE((int64_t)H * W * C3_C < ((int64_t)1 << 31), "msm_conv3x3_c64_nchw_f32: image too large");
    MSM_REQUIRE(((((uintptr_t)in) | ((uintptr_t)w_tap_major) | ((uintptr_t)out)) & 15) == 0, "msm_conv3x3_c64_nchw_f32: misaligned pointer");
    const int units = cdiv(W, 16) * H;
    const int slices = Cout / C3_C;
    int per_image = max(1, 256 / (B * slices));
    per_image = min(per_image, cdiv(units, C3_W));
    if (bf) {
        auto go = [&](auto tag) -> int {
            constexpr int BFV = decltype(tag)::value;
            const size_t lds = sizeof(unsigned short) * (size_t)C3_C * C3_LDB + sizeof(float) * (size_t)C3_W * C3_C * 2;
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<true, BFV>, lds));
            hipLaunchKernelGGL((conv3x3_c64_kernel<true, BFV>), dim3(per_image, B, slices), dim3(C3_W * 64), lds, (hipStream_t)stream, in, w_tap_major,
                               bias, out, nullptr, H, W);
            return MSM_OK;
        };
        const int rc = bf == 2 ? go(std::integral_constant<int, 2>{}) : go(std::integral_constant<int, 1>{});
        if (rc != MSM_OK) return rc;
    } else {
        const size_t lds = sizeof(float) * ((size_t)C3_C * C3_LD + (size_t)C3_W * C3_C * 2);
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_kernel<true>, lds));
        hipLaunchKernelGGL(conv3x3_c64_kernel<true>, dim3(per_image, B, slices), dim3(C3_W * 64), lds, (hipStream_t)stream, in, w_tap_major,
                           bias, out, nullptr, H, W);
    }
    MSM_CHECK_LAUNCH("msm_conv3x3_c64_nchw_f32");
    return MSM_OK;
}

extern "C" int msm_conv3x3_c64_nchw_f32(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                                        int Cout, void* stream) {
    return conv3x3_c64_nchw_launch(in, w_tap_major, bias, out, B, H, W, Cout, 0, stream);
}

extern "C" int msm_conv3x3_c64_nchw_bf16(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                                         int Cout, void* stream) {
    return conv3x3_c64_nchw_launch(in, w_tap_major, bias, out, B, H, W, Cout, 1, stream);
}
extern "C" int msm_conv3x3_c64_f16(const float* in, const float* w_tap_major, float* out, double* stats,
                                   int stats_cleared, int B, int H, int W, void* stream) {
    return conv3x3_c64_launch(in, w_tap_major, out, stats, stats_cleared, B, H, W, 2, stream);
}
extern "C" int msm_conv3x3_c64_nchw_f16(const float* in, const float* w_tap_major, const float* bias, float* out, int B, int H, int W,
                                        int Cout, void* stream) {
    return conv3x3_c64_nchw_launch(in, w_tap_major, bias, out, B, H, W, Cout, 2, stream);
}

extern "C" int msm_conv3x3_c64_f16h(const void* in_f16, const float* w_tap_major, float* out, double* stats, int stats_cleared, int B, int H,
                                    int W, void* stream) {
    MSM_REQUIRE(in_f16 && w_tap_major && out, "msm_conv3x3_c64_f16h: null pointer");
    MSM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W * C3_C < ((int64_t)1 << 31), "msm_conv3x3_c64_f16h: bad sizes B=%d H=%d W=%d", B, H, W);
    MSM_REQUIRE(((((uintptr_t)in_f16) | ((uintptr_t)w_tap_major) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "msm_conv3x3_c64_f16h: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C3_C * (size_t)B, st));
    // workgroups per image: about one round of the chip over the batch (a multiple of 8 when there are that many: the XCD-contiguous unit
    // ranges); rows per unit: as few as keep every unit on its own wave, at most C3R_R
    int pmax = max(1, 256 / B);
    if (pmax >= 8) pmax &= ~7;
    const int xt = cdiv(W, 32);
    int R = 1;
    while (R < C3R_R && xt * cdiv(H, R) > pmax * C3R_W) ++R;
    const int units = xt * cdiv(H, R);
    const int per_image = pmax >= 8 ? min(pmax, 8 * cdiv(cdiv(units, 8), C3R_W)) : min(pmax, cdiv(units, C3R_W));
    const size_t lds = sizeof(unsigned short) * (size_t)C3_C * C3_LDB + sizeof(float) * (size_t)C3R_W * C3_C * 2;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv3x3_c64_rows_kernel, lds));
    hipLaunchKernelGGL(conv3x3_c64_rows_kernel, dim3(per_image, B), dim3(C3R_W * 64), lds, st, (const uint16_t*)in_f16, w_tap_major, out, stats, H,
                       W, R);
    MSM_CHECK_LAUNCH("msm_conv3x3_c64_f16h");
    return MSM_OK;
}
