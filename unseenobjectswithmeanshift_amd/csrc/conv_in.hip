// Input projections of the pixel decoder (see include/msm_hip.h: msm_conv1x1_in_f32):
//     out[b][p][o] = sum_k w[o][k] * x[b][k][p] + bias[o],   o < 64,  k < Cin in {256 .. 2048},  x NCHW
//     stats[b][o]  += (sum_p out, sum_p out^2)               (the GroupNorm statistics of the result)
//
// Reference: `input_proj[l] = Conv2d(Cin, 64, 1) + GroupNorm(32, 64)` on res3/res4/res5 (msdeformattn.py:212-220,
// 326-329) and the FPN lateral `Conv2d(256, 64, 1, bias=False) + GroupNorm` on res2 (:225-238, 343-347).
//
// These are deep-K, 64-wide products: a stream over the backbone features (137 + 157 MB at B = 8) with 32 FLOP per
// byte.  The tiled GEMM has too few 64x64 tiles to cover the chip on the coarse levels (res5: 40 tiles with a
// 2048-deep K loop each -> 33 us for 20 MB) and needs a second pass over its output for the GroupNorm statistics.
// Here a workgroup owns ONE 64-pixel tile of one image and its 8 waves split K:
//   * MFMA orientation D^T: rows = output channels (A = w, pre-packed in fragment order so that a wave's operand load
//     is 512 contiguous bytes), cols = 16 pixels (B = x: a lane loads NT consecutive pixels of one channel row with
//     one 4*NT-byte load, pixel block nt of the tile being the strided set {px0 + NT*n + nt}).  K order inside an
//     8-deep group is k = lq*2 + j for step j on both operands;
//   * every wave accumulates the full 64 x 64 tile over its K slice (64 accumulator registers, no redundant loads
//     of x), then the eight partial tiles are summed through LDS in two half rounds, in a fixed order
//     (deterministic), wave w finishing blocks (channels 16*(w&3).., pixels 16*(w>>2).. and 32 + 16*(w>>2)..);
//   * the finishing wave adds the bias, stores token-major float4s and reduces sum / sum-of-squares of its 16
//     channels over its pixels; one double atomic per (workgroup, channel, moment) lands in stats.
#include <math.h>
#include <stdlib.h>

#include "bf16.h"
#include "common.h"

namespace msm {

constexpr int CI_W = 8;              // waves per workgroup = K slices
constexpr int CI_O = 64;             // output channels

// NT = 16-pixel blocks per workgroup (64 accumulator registers at NT = 4), D = K groups (of 8) in flight per wave.
// Wide tiles (NT = 4, D = 2) read w once per 64 pixels and suit the fine levels; the coarse levels have too few
// pixels to cover the chip with them, so they take NT = 1 with a deep ring of loads (D = 8) instead.
// One tile: image b of x (Cin x HW), pixels [tile*16*NT, +16*NT).  out / stats point at this LEVEL (image b is indexed here).
template <int NT, int D>
__device__ __forceinline__ void conv_in_tile(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                             float* __restrict__ out, int64_t out_sb, double* __restrict__ stats, int Cin, int HW,
                                             int b, int tile, float4* red) {
    constexpr int NB = 4 * NT;                        // 16 x 16 blocks of the tile
    constexpr int HR = NT >= 2 ? 2 : 1;               // half rounds of the reduction (8 blocks each; NB = 4: one round of 4)
    constexpr int RB = NB / HR;                       // blocks per round
    float* st = reinterpret_cast<float*>(red + CI_W * RB * 64);    // [2][64 ch][2], after the [CI_W][RB blocks][64 lanes] partial tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int px0 = tile * (16 * NT);
    const int kw = Cin / CI_W;                        // K slice of this wave (a multiple of 8*D)
    const int k0 = wave * kw;
    // buffer descriptors (SGPRs) over this image of x and over w; per-lane byte offsets are loop invariant,
    // the walk along K goes through the scalar offset of the load
    auto uniform_ptr = [](const void* p) {
        const uint64_t u = (uint64_t)p;
        return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
    };
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + (int64_t)b * Cin * HW), 0, Cin * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(w), 0, CI_O * Cin * 4, 0x00020000);
    // x: a lane loads NT consecutive pixels of one channel row (one 4*NT-byte load; 16 lanes = 64*NT contiguous bytes);
    // pixel block nt of the tile is therefore the strided set {px0 + NT*n + nt}.  w is pre-packed in fragment order
    // (include/msm_hip.h): the four A operands of a group are 512-byte contiguous wave loads.
    const unsigned xo = 4u * (unsigned)(lq * 2 * HW + min(px0 + NT * lj, HW - NT));
    const unsigned wo = 8u * (unsigned)lane;

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K is walked in groups of 8: step j in {0, 1} of a group covers k = lq*2 + j on both operands, so a lane's A
    // operands of a group are one 8-byte load per channel block.  D register sets form a ring: the loads of group
    // g + D are issued right after the MFMAs of group g (the loop is unrolled by D, so no register copies).
    float2 wa[D][4];
    float xa[D][NT][2];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load = [&](int kg, float2 (&wf)[4], float (&xf)[NT][2]) {
        const unsigned ks = (unsigned)(k0 + kg * 8);            // wave-uniform
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(wr, wo, (ks / 8 * 4 + mt) * 512u, 0);
            wf[mt] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned so = (ks + j) * (unsigned)HW * 4u;
            if constexpr (NT == 4) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, so, 0);
                xf[0][j] = __uint_as_float(t.x); xf[1][j] = __uint_as_float(t.y);
                xf[2][j] = __uint_as_float(t.z); xf[3][j] = __uint_as_float(t.w);
            } else if constexpr (NT == 2) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(xr, xo, so, 0);
                xf[0][j] = __uint_as_float(t.x); xf[1][j] = __uint_as_float(t.y);
            } else {
                xf[0][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, xo, so, 0));
            }
        }
    };
    auto mma = [&](const float2 (&wf)[4], const float (&xf)[NT][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float a = j == 0 ? wf[mt].x : wf[mt].y;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(a, xf[nt][j], acc[mt][nt]);
            }
    };
    const int groups = kw / 8;                        // a multiple of D
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, wa[d], xa[d]);
#pragma unroll 1
    for (int kg = 0; kg < groups; kg += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            mma(wa[d], xa[d]);
            // unconditional (the last trips re-read the final group): with a conditional load the number of loads in flight at the
            // loop head depends on the path and hipcc waits for ALL of them there (s_waitcnt vmcnt(0)) -- the ring then exposes
            // one memory round trip per D groups
            load(min(kg + D + d, groups - 1), wa[d], xa[d]);
        }
    }

    // ---- epilogue of one finished 16 x 16 block: bias, token-major store, GroupNorm moments into the LDS table ----
    if (stats) {
        for (int i = tid; i < 2 * CI_O * 2; i += CI_W * 64) st[i] = 0.f;
        __syncthreads();
    }
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};     // moments of this wave's finished blocks (one channel block)
    auto finish = [&](const f32x4& v, int mt, int nt) {
        const int ch = mt * 16 + lq * 4;
        const int px = px0 + NT * lj + nt;                 // pixel block nt holds pixels px0 + NT*n + nt
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + ch);
        const float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
        if (px < HW) {
            *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * CI_O + ch) = make_float4(v0, v1, v2, v3);
            sm[0] += v0; sm[1] += v1; sm[2] += v2; sm[3] += v3;
            sq[0] += v0 * v0; sq[1] += v1 * v1; sq[2] += v2 * v2; sq[3] += v3 * v3;
        }
    };
    {
        // sum the 8 partial tiles through LDS in HR rounds of RB blocks, fixed order (deterministic).  Block id inside
        // a round = mt + 4*q with pixel block nt = h*(NT/HR) + q; wave w finishes block w of the round
#pragma unroll
        for (int h = 0; h < HR; ++h) {
            if (h) __syncthreads();
#pragma unroll
            for (int q = 0; q < NT / HR; ++q)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 v = acc[mt][h * (NT / HR) + q];
                    red[(wave * RB + mt + 4 * q) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
            if (wave < RB) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < CI_W; ++s) {
                    const float4 v = red[(s * RB + wave) * 64 + lane];
                    t += f32x4{v.x, v.y, v.z, v.w};
                }
                finish(t, wave & 3, h * (NT / HR) + (wave >> 2));
            }
        }
    }
    if (stats) {
        // a wave's blocks all belong to channel block wave & 3: reduce over its 16 pixels per lane quarter, one LDS slot
        // per (wave >> 2, channel) -- no atomics below the per-workgroup double adds, so the sums are reproducible
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            {          // (over the 16-lane row in the order 1, 2, 4, 8 of the shfl_xor loop it replaces, on DPP: bitwise the same sums)
                sm[r] += wave_xor_dpp1(sm[r]);
                sq[r] += wave_xor_dpp1(sq[r]);
                sm[r] += wave_xor_dpp2(sm[r]);
                sq[r] += wave_xor_dpp2(sq[r]);
                sm[r] += wave_xor_dpp4(sm[r]);
                sq[r] += wave_xor_dpp4(sq[r]);
                sm[r] += wave_xor_dpp8(sm[r]);
                sq[r] += wave_xor_dpp8(sq[r]);
            }
        }
        if (lj == 0 && wave < RB) {
            const int ch = (wave & 3) * 16 + lq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[((wave >> 2) * CI_O + ch + r) * 2 + 0] = sm[r];
                st[((wave >> 2) * CI_O + ch + r) * 2 + 1] = sq[r];
            }
        }
        __syncthreads();
        if (tid < CI_O * 2) {
            const double v = (double)st[tid] + (double)st[CI_O * 2 + tid];
            atomicAdd(stats + (int64_t)b * CI_O * 2 + tid, v);
        }
    }
}

template <int NT, int D>
__global__ __launch_bounds__(CI_W * 64, 4) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int64_t out_sb, double* __restrict__ stats, int Cin, int HW) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    conv_in_tile<NT, D>(x, w, bias, out, out_sb, stats, Cin, HW, blockIdx.y, blockIdx.x, red);
}

// ---- shallow K (the FPN lateral on res2: Cin = 256, 19 200 pixels per image) -------------------------------------------------
// With K = 256 the eight-way K split above is all reduction (70 us at B = 8), and the tiled GEMM needs a second pass over its
// output for the GroupNorm moments (61 + 10 us).  Here every wave owns its own 16*NT-pixel tile over the full K -- no partial
// tiles, no reduction --, x and the packed weight (64 KB, L2 resident) stream through a ring of D k-groups per wave, four-wave
// workgroups are dispatched as slots free up (4800 tiles over 1024 SIMDs: no fixed assignment to round up), and the moments
// leave as one double atomic per (workgroup, channel, moment) as above.
// WV: waves per workgroup (WV x 16 NT consecutive pixels of every channel row)
template <int NT, int D, int WV>
__global__ __launch_bounds__(WV * 64, 5) void conv_in_shallow_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int64_t out_sb, double* __restrict__ stats, int Cin, int HW) {
    __shared__ float st[WV * CI_O * 2];                                 // [wave][64 ch][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int tile = blockIdx.x * WV + wave;
    const int px0 = tile * (16 * NT);
    const bool live = px0 < HW;                                           // wave-uniform
    const bool has_stats = stats != nullptr;
    if (live) {
        auto uniform_ptr = [](const void* p) {
            const uint64_t u = (uint64_t)p;
            return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                           (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
        };
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + (int64_t)b * Cin * HW), 0, Cin * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(w), 0, CI_O * Cin * 4, 0x00020000);
        const unsigned xo = 4u * (unsigned)(lq * 2 * HW + min(px0 + NT * lj, HW - NT));
        const unsigned wo = 8u * (unsigned)lane;
        f32x4 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float2 wa[D][4];
        float xa[D][NT][2];
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        auto load = [&](int kg, float2 (&wf)[4], float (&xf)[NT][2]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(wr, wo, (unsigned)(kg * 4 + mt) * 512u, 0);
                wf[mt] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned so = (unsigned)(kg * 8 + j) * (unsigned)HW * 4u;
                if constexpr (NT == 4) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, so, 0);
                    xf[0][j] = __uint_as_float(t.x); xf[1][j] = __uint_as_float(t.y);
                    xf[2][j] = __uint_as_float(t.z); xf[3][j] = __uint_as_float(t.w);
                } else {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(xr, xo, so, 0);
                    xf[0][j] = __uint_as_float(t.x); xf[1][j] = __uint_as_float(t.y);
                }
            }
        };
        auto mma = [&](const float2 (&wf)[4], const float (&xf)[NT][2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float a = j == 0 ? wf[mt].x : wf[mt].y;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(a, xf[nt][j], acc[mt][nt]);
                }
        };
        const int groups = Cin / 8;                       // a multiple of D
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, wa[d], xa[d]);
#pragma unroll 1
        for (int kg = 0; kg < groups; kg += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                mma(wa[d], xa[d]);
                load(min(kg + D + d, groups - 1), wa[d], xa[d]);      // unconditional: see conv_in_tile
            }
        }
        // bias, token-major store, and the tile's moments: reduced over the 16 pixels of a lane quarter, one LDS slot per
        // (wave, channel)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int ch = mt * 16 + lq * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + ch);
            float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int px = px0 + NT * lj + nt;             // pixel block nt holds pixels px0 + NT*n + nt
                const f32x4 v = acc[mt][nt];
                const float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
                if (px < HW) {
                    *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * CI_O + ch) = make_float4(v0, v1, v2, v3);
                    sm[0] += v0; sm[1] += v1; sm[2] += v2; sm[3] += v3;
                    sq[0] += v0 * v0; sq[1] += v1 * v1; sq[2] += v2 * v2; sq[3] += v3 * v3;
                }
            }
            if (has_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    {          // (over the 16-lane row in the order 1, 2, 4, 8 of the shfl_xor loop it replaces, on DPP: bitwise the same sums)
                        sm[r] += wave_xor_dpp1(sm[r]);
                        sq[r] += wave_xor_dpp1(sq[r]);
                        sm[r] += wave_xor_dpp2(sm[r]);
                        sq[r] += wave_xor_dpp2(sq[r]);
                        sm[r] += wave_xor_dpp4(sm[r]);
                        sq[r] += wave_xor_dpp4(sq[r]);
                        sm[r] += wave_xor_dpp8(sm[r]);
                        sq[r] += wave_xor_dpp8(sq[r]);
                    }
                    if (lj == 0) {
                        st[(wave * CI_O + ch + r) * 2 + 0] = sm[r];
                        st[(wave * CI_O + ch + r) * 2 + 1] = sq[r];
                    }
                }
            }
        }
    } else if (has_stats) {
        for (int i = lane; i < CI_O * 2; i += 64) st[wave * CI_O * 2 + i] = 0.f;
    }
    if (has_stats) {
        // the workgroup's slots are summed in a fixed order, so nothing below the per-workgroup double adds depends on timing
        __syncthreads();
        if (tid < CI_O * 2) {
            double v = 0.0;
#pragma unroll
            for (int s_ = 0; s_ < WV; ++s_) v += (double)st[s_ * CI_O * 2 + tid];
            atomicAdd(stats + (int64_t)b * CI_O * 2 + tid, v);
        }
    }
}

// All input projections of the pixel decoder in ONE launch: the levels are independent, and on its own each coarse level
// fills a fraction of the chip (res5 at B = 8: 152 workgroups).  Workgroups are numbered level by level in the order given
// (deepest K first, so the longest-running tiles start first); cfg selects the tile shape per level.
constexpr int CI_MAXL = 4;
struct ConvInLevels {
    int n;
    const float* x[CI_MAXL];
    const float* w[CI_MAXL];
    const float* bias[CI_MAXL];
    float* out[CI_MAXL];
    double* stats[CI_MAXL];
    int Cin[CI_MAXL], HW[CI_MAXL], tiles[CI_MAXL], cfg[CI_MAXL];
    int first[CI_MAXL + 1];       // first workgroup of each level
};

__global__ __launch_bounds__(CI_W * 64, 4) void conv_in_multi_kernel(ConvInLevels lv, int64_t out_sb) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    int l = 0;
#pragma unroll
    for (int i = 1; i < CI_MAXL; ++i) l += (i < lv.n && (int)blockIdx.x >= lv.first[i]) ? 1 : 0;
    const int local = (int)blockIdx.x - lv.first[l];
    const int b = local / lv.tiles[l], tile = local - b * lv.tiles[l];
    switch (lv.cfg[l]) {
        case 0: conv_in_tile<4, 2>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, tile, red); break;
        case 1: conv_in_tile<2, 4>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, tile, red); break;
        case 2: conv_in_tile<2, 2>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, tile, red); break;
        case 3: conv_in_tile<1, 8>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, tile, red); break;
        default: conv_in_tile<1, 2>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, tile, red); break;
    }
}

// ---- the same projections on the bf16 matrix pipe (round 4; the bf16 plan) ------------------------------------------------------
// The fp32 kernels above are bound by the fp32 matrix pipe (4.4 + 5.0 GFLOP at 61 % of its peak: 66 + 62 us for 137 + 157 MB of
// backbone features, a quarter of HBM speed).  Here both operands enter as hi + lo bf16 pairs -- x split in registers as it
// arrives (x = x_h + x_l up to 2^-17 |x|), the weight pre-split and pre-packed in fragment order -- and a product is three
// v_mfma_f32_16x16x32_bf16 (w_l x_h + w_h x_l + w_h x_h): 48 MFMAs of 16 cycles per 32-deep K group and 64-pixel tile where the
// fp32 form issues 128 of 32 cycles, with fp32-level accuracy (the dropped w_l x_l term is 2^-18 of a product).  What is left is
// the stream.
//   * K order k = 32 g + 8 lq + e on both operands: a lane's B operand of pixel block nt is eight channel rows of x at its NT
//     consecutive pixels (eight 4 NT-byte loads per group), its A operand of channel block mt 16 bytes of the packed weight:
//       packed[(((g * 4 + mt) * 2 + plane) * 64 + lane) * 8 + e] = plane(w)[16 mt + lj][32 g + 8 lq + e],  plane 0 = hi, 1 = lo
//   * four waves per workgroup, two workgroups per CU (<= 256 registers): deep levels split K over the four waves of a tile and
//     meet in LDS once (every wave finishes one channel block of all pixel blocks; one barrier), the shallow lateral (K = 256)
//     gives every wave its own 32-pixel tile over the full K.
constexpr int CL_W = 4;
#ifndef CL_EXP
#define CL_EXP 0   // tuning builds only (tools/probes/conv_in_parts.sh): 1 no MFMAs, 2 no weight loads, 3 no hi / lo split, 4 no stores
#endif

// One workgroup (KSPLIT) or wave walks tiles first, first + stride, ... of image b.  The (tile, K group) pairs form ONE sequence
// through the ring of D load groups: the first groups of the next tile are in flight while a tile is reduced and stored, and the
// moments of all its tiles leave in one set of atomics at the end.
template <int NT, int D, bool KSPLIT, bool WLDS = false>
__device__ __forceinline__ void conv_in_lp_stream(const float* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ out, int64_t out_sb, double* __restrict__ stats, int Cin, int HW,
                                                  int b, int first, int stride, float4* red) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int ntiles = (HW + 16 * NT - 1) / (16 * NT);
    const int t_first = KSPLIT ? first : first * CL_W + wave, t_stride = KSPLIT ? stride : stride * CL_W;
    const int count = t_first < ntiles ? (ntiles - 1 - t_first) / t_stride + 1 : 0;      // wave-uniform (workgroup-uniform with KSPLIT)
    const int groups = (KSPLIT ? Cin / CL_W : Cin) / 32;          // K groups per tile of this wave (a multiple of D)
    const int g0 = KSPLIT ? wave * groups : 0;
    float sm[KSPLIT ? 1 : 4][4], sq[KSPLIT ? 1 : 4][4];           // moments of the finished blocks, per channel block this wave finishes
#pragma unroll
    for (int i = 0; i < (KSPLIT ? 1 : 4); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[i][r] = sq[i][r] = 0.f;
    // WLDS (shallow K, no K split): the packed weight (256 B per input channel) is copied into LDS once per workgroup -- read from L2 by
    // every wave for every tile it is as many bytes as x at 64 pixels per wave and twice as many at 32, and thousands of waves ask
    // the same 64 lines of a K group at the same time (measured: 62 -> 45 us for the FPN lateral without the weight loads)
    const unsigned char* wl = reinterpret_cast<const unsigned char*>(red) + (KSPLIT ? 0 : sizeof(float) * CL_W * CI_O * 2);
    if constexpr (WLDS) {
        static_assert(!KSPLIT, "the weight copy in LDS is for the full-K form");
        uint4* dst = reinterpret_cast<uint4*>(const_cast<unsigned char*>(wl));
        const uint4* src = reinterpret_cast<const uint4*>(w);
        for (int i = tid; i < Cin * 16; i += CL_W * 64) dst[i] = src[i];
        __syncthreads();
    }
    if (count > 0) {
        auto uniform_ptr = [](const void* p) {
            const uint64_t u = (uint64_t)p;
            return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                           (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
        };
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + (int64_t)b * Cin * HW), 0, Cin * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(w), 0, CI_O * Cin * 4, 0x00020000);
        const unsigned wo = 16u * (unsigned)lane;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 wf[WLDS ? 1 : D][4][2];
        float xf[D][8][NT];
        int cg = 0;                                                 // K group of the next mma (WLDS: its weight fragments are read there)
        // load cursor: (tile, group) of the next load; past the last pair it stays on the last one (unconditional loads: see conv_in_tile)
        int lt = t_first, lg = 0;
        const int t_last = t_first + (count - 1) * t_stride;
        auto load = [&](u32x4 (&wv)[4][2], float (&xv)[8][NT]) {
            const unsigned gg = (unsigned)(g0 + lg);                // wave-uniform
            const unsigned xo = 4u * (unsigned)(lq * 8 * HW + min(lt * (16 * NT) + NT * lj, HW - NT));
            if constexpr (!WLDS)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#if CL_EXP == 2
                    wv[mt][pl] = u32x4{gg, wo, (unsigned)mt, (unsigned)pl};
#else
                    wv[mt][pl] = __builtin_amdgcn_raw_buffer_load_b128(wr, wo, ((gg * 4 + mt) * 2 + pl) * 1024u, 0);
#endif
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned so = (gg * 32 + e) * (unsigned)HW * 4u;
                if constexpr (NT == 4) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, so, 0);
                    xv[e][0] = __uint_as_float(t.x); xv[e][1] = __uint_as_float(t.y);
                    xv[e][2] = __uint_as_float(t.z); xv[e][3] = __uint_as_float(t.w);
                } else if constexpr (NT == 2) {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(xr, xo, so, 0);
                    xv[e][0] = __uint_as_float(t.x); xv[e][1] = __uint_as_float(t.y);
                } else {
                    xv[e][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, xo, so, 0));
                }
            }
            // advance, sticking to the last pair
            const bool wrap = lg + 1 == groups;
            const bool end = wrap && lt == t_last;
            lg = end ? lg : (wrap ? 0 : lg + 1);
            lt = (wrap && !end) ? lt + t_stride : lt;
        };
        f32x4 acc[4][NT];
        auto mma = [&](u32x4 (&wv)[4][2], const float (&xv)[8][NT]) {
            if constexpr (WLDS) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        wv[mt][pl] = *reinterpret_cast<const u32x4*>(wl + ((cg * 4 + mt) * 2 + pl) * 1024 + lane * 16);
                cg = cg + 1 == groups ? 0 : cg + 1;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#if CL_EXP == 3
                const bf16x8 xh = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(xv[0][nt]), __float_as_uint(xv[1][nt]), __float_as_uint(xv[2][nt]), __float_as_uint(xv[3][nt])});
                const bf16x8 xl = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(xv[4][nt]), __float_as_uint(xv[5][nt]), __float_as_uint(xv[6][nt]), __float_as_uint(xv[7][nt])});
#else
                const Split4 s0 = split4(xv[0][nt], xv[1][nt], xv[2][nt], xv[3][nt]), s1 = split4(xv[4][nt], xv[5][nt], xv[6][nt], xv[7][nt]);
                const bf16x8 xh = cat8(s0.hi, s1.hi), xl = cat8(s0.lo, s1.lo);
#endif
#if CL_EXP == 1
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const u32x4 a = __builtin_bit_cast(u32x4, xh), c = __builtin_bit_cast(u32x4, xl);
                    acc[mt][nt][0] += __uint_as_float((a.x ^ c.x ^ wv[mt][0].x ^ wv[mt][1].x) & 0x3fffffffu);
                    acc[mt][nt][1] += __uint_as_float((a.y ^ c.y ^ wv[mt][0].y ^ wv[mt][1].y) & 0x3fffffffu);
                    acc[mt][nt][2] += __uint_as_float((a.z ^ c.z ^ wv[mt][0].z ^ wv[mt][1].z) & 0x3fffffffu);
                    acc[mt][nt][3] += __uint_as_float((a.w ^ c.w ^ wv[mt][0].w ^ wv[mt][1].w) & 0x3fffffffu);
                }
#else
                // the three terms of a product walk the four channel blocks: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][1]), xh, acc[mt][nt]);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][0]), xl, acc[mt][nt]);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][0]), xh, acc[mt][nt]);
#endif
            }
        };
        // bias, token-major store, moments of one finished 16 x 16 block
        auto finish = [&](const f32x4& v, int mt, int nt, int px0, float (&s1)[4], float (&s2)[4]) {
            const int ch = mt * 16 + lq * 4;
            const int px = px0 + NT * lj + nt;                 // pixel block nt holds pixels px0 + NT*n + nt
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + ch);
            const float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
            if (px < HW) {
#if CL_EXP == 4
                if (v0 == 1.2345e-20f)
#endif
                *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * CI_O + ch) = make_float4(v0, v1, v2, v3);
                s1[0] += v0; s1[1] += v1; s1[2] += v2; s1[3] += v3;
                s2[0] += v0 * v0; s2[1] += v1 * v1; s2[2] += v2 * v2; s2[3] += v3 * v3;
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load(wf[WLDS ? 0 : d], xf[d]);
#pragma unroll 1
        for (int it = 0; it < count; ++it) {
            const int px0 = (t_first + it * t_stride) * (16 * NT);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int g = 0; g < groups; g += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    mma(wf[WLDS ? 0 : d], xf[d]);
                    load(wf[WLDS ? 0 : d], xf[d]);
                }
            }
            if constexpr (KSPLIT) {
                // the four partial tiles meet in LDS; wave w finishes channel block w of every pixel block (fixed order)
                if (it) __syncthreads();                       // the previous tile's sums have been read
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4 v = acc[mt][nt];
                        red[((wave * 4 + mt) * NT + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                __syncthreads();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < CL_W; ++s) {
                        const float4 v = red[((s * 4 + wave) * NT + nt) * 64 + lane];
                        t += f32x4{v.x, v.y, v.z, v.w};
                    }
                    finish(t, wave, nt, px0, sm[0], sq[0]);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) finish(acc[mt][nt], mt, nt, px0, sm[mt], sq[mt]);
            }
        }
    }
    if (!stats) return;
    // moments: over the 16 pixels of a lane quarter, then one double atomic per (workgroup, channel, moment)
#pragma unroll
    for (int i = 0; i < (KSPLIT ? 1 : 4); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            {          // (over the 16-lane row in the order 1, 2, 4, 8 of the shfl_xor loop it replaces, on DPP: bitwise the same sums)
                sm[i][r] += wave_xor_dpp1(sm[i][r]);
                sq[i][r] += wave_xor_dpp1(sq[i][r]);
                sm[i][r] += wave_xor_dpp2(sm[i][r]);
                sq[i][r] += wave_xor_dpp2(sq[i][r]);
                sm[i][r] += wave_xor_dpp4(sm[i][r]);
                sq[i][r] += wave_xor_dpp4(sq[i][r]);
                sm[i][r] += wave_xor_dpp8(sm[i][r]);
                sq[i][r] += wave_xor_dpp8(sq[i][r]);
            }
    if constexpr (KSPLIT) {
        if (lj == 0 && count > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                atomicAdd(stats + ((int64_t)b * CI_O + wave * 16 + lq * 4 + r) * 2 + 0, (double)sm[0][r]);
                atomicAdd(stats + ((int64_t)b * CI_O + wave * 16 + lq * 4 + r) * 2 + 1, (double)sq[0][r]);
            }
        }
    } else {
        float* st = reinterpret_cast<float*>(red);             // [CL_W][64 ch][2]
        if (lj == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[(wave * CI_O + mt * 16 + lq * 4 + r) * 2 + 0] = sm[mt][r];
                    st[(wave * CI_O + mt * 16 + lq * 4 + r) * 2 + 1] = sq[mt][r];
                }
        }
        __syncthreads();
        if (tid < CI_O * 2) {
            double v = 0.0;
#pragma unroll
            for (int s_ = 0; s_ < CL_W; ++s_) v += (double)st[s_ * CI_O * 2 + tid];
            atomicAdd(stats + (int64_t)b * CI_O * 2 + tid, v);
        }
    }
}

template <int NT, int D, bool KSPLIT, bool WLDS = false>
__global__ __launch_bounds__(CL_W * 64, 2) void conv_in_lp_kernel(const float* __restrict__ x, const uint16_t* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int64_t out_sb,
                                                                  double* __restrict__ stats, int Cin, int HW) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    conv_in_lp_stream<NT, D, KSPLIT, WLDS>(x, w, bias, out, out_sb, stats, Cin, HW, blockIdx.y, blockIdx.x, gridDim.x, red);
}

struct ConvInLpLevels {
    int n;
    const float* x[CI_MAXL];
    const uint16_t* w[CI_MAXL];
    const float* bias[CI_MAXL];
    float* out[CI_MAXL];
    double* stats[CI_MAXL];
    int Cin[CI_MAXL], HW[CI_MAXL], wgs[CI_MAXL] /* workgroups per image */, cfg[CI_MAXL];
    int first[CI_MAXL + 1];
};

__global__ __launch_bounds__(CL_W * 64, 2) void conv_in_lp_multi_kernel(ConvInLpLevels lv, int64_t out_sb) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    int l = 0;
#pragma unroll
    for (int i = 1; i < CI_MAXL; ++i) l += (i < lv.n && (int)blockIdx.x >= lv.first[i]) ? 1 : 0;
    const int local = (int)blockIdx.x - lv.first[l];
    const int b = local / lv.wgs[l], j = local - b * lv.wgs[l];
    switch (lv.cfg[l]) {
        case 0: conv_in_lp_stream<4, 2, true>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, j, lv.wgs[l], red); break;
        case 1: conv_in_lp_stream<2, 2, true>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, j, lv.wgs[l], red); break;
        case 2: conv_in_lp_stream<1, 4, true>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, j, lv.wgs[l], red); break;
        default: conv_in_lp_stream<1, 2, true>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, j, lv.wgs[l], red); break;
    }
}


// ---- the deep levels on the bf16 matrix pipe with the weight broadcast through LDS (round 6: msm_conv1x1_in_multi_wide) ------------
// What bounds conv_in_lp_stream on the deep levels is its weight traffic (k31): every wave re-reads its weight fragments from L2 for
// every 64-pixel tile -- as many bytes as x -- and thousands of waves ask for the same lines at the same moment.  Here an eight-wave
// workgroup covers PW = 8 / KW adjacent 64-pixel tiles x KW slices of K (KW = Cin / the shallowest level's Cin, so that every wave of
// every level walks the same K depth: 131 KB of x per wave at the shipped sizes, one round of workgroups), and the packed weight of a
// 32-deep K group (8 KiB per K slice) arrives ONCE per workgroup by LDS-DMA, in a ring of four stages three groups ahead; the waves
// step through K together (one barrier per group) and read their A fragments from LDS.  x: a ring of two K groups per wave (16
// 16-byte loads in flight per lane).  Waits are counted: loads return in order, a wave issues KW DMA pieces + 8 x loads per group,
// and the x group it is about to use is followed by 8 + KW younger loads (sticky re-reads past the end keep the count
// constant).  K splits meet in LDS at the end (fixed order: slice 0, 1, ...), the moments leave as one double atomic per
// (workgroup, channel, moment).  Arithmetic as conv_in_lp_stream: x = hi + lo in registers, three bf16 MFMAs per product.
constexpr int CW_W = 8;                       // waves per workgroup
constexpr int CW_NBUF = 4;                    // ring stages (a stage is requested three groups ahead)
constexpr int CW_D = 2;                       // x groups in flight per wave (16 16-byte loads per lane; three measured slower, four spill)
constexpr int CW_KWMAX = 4;
constexpr int CW_STAGE = CW_KWMAX * 8192;     // bytes of a stage: one 8-KiB packed K group per K slice

struct ConvInWideLevels {
    int n;
    const float* x[CI_MAXL];
    const uint16_t* w[CI_MAXL];
    const float* bias[CI_MAXL];
    float* out[CI_MAXL];
    double* stats[CI_MAXL];
    int Cin[CI_MAXL], HW[CI_MAXL], kw[CI_MAXL] /* K slices: 1, 2, 4 */, chunks[CI_MAXL] /* workgroups per image */;
    int first[CI_MAXL + 1];
};

__device__ __forceinline__ void cw_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int KW>
__device__ __forceinline__ void conv_in_wide_body(const float* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ out, int64_t out_sb, double* __restrict__ stats, int Cin, int HW, int b,
                                                  int chunk, char* lds) {
    constexpr int PW = CW_W / KW;
    constexpr int NT = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int pt = wave % PW, ks = wave / PW;
    const int px0 = (chunk * PW + pt) * (16 * NT);
    const int groups = Cin / KW / 32;                               // K groups per slice (>= 2)
    const int g0 = ks * groups;
    auto uniform_ptr = [](const void* p) {
        const uint64_t u = (uint64_t)p;
        return (void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
    };
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + (int64_t)b * Cin * HW), 0, Cin * HW * 4, 0x00020000);
    const void* wb = uniform_ptr(w);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned xo = 4u * (unsigned)(lq * 8 * HW + min(px0 + NT * lj, HW - NT));      // (a tile past the map re-reads the last pixels: nothing is stored)
    u32x4 xf[CW_D][8];                                              // raw 16-byte results: component nt of row e = pixel block nt
    int lg = 0;                                                     // group of the next x load / DMA request (sticky at the last one)
    int dg = 0;
    auto load_x = [&](u32x4 (&xv)[8]) {
        const unsigned gg = (unsigned)(g0 + lg);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, (gg * 32 + e) * (unsigned)HW * 4u, 0);
        lg = lg + 1 < groups ? lg + 1 : lg;
    };
    auto request = [&](int stage) {                                 // the packed K group dg of every slice -> ring stage `stage`
#pragma unroll
        for (int s = 0; s < KW; ++s)
            cw_glds16(wb, (unsigned)(s * groups + dg) * 8192u + (unsigned)tid * 16u, (unsigned)(stage * CW_STAGE + s * 8192 + wave * 1024));
        dg = dg + 1 < groups ? dg + 1 : dg;
    };
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const char* stage, const u32x4 (&xv)[8], bool live) {
        u32x4 wv[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(stage + ks * 8192 + (mt * 2 + pl) * 1024 + lane * 16);
                wv[mt][pl] = live ? t : u32x4{0u, 0u, 0u, 0u};       // (a padding round past the last K group adds zeros)
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            auto f = [&](int e) { return __uint_as_float(xv[e][nt]); };
            const Split4 s0 = split4(f(0), f(1), f(2), f(3)), s1 = split4(f(4), f(5), f(6), f(7));
            const bf16x8 xh = cat8(s0.hi, s1.hi), xl = cat8(s0.lo, s1.lo);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][1]), xh, acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][0]), xl, acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16k32(__builtin_bit_cast(bf16x8, wv[mt][0]), xh, acc[mt][nt]);
        }
    };
    // issue order: DMA(0), DMA(1), x(0), DMA(2), x(1); then per group g: [wait, barrier, fragments, MFMAs] DMA(g + 3), x(g + 2).  The x
    // group about to be used is followed by DMA(g + 2), x(g + 1): KW + 8 younger loads (its stage is older still).  Cursors are sticky
    // past the last group (redundant re-reads keep the count constant).  No branch around loads in the loop: with loads on a conditional
    // path hipcc waits for ALL loads at the join.
    request(0);
    request(1);
    load_x(xf[0]);
    request(2);
    load_x(xf[1]);
    int st = 0;                                                     // ring stage of the group in use
#pragma unroll 1
    for (int r = 0; r < groups / CW_D; ++r) {                       // (groups is even: see the launcher)
#pragma unroll
        for (int d = 0; d < CW_D; ++d) {
            if constexpr (KW == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if constexpr (KW == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // every wave's pieces of this stage have landed; everyone is done with the stage before it
            const int nst = st + 3 >= CW_NBUF ? st + 3 - CW_NBUF : st + 3;
            mma(lds + st * CW_STAGE, xf[d], true);
            __builtin_amdgcn_sched_barrier(0);                      // (the reloads stay behind the MFMAs that read the old values: no copies at the back edge)
            request(nst);
            load_x(xf[d]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads are done before it reaches the next barrier
            st = st + 1 == CW_NBUF ? 0 : st + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // the ring is free: reduction scratch from here on

    float sm[4][4], sq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[i][r] = sq[i][r] = 0.f;
    auto finish = [&](const f32x4& v, int mt, int nt) {
        const int ch = mt * 16 + lq * 4;
        const int px = px0 + NT * lj + nt;                 // pixel block nt holds pixels px0 + NT*n + nt
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + ch);
        const float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
        if (px < HW) {
            *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * CI_O + ch) = make_float4(v0, v1, v2, v3);
            sm[mt][0] += v0; sm[mt][1] += v1; sm[mt][2] += v2; sm[mt][3] += v3;
            sq[mt][0] += v0 * v0; sq[mt][1] += v1 * v1; sq[mt][2] += v2 * v2; sq[mt][3] += v3 * v3;
        }
    };
    float4* red = reinterpret_cast<float4*>(lds);                   // [wave][mt][2 pixel blocks][64 lanes] per half: 64 KiB
    if constexpr (KW == 1) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) finish(acc[mt][nt], mt, nt);
    } else {
        // the KW partial tiles of a pixel tile meet in LDS, two pixel blocks at a time; wave (pt, ks) finishes channel blocks ks, ks + KW, ..
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    const f32x4 v = acc[mt][h * 2 + n2];
                    red[((wave * 4 + mt) * 2 + n2) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 4 / KW; ++mi) {
                const int mt = ks + mi * KW;
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KW; ++s) {
                        const float4 v = red[(((s * PW + pt) * 4 + mt) * 2 + n2) * 64 + lane];
                        t += f32x4{v.x, v.y, v.z, v.w};
                    }
                    // (mt is not a compile-time constant here: the moments go through a switch-free table below)
#pragma unroll
                    for (int m2 = 0; m2 < 4; ++m2)
                        if (m2 == mt) finish(t, m2, h * 2 + n2);
                }
            }
        }
    }
    if (!stats) return;
    __syncthreads();
    float* stt = reinterpret_cast<float*>(lds);                     // [CW_W][64 ch][2]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm[i][r] += wave_xor_dpp1(sm[i][r]);
            sq[i][r] += wave_xor_dpp1(sq[i][r]);
            sm[i][r] += wave_xor_dpp2(sm[i][r]);
            sq[i][r] += wave_xor_dpp2(sq[i][r]);
            sm[i][r] += wave_xor_dpp4(sm[i][r]);
            sq[i][r] += wave_xor_dpp4(sq[i][r]);
            sm[i][r] += wave_xor_dpp8(sm[i][r]);
            sq[i][r] += wave_xor_dpp8(sq[i][r]);
        }
    if (lj == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                stt[(wave * CI_O + mt * 16 + lq * 4 + r) * 2 + 0] = sm[mt][r];
                stt[(wave * CI_O + mt * 16 + lq * 4 + r) * 2 + 1] = sq[mt][r];
            }
    }
    __syncthreads();
    if (tid < CI_O * 2) {
        double v = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < CW_W; ++s_) v += (double)stt[s_ * CI_O * 2 + tid];
        atomicAdd(stats + (int64_t)b * CI_O * 2 + tid, v);
    }
}

__global__ __launch_bounds__(CW_W * 64, 1) void conv_in_wide_kernel(ConvInWideLevels lv, int64_t out_sb) {
    extern __shared__ __attribute__((aligned(16))) char cw_lds[];
    int l = 0;
#pragma unroll
    for (int i = 1; i < CI_MAXL; ++i) l += (i < lv.n && (int)blockIdx.x >= lv.first[i]) ? 1 : 0;
    const int local = (int)blockIdx.x - lv.first[l];
    const int b = local / lv.chunks[l], c = local - b * lv.chunks[l];
    switch (lv.kw[l]) {
        case 1: conv_in_wide_body<1>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, c, cw_lds); break;
        case 2: conv_in_wide_body<2>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, c, cw_lds); break;
        default: conv_in_wide_body<4>(lv.x[l], lv.w[l], lv.bias[l], lv.out[l], out_sb, lv.stats[l], lv.Cin[l], lv.HW[l], b, c, cw_lds); break;
    }
}

}  // namespace msm

using namespace msm;

// tile shape for one level: 64 pixels when such tiles cover the chip about twice, else 32 or 16 pixels with a deeper ring of
// loads (the coarse levels are latency bound: few pixels, K up to 2048).  Returns cfg (see conv_in_multi_kernel), sets nt.
static int conv_in_config(int B, int Cin, int HW, int& nt) {
    const int64_t t64 = (int64_t)cdiv(HW, 64) * B;
    nt = t64 >= 512 ? 4 : (t64 >= 128 ? 2 : 1);
    if (const int o = opt(MSM_OPT_CONVIN_NT); o != MSM_OPT_AUTO) nt = o == 4 ? 4 : (o == 2 ? 2 : 1);
    const int kw = Cin / CI_W;
    if (nt == 4) return 0;
    if (nt == 2) return kw % 32 == 0 ? 1 : 2;
    return kw % 64 == 0 ? 3 : 4;
}
static size_t conv_in_lds(int nt) { return sizeof(float4) * CI_W * (4 * nt / (nt >= 2 ? 2 : 1)) * 64 + sizeof(float) * 2 * CI_O * 2; }

static int conv_in_check(const float* x, const float* w, const float* bias, const float* out, const double* stats, int B, int Cin,
                         int HW, int64_t out_batch_stride) {
    MSM_REQUIRE(x && w && out, "msm_conv1x1_in_f32: null pointer");
    MSM_REQUIRE(HW % 4 == 0, "msm_conv1x1_in_f32: HW=%d must be a multiple of 4", HW);
    MSM_REQUIRE(B > 0 && HW > 0 && Cin >= 128 && Cin % (CI_W * 16) == 0, "msm_conv1x1_in_f32: Cin=%d must be a multiple of %d", Cin,
                CI_W * 16);
    MSM_REQUIRE(out_batch_stride >= (int64_t)HW * CI_O && out_batch_stride % 4 == 0, "msm_conv1x1_in_f32: bad output batch stride");
    MSM_REQUIRE((int64_t)Cin * HW < ((int64_t)1 << 30), "msm_conv1x1_in_f32: one image of x must be < 4 GiB (32-bit buffer offsets)");
    MSM_REQUIRE(((((uintptr_t)w) | ((uintptr_t)out) | ((uintptr_t)bias) | ((uintptr_t)x)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "msm_conv1x1_in_f32: misaligned pointer");
    return MSM_OK;
}

extern "C" int msm_conv1x1_in_f32(const float* x, const float* w, const float* bias, float* out, int64_t out_batch_stride,
                                  double* stats, int stats_cleared, int B, int Cin, int HW, void* stream) {
    if (int rc = conv_in_check(x, w, bias, out, stats, B, Cin, HW, out_batch_stride)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * CI_O * (size_t)B, st));
    int nt;
    const int cfg = conv_in_config(B, Cin, HW, nt);
    if (Cin <= 384 && Cin % 64 == 0 && (int64_t)B * HW >= 32 * 1024) {
        // shallow K over many pixels (the FPN lateral): one tile per wave, no K split
        // (measured at B = 8, 120x160: 64-66 us whether a lane takes 2 or 4 pixels, a workgroup 4 or 8 waves, the weight comes from
        // L2 or LDS -- the tiled GEMM's time, without its 10-us moments pass)
        constexpr int SNT = 2, SW = 4;
        dim3 sgrid(cdiv(cdiv(HW, 16 * SNT), SW), B);
        hipLaunchKernelGGL((conv_in_shallow_kernel<SNT, 4, SW>), sgrid, dim3(SW * 64), 0, st, x, w, bias, out, out_batch_stride, stats, Cin, HW);
        MSM_CHECK_LAUNCH("msm_conv1x1_in_f32");
        return MSM_OK;
    }
    dim3 grid(cdiv(HW, 16 * nt), B), block(CI_W * 64);
    const size_t lds = conv_in_lds(nt);
#define CI_LAUNCH(NT_, D_)                                                                                        \
    {                                                                                                             \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_kernel<NT_, D_>, lds));                 \
        hipLaunchKernelGGL((conv_in_kernel<NT_, D_>), grid, block, lds, st, x, w, bias, out, out_batch_stride, stats, Cin, HW); \
    }
    switch (cfg) {
        case 0: CI_LAUNCH(4, 2) break;
        case 1: CI_LAUNCH(2, 4) break;
        case 2: CI_LAUNCH(2, 2) break;
        case 3: CI_LAUNCH(1, 8) break;
        default: CI_LAUNCH(1, 2) break;
    }
#undef CI_LAUNCH
    MSM_CHECK_LAUNCH("msm_conv1x1_in_f32");
    return MSM_OK;
}

extern "C" int msm_conv1x1_in_multi_f32(int n_levels, const float* const* x, const float* const* w_packed, const float* const* bias,
                                        const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                                        int stats_cleared, int B, void* stream) {
    MSM_REQUIRE(n_levels >= 1 && n_levels <= CI_MAXL && x && w_packed && bias && Cin && HW && out,
                "msm_conv1x1_in_multi_f32: bad arguments (1..%d levels)", CI_MAXL);
    hipStream_t st = (hipStream_t)stream;
    ConvInLevels lv;
    lv.n = n_levels;
    int64_t tok = 0;
    int wg = 0, max_nt = 1;
    for (int l = 0; l < n_levels; ++l) {
        float* o = out + tok * CI_O;
        double* s = stats ? stats + (size_t)l * B * CI_O * 2 : nullptr;
        if (int rc = conv_in_check(x[l], w_packed[l], bias[l], o, s, B, Cin[l], HW[l], out_batch_stride)) return rc;
        int nt;
        lv.cfg[l] = conv_in_config(B, Cin[l], HW[l], nt);
        max_nt = max(max_nt, nt);
        lv.x[l] = x[l]; lv.w[l] = w_packed[l]; lv.bias[l] = bias[l]; lv.out[l] = o; lv.stats[l] = s;
        lv.Cin[l] = Cin[l]; lv.HW[l] = HW[l];
        lv.tiles[l] = cdiv(HW[l], 16 * nt);
        lv.first[l] = wg;
        wg += lv.tiles[l] * B;
        tok += HW[l];
    }
    for (int l = n_levels; l <= CI_MAXL; ++l) lv.first[l] = wg;
    for (int l = n_levels; l < CI_MAXL; ++l) {
        lv.x[l] = nullptr; lv.w[l] = nullptr; lv.bias[l] = nullptr; lv.out[l] = nullptr; lv.stats[l] = nullptr;
        lv.Cin[l] = lv.HW[l] = lv.tiles[l] = lv.cfg[l] = 0;
    }
    MSM_REQUIRE(out_batch_stride >= tok * CI_O, "msm_conv1x1_in_multi_f32: output batch stride smaller than the %lld tokens of an image",
                (long long)tok);
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * CI_O * (size_t)B * n_levels, st));
    const size_t lds = conv_in_lds(max_nt);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_multi_kernel, lds));
    hipLaunchKernelGGL(conv_in_multi_kernel, dim3(wg), dim3(CI_W * 64), lds, st, lv, out_batch_stride);
    MSM_CHECK_LAUNCH("msm_conv1x1_in_multi_f32");
    return MSM_OK;
}

// ---- bf16 matrix pipe (hi + lo operands): see conv_in_lp_stream -------------------------------------------------------------------
static int conv_in_lp_config(int B, int Cin, int HW, int& nt) {
    // wide tiles read the packed weight (as many bytes as x at 64 pixels, twice / four times as many at 32 / 16) once per 64 pixels
    const int64_t t64 = (int64_t)cdiv(HW, 64) * B;
    nt = t64 >= 128 ? 4 : (t64 >= 32 ? 2 : 1);       // (measured at B = 8: 67 -> 60 us against the fp32 kernel's 512 / 128 thresholds, inputs warm)
    if (nt == 4) return 0;
    if (nt == 2) return 1;
    return Cin % 512 == 0 ? 2 : 3;
}
static size_t conv_in_lp_lds(int nt) { return sizeof(float4) * CL_W * 4 * nt * 64; }
// workgroups per image when this launch (level) may hold `slots` workgroups at once: all of them start together and walk
// `per` tiles each (a second round of workgroups would cost a whole tile time for a few stragglers)
static int conv_in_lp_wgs(int tiles, int B, double slots) {
    const int per = max(1, (int)ceil(tiles * (double)B / max(1.0, slots)));
    return cdiv(tiles, per);
}

static int conv_in_lp_check(const char* who, const float* x, const void* w, const float* bias, const float* out, const double* stats, int B,
                            int Cin, int HW, int64_t out_batch_stride) {
    MSM_REQUIRE(x && w && out, "%s: null pointer", who);
    MSM_REQUIRE(HW % 4 == 0 && B > 0 && HW >= 4 && Cin >= 256 && Cin % 256 == 0, "%s: Cin=%d must be a multiple of 256, HW=%d of 4", who, Cin, HW);
    MSM_REQUIRE(out_batch_stride >= (int64_t)HW * CI_O && out_batch_stride % 4 == 0, "%s: bad output batch stride", who);
    MSM_REQUIRE((int64_t)Cin * HW < ((int64_t)1 << 29), "%s: one image of x must be < 2 GiB (32-bit buffer offsets)", who);
    MSM_REQUIRE(((((uintptr_t)w) | ((uintptr_t)out) | ((uintptr_t)bias) | ((uintptr_t)x)) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                "%s: misaligned pointer", who);
    return MSM_OK;
}

extern "C" int msm_conv1x1_in_lp(const float* x, const void* w_packed, const float* bias, float* out, int64_t out_batch_stride,
                                 double* stats, int stats_cleared, int B, int Cin, int HW, void* stream) {
    if (int rc = conv_in_lp_check("msm_conv1x1_in_lp", x, w_packed, bias, out, stats, B, Cin, HW, out_batch_stride)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * CI_O * (size_t)B, st));
    const uint16_t* w = (const uint16_t*)w_packed;
    if (Cin <= 512 && (int64_t)B * HW >= 32 * 1024) {
        // shallow K over many pixels (the FPN lateral): 32-pixel tiles, a wave walks its own tiles over the full K (no reduction);
        // two workgroups of four waves per CU
        const size_t st_lds = sizeof(float) * CL_W * CI_O * 2;
        constexpr int SNT = 2;
        const int wgs = conv_in_lp_wgs(cdiv(cdiv(HW, 16 * SNT), CL_W), B, 512.0);
        if (Cin == 256) {
            // the weight (64 KiB) lives in LDS: two workgroups per CU (measured at B = 8, 120x160, inputs cold: 48.9 us; ring of 4 groups
            // 51.7, 64-pixel wave tiles 55.0, weight from L2 62.5; the fp32 MFMA kernel 66-69)
            const size_t lds = st_lds + (size_t)Cin * 256;
            MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_lp_kernel<SNT, 2, false, true>, lds));
            hipLaunchKernelGGL((conv_in_lp_kernel<SNT, 2, false, true>), dim3(wgs, B), dim3(CL_W * 64), lds, st, x, w, bias, out, out_batch_stride, stats, Cin, HW);
        } else {
            hipLaunchKernelGGL((conv_in_lp_kernel<SNT, 2, false>), dim3(wgs, B), dim3(CL_W * 64), st_lds, st, x, w, bias, out, out_batch_stride, stats, Cin, HW);
        }
        MSM_CHECK_LAUNCH("msm_conv1x1_in_lp");
        return MSM_OK;
    }
    int nt;
    const int cfg = conv_in_lp_config(B, Cin, HW, nt);
    dim3 grid(conv_in_lp_wgs(cdiv(HW, 16 * nt), B, 512.0), B), block(CL_W * 64);
    const size_t lds = conv_in_lp_lds(nt);
#define CL_LAUNCH(NT_, D_)                                                                                              \
    {                                                                                                                   \
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_lp_kernel<NT_, D_, true>, lds));              \
        hipLaunchKernelGGL((conv_in_lp_kernel<NT_, D_, true>), grid, block, lds, st, x, w, bias, out, out_batch_stride, stats, Cin, HW); \
    }
    switch (cfg) {
        case 0: CL_LAUNCH(4, 2) break;
        case 1: CL_LAUNCH(2, 2) break;
        case 2: CL_LAUNCH(1, 4) break;
        default: CL_LAUNCH(1, 2) break;
    }
#undef CL_LAUNCH
    MSM_CHECK_LAUNCH("msm_conv1x1_in_lp");
    return MSM_OK;
}

extern "C" int msm_conv1x1_in_multi_lp(int n_levels, const float* const* x, const void* const* w_packed, const float* const* bias,
                                       const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                                       int stats_cleared, int B, void* stream) {
    MSM_REQUIRE(n_levels >= 1 && n_levels <= CI_MAXL && x && w_packed && bias && Cin && HW && out,
                "msm_conv1x1_in_multi_lp: bad arguments (1..%d levels)", CI_MAXL);
    hipStream_t st = (hipStream_t)stream;
    ConvInLpLevels lv;
    lv.n = n_levels;
    int64_t tok = 0;
    int wg = 0, max_nt = 1;
    double bytes = 0.0;
    for (int l = 0; l < n_levels; ++l) bytes += (double)Cin[l] * HW[l];
    for (int l = 0; l < n_levels; ++l) {
        float* o = out + tok * CI_O;
        double* s = stats ? stats + (size_t)l * B * CI_O * 2 : nullptr;
        if (int rc = conv_in_lp_check("msm_conv1x1_in_multi_lp", x[l], w_packed[l], bias[l], o, s, B, Cin[l], HW[l], out_batch_stride)) return rc;
        int nt;
        lv.cfg[l] = conv_in_lp_config(B, Cin[l], HW[l], nt);
        max_nt = max(max_nt, nt);
        lv.x[l] = x[l]; lv.w[l] = (const uint16_t*)w_packed[l]; lv.bias[l] = bias[l]; lv.out[l] = o; lv.stats[l] = s;
        lv.Cin[l] = Cin[l]; lv.HW[l] = HW[l];
        // the chip's 512 workgroup slots are shared out in proportion to the bytes a level streams
        lv.wgs[l] = conv_in_lp_wgs(cdiv(HW[l], 16 * nt), B, 512.0 * (double)Cin[l] * HW[l] / bytes);
        lv.first[l] = wg;
        wg += lv.wgs[l] * B;
        tok += HW[l];
    }
    for (int l = n_levels; l <= CI_MAXL; ++l) lv.first[l] = wg;
    for (int l = n_levels; l < CI_MAXL; ++l) {
        lv.x[l] = nullptr; lv.w[l] = nullptr; lv.bias[l] = nullptr; lv.out[l] = nullptr; lv.stats[l] = nullptr;
        lv.Cin[l] = lv.HW[l] = lv.cfg[l] = 0;
        lv.wgs[l] = 1;
    }
    MSM_REQUIRE(out_batch_stride >= tok * CI_O, "msm_conv1x1_in_multi_lp: output batch stride smaller than the %lld tokens of an image",
                (long long)tok);
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * CI_O * (size_t)B * n_levels, st));
    const size_t lds = conv_in_lp_lds(max_nt);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_lp_multi_kernel, lds));
    hipLaunchKernelGGL(conv_in_lp_multi_kernel, dim3(wg), dim3(CL_W * 64), lds, st, lv, out_batch_stride);
    MSM_CHECK_LAUNCH("msm_conv1x1_in_multi_lp");
    return MSM_OK;
}

extern "C" int msm_conv1x1_in_multi_wide(int n_levels, const float* const* x, const void* const* w_packed, const float* const* bias,
                                         const int32_t* Cin, const int32_t* HW, float* out, int64_t out_batch_stride, double* stats,
                                         int stats_cleared, int B, void* stream) {
    MSM_REQUIRE(n_levels >= 1 && n_levels <= CI_MAXL && x && w_packed && bias && Cin && HW && out,
                "msm_conv1x1_in_multi_wide: bad arguments (1..%d levels)", CI_MAXL);
    hipStream_t st = (hipStream_t)stream;
    ConvInWideLevels lv;
    lv.n = n_levels;
    int cmin = Cin[0];
    for (int l = 1; l < n_levels; ++l) cmin = min(cmin, Cin[l]);
    // K slices per level: every wave walks about the shallowest level's K (a power of two, at most CW_KWMAX, whole rounds of CW_D groups per
    // slice) -- then, while the launch stays under 160 workgroups (one per CU: 204 registers), the level with the most pixels is split further
    // (measured at B = 8, 640x480: slices 4 / 2 / 1 = 144 workgroups 45 us, 4 / 4 / 2 = 256 workgroups 49 us; res3 on its own 35 -> 26 us)
    int kws[CI_MAXL];
    auto can_double = [&](int l) { return kws[l] < CW_KWMAX && Cin[l] % (2 * kws[l] * 64) == 0; };      // (a slice: at least two 32-deep groups)
    auto wgs_of = [&](int l, int kw) { return cdiv(HW[l], 64 * (CW_W / kw)) * B; };
    int total = 0;
    for (int l = 0; l < n_levels; ++l) {
        kws[l] = 1;
        while (can_double(l) && Cin[l] / (2 * kws[l]) >= cmin) kws[l] *= 2;
        if (const int o_ = opt(MSM_OPT_CONVIN_NT); (o_ == 1 || o_ == 2 || o_ == 4) && Cin[l] % (o_ * 64) == 0) kws[l] = o_;      // (tuning: the same K split everywhere)
        if (const int o_ = opt(MSM_OPT_CONVIN_NT); o_ >= 100) {                   // (tuning: one decimal digit per level, e.g. 421)
            const int dgt = l == 0 ? o_ / 100 : (l == 1 ? o_ / 10 % 10 : o_ % 10);
            if ((dgt == 1 || dgt == 2 || dgt == 4) && Cin[l] % (dgt * 64) == 0) kws[l] = dgt;
        }
        total += wgs_of(l, kws[l]);
    }
    for (bool again = opt(MSM_OPT_CONVIN_NT) == MSM_OPT_AUTO; again;) {
        again = false;
        int best = -1;
        for (int l = 0; l < n_levels; ++l)
            if (can_double(l) && total - wgs_of(l, kws[l]) + wgs_of(l, 2 * kws[l]) <= 160 && (best < 0 || HW[l] > HW[best])) best = l;
        if (best >= 0) {
            total += wgs_of(best, 2 * kws[best]) - wgs_of(best, kws[best]);
            kws[best] *= 2;
            again = true;
        }
    }
    int64_t tok = 0;
    int wg = 0;
    for (int l = 0; l < n_levels; ++l) {
        float* o = out + tok * CI_O;
        double* s = stats ? stats + (size_t)l * B * CI_O * 2 : nullptr;
        if (int rc = conv_in_lp_check("msm_conv1x1_in_multi_wide", x[l], w_packed[l], bias[l], o, s, B, Cin[l], HW[l], out_batch_stride)) return rc;
        const int kw = kws[l];
        lv.x[l] = x[l]; lv.w[l] = (const uint16_t*)w_packed[l]; lv.bias[l] = bias[l]; lv.out[l] = o; lv.stats[l] = s;
        lv.Cin[l] = Cin[l]; lv.HW[l] = HW[l]; lv.kw[l] = kw;
        lv.chunks[l] = cdiv(HW[l], 64 * (CW_W / kw));
        lv.first[l] = wg;
        wg += lv.chunks[l] * B;
        tok += HW[l];
    }
    for (int l = n_levels; l <= CI_MAXL; ++l) lv.first[l] = wg;
    for (int l = n_levels; l < CI_MAXL; ++l) {
        lv.x[l] = nullptr; lv.w[l] = nullptr; lv.bias[l] = nullptr; lv.out[l] = nullptr; lv.stats[l] = nullptr;
        lv.Cin[l] = lv.HW[l] = 0;
        lv.kw[l] = lv.chunks[l] = 1;
    }
    MSM_REQUIRE(out_batch_stride >= tok * CI_O, "msm_conv1x1_in_multi_wide: output batch stride smaller than the %lld tokens of an image",
                (long long)tok);
    if (stats && !stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * CI_O * (size_t)B * n_levels, st));
    const size_t lds = (size_t)CW_NBUF * CW_STAGE;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)conv_in_wide_kernel, lds));
    hipLaunchKernelGGL(conv_in_wide_kernel, dim3(wg), dim3(CW_W * 64), lds, st, lv, out_batch_stride);
    MSM_CHECK_LAUNCH("msm_conv1x1_in_multi_wide");
    return MSM_OK;
}
