// Generic fp32 MFMA GEMM with fused prologue/epilogue (see include/msm_hip.h: msm_gemm_f32).
//
// Replaces the torch ops around the hot kernels of the reference: F.linear (attention_util.py:134-140,
// 425; meanshiftformer_transformer_decoder.py:301,336-340,663-664; ops/modules/ms_deform_attn.py:95-104,
// 123), 1x1 Conv2d (meanshiftformer_transformer_decoder.py:499,575; msdeformattn.py:212-220,245-252,
// 264-266) and the 3x3 output conv (msdeformattn.py:268-277, as an implicit GEMM over NHWC tokens).
//
// Structure: 256 threads = 4 waves (2x2), each wave owns a (16*MI)x(16*NI) block of C built from
// v_mfma_f32_16x16x4_f32 tiles (exact fp32).  A/W tiles of BK=32 are staged through LDS with
// register prefetch of the next tile.  LDS row strides are chosen so that the fragment reads
// (lane = (row l&15, k-slot l>>4)) are bank-conflict free for ds_read_b32:
//   K-contiguous tiles  [rows][34]   : bank = (2*row + k) mod 32, distinct over the two 32-lane halves
//   M-contiguous tiles  [32][BM+16]  : bank = (16*k + row) mod 32, likewise.
#include <stdlib.h>

#include "common.h"

namespace msm {

struct GemmArgs {
    const float* A;
    const float* A2;
    const float* W;
    const float* bias;
    float* C;
    int M, N, K, batch;
    int64_t a_sm, a_sk, a_sb, a2_sb, w_sb, c_sm, c_sn, c_sb, c_ss;
    int conv_h, conv_w, conv_c;
    int bias_mode, act, split_k, k_per_split;
    int vec_a, vec_w, vec_c;
};

constexpr int BK = 32;
constexpr int SK = BK + 2;  // row stride of K-contiguous LDS tiles

// AMODE 0: A rows K-contiguous; 1: A is M-contiguous ([K][M]); 2: implicit 3x3 conv over NHWC tokens
// SWAP: the MFMA operands are exchanged so that the accumulator holds C^T tiles (lane = row m, four
// consecutive columns n in its registers): row-major outputs are then written with 16-byte stores.
// Without SWAP a lane holds four consecutive rows m of one column: 16-byte stores for m-contiguous
// (NCHW) outputs.
// VEC: every tile load is one unconditional 16-byte load from a clamped address followed by a select
// (no branches: hipcc otherwise puts each guarded load in its own basic block and the loads of a tile
// are issued one latency after the other).  !VEC is the element-wise, fully guarded fallback for
// unaligned / odd shapes.
// KT: LDS tile depth in units of 32 (BKX = 32*KT).  KT = 4 is used for the small, latency-bound GEMMs of
// the decoder's per-query chain: K = 256 becomes two load phases with 16+ loads in flight per thread instead
// of eight dependent ones.
template <int MI, int NI, int AMODE, bool SWAP, bool VEC, bool HAS_A2, int KT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int BM = 32 * MI, BN = 32 * NI;
    constexpr int BKX = BK * KT, SKX = BKX + 2, F4R = 8 * KT;   // tile depth, K-contiguous row stride, float4 per row
    constexpr int LA = MI * KT, LW = NI * KT;                  // float4 loads per thread per tile
    constexpr int SM = BM + 16;  // row stride of the M-contiguous A tile
    constexpr int A_ELEMS = (AMODE == 1) ? BKX * SM : BM * SKX;
    __shared__ __attribute__((aligned(16))) float As[A_ELEMS];
    __shared__ __attribute__((aligned(16))) float Bs[BN * SKX];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    const int wm = (wave >> 1) * (16 * MI), wn = (wave & 1) * (16 * NI);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int bz = blockIdx.z;
    const int b = bz / p.split_k, ks = bz - b * p.split_k;
    const int kbeg = ks * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const float* __restrict__ Ab = p.A + (int64_t)b * p.a_sb;
    const float* __restrict__ A2b = HAS_A2 ? p.A2 + (int64_t)b * p.a2_sb : nullptr;
    const float* __restrict__ Wb = p.W + (int64_t)b * p.w_sb;

    // Raw tile registers.  Loads are unconditional 16-byte loads from clamped addresses; the
    // out-of-range select and the A2 add are applied when the tile is written to LDS (i.e. AFTER the
    // MFMAs of the previous tile), so that the s_waitcnt for a prefetched tile sits behind the compute.
    float4 ra[LA], ra2[HAS_A2 ? LA : 1], rb[LW];
    // component-wise select (a float4 ?: is lowered through scratch memory by hipcc)
    auto sel4 = [](bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
    auto ld4 = [&](const float* ptr) { return *reinterpret_cast<const float4*>(ptr); };
    auto ld4_guarded = [&](const float* base, const float* base2, int64_t off, int count) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[e] = 0.f;
            if (e < count) {
                t[e] = base[off + e];
                if (base2) t[e] += base2[off + e];
            }
        }
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    // validity of the float4 that thread `tid` loads for slot i of the tile starting at k0
    auto a_ok = [&](int i, int k0) {
        const int f = tid + 256 * i;
        if constexpr (AMODE == 1) {
            return (k0 + f / (BM / 4) < kend) && (m0 + (f % (BM / 4)) * 4 < p.M);
        } else if constexpr (AMODE == 0) {
            return (m0 + (f / F4R) < p.M) && (k0 + (f % F4R) * 4 < kend);
        } else {
            const int k = k0 + (f % F4R) * 4, m = m0 + (f / F4R);
            const int tap = min(k, kend - 4) / p.conv_c;
            const int mc = min(m, p.M - 1);
            const int y = mc / p.conv_w, x = mc - y * p.conv_w;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            return m < p.M && k < kend && yy >= 0 && yy < p.conv_h && xx >= 0 && xx < p.conv_w;
        }
    };
    auto w_ok = [&](int i, int k0) {
        const int f = tid + 256 * i;
        return (n0 + (f / F4R) < p.N) && (k0 + (f % F4R) * 4 < kend);
    };

    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int f = tid + 256 * i;
            float4 v;
            if constexpr (AMODE == 0) {
                const int row = f / F4R, k = k0 + (f % F4R) * 4;
                const int m = m0 + row;
                if constexpr (VEC) {
                    const int64_t off = (int64_t)min(m, p.M - 1) * p.a_sm + min(k, kend - 4);
                    v = ld4(Ab + off);
                    if constexpr (HAS_A2) ra2[i] = ld4(A2b + off);
                } else {
                    v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < p.M && k < kend) v = ld4_guarded(Ab, A2b, (int64_t)m * p.a_sm + k, kend - k);
                }
            } else if constexpr (AMODE == 1) {
                const int kk = f / (BM / 4), m = m0 + (f % (BM / 4)) * 4;
                const int k = k0 + kk;
                if constexpr (VEC) {
                    v = ld4(Ab + (int64_t)min(k, kend - 1) * p.a_sk + min(m, p.M - 4));
                } else {
                    v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < kend && m < p.M) v = ld4_guarded(Ab, nullptr, (int64_t)k * p.a_sk + m, p.M - m);
                }
            } else {
                // implicit im2col: k = tap*C + c, pixel (y,x) = (m / W, m % W), zero padding 1
                const int k = min(k0 + (f % F4R) * 4, kend - 4);
                const int m = min(m0 + (f / F4R), p.M - 1);
                const int tap = k / p.conv_c, c = k - tap * p.conv_c;
                const int y = m / p.conv_w, x = m - y * p.conv_w;
                const int yc = min(max(y + tap / 3 - 1, 0), p.conv_h - 1), xc = min(max(x + tap % 3 - 1, 0), p.conv_w - 1);
                v = ld4(Ab + ((int64_t)yc * p.conv_w + xc) * p.conv_c + c);
            }
            ra[i] = v;
        }
    };
    auto load_w = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LW; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4R, k = k0 + (f % F4R) * 4;
            const int n = n0 + row;
            if constexpr (VEC) {
                rb[i] = ld4(Wb + (int64_t)min(n, p.N - 1) * p.K + min(k, kend - 4));
            } else {
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N && k < kend) rb[i] = ld4_guarded(Wb, nullptr, (int64_t)n * p.K + k, kend - k);
            }
        }
    };
    auto store_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int f = tid + 256 * i;
            float4 v = ra[i];
            if constexpr (VEC) {
                if constexpr (HAS_A2) {
                    v.x += ra2[i].x; v.y += ra2[i].y; v.z += ra2[i].z; v.w += ra2[i].w;
                }
                v = sel4(a_ok(i, k0), v);
            }
            if constexpr (AMODE == 1) {
                const int kk = f / (BM / 4), mm = (f % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(&As[kk * SM + mm]) = v;
            } else {
                const int row = f / F4R, c4 = (f % F4R) * 4;
                float2* d = reinterpret_cast<float2*>(&As[row * SKX + c4]);
                d[0] = make_float2(v.x, v.y);
                d[1] = make_float2(v.z, v.w);
            }
        }
#pragma unroll
        for (int i = 0; i < LW; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4R, c4 = (f % F4R) * 4;
            float4 v = rb[i];
            if constexpr (VEC) v = sel4(w_ok(i, k0), v);
            float2* d = reinterpret_cast<float2*>(&Bs[row * SKX + c4]);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_a(kbeg);
    load_w(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BKX) {
        store_tiles(k0);
        __syncthreads();
        if (k0 + BKX < kend) {
            load_a(k0 + BKX);
            load_w(k0 + BKX);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above, the selects/LDS writes below the MFMAs
#pragma unroll
        for (int kk = 0; kk < BKX; kk += 4) {
            float a[MI], bb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (AMODE == 1)
                    a[i] = As[(kk + lq) * SM + wm + i * 16 + lj];
                else
                    a[i] = As[(wm + i * 16 + lj) * SKX + kk + lq];
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) bb[j] = Bs[(wn + j * 16 + lj) * SKX + kk + lq];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = SWAP ? mfma16(bb[j], a[i], acc[i][j]) : mfma16(a[i], bb[j], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    // epilogue.  !SWAP: lane holds rows m = .. + lq*4 + r, column n = .. + lj.
    //            SWAP: lane holds row m = .. + lj, columns n = .. + lq*4 + r.
    float* __restrict__ Cb = p.C + (int64_t)b * p.c_sb + (int64_t)ks * p.c_ss;
    const bool raw = p.split_k > 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int mb = m0 + wm + i * 16 + (SWAP ? lj : lq * 4);
            const int nb = n0 + wn + j * 16 + (SWAP ? lq * 4 : lj);
            if (mb >= p.M || nb >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = SWAP ? mb : mb + r, n = SWAP ? nb + r : nb;
                float t = acc[i][j][r];
                if (!raw && m < p.M && n < p.N) {
                    if (p.bias_mode == 1) t += p.bias[n];
                    else if (p.bias_mode == 2) t += p.bias[m];
                    else if (p.bias_mode == 3) t += p.bias[(int64_t)m * p.N + n];
                    if (p.act == 1) t = fmaxf(t, 0.f);
                }
                v[r] = t;
            }
            float* dst = Cb + (int64_t)mb * p.c_sm + (int64_t)nb * p.c_sn;
            const bool full = SWAP ? (nb + 3 < p.N) : (mb + 3 < p.M);
            if (p.vec_c && full) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const int64_t step = SWAP ? p.c_sn : p.c_sm;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (SWAP ? (nb + r < p.N) : (mb + r < p.M)) dst[r * step] = v[r];
            }
        }
    }
}

template <int AMODE, bool SWAP, bool HAS_A2>
static int launch_gemm_o(const GemmArgs& p, hipStream_t st) {
    dim3 block(256);
    if (!(p.vec_a && p.vec_w)) {   // unaligned / odd shapes: guarded element-wise loads, smallest tile
        dim3 grid(cdiv(p.N, 32), cdiv(p.M, 32), p.batch * p.split_k);
        hipLaunchKernelGGL((gemm_kernel<1, 1, AMODE, SWAP, false, HAS_A2, 1>), grid, block, 0, st, p);
        MSM_CHECK_LAUNCH("msm_gemm_f32");
        return MSM_OK;
    }
    // tile choice (measured on MI355X, tools/microbench.py): 64x64 workgroup tiles beat 128x128 and 64x128
    // on every shape of this model (more resident workgroups hide the LDS/barrier phases), so the order
    // of preference is 64x64, 32x64, 32x32: the first that wastes < 20 % of its MFMA work on M/N padding
    // and still yields >= 2 workgroups per CU; failing that, the low-waste tile with the most workgroups.
    // 128x128 / 64x128 stay reachable through MSM_GEMM_TILE=0/1 for tuning.
    const int cfgs[5][2] = {{4, 4}, {2, 4}, {2, 2}, {1, 2}, {1, 1}};
    int pick = -1, best_blocks = -1, fallback = 4;
    for (int c = 2; c < 5; ++c) {
        const int64_t bm = 32 * cfgs[c][0], bn = 32 * cfgs[c][1];
        const int64_t gm = cdiv(p.M, bm), gn = cdiv(p.N, bn);
        const double waste = 1.0 - (double)p.M * p.N / ((double)gm * bm * gn * bn);
        const int64_t blocks = gm * gn * p.batch * p.split_k;
        if (waste > 0.2 && c < 4) continue;
        if (blocks >= 512) { pick = c; break; }
        if (blocks > best_blocks) { best_blocks = (int)blocks; fallback = c; }
    }
    if (pick < 0) pick = fallback;
    if (const int o = opt(MSM_OPT_GEMM_TILE); o >= 0 && o <= 4) pick = o;
    const int mi = cfgs[pick][0], ni = cfgs[pick][1];
    dim3 grid(cdiv(p.N, 32 * ni), cdiv(p.M, 32 * mi), p.batch * p.split_k);
    // deep LDS tiles for the small latency-bound shapes (row-major activations only)
    const bool deep = AMODE == 0 && pick >= 3 && p.k_per_split >= 128 && opt(MSM_OPT_GEMM_SHALLOW) != 1;
    switch (pick) {
        case 0: hipLaunchKernelGGL((gemm_kernel<4, 4, AMODE, SWAP, true, HAS_A2, 1>), grid, block, 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<2, 4, AMODE, SWAP, true, HAS_A2, 1>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<2, 2, AMODE, SWAP, true, HAS_A2, 1>), grid, block, 0, st, p); break;
        case 3:
            if constexpr (AMODE == 0) {
                if (deep) { hipLaunchKernelGGL((gemm_kernel<1, 2, 0, SWAP, true, HAS_A2, 4>), grid, block, 0, st, p); break; }
            }
            hipLaunchKernelGGL((gemm_kernel<1, 2, AMODE, SWAP, true, HAS_A2, 1>), grid, block, 0, st, p);
            break;
        default:
            if constexpr (AMODE == 0) {
                if (deep) { hipLaunchKernelGGL((gemm_kernel<1, 1, 0, SWAP, true, HAS_A2, 4>), grid, block, 0, st, p); break; }
            }
            hipLaunchKernelGGL((gemm_kernel<1, 1, AMODE, SWAP, true, HAS_A2, 1>), grid, block, 0, st, p);
            break;
    }
    MSM_CHECK_LAUNCH("msm_gemm_f32");
    return MSM_OK;
}

template <int AMODE, bool HAS_A2>
static int launch_gemm(GemmArgs& p, hipStream_t st) {
    const bool c16 = (((uintptr_t)p.C) & 15) == 0 && p.c_sb % 4 == 0 && p.c_ss % 4 == 0;
    if (p.c_sn == 1) {                     // row-major output: 4 consecutive n per lane
        p.vec_c = c16 && p.c_sm % 4 == 0;
        return launch_gemm_o<AMODE, true, HAS_A2>(p, st);
    }
    p.vec_c = c16 && p.c_sm == 1 && p.c_sn % 4 == 0;   // m-contiguous (NCHW) output
    return launch_gemm_o<AMODE, false, HAS_A2>(p, st);
}

}  // namespace msm

using namespace msm;

extern "C" int msm_gemm_f32(const float* A, const float* A2, const float* W, const float* bias, float* C,
                            int M, int N, int K, int batch,
                            int64_t a_sm, int64_t a_sk, int64_t a_sb, int64_t a2_sb, int64_t w_sb,
                            int64_t c_sm, int64_t c_sn, int64_t c_sb, int64_t c_ss,
                            int a_mode, int conv_h, int conv_w, int conv_c,
                            int bias_mode, int act, int split_k, void* stream) {
    MSM_REQUIRE(A && W && C, "msm_gemm_f32: null pointer");
    MSM_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "msm_gemm_f32: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
    MSM_REQUIRE(split_k >= 1, "msm_gemm_f32: split_k must be >= 1");
    MSM_REQUIRE(bias_mode == 0 || bias != nullptr, "msm_gemm_f32: bias_mode set without bias");
    GemmArgs p;
    p.A = A; p.A2 = A2; p.W = W; p.bias = bias; p.C = C;
    p.M = M; p.N = N; p.K = K; p.batch = batch;
    p.a_sm = a_sm; p.a_sk = a_sk; p.a_sb = a_sb; p.a2_sb = a2_sb; p.w_sb = w_sb;
    p.c_sm = c_sm; p.c_sn = c_sn; p.c_sb = c_sb; p.c_ss = c_ss;
    p.conv_h = conv_h; p.conv_w = conv_w; p.conv_c = conv_c;
    p.bias_mode = bias_mode; p.act = act; p.split_k = split_k;
    int kps = cdiv(K, split_k);
    kps = cdiv(kps, BK) * BK;  // whole 32-deep sub-tiles per split (deep tiles zero-fill their tail)
    p.k_per_split = kps;
    MSM_REQUIRE((int64_t)kps * (split_k - 1) < K, "msm_gemm_f32: split_k=%d too large for K=%d", split_k, K);
    const bool a16 = (((uintptr_t)A) & 15) == 0 && (!A2 || (((uintptr_t)A2) & 15) == 0);
    p.vec_w = (K % 4 == 0) && ((((uintptr_t)W) & 15) == 0) && (w_sb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (a_mode == 2) {
        MSM_REQUIRE(conv_c % 4 == 0 && K == 9 * conv_c && M == conv_h * conv_w && a16 && a_sb % 4 == 0 && !A2,
                    "msm_gemm_f32: bad implicit-conv arguments");
        p.vec_a = 1;
        MSM_REQUIRE(p.vec_w, "msm_gemm_f32: implicit-conv weights must be 16-byte aligned");
        return launch_gemm<2, false>(p, st);
    }
    MSM_REQUIRE(a_mode == 0, "msm_gemm_f32: a_mode must be 0 or 2");
    if (a_sk == 1) {
        p.vec_a = a16 && (K % 4 == 0) && (a_sm % 4 == 0) && (a_sb % 4 == 0) && (a2_sb % 4 == 0);
        return A2 ? launch_gemm<0, true>(p, st) : launch_gemm<0, false>(p, st);
    }
    MSM_REQUIRE(a_sm == 1, "msm_gemm_f32: one of a_sm/a_sk must be 1");
    MSM_REQUIRE(!A2, "msm_gemm_f32: A2 is only supported for K-contiguous A");
    p.vec_a = a16 && (M % 4 == 0) && (a_sk % 4 == 0) && (a_sb % 4 == 0);
    return launch_gemm<1, false>(p, st);
}
