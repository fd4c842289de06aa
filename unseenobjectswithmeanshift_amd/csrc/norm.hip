// Row / group normalisation kernels (see include/msm_hip.h).
//
//   msm_layernorm_f32      : residual + split-K parts + bias -> LayerNorm [-> L2 normalise] [-> LayerNorm]
//                            (meanshiftformer_transformer_decoder.py:255-257,178-179,300-304,637-638,661;
//                             msdeformattn.py:116-118,124-126)
//   msm_groupnorm_stats/apply : GroupNorm(32, C) over NHWC token maps with the FPN top-down
//                            bilinear add and ReLU fused in (msdeformattn.py:212-220,262-277,343-351)
//   msm_pos_embed_sine, msm_transpose_f32 : position_encoding.py:29-52 and layout glue.
//
// All of these are HBM-bound streaming kernels: one wave per row (LayerNorm) or one lane per
// channel (GroupNorm) so that every global access is a full coalesced 256 B segment.
#include "bf16.h"
#include "common.h"

namespace msm {

template <int VPT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ parts,
                                                        int n_parts, int64_t part_stride,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        int l2norm, const float* __restrict__ g2,
                                                        const float* __restrict__ b2, float* __restrict__ y,
                                                        float* __restrict__ y2, int rows, float eps) {
    constexpr int E = VPT * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = (int64_t)row * E;
    float v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int e = lane + 64 * i;
        float t = x ? x[base + e] : 0.f;
        for (int s = 0; s < n_parts; ++s) t += parts[(int64_t)s * part_stride + base + e];
        if (bias) t += bias[e];
        v[i] = t;
    }
    auto ln = [&](const float* __restrict__ g, const float* __restrict__ b) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) s += v[i];
        const float mean = wave_sum(s) * (1.0f / E);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const float d = v[i] - mean;
            q += d * d;
        }
        const float var = wave_sum(q) * (1.0f / E);
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int e = lane + 64 * i;
            v[i] = (v[i] - mean) * rstd * g[e] + b[e];
        }
    };
    ln(g1, b1);
    if (l2norm) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) q += v[i] * v[i];
        const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
#pragma unroll
        for (int i = 0; i < VPT; ++i) v[i] = v[i] / nrm;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) y[base + lane + 64 * i] = v[i];
    if (g2) {
        ln(g2, b2);
#pragma unroll
        for (int i = 0; i < VPT; ++i) y2[base + lane + 64 * i] = v[i];
    }
}

// ---- GroupNorm --------------------------------------------------------------------------------
// stats[b][c] = (sum, sumsq) in double.  Thread = (4 consecutive channels, pixel stream): 16-byte
// loads, fp32 partial sums over a short run of pixels, then double for the cross-thread reduction and
// one double atomic per (block, channel).  Double accumulation keeps E[x^2]-E[x]^2 well conditioned and
// makes the result insensitive to the (unordered) atomic arrival order at fp32 precision.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                       int HW, int C, int pix_per_block) {
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    const int c4n = C >> 2;                    // float4 per pixel
    const int streams = 256 / c4n;
    const int c4 = threadIdx.x % c4n, st = threadIdx.x / c4n;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (st < streams) {
        const float* xb = x + ((int64_t)b * HW) * C + c4 * 4;
        for (int p = p0 + st; p < p1; p += streams) {
            const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * C);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
        }
    }
    __shared__ double red[2][4][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][e][threadIdx.x] = (double)s[e];
        red[1][e][threadIdx.x] = (double)q[e];
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, cc = c >> 2, e = c & 3;
        double sum = 0.0, sq = 0.0;
        for (int k = 0; k < streams; ++k) {
            sum += red[0][e][k * c4n + cc];
            sq += red[1][e][k * c4n + cc];
        }
        double* d = stats + ((int64_t)b * C + c) * 2;
        atomicAdd(d, sum);
        atomicAdd(d + 1, sq);
    }
}

__device__ __forceinline__ void bilin_src(int dst, int in, int out, int& i0, int& i1, float& l1) {
    // PyTorch area_pixel_compute_source_index, align_corners=False
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

// y = (x - mean_g) * rstd_g * gamma + beta [+ bilinear(up)] [relu]; per-channel scale/shift are derived once
// per block into LDS, the body is 16-byte loads/stores (thread = 4 channels of one pixel).
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ up, int uh, int uw, int64_t up_sb,
                                                       float* __restrict__ y, int H, int W, int C, int groups,
                                                       float eps, int relu, uint16_t* __restrict__ planes, int64_t plane_stride, int h16) {
    __shared__ float sc[256], sh[256], mn[256];
    const int b = blockIdx.y;
    const int HW = H * W;
    const int cpg = C / groups;
    if (threadIdx.x < C) {
        const int c = threadIdx.x, g0 = (c / cpg) * cpg;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < cpg; ++k) {
            s += stats[((int64_t)b * C + g0 + k) * 2];
            q += stats[((int64_t)b * C + g0 + k) * 2 + 1];
        }
        const double cnt = (double)cpg * (double)HW;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float a = rstd * gamma[c];
        sc[c] = a;
        sh[c] = beta[c];
        mn[c] = (float)mean;
    }
    __syncthreads();
    const int c4n = C >> 2;
    const int64_t total4 = (int64_t)HW * c4n;
    const float* xb = x + (int64_t)b * HW * C;
    float* yb = y + (int64_t)b * HW * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const int p = (int)(idx / c4n);
        float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * C + c);
        v.x = (v.x - mn[c]) * sc[c] + sh[c];
        v.y = (v.y - mn[c + 1]) * sc[c + 1] + sh[c + 1];
        v.z = (v.z - mn[c + 2]) * sc[c + 2] + sh[c + 2];
        v.w = (v.w - mn[c + 3]) * sc[c + 3] + sh[c + 3];
        if (up) {
            const int yy = p / W, xx = p - yy * W;
            int y0, y1, x0, x1;
            float ly, lx;
            bilin_src(yy, uh, H, y0, y1, ly);
            bilin_src(xx, uw, W, x0, x1, lx);
            const float* ub = up + (int64_t)b * up_sb + c;
            const float4 v00 = *reinterpret_cast<const float4*>(ub + ((int64_t)y0 * uw + x0) * C);
            const float4 v01 = *reinterpret_cast<const float4*>(ub + ((int64_t)y0 * uw + x1) * C);
            const float4 v10 = *reinterpret_cast<const float4*>(ub + ((int64_t)y1 * uw + x0) * C);
            const float4 v11 = *reinterpret_cast<const float4*>(ub + ((int64_t)y1 * uw + x1) * C);
            const float hy = 1.f - ly, hx = 1.f - lx;
            v.x += hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            v.y += hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            v.z += hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            v.w += hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (planes && h16) {
            // the result as ONE plane of clamped IEEE halves: the operand bits msm_conv3x3_c64_f16 rounds to, written once by the producer
            *reinterpret_cast<u32x2b*>(planes + ((int64_t)b * HW + p) * C + c) = pack4h(v.x, v.y, v.z, v.w);
        } else if (planes) {
            // the result as three bf16 planes (v = h + m + l exactly): the operand of msm_conv3x3_c64_split, split once here
            // instead of nine times (once per tap) in the consumer
            const Split3 t3 = split3(v.x, v.y, v.z, v.w);
            uint16_t* pb = planes + ((int64_t)b * HW + p) * C + c;
            *reinterpret_cast<u32x2b*>(pb) = __builtin_bit_cast(u32x2b, t3.h);
            *reinterpret_cast<u32x2b*>(pb + plane_stride) = __builtin_bit_cast(u32x2b, t3.m);
            *reinterpret_cast<u32x2b*>(pb + 2 * plane_stride) = __builtin_bit_cast(u32x2b, t3.l);
        } else {
            *reinterpret_cast<float4*>(yb + (int64_t)p * C + c) = v;
        }
    }
}

// y[b][c][p] = act(GN(x))[b][p][c]: GroupNorm (+ReLU) of a token map written as NCHW planes -- the 64-channel activation the
// folded mask step contracts with directly (mask_features = Wm a + bm is never materialised, see msm_mask_logits_fwd).
// A block normalises a 64-token x C tile and transposes it through LDS; 16-byte loads and stores on both sides.
__global__ __launch_bounds__(256) void gn_apply_nchw_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, int HW, int C, int groups, float eps, int relu) {
    extern __shared__ float gl[];                // sc[C], sh[C], mn[C], tile[C][64 + 4]
    float *sc = gl, *sh = gl + C, *mn = gl + 2 * C, *tile = gl + 3 * C;
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g0 = (c / cpg) * cpg;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < cpg; ++k) {
            s += stats[((int64_t)b * C + g0 + k) * 2];
            q += stats[((int64_t)b * C + g0 + k) * 2 + 1];
        }
        const double cnt = (double)cpg * (double)HW;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sc[c] = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
        sh[c] = beta[c];
        mn[c] = (float)mean;
    }
    __syncthreads();
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < 64 * c4n; i += 256) {
        const int t = i / c4n, c = (i - t * c4n) * 4;
        const int p = min(p0 + t, HW - 1);
        float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)b * HW + p) * C + c);
        v.x = (v.x - mn[c]) * sc[c] + sh[c];
        v.y = (v.y - mn[c + 1]) * sc[c + 1] + sh[c + 1];
        v.z = (v.z - mn[c + 2]) * sc[c + 2] + sh[c + 2];
        v.w = (v.w - mn[c + 3]) * sc[c + 3] + sh[c + 3];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        tile[(c + 0) * 68 + t] = v.x;
        tile[(c + 1) * 68 + t] = v.y;
        tile[(c + 2) * 68 + t] = v.z;
        tile[(c + 3) * 68 + t] = v.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 16; i += 256) {
        const int c = i >> 4, t4 = (i & 15) * 4;
        if (p0 + t4 < HW)          // HW % 4 == 0: a float4 is inside or outside as a whole
            *reinterpret_cast<float4*>(y + ((int64_t)b * C + c) * HW + p0 + t4) = *reinterpret_cast<const float4*>(tile + c * 68 + t4);
    }
}

// ---- position encoding -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pos_embed_kernel(float* __restrict__ out, int H, int W, int npf, int64_t s_c,
                                                        int64_t s_p, const float* __restrict__ add_c,
                                                        float temperature, float scale) {
    const int64_t total = (int64_t)H * W * 2 * npf;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        // idx = p * (2*npf) + c  (channel fastest: coalesced for token-major outputs)
        const int c = (int)(idx % (2 * npf));
        const int p = (int)(idx / (2 * npf));
        const int yy = p / W, xx = p - yy * W;
        const bool is_y = c < npf;
        const int i = is_y ? c : c - npf;
        const float eps = 1e-6f;
        const float e = is_y ? ((float)(yy + 1) / ((float)H + eps) * scale) : ((float)(xx + 1) / ((float)W + eps) * scale);
        const float dim_t = powf(temperature, (float)(2 * (i / 2)) / (float)npf);
        const float a = e / dim_t;
        float v = (i & 1) ? cosf(a) : sinf(a);
        if (add_c) v += add_c[c];
        out[(int64_t)c * s_c + (int64_t)p * s_p] = v;
    }
}

// ---- transpose ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* ib = in + (int64_t)b * R * C;
    float* ob = out + (int64_t)b * R * C;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? ib[(int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) ob[(int64_t)c * R + r] = tile[tx][k];
    }
}

// ---- a 64-channel NCHW fp32 map as fp16 tokens [B][HW][64] in one pass (the feature form msm_hypersphere_attn_fused_kv_fwd streams) ----
// A block = 64 pixels: channel rows read as 256 contiguous bytes, transposed through LDS, a pixel's 64 halves written as 128 contiguous bytes.
__global__ __launch_bounds__(256) void nchw_to_tokens_f16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int HW) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const float* ib = in + (int64_t)b * 64 * HW;
    {
        const int p = tid & 63, cg = tid >> 6;
        const int pp = min(p0 + p, HW - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[cg * 16 + i][p] = ib[(int64_t)(cg * 16 + i) * HW + pp];
    }
    __syncthreads();
    const int pix = tid >> 2, qd = tid & 3;
    if (p0 + pix < HW) {
        unsigned w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = __builtin_amdgcn_fmed3f(tile[qd * 16 + 2 * j][pix], -65504.f, 65504.f);
            const float c = __builtin_amdgcn_fmed3f(tile[qd * 16 + 2 * j + 1][pix], -65504.f, 65504.f);
            w[j] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)c) << 16);
        }
        uint4* dst = reinterpret_cast<uint4*>(out + ((int64_t)b * HW + p0 + pix) * 64 + qd * 16);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// ---- x / max(||x||_2 over channels, eps) for an NCHW map (F.normalize(x, p=2, dim=1)) -------------------------------------------
// The UCN meta-arch normalises the backbone's 64-channel full-resolution embedding before the head
// (pretrained_meanshiftformer_model.py:298-300): one pass -- a lane owns one pixel (a wave reads 256 contiguous bytes of each
// channel plane), the C values stay in registers between the norm and the division (C <= 64), otherwise the map is read twice.
template <int CMAX>
__global__ __launch_bounds__(256) void l2norm_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t hw, float eps) {
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * C * hw;
    float* yb = y + (int64_t)b * C * hw;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < hw; p += (int64_t)gridDim.x * 256) {
        float ss = 0.f;
        float v[CMAX > 0 ? CMAX : 1];
        if constexpr (CMAX > 0) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) {
                    v[c] = xb[(int64_t)c * hw + p];
                    ss += v[c] * v[c];
                }
        } else {
            for (int c = 0; c < C; ++c) {
                const float t = xb[(int64_t)c * hw + p];
                ss += t * t;
            }
        }
        const float d = fmaxf(sqrtf(ss), eps);
        if constexpr (CMAX > 0) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) yb[(int64_t)c * hw + p] = v[c] / d;
        } else {
            for (int c = 0; c < C; ++c) yb[(int64_t)c * hw + p] = xb[(int64_t)c * hw + p] / d;
        }
    }
}

// ---- the location / softmax glue of the general MSDeformAttn.forward (ops/modules/ms_deform_attn.py:101-109) -------------------
//   attn = softmax over the L*P logits of a (query, head);  loc = ref[:, :, None, :, None, :] + off / (W_l, H_l)
// off [N*Lq][M][L][P][2], logits [N*Lq][M][L*P], ref [N*Lq][L][2] -> loc [N*Lq][M][L][P][2], attn [N*Lq][M][L][P];
// one lane per (query, head), L*P <= 64.
__global__ __launch_bounds__(256) void msda_locations_kernel(const float* __restrict__ off, const float* __restrict__ logits,
                                                             const float* __restrict__ ref, const int64_t* __restrict__ shapes,
                                                             float* __restrict__ loc, float* __restrict__ attn, int64_t rows, int M,
                                                             int L, int P) {
    const int LP = L * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * M; i += (int64_t)gridDim.x * 256) {
        const int64_t q = i / M;
        const float* lg = logits + i * LP;
        float mx = -INFINITY;
        for (int j = 0; j < LP; ++j) mx = fmaxf(mx, lg[j]);
        float den = 0.f;
        for (int j = 0; j < LP; ++j) den += expf(lg[j] - mx);
        for (int l = 0; l < L; ++l) {
            const float Hf = (float)shapes[2 * l], Wf = (float)shapes[2 * l + 1];
            const float rx = ref[(q * L + l) * 2], ry = ref[(q * L + l) * 2 + 1];
            for (int p = 0; p < P; ++p) {
                const int j = l * P + p;
                loc[(i * LP + j) * 2] = rx + off[(i * LP + j) * 2] / Wf;              // offset_normalizer = (W_l, H_l), :106-109
                loc[(i * LP + j) * 2 + 1] = ry + off[(i * LP + j) * 2 + 1] / Hf;
                attn[i * LP + j] = expf(lg[j] - mx) / den;
            }
        }
    }
}

}  // namespace msm

using namespace msm;

extern "C" int msm_layernorm_f32(const float* x, const float* parts, int n_parts, int64_t part_stride,
                                 const float* bias, const float* g1, const float* b1, int l2norm,
                                 const float* g2, const float* b2, float* y, float* y2,
                                 int rows, int E, float eps, void* stream) {
    MSM_REQUIRE(g1 && b1 && y, "msm_layernorm_f32: null pointer");
    MSM_REQUIRE(rows > 0, "msm_layernorm_f32: rows=%d", rows);
    MSM_REQUIRE(n_parts == 0 || parts, "msm_layernorm_f32: parts missing");
    MSM_REQUIRE(!g2 || (b2 && y2), "msm_layernorm_f32: second norm needs b2 and y2");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(rows, 4)), block(256);
#define LN_CASE(V)                                                                                             \
    hipLaunchKernelGGL((layernorm_kernel<V>), grid, block, 0, st, x, parts, n_parts, part_stride, bias, g1, b1, \
                       l2norm, g2, b2, y, y2, rows, eps)
    switch (E) {
        case 64: LN_CASE(1); break;
        case 128: LN_CASE(2); break;
        case 256: LN_CASE(4); break;
        case 512: LN_CASE(8); break;
        default: MSM_REQUIRE(false, "msm_layernorm_f32: unsupported E=%d", E);
    }
#undef LN_CASE
    MSM_CHECK_LAUNCH("msm_layernorm_f32");
    return MSM_OK;
}

extern "C" int msm_groupnorm_stats_f32(const float* x, double* stats, int stats_cleared, int B, int HW, int C, void* stream) {
    MSM_REQUIRE(x && stats && B > 0 && HW > 0, "msm_groupnorm_stats_f32: bad arguments");
    MSM_REQUIRE(C >= 4 && C <= 256 && C % 4 == 0 && 1024 % C == 0, "msm_groupnorm_stats_f32: C=%d must be a multiple of 4 dividing 1024", C);
    MSM_REQUIRE((((uintptr_t)x) & 15) == 0, "msm_groupnorm_stats_f32: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (!stats_cleared) MSM_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * (size_t)B * C, st));
    const int ppb = 256;
    dim3 grid(cdiv(HW, ppb), B), block(256);
    hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, st, x, stats, HW, C, ppb);
    MSM_CHECK_LAUNCH("msm_groupnorm_stats_f32");
    return MSM_OK;
}

static int groupnorm_apply_impl(const char* who, const float* x, const double* stats, const float* gamma, const float* beta,
                                       const float* up, int uh, int uw, int64_t up_batch_stride, float* y, int B, int H, int W,
                                       int C, int groups, float eps, int relu, uint16_t* planes, void* stream, int h16 = 0) {
    MSM_REQUIRE(x && stats && gamma && beta && (y || planes), "%s: null pointer", who);
    MSM_REQUIRE(groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 256, "%s: C=%d groups=%d", who, C, groups);
    MSM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)up)) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
    MSM_REQUIRE(!up || (uh > 0 && uw > 0), "%s: bad upsample source size", who);
    if (up_batch_stride == 0) up_batch_stride = (int64_t)uh * uw * C;
    MSM_REQUIRE(!up || (up_batch_stride >= (int64_t)uh * uw * C && up_batch_stride % 4 == 0), "%s: bad upsample batch stride", who);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)H * W * (C / 4);
    dim3 grid((unsigned)min((int64_t)1024, (total + 255) / 256), B), block(256);
    hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, st, x, stats, gamma, beta, up, uh, uw, up_batch_stride, y, H, W, C, groups, eps,
                       relu, planes, (int64_t)B * H * W * C, h16);
    MSM_CHECK_LAUNCH(who);
    return MSM_OK;
}

extern "C" int msm_groupnorm_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta,
                                       const float* up, int uh, int uw, int64_t up_batch_stride, float* y, int B, int H, int W,
                                       int C, int groups, float eps, int relu, void* stream) {
    MSM_REQUIRE(y, "msm_groupnorm_apply_f32: null pointer");
    return groupnorm_apply_impl("msm_groupnorm_apply_f32", x, stats, gamma, beta, up, uh, uw, up_batch_stride, y, B, H, W, C, groups, eps, relu,
                                nullptr, stream);
}

extern "C" int msm_groupnorm_apply_split(const float* x, const double* stats, const float* gamma, const float* beta,
                                         const float* up, int uh, int uw, int64_t up_batch_stride, uint16_t* planes, int B, int H, int W,
                                         int C, int groups, float eps, int relu, void* stream) {
    MSM_REQUIRE(planes && (((uintptr_t)planes) & 15) == 0, "msm_groupnorm_apply_split: planes must be a 16-byte aligned pointer");
    return groupnorm_apply_impl("msm_groupnorm_apply_split", x, stats, gamma, beta, up, uh, uw, up_batch_stride, nullptr, B, H, W, C, groups, eps,
                                relu, planes, stream);
}

extern "C" int msm_groupnorm_apply_f16(const float* x, const double* stats, const float* gamma, const float* beta,
                                       const float* up, int uh, int uw, int64_t up_batch_stride, void* y_f16, int B, int H, int W,
                                       int C, int groups, float eps, int relu, void* stream) {
    MSM_REQUIRE(y_f16 && (((uintptr_t)y_f16) & 15) == 0, "msm_groupnorm_apply_f16: y must be a 16-byte aligned pointer");
    return groupnorm_apply_impl("msm_groupnorm_apply_f16", x, stats, gamma, beta, up, uh, uw, up_batch_stride, nullptr, B, H, W, C, groups, eps,
                                relu, (uint16_t*)y_f16, stream, 1);
}

extern "C" int msm_groupnorm_apply_nchw_f32(const float* x, const double* stats, const float* gamma, const float* beta, float* y,
                                            int B, int HW, int C, int groups, float eps, int relu, void* stream) {
    MSM_REQUIRE(x && stats && gamma && beta && y && x != y, "msm_groupnorm_apply_nchw_f32: null or aliased pointer");
    MSM_REQUIRE(B > 0 && HW > 0 && HW % 4 == 0, "msm_groupnorm_apply_nchw_f32: HW=%d must be a positive multiple of 4", HW);
    MSM_REQUIRE(groups > 0 && C % groups == 0 && C % 4 == 0 && C >= 4 && C <= 128, "msm_groupnorm_apply_nchw_f32: C=%d groups=%d", C, groups);
    MSM_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0, "msm_groupnorm_apply_nchw_f32: pointers must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)3 * C + (size_t)C * 68);
    dim3 grid(cdiv(HW, 64), B), block(256);
    hipLaunchKernelGGL(gn_apply_nchw_kernel, grid, block, lds, (hipStream_t)stream, x, stats, gamma, beta, y, HW, C, groups, eps, relu);
    MSM_CHECK_LAUNCH("msm_groupnorm_apply_nchw_f32");
    return MSM_OK;
}

extern "C" int msm_pos_embed_sine(float* out, int H, int W, int npf, int64_t s_c, int64_t s_p, const float* add_c,
                                  float temperature, float scale, void* stream) {
    MSM_REQUIRE(out && H > 0 && W > 0 && npf > 0, "msm_pos_embed_sine: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)H * W * 2 * npf;
    dim3 grid((unsigned)min((int64_t)2048, (total + 255) / 256)), block(256);
    hipLaunchKernelGGL(pos_embed_kernel, grid, block, 0, st, out, H, W, npf, s_c, s_p, add_c, temperature, scale);
    MSM_CHECK_LAUNCH("msm_pos_embed_sine");
    return MSM_OK;
}

extern "C" int msm_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream) {
    MSM_REQUIRE(in && out && B > 0 && R > 0 && C > 0, "msm_transpose_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(C, 32), cdiv(R, 32), B), block(256);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, in, out, R, C);
    MSM_CHECK_LAUNCH("msm_transpose_f32");
    return MSM_OK;
}

extern "C" int msm_nchw_to_tokens_f16(const float* in, void* out, int B, int C, int HW, void* stream) {
    MSM_REQUIRE(in && out && B > 0 && HW > 0, "msm_nchw_to_tokens_f16: bad arguments");
    MSM_REQUIRE(C == 64, "msm_nchw_to_tokens_f16: C=%d, only 64 channels", C);
    MSM_REQUIRE((((uintptr_t)out) & 15) == 0, "msm_nchw_to_tokens_f16: out must be 16-byte aligned");
    hipLaunchKernelGGL(nchw_to_tokens_f16_kernel, dim3(cdiv(HW, 64), B), dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out, HW);
    MSM_CHECK_LAUNCH("msm_nchw_to_tokens_f16");
    return MSM_OK;
}

extern "C" int msm_l2_normalize_nchw_f32(const float* x, float* y, int B, int C, int HW, float eps, void* stream) {
    MSM_REQUIRE(x && y, "msm_l2_normalize_nchw_f32: null pointer");
    MSM_REQUIRE(B > 0 && C > 0 && HW > 0, "msm_l2_normalize_nchw_f32: bad shape");
    const int64_t hw = HW;
    dim3 grid((unsigned)max((int64_t)1, min((hw + 255) / 256, (int64_t)max(1, 8192 / B))), B), block(256);
    if (C <= 64) hipLaunchKernelGGL(l2norm_nchw_kernel<64>, grid, block, 0, (hipStream_t)stream, x, y, C, hw, eps);
    else hipLaunchKernelGGL(l2norm_nchw_kernel<0>, grid, block, 0, (hipStream_t)stream, x, y, C, hw, eps);
    MSM_CHECK_LAUNCH("msm_l2_normalize_nchw_f32");
    return MSM_OK;
}

extern "C" int msm_msda_locations(const float* offsets, const float* logits, const float* reference_points, const int64_t* spatial_shapes,
                                  float* sampling_loc, float* attn_weight, int64_t rows, int M, int L, int P, void* stream) {
    MSM_REQUIRE(offsets && logits && reference_points && spatial_shapes && sampling_loc && attn_weight, "msm_msda_locations: null pointer");
    MSM_REQUIRE(rows > 0 && M > 0 && L > 0 && P > 0, "msm_msda_locations: bad sizes");
    hipLaunchKernelGGL(msda_locations_kernel, dim3((unsigned)min((int64_t)4096, (rows * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       offsets, logits, reference_points, spatial_shapes, sampling_loc, attn_weight, rows, M, L, P);
    MSM_CHECK_LAUNCH("msm_msda_locations");
    return MSM_OK;
}
