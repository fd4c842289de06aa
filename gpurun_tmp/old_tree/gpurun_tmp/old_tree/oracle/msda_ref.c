/* CPU oracle (plain C) for multi-scale deformable attention forward.  TEST INFRASTRUCTURE ONLY:
 * loaded by tests/ (ctypes) and compiled by __graft_entry__.build(); never linked into or called
 * by the product library.
 *
 * Restates the arithmetic of the reference CUDA kernel
 *   MSMFormer/meanshiftformer/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304
 * (output element loop, pixel coordinates h = y*H - 0.5, inside test) and its bilinear helper
 * :38-89 (4 taps, zero outside), as straightforward nested loops over (b, q, m, c, l, p).
 * Pinned by tests/test_oracle_vs_golden.py against the reference's own PyTorch implementation
 * (ops/functions/ms_deform_attn_func.py:52-72) on the inputs of ops/test.py:24-63.
 */
#include <math.h>
#include <stdint.h>

#define DEFINE_MSDA(NAME, T)                                                                          \
    static T NAME##_tap(const T* v, int H, int W, int stride, T h, T w) {                              \
        const int h0 = (int)floor((double)h), w0 = (int)floor((double)w);                              \
        const int h1 = h0 + 1, w1 = w0 + 1;                                                            \
        const T lh = h - (T)h0, lw = w - (T)w0, hh = (T)1 - lh, hw = (T)1 - lw;                        \
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                              \
        if (h0 >= 0 && w0 >= 0) v1 = v[(h0 * W + w0) * stride];                                        \
        if (h0 >= 0 && w1 <= W - 1) v2 = v[(h0 * W + w1) * stride];                                    \
        if (h1 <= H - 1 && w0 >= 0) v3 = v[(h1 * W + w0) * stride];                                    \
        if (h1 <= H - 1 && w1 <= W - 1) v4 = v[(h1 * W + w1) * stride];                                \
        return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;                              \
    }                                                                                                  \
    void NAME(const T* value, const int64_t* shapes, const int64_t* start, const T* loc, const T* wgt, \
              T* out, int B, int S, int M, int D, int L, int Lq, int P) {                              \
        for (int b = 0; b < B; ++b)                                                                    \
            for (int q = 0; q < Lq; ++q)                                                               \
                for (int m = 0; m < M; ++m)                                                            \
                    for (int c = 0; c < D; ++c) {                                                      \
                        T col = 0;                                                                     \
                        const int64_t base = (((int64_t)b * Lq + q) * M + m) * L * P;                  \
                        for (int l = 0; l < L; ++l) {                                                  \
                            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];              \
                            const T* v = value + (((int64_t)b * S + start[l]) * M + m) * D + c;        \
                            for (int p = 0; p < P; ++p) {                                              \
                                const T x = loc[(base + l * P + p) * 2], y = loc[(base + l * P + p) * 2 + 1]; \
                                const T him = y * (T)H - (T)0.5, wim = x * (T)W - (T)0.5;              \
                                if (him > -1 && wim > -1 && him < H && wim < W)                        \
                                    col += NAME##_tap(v, H, W, M * D, him, wim) * wgt[base + l * P + p]; \
                            }                                                                          \
                        }                                                                              \
                        out[(((int64_t)b * Lq + q) * M + m) * D + c] = col;                            \
                    }                                                                                  \
    }

DEFINE_MSDA(msda_ref_f32, float)
DEFINE_MSDA(msda_ref_f64, double)
