// Where do the 64 us of the FPN lateral (x NCHW 8 x 256 x 19200 fp32 -> 64 channels, K = 256) go?  The product kernel's structure
// (one 32-pixel tile per wave over the full K, ring of 4 k-groups) with parts switched off:
//   MODE 0 as shipped; 1 no MFMAs (loads only, xor-summed); 2 no x loads (MFMAs on stale registers); 3 x read as if it were
//   tile-major ([tile][k][32 pixels]: the same bytes, linear per wave) to price the NCHW stride pattern; 4 no loads in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lateral_probe.hip -o /tmp/lateral_probe && /tmp/lateral_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 4) void lateral(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int Cin, int HW) {
    constexpr int NT = 2, D = 4, WV = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const int tile = blockIdx.x * WV + wave;
    const int px0 = tile * 32;
    if (px0 >= HW) return;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * Cin * HW), 0, Cin * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 64 * Cin * 4, 0x00020000);
    const unsigned xo = MODE == 3 ? 4u * (unsigned)(tile * Cin * 32 + lq * 2 * 32 + NT * lj) : 4u * (unsigned)(lq * 2 * HW + px0 + NT * lj);
    const unsigned wo = 8u * (unsigned)lane;
    f32x4 acc[4][NT];
    for (int mt = 0; mt < 4; ++mt)
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float wa[D][4][2], xa[D][NT][2];
    auto load = [&](int kg, float (&wf)[4][2], float (&xf)[NT][2]) {
        if (MODE == 4) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(wr, wo, (unsigned)(kg * 4 + mt) * 512u, 0);
            wf[mt][0] = __uint_as_float(t.x); wf[mt][1] = __uint_as_float(t.y);
        }
        if (MODE == 2) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned so = MODE == 3 ? (unsigned)(kg * 8 + j) * 32u * 4u : (unsigned)(kg * 8 + j) * (unsigned)HW * 4u;
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(xr, xo, so, 0);
            xf[0][j] = __uint_as_float(t.x); xf[1][j] = __uint_as_float(t.y);
        }
    };
    auto mma = [&](const float (&wf)[4][2], const float (&xf)[NT][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (MODE == 1) acc[mt][nt][0] += wf[mt][j] * xf[nt][j];
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[mt][j], xf[nt][j], acc[mt][nt], 0, 0, 0);
                }
    };
    for (int d = 0; d < D; ++d) {
        for (int nt = 0; nt < NT; ++nt) xa[d][nt][0] = xa[d][nt][1] = lane * 1e-3f;
        for (int mt = 0; mt < 4; ++mt) wa[d][mt][0] = wa[d][mt][1] = lane * 2e-3f + mt;
    }
    const int groups = Cin / 8;
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, wa[d], xa[d]);
#pragma unroll 1
    for (int kg = 0; kg < groups; kg += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            mma(wa[d], xa[d]);
            load(min(kg + D + d, groups - 1), wa[d], xa[d]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int px = px0 + NT * lj + nt;
            const f32x4 v = acc[mt][nt];
            *reinterpret_cast<float4*>(out + ((int64_t)b * HW + px) * 64 + mt * 16 + lq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
}

template <int MODE>
static void run(const char* name, const float* x, const float* w, float* out, int B, int Cin, int HW) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((HW / 32 + 7) / 8, B);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lateral<MODE>, grid, dim3(512), 0, 0, x, w, out, Cin, HW);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lateral<MODE>, grid, dim3(512), 0, 0, x, w, out, Cin, HW);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.1f us\n", name, 1e3 * ms / reps);
}

int main() {
    const int B = 8, Cin = 256, HW = 19200;
    float *x, *w, *out;
    hipMalloc(&x, sizeof(float) * B * Cin * HW);
    hipMalloc(&w, sizeof(float) * 64 * Cin);
    hipMalloc(&out, sizeof(float) * B * HW * 64);
    hipMemset(x, 0, sizeof(float) * B * Cin * HW);
    hipMemset(w, 0, sizeof(float) * 64 * Cin);
    run<0>("as shipped", x, w, out, B, Cin, HW);
    run<1>("no MFMAs (loads + one VALU fma per product)", x, w, out, B, Cin, HW);
    run<2>("no x loads", x, w, out, B, Cin, HW);
    run<3>("x as tile-major (linear per wave)", x, w, out, B, Cin, HW);
    run<4>("no loads in the loop at all (MFMAs + stores)", x, w, out, B, Cin, HW);
    run<0>("as shipped (again)", x, w, out, B, Cin, HW);
    return 0;
}
