// fp32 MFMA ceiling probe: what rate does v_mfma_f32_16x16x4_f32 sustain with W waves per SIMD and A independent
// accumulators per wave, with no memory traffic at all?  (The roofline `peak` in bench.py is the guide's 157.3 TFLOP/s =
// 256 CUs x 4 SIMDs x 2048 FLOP / 32 cycles x 2.4 GHz; this probe shows how much of it a pure-MFMA loop reaches for the
// duration of a ~150 us kernel and for a ~10 ms one, i.e. what clock the part actually holds.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int A>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x4 acc[A];
    for (int i = 0; i < A; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16 / A; ++k)
#pragma unroll
            for (int i = 0; i < A; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < A; ++i) s += acc[i];
    if (s[0] == 12345.f) out[0] = s[1] + s[2] + s[3];
}

// The encoder block's FFN stage on registers only: 32 linear1 MFMAs into 4 accumulators, bias + ReLU on the VALU (a true
// MFMA -> VALU -> MFMA dependency), 32 linear2 MFMAs into 4 running accumulators.  MODE 0: as is; 1: without the VALU
// step (linear1's output feeds linear2 directly).
template <int MODE>
__global__ __launch_bounds__(256) void ffn_probe(float* out, int stages) {
    float x[16], w[32];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 32; ++i) w[i] = blockIdx.x * 1e-3f + i;
    f32x4 acc2[4];
    for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < stages; ++s) {
        f32x4 dd[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            dd[q][0] = dd[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) dd[q][k & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(k + 16 * q) & 31], x[k], dd[q][k & 1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 h = MODE >= 2 ? dd[q][0] : dd[q][0] + dd[q][1];
            if (MODE >= 2) acc2[q] += dd[q][1] * 0.f;   // keep the second accumulator alive without a VALU step in front of linear2
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r] + w[r + s % 7], 0.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(ob * 4 + r + 16 * q) & 31], h[r], acc2[ob], 0, 0, 0);
        }
        if (MODE < 2)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] += 1e-9f * acc2[i & 3][i >> 2];      // keeps the stages dependent like the real chain's LayerNorm input
    }
    f32x4 t = acc2[0] + acc2[1] + acc2[2] + acc2[3];
    if (t[0] == 12345.f) out[0] = t[1] + t[2] + t[3];
}

template <int MODE>
void run_ffn(float* d, int wgs_per_cu, int stages) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    ffn_probe<MODE><<<grid, 256>>>(d, stages);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    ffn_probe<MODE><<<grid, 256>>>(d, stages);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * stages * 64 * 2048.0;
    printf("ffn stage pattern, mode %d, waves/SIMD %d, %5d stages: %8.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)\n", MODE, wgs_per_cu, stages,
           ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3 * 100);
}

template <int A>
void run(float* d, int wgs_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    probe<A><<<grid, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<A><<<grid, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 16 * 2048.0;
    printf("accumulators %d, waves/SIMD %d, %6d x 16 MFMAs per wave: %8.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)\n", A, wgs_per_cu,
           iters, ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3 * 100);
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    for (int w = 1; w <= 4; ++w) run<4>(d, w, 600 / w);          // ~ 100 us
    for (int w = 1; w <= 4; ++w) run<2>(d, w, 600 / w);
    run<1>(d, 1, 600);
    run<1>(d, 4, 150);
    run<4>(d, 4, 15000);                                            // ~ 10 ms
    run<4>(d, 4, 150000);                                           // ~ 100 ms
    for (int w = 1; w <= 4; ++w) run_ffn<0>(d, w, 3200 / w);          // ~ 3 ms: the launch overhead is out of the picture
    for (int w = 1; w <= 4; ++w) run_ffn<1>(d, w, 3200 / w);
    for (int w = 1; w <= 4; ++w) run_ffn<2>(d, w, 3200 / w);
    return 0;
}
