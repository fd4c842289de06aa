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
    return 0;
}
