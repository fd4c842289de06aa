// Write-bandwidth probe: how fast can a wave write when one store instruction covers SEG contiguous bytes per row and rows
// are `stride` bytes apart?  (K/V projection, mask_features and the token-major convolutions write 64-byte segments.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

// each wave instruction: 64 lanes x 16 B; lanes are grouped LPR per row -> LPR*16 contiguous bytes per row, 64/LPR rows
template <int LPR>
__global__ void probe(float4* out, long rows, long row_f4) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    constexpr int RPI = 64 / LPR;                 // rows per instruction
    const int r = lane / LPR, c = lane % LPR;
    // a wave owns RPI rows at a time and walks them left to right, like a kernel finishing column blocks of its tile
    for (long rb = wave * RPI; rb + RPI <= rows; rb += nwaves * RPI)
        for (long x = 0; x < row_f4; x += LPR) out[(rb + r) * row_f4 + x + c] = make_float4(1.f, 2.f, 3.f, (float)x);
}

template <int LPR>
void run(float4* d, long rows, long row_f4) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<LPR><<<2048, 256>>>(d, rows, row_f4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) probe<LPR><<<2048, 256>>>(d, rows, row_f4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("segment %4d B per row and instruction, row stride %ld B: %6.2f TB/s\n", LPR * 16, row_f4 * 16, rows * row_f4 * 16.0 * 10 / (ms * 1e-3) / 1e12);
}

int main() {
    const long row_f4 = 128;                      // 2 KiB rows (512 floats: a K|V row)
    const long rows = 1 << 18;                    // 512 MiB
    float4* d;
    hipMalloc(&d, rows * row_f4 * 16);
    run<4>(d, rows, row_f4);
    run<8>(d, rows, row_f4);
    run<16>(d, rows, row_f4);
    run<32>(d, rows, row_f4);
    run<64>(d, rows, row_f4);
    return 0;
}
