// What read rate does the NCHW tile pattern of the input projections reach from HBM, whatever the arithmetic?  x [8][256][19200] fp32
// (157 MB), four copies in rotation so that no launch finds its input in the 256-MB MALL.  Patterns:
//   0  linear: a wave reads 1 KiB contiguous per load, grid-stride over the tensor
//   1  the lateral kernel's tile: a wave owns 32 pixels, a K group = 8 loads of 8 B per lane (lane = (pixel pair lj, row block lq): 4 rows x 128 B)
//   2  the same with 64 pixels per wave, 16 B per lane (4 rows x 256 B per load)
//   3  pattern 2, but a wave's loads of a group cover ONE row each: lane -> 16 B of a 1-KiB run (256 pixels per workgroup-row... per wave)
// DEPTH groups are requested before the first is consumed.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CIN = 256, HW = 19200, B = 8;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, unsigned* __restrict__ sink, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * CIN * HW), 0, CIN * HW * 4, 0x00020000);
    unsigned acc = 0;
    if (PAT == 0) {
        // the image as 16-byte chunks: wave w of the grid takes chunks [i*64 .. +64) round robin
        const int waves = gridDim.x * 4, me = blockIdx.x * 4 + wave;
        const int chunks = CIN * HW / 4 / 64;           // wave-loads per image
        u32x4 v[DEPTH];
        int i = me;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { v[d] = __builtin_amdgcn_raw_buffer_load_b128(xr, 16u * lane, (unsigned)min(i, chunks - 1) * 1024u, 0); i += waves; }
        for (int c = me; c < chunks; c += waves * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
                v[d] = __builtin_amdgcn_raw_buffer_load_b128(xr, 16u * lane, (unsigned)min(i, chunks - 1) * 1024u, 0);
                i += waves;
            }
        }
    } else {
        constexpr int NT = PAT == 1 ? 2 : 4;
        const int ntiles = HW / (16 * NT);
        const int t_first = blockIdx.x * 4 + wave, t_stride = gridDim.x * 4;
        constexpr int G = CIN / 32;
        u32x4 v[DEPTH][8];
        int lt = t_first, lg = 0;
        auto load = [&](u32x4 (&r)[8]) {
            const int tt = min(lt, ntiles - 1);
            unsigned xo;
            if (PAT == 3) xo = 4u * (unsigned)(tt * 64 + 4 * lj) + 0u;      // rows come from the scalar offset + lq below
            else xo = 4u * (unsigned)(lq * 8 * HW + tt * (16 * NT) + NT * lj);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (PAT == 1) {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(xr, xo, (unsigned)(lg * 32 + e) * (unsigned)HW * 4u, 0);
                    r[e] = u32x4{t.x, t.y, 0, 0};
                } else if (PAT == 2) {
                    r[e] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo, (unsigned)(lg * 32 + e) * (unsigned)HW * 4u, 0);
                } else {
                    // 16 lanes x 16 B = 256 B of one row per quarter wave; quarter lq takes row 4 e' + lq of the group: same bytes as PAT 2,
                    // but ... (kept identical on purpose: the control for the address arithmetic)
                    r[e] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo + 4u * (unsigned)(lq * 8 * HW), (unsigned)(lg * 32 + e) * (unsigned)HW * 4u, 0);
                }
            }
            if (++lg == G) { lg = 0; lt += t_stride; }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(v[d]);
        for (int it = 0; it < tiles_per_wave; ++it)
            for (int g = 0; g < G; g += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc ^= v[d][e].x ^ v[d][e].y ^ v[d][e].z ^ v[d][e].w;
                    load(v[d]);
                }
            }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int PAT, int DEPTH>
static void run(const char* name, std::vector<float*>& xs, unsigned* sink, int wgs_per_image) {
    constexpr int NT = PAT == 1 ? 2 : 4;
    const int ntiles = HW / (16 * NT);
    const int tpw = PAT == 0 ? 0 : (ntiles + wgs_per_image * 4 - 1) / (wgs_per_image * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<PAT, DEPTH>), dim3(wgs_per_image, B), dim3(256), 0, 0, xs[i % xs.size()], sink, tpw);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<PAT, DEPTH>), dim3(wgs_per_image, B), dim3(256), 0, 0, xs[i % xs.size()], sink, tpw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)B * CIN * HW * 4;
    printf("%-34s depth %d, %3d workgroups/image: %6.1f us  %5.2f TB/s\n", name, DEPTH, wgs_per_image, us, bytes / us / 1e6);
}

int main() {
    std::vector<float*> xs(4);
    const size_t n = (size_t)B * CIN * HW;
    for (auto& p : xs) { hipMalloc(&p, n * 4); hipMemset(p, 1, n * 4); }
    unsigned* sink;
    hipMalloc(&sink, 4);
    run<0, 4>("linear", xs, sink, 128);
    run<0, 8>("linear", xs, sink, 128);
    run<0, 8>("linear", xs, sink, 256);
    run<1, 2>("32-pixel tiles, 8 B/lane", xs, sink, 50);
    run<1, 2>("32-pixel tiles, 8 B/lane", xs, sink, 75);
    run<1, 4>("32-pixel tiles, 8 B/lane", xs, sink, 75);
    run<1, 4>("32-pixel tiles, 8 B/lane", xs, sink, 150);
    run<2, 2>("64-pixel tiles, 16 B/lane", xs, sink, 38);
    run<2, 4>("64-pixel tiles, 16 B/lane", xs, sink, 38);
    run<2, 2>("64-pixel tiles, 16 B/lane", xs, sink, 75);
    run<2, 4>("64-pixel tiles, 16 B/lane", xs, sink, 75);
    return 0;
}
