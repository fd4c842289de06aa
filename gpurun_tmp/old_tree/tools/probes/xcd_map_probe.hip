// Which XCD does workgroup (x, y) of a 2-D grid run on?  (csrc/dec_chain.hip's prefetch rows assume linear id % 8.)
// hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_map_probe.hip -o /tmp/xcd_map_probe && /tmp/xcd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(int* out, int spin) {
    const int id = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0) out[id] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;      // HW_REG_XCC_ID[3:0]
    // working rows spin for a while (like a tail), the last row exits at once
    if ((int)blockIdx.y < (int)gridDim.y - 1) {
        volatile int s = 0;
        for (int i = 0; i < spin; ++i) s += i;
    }
}
int main() {
    for (int gx : {50, 56, 25}) {
        for (int gy : {4, 5, 2}) {
            const int n = gx * gy;
            int* d;
            hipMalloc(&d, n * sizeof(int));
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(d, 0xff, n * sizeof(int));
                hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(512), 0, 0, d, 20000);
                hipDeviceSynchronize();
            }
            std::vector<int> h(n);
            hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
            int agree = 0, per[8] = {0};
            const int first = gx * (gy - 1);
            for (int i = 0; i < n; ++i) agree += h[i] == (i & 7);
            for (int i = first; i < n; ++i) per[h[i] & 7]++;
            printf("grid (%d, %d): %d of %d workgroups on XCD id %% 8; last row per XCD:", gx, gy, agree, n);
            for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
