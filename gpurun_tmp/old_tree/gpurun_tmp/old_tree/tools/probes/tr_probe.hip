// ds_read_b64_tr_b16 semantics probe: LDS holds u16 value = its own element index; lane l supplies byte address addr[l];
// prints for lanes 0..63 the four returned elements.  hipcc --offload-arch=gfx950 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
    int h[64]; unsigned short o[256];
    int *d; unsigned short* od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: lane l -> row (l%16)/4 + 4*(l/16) of a [rows][64 u16] image, chunk (l%4): address = row*128 + (l%4)*8
        // mode 1: every lane its own row l (address l*128)
        for (int l = 0; l < 64; ++l) h[l] = mode == 0 ? (((l % 16) / 4 + 4 * (l / 16)) * 128 + (l % 4) * 8) : l * 128;
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
        hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
    return 0;
}
