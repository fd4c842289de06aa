// msm_l2_prefetch (see include/msm_hip.h): pull up to 8 byte ranges into EVERY XCD's L2 ahead of the kernel that streams them.
//
// Why: the decoder's row-local tails (csrc/dec_chain.hip) are chains of 256 x 256 GEMM stages whose workgroups stream 0.4 - 2.2 MB of
// packed weights per launch through one CU each; between two uses of a layer's weights a whole forward pass goes through the 4-MiB
// per-XCD L2s, so every launch starts on HBM / Infinity-Cache latency -- measured (tools/probes/tails_graph_time.py): post_self 9.7 us
// with its weights L2-resident against 14.1 us inside a pass.  The launches in FRONT of a tail (the attention cores: a few dozen
// workgroups, K/V streams of a few MB) leave the fabric idle, so the decoder forks a side stream there and this kernel touches the
// next tail's weights: block b reads slice b / 8 of every range -- blocks are dealt to the XCDs round robin (block b runs on XCD
// b % 8: an observed placement, relied on for SPEED only; a wrong guess costs the prefetch, never a result).
#include "common.h"

namespace msm {

constexpr int PF_MAX = 8;          // ranges per launch
constexpr int PF_SLICES = 16;      // workgroups per XCD
struct PfJobs {
    const unsigned char* p[PF_MAX];
    int64_t bytes[PF_MAX];
    int n;
};
typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void l2_prefetch_kernel(PfJobs jobs, unsigned* __restrict__ sink) {
    const int slice = blockIdx.x >> 3;                       // (blockIdx.x & 7 = the XCD this block is expected on)
    unsigned acc = 0;
    for (int j = 0; j < jobs.n; ++j) {
        const int64_t units = jobs.bytes[j] >> 4;            // 16-byte units
        const int64_t per = (units + PF_SLICES - 1) / PF_SLICES;
        const int64_t u0 = (int64_t)slice * per, u1 = min(units, u0 + per);
        const pf_u32x4* src = reinterpret_cast<const pf_u32x4*>(jobs.p[j]);
        // one 16-byte load per 128-byte line is enough to allocate it: a lane touches line (u0 / 8 + i), 8 units apart
        for (int64_t u = u0 + (int64_t)threadIdx.x * 8; u < u1; u += 256 * 8) {
            const pf_u32x4 v = src[u];
            acc ^= v.x;
        }
    }
    if (acc == 0x9e3779b9u && sink) *sink = acc;             // (never true in practice: keeps the loads alive)
}

}  // namespace msm

using namespace msm;

extern "C" int msm_l2_prefetch(const void* const* ptrs, const int64_t* bytes, int n, void* stream) {
    MSM_REQUIRE(ptrs && bytes && n >= 1 && n <= PF_MAX, "msm_l2_prefetch: 1..%d ranges", PF_MAX);
    PfJobs jobs{};
    jobs.n = n;
    for (int j = 0; j < n; ++j) {
        MSM_REQUIRE(ptrs[j] && bytes[j] >= 0 && (((uintptr_t)ptrs[j]) & 15) == 0, "msm_l2_prefetch: range %d must be a 16-byte aligned device pointer", j);
        jobs.p[j] = (const unsigned char*)ptrs[j];
        jobs.bytes[j] = bytes[j];
    }
    hipLaunchKernelGGL(l2_prefetch_kernel, dim3(8 * PF_SLICES), dim3(256), 0, (hipStream_t)stream, jobs, (unsigned*)nullptr);
    MSM_CHECK_LAUNCH("msm_l2_prefetch");
    return MSM_OK;
}
