// Thread-local error string + ABI version for libmsm_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "../../include/msm_hip.h"

namespace msm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static const void* fns[64];
    static size_t sizes[64];
    static int n = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n; ++i)
        if (fns[i] == kernel) {
            if (sizes[i] >= bytes) return 0;
            hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) sizes[i] = bytes;
            return (int)e;
        }
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && n < 64) {
        fns[n] = kernel;
        sizes[n] = bytes;
        ++n;
    }
    return (int)e;
}
}  // namespace msm

extern "C" const char* msm_last_error_string(void) { return msm::g_err; }
extern "C" int msm_abi_version(void) { return MSM_ABI_VERSION; }
