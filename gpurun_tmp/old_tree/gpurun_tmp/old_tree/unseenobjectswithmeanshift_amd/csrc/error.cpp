// Thread-local error string + ABI version for libmsm_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
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

static std::atomic<int> g_opt[MSM_OPT_COUNT];
static struct OptInit {
    OptInit() {
        for (auto& o : g_opt) o.store(MSM_OPT_AUTO, std::memory_order_relaxed);
    }
} g_opt_init;
int opt(int key) { return g_opt[key].load(std::memory_order_relaxed); }

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
extern "C" int msm_set_option(int key, int value) {
    if (key < 0 || key >= MSM_OPT_COUNT) {
        msm::set_error("msm_set_option: unknown key %d", key);
        return MSM_E_INVALID;
    }
    msm::g_opt[key].store(value, std::memory_order_relaxed);
    return MSM_OK;
}
extern "C" int msm_get_option(int key) {
    if (key < 0 || key >= MSM_OPT_COUNT) return MSM_OPT_AUTO;
    return msm::opt(key);
}
