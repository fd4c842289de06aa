// Thread-local error string + ABI version for libmsm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/msm_hip.h"

namespace msm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace msm

extern "C" const char* msm_last_error_string(void) { return msm::g_err; }
extern "C" int msm_abi_version(void) { return 1; }
