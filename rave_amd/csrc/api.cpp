// Error reporting + version for the C ABI (include/rave_hip.h).  No mutable global state
// besides the thread-local message buffer.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "common.hpp"

static thread_local char g_err[512] = "";

void rh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rh_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return RH_OK;
}

extern "C" int rh_version(void) { return RH_VERSION; }
extern "C" const char* rh_last_error(void) { return g_err; }
