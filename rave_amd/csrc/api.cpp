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

thread_local hipEvent_t rh_ev_start = nullptr, rh_ev_stop = nullptr;
thread_local int rh_ev_used = 0;

extern "C" int rh_set_kernel_events(void* start, void* stop) {
    rh_ev_start = (hipEvent_t)start;
    rh_ev_stop = (hipEvent_t)stop;
    rh_ev_used = 0;
    return RH_OK;
}
// 1 if a main-kernel launch took the events since rh_set_kernel_events (also disarms them)
extern "C" int rh_kernel_events_used(void) {
    const int u = rh_ev_used;
    rh_ev_start = rh_ev_stop = nullptr;
    rh_ev_used = 0;
    return u;
}

// Range slots of the NEXT convolution / residual-unit / weight-gradient call of this thread (common.hpp: RH_X6_F16).
thread_local const unsigned* rh_rng_a = nullptr;
thread_local const unsigned* rh_rng_b = nullptr;
thread_local unsigned* rh_rng_out = nullptr;
thread_local unsigned* rh_rng_out2 = nullptr;
extern "C" int rh_x6_set_ranges(const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t* out2) {
    rh_rng_a = a; rh_rng_b = b; rh_rng_out = out; rh_rng_out2 = out2;
    return RH_OK;
}
void rh_take_ranges(const unsigned** a, const unsigned** b, unsigned** out, unsigned** out2) {
    if (a) *a = rh_rng_a;
    if (b) *b = rh_rng_b;
    if (out) *out = rh_rng_out;
    if (out2) *out2 = rh_rng_out2;
    rh_rng_a = rh_rng_b = nullptr;
    rh_rng_out = rh_rng_out2 = nullptr;
}
extern "C" int rh_x6_uses_ranges(void) { return RH_X6_F16 ? 1 : 0; }
extern "C" int rh_x6_range_words(void) { return kRangeSlotWords; }

extern "C" int rh_event_create(void** ev) {
    RH_REQUIRE(ev, RH_ERR_INVALID, "event_create: null pointer");
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { rh_set_error("event_create: hipEventCreate failed"); return RH_ERR_INVALID; }
    *ev = e;
    return RH_OK;
}
extern "C" int rh_event_destroy(void* ev) { return ev && hipEventDestroy((hipEvent_t)ev) != hipSuccess ? RH_ERR_INVALID : RH_OK; }
extern "C" int rh_event_elapsed_ms(void* start, void* stop, float* ms) {
    RH_REQUIRE(start && stop && ms, RH_ERR_INVALID, "event_elapsed: null pointer");
    const hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (e != hipSuccess) { rh_set_error("event_elapsed: %s", hipGetErrorString(e)); return RH_ERR_INVALID; }
    return RH_OK;
}

extern "C" int rh_version(void) { return RH_VERSION; }
extern "C" const char* rh_last_error(void) { return g_err; }
