// Shared device/host helpers for librave_hip (gfx950 only: wave = 64 lanes, f32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/rave_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void rh_set_error(const char* fmt, ...);
int rh_check_launch(const char* what);

// Measurement hook (rh_set_kernel_events, bench.py's roofline leg): the next MAIN kernel launch of this thread -- the
// convolution / weight-gradient kernel itself, not its split-K finalize or reduction launches -- carries the pair of HIP
// events as its own start / stop events (hipExtLaunchKernelGGL: the dispatch's own timestamps, what rocprofv3 reports).
extern thread_local hipEvent_t rh_ev_start, rh_ev_stop;
extern thread_local int rh_ev_used;
#ifdef __HIP__
template <typename K, typename... A>
inline void rh_launch_main(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
    if (rh_ev_start) {
        hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, stream, rh_ev_start, rh_ev_stop, 0, args...);
        rh_ev_start = rh_ev_stop = nullptr;
        rh_ev_used = 1;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, stream, args...);
    }
}
#endif

#define RH_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            rh_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

// Activation applied to the conv input (rave/blocks.py: LeakyReLU(.2) / Snake).
__device__ __forceinline__ float rh_act_apply(float x, int act, float slope, float alpha) {
    if (act == RH_ACT_LEAKY) return x > 0.f ? x : x * slope;
    if (act == RH_ACT_SNAKE) {
        const float s = sinf(alpha * x);
        return x + s * s / (alpha + 1e-9f);
    }
    return x;
}
// d act(x) / dx.  LeakyReLU uses torch's convention (slope at x <= 0).
__device__ __forceinline__ float rh_act_grad(float x, int act, float slope, float alpha) {
    if (act == RH_ACT_LEAKY) return x > 0.f ? 1.f : slope;
    if (act == RH_ACT_SNAKE) return 1.f + alpha * sinf(2.f * alpha * x) / (alpha + 1e-9f);
    return 1.f;
}

// max(a, b) as ONE v_max_f32.  fmaxf() on a value that came from memory costs two: the compiler first canonicalises the
// operand (v_max x, x -- a loaded float may be a signalling NaN), 1 of the ~8.5 VALU instructions per converted sample of
// the bf16x6 kernels.  NaN behaviour is the instruction's (IEEE maxNum), which is what fmaxf lowers to anyway.
__device__ __forceinline__ float rh_max1(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

static inline int rh_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rh_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
