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

// Measurement builds only (tools/x6_products.sh, never the product library): RH_X6_PRODUCTS = 3 | 4 replaces the exact
// 3-way bf16 split of the bf16x6 forward / data-gradient kernels by a 2-piece split -- a1 = the high half of the float
// (truncation), a2 = the remainder rounded to nearest-even bf16 -- and the six partial products by a1b1 + a1b2 + a2b1
// (3) or those + a2b2 (4), smallest terms first.  a1 + a2 represents a to 2^-17 relative, the dropped a2b2 is <= 2^-16 |ab|:
// NOT the f32 numerics class of the 6-product kernels; SURVEY.md section 7 option (i), measured against north_star's
// 1e-4 bar in profiles/round4_x6_products.md.  Weight gradients keep their six products.
#ifndef RH_X6_PRODUCTS
#define RH_X6_PRODUCTS 6
#endif
#if RH_X6_PRODUCTS == 6
#define RH_X6_NPROD 6
#define RH_X6_NPIECE 3
#define RH_X6_SA {2, 0, 1, 1, 0, 0}
#define RH_X6_SB {0, 2, 1, 0, 1, 0}
#elif RH_X6_PRODUCTS == 4
#define RH_X6_NPROD 4
#define RH_X6_NPIECE 2
#define RH_X6_SA {1, 1, 0, 0}
#define RH_X6_SB {1, 0, 1, 0}
#elif RH_X6_PRODUCTS == 3
#define RH_X6_NPROD 3
#define RH_X6_NPIECE 2
#define RH_X6_SA {1, 0, 0}
#define RH_X6_SB {0, 1, 0}
#else
#error "RH_X6_PRODUCTS must be 6, 4 or 3"
#endif
// The three bf16 pieces of v as bit patterns whose HIGH halves are the pieces (the packers take the high halves).
__device__ __forceinline__ void rh_x6_split(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
    h0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(h0 & 0xffff0000u);
#if RH_X6_PRODUCTS == 6
    h1 = __float_as_uint(r1);
    h2 = __float_as_uint(r1 - __uint_as_float(h1 & 0xffff0000u));
#else
    const unsigned u = __float_as_uint(r1);
    h1 = u + 0x7fffu + ((u >> 16) & 1u);      // round to nearest-even at bit 16
    h2 = 0u;
#endif
}

static inline int rh_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rh_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
