// Shared device/host helpers for librave_hip (gfx950 only: wave = 64 lanes, f32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rave_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void rh_set_error(const char* fmt, ...);
int rh_check_launch(const char* what);

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

static inline int rh_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rh_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
