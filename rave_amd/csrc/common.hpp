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

// consumes the thread's pending range slots (rh_x6_set_ranges): every conv entry point takes them exactly once
void rh_take_ranges(const unsigned** a, const unsigned** b, unsigned** out, unsigned** out2);

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

// ---- exact-f32 products on the 16-bit matrix cores ("x6" kernels: conv_x6_kernel.inc, unit_x6.hip, conv_wgrad_x6.hip) --------
// RH_X6_F16 = 1 (round 6, the product build): every f32 operand is scaled by a power of two that puts the TENSOR's largest
//   magnitude into [2^14, 2^15) and split into TWO f16 pieces, hi = RN16(x'), lo = RN16(x' - hi); a product block is the three
//   partial products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 (which keeps subnormal f16 inputs: tools/probe/f16x3.hip),
//   smallest terms first.  hi + lo represents x' to max(2^-24 |x'|, 2^-25) -- i.e. to f32 precision for every element within
//   2^-15 of the tensor's maximum and to 2^-39 of that maximum below -- and the dropped lo*lo term is <= 2^-22 |ab| (rms
//   2^-23.6): the error class of an f32 fmaf chain (measured: tools/check_x6.py, profiles/round6_*), at HALF the matrix
//   instructions and two thirds of the fragment traffic of the bf16 scheme.  The scale needs the tensor's max |x|:
//   producers leave it in a "range slot" (rh_x6_set_ranges, include/rave_hip.h), the weights' is in their packed operand.
// RH_X6_F16 = 0 (rounds 2-5, kept as the comparison build rave_amd/_var/librave_hip_bf16.so): three bf16 truncation pieces
//   (x == h1 + h2 + h3 exactly), six products with i + j <= 4 on v_mfma_f32_32x32x16_bf16; no scales, no range slots.
#ifndef RH_X6_F16
#define RH_X6_F16 1
#endif
#if RH_X6_F16
#define RH_X6_NPROD 3
#define RH_X6_NPIECE 2
#define RH_X6_SA {1, 0, 0}
#define RH_X6_SB {0, 1, 0}
typedef _Float16 rh_x6_frag __attribute__((ext_vector_type(8)));
#define RH_X6_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
#define RH_X6_NPROD 6
#define RH_X6_NPIECE 3
#define RH_X6_SA {2, 0, 1, 1, 0, 0}
#define RH_X6_SB {0, 2, 1, 0, 1, 0}
typedef __bf16 rh_x6_frag __attribute__((ext_vector_type(8)));
#define RH_X6_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
constexpr int kX6P = RH_X6_NPIECE;             // pieces (16-byte fragments) stored per octet of 8 K values

// A range slot: kRangeWords words, ONE PER 128-BYTE LINE (kRangeStride words apart), each the bit pattern of a non-negative
// float; the tensor's max |x| is the largest of them.  Producers atomicMax their workgroups' maxima into word
// (workgroup id) % kRangeWords.  Agent-scope atomics are performed memory-side on this chip and serialise per LINE: 1024
// workgroups publishing into 32 words of one line cost +5.4 us per launch, into 32 lines +0.4 us (tools/probe/atomic_fanin.hip,
// profiles/round6_probe_atomic_fanin.txt).  Must be zero before the producer runs.
constexpr int kRangeWords = 32;
constexpr int kRangeStride = 32;
constexpr int kRangeSlotWords = kRangeWords * kRangeStride;      // uint32 per slot (4 KB)
// Power-of-two scale that takes a tensor with max |x| = float(bits) into [2^14, 2^15), as a float bit pattern; *inv_exp = the
// biased exponent of its inverse.  Exponents are clamped to normal floats: tensors whose maximum is below 2^-111 lose
// precision (their products underflow f32 anyway), Inf / NaN maxima scale like the largest finite float.
__host__ __device__ __forceinline__ unsigned rh_x6_scale_bits(unsigned amax_bits, int* inv_exp) {
    int e = (int)((amax_bits >> 23) & 0xffu);
    if (e > 254) e = 254;
    int se = 268 - e;                           // 2^(14 - (e - 127)) has biased exponent 127 + 14 - (e - 127)
    if (se > 252) se = 252;
    if (se < 2) se = 2;
    *inv_exp = 254 - se;
    return (unsigned)se << 23;
}
// scale of the product: 1 / (scale_a * scale_b) as a float, exponent clamped to the normal range
__host__ __device__ __forceinline__ unsigned rh_x6_unscale_bits(int inv_exp_a, int inv_exp_b) {
    int e = inv_exp_a + inv_exp_b - 127;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    return (unsigned)e << 23;
}
#ifdef __HIP__
__device__ __forceinline__ unsigned rh_range_max(const unsigned* slot) {       // uniform (scalar loads)
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < kRangeWords; ++i) m = max(m, slot[i * kRangeStride]);
    return m;
}
// max |v| of the WORKGROUP (256 threads, every thread must call) -> ONE atomic on slot word `salt` % kRangeWords (12 k per-wave
// atomics into one line at the end of a split-K finalize launch cost 60 us: measured, round 6 -- hence one per workgroup, one
// line per word).
// `red`: 4 floats of LDS nobody else touches any more.
__device__ __forceinline__ void rh_range_publish(unsigned* slot, float amax_lane, unsigned salt, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax_lane = fmaxf(amax_lane, __shfl_xor(amax_lane, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax_lane;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        atomicMax(slot + (salt % kRangeWords) * kRangeStride, __float_as_uint(m));
    }
}
__device__ __forceinline__ float rh_absmax(float m, float v) {                 // max(m, |v|) as one v_max_f32
    float r;
    asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(v));
    return r;
}
#if RH_X6_F16
typedef _Float16 rh_f16x2 __attribute__((ext_vector_type(2)));
typedef float rh_f32x2 __attribute__((ext_vector_type(2)));
// the two f16 pieces of the (already scaled) pair (a, b), packed low half = a: v_cvt_pk_f16_f32 (round to nearest even),
// two v_cvt_f32_f16, two subtractions, v_cvt_pk_f16_f32
struct rh_h2 { unsigned hi, lo; };
__device__ __forceinline__ rh_h2 rh_h2_split(float a, float b) {
    const rh_f32x2 v = {a, b};
    const rh_f16x2 h = __builtin_convertvector(v, rh_f16x2);
    const rh_f32x2 r = v - __builtin_convertvector(h, rh_f32x2);
    const rh_f16x2 l = __builtin_convertvector(r, rh_f16x2);
    return rh_h2{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}
#endif
#endif
// The three bf16 pieces of v as bit patterns whose HIGH halves are the pieces (the packers take the high halves): the
// 2-D kernels (conv2d_x6.hip, wgrad2d_x6.hip) and the RH_X6_F16 = 0 build.
__device__ __forceinline__ void rh_bf3_split(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
    h0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(h0 & 0xffff0000u);
    h1 = __float_as_uint(r1);
    h2 = __float_as_uint(r1 - __uint_as_float(h1 & 0xffff0000u));
}

static inline int rh_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rh_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
