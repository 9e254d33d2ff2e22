// Round 6: can the exact-f32 convolution kernels run on the f16 matrix cores with TWO pieces per operand and THREE partial
// products (a1b1 + a1b2 + a2b1) instead of three bf16 pieces and six products?  Needs (1) v_mfma_f32_32x32x16_f16 to keep
// SUBNORMAL f16 inputs (the low piece of a small value is subnormal: with them the pair is a 40-bit fixed-point window below the
// tensor's scaled maximum, without them only 30 bits), (2) the f16 instruction to run at the bf16 rate under the power limit.
//   hipcc --offload-arch=gfx950 -O3 -o f16x3 tools/probe/f16x3.hip && ./f16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// (1) one wave: A[i][k] = a for k == 0 else 0, B[k][j] = b for k == 0 else 0  ->  every C element = a * b
__global__ void sub_k(float* out, float a, float b) {
    const int lane = threadIdx.x;
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)0.f; bv[i] = (_Float16)0.f; }
    if (lane < 32) { av[0] = (_Float16)a; bv[0] = (_Float16)b; }      // k = 0 lives in lanes 0..31, element 0
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    out[lane] = acc[0];
}

template <bool F16>
__global__ void rate_k(float* out, unsigned long long* ticks, const float* seed, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf16x8 a, b; f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) {
        const float x = seed[(threadIdx.x * 8 + i) & 4095], y = seed[(threadIdx.x * 8 + i + 2048) & 4095];
        a[i] = (__bf16)x; b[i] = (__bf16)y; ah[i] = (_Float16)x; bh[i] = (_Float16)y;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (F16) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int t = 0; t < 4; ++t) r += acc[t][0];
    if (r == 123456.789f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

template <bool F16>
void rate(float* out, unsigned long long* ticks, const float* seed, int threads, const char* tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_k<F16>, dim3(256), dim3(threads), 0, 0, out, ticks, seed, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[8];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = 8.0 * iters, tk = (double)h[0];
    printf("%-6s threads=%3d: wall %7.3f ms, clock %.3f GHz, %.1f ticks/MFMA, %.2f ns/MFMA per SIMD-wave\n", tag, threads, ms,
           tk / (ms * 1e6), tk / nm, ms * 1e6 / nm);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const float as[] = {1.f, 0x1p-14f, 0x1p-15f, 0x1p-20f, 0x1p-24f, 3 * 0x1p-24f, 1023 * 0x1p-24f};
    const float bs[] = {1.f, 1.f, 1.f, 1.f, 1.f, 0x1p10f, 0x1p-14f};
    printf("(1) subnormal f16 inputs of v_mfma_f32_32x32x16_f16 (C = a * b expected)\n");
    for (int i = 0; i < 7; ++i) {
        for (int swap = 0; swap < 2; ++swap) {
            const float a = swap ? bs[i] : as[i], b = swap ? as[i] : bs[i];
            hipLaunchKernelGGL(sub_k, dim3(1), dim3(64), 0, 0, out, a, b);
            float h[64];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            printf("   a = %-12g b = %-12g  C = %-14g expected %-14g %s\n", a, b, h[5], a * b, h[5] == a * b ? "ok" : "FLUSHED / WRONG");
        }
    }
    float* seed;
    unsigned long long* ticks;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&ticks, 64);
    float hs[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) hs[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    printf("(2) rate under load, random operands, 256 workgroups\n");
    for (int rep = 0; rep < 2; ++rep) {
        rate<false>(out, ticks, seed, 256, "bf16");
        rate<true>(out, ticks, seed, 256, "f16");
        rate<false>(out, ticks, seed, 512, "bf16");
        rate<true>(out, ticks, seed, 512, "f16");
    }
    return 0;
}
