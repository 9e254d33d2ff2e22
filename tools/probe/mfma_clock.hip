// Settles VERDICT r2 weak #3 with measurements instead of an assumed clock:
//   * the shader clock under a pure v_mfma_f32_32x32x16_bf16 load = s_memtime ticks / wall time (the probe
//     mfma_valu_overlap.hip priced everything at an ASSUMED 2.4 GHz);
//   * cycles per MFMA (ticks, not wall) with F = 0 .. 8 independent VALU fillers per MFMA gap, interleaved by
//     sched_group_barrier, with ONE wave per SIMD (256-thread workgroup per CU) and with TWO (512 threads);
//   * the same fillers issued by the PARTNER wave of the SIMD instead (role split).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock tools/probe/mfma_clock.hip && ./mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int F>
__device__ __forceinline__ void body(f32x16 (&acc)[4], bf16x8 a, bf16x8 b, float (&v)[16], int iters) {
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < F; ++w) v[(u * F + w) & 15] = __builtin_fmaf(v[(u * F + w) & 15], c, d);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (F > 0) __builtin_amdgcn_sched_group_barrier(0x2, F, 0);
        }
    }
}

// role: 0 = MFMA + F fillers in the same wave (all waves); 1 = waves 0-3 MFMA only, waves 4-7 the fillers (8 F per 8 MFMA slots)
template <int F>
__global__ void k(float* out, unsigned long long* ticks, const float* seed, int iters, int role) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 4095]; b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 2048) & 4095]; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed[(threadIdx.x + i) & 4095];
    const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime
    if (role == 0) body<F>(acc, a, b, v, iters);
    else if (wave < 4) body<0>(acc, a, b, v, iters);
    else {
        const float c = 1.0001f, d = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8 * (F > 0 ? F : 1); ++u) v[u & 15] = __builtin_fmaf(v[u & 15], c, d);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int t = 0; t < 4; ++t) r += acc[t][0];
    for (int i = 0; i < 16; ++i) r += v[i];
    if (r == 123456.789f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

template <int F>
void run(float* out, unsigned long long* ticks, const float* seed, int threads, int role, const char* tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<F>, dim3(256), dim3(threads), 0, 0, out, ticks, seed, iters, role);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[8];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = 8.0 * iters;
    const double tk = (double)h[0];
    printf("%-28s F=%d threads=%3d: wall %7.3f ms, wave0 %9.0f ticks -> clock %.3f GHz, %.1f ticks/MFMA (%.1f ns), last wave %9.0f ticks\n",
           tag, F, threads, ms, tk, tk / (ms * 1e6), tk / nm, ms * 1e6 / nm, (double)h[threads / 64 - 1]);
}

int main() {
    float *out, *seed;
    unsigned long long* ticks;
    hipMalloc(&out, 4096);
    hipMalloc(&ticks, 64);
    hipMalloc(&seed, 4096 * 4);
    float hs[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) hs[i] = (float)rand() / RAND_MAX * 2.f - 1.f;     // full-range random operands (DVFS: rule 25)
    hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    puts("== one wave per SIMD (256 threads), fillers in the MFMA wave");
    run<0>(out, ticks, seed, 256, 0, "same-wave");
    run<1>(out, ticks, seed, 256, 0, "same-wave");
    run<2>(out, ticks, seed, 256, 0, "same-wave");
    run<3>(out, ticks, seed, 256, 0, "same-wave");
    run<4>(out, ticks, seed, 256, 0, "same-wave");
    run<5>(out, ticks, seed, 256, 0, "same-wave");
    run<6>(out, ticks, seed, 256, 0, "same-wave");
    run<8>(out, ticks, seed, 256, 0, "same-wave");
    puts("== two waves per SIMD (512 threads), every wave MFMA + F fillers (ticks/MFMA is per wave: the SIMD issues twice that)");
    run<0>(out, ticks, seed, 512, 0, "two same-wave");
    run<2>(out, ticks, seed, 512, 0, "two same-wave");
    run<4>(out, ticks, seed, 512, 0, "two same-wave");
    run<6>(out, ticks, seed, 512, 0, "two same-wave");
    puts("== two waves per SIMD, role split: waves 0-3 MFMA only, waves 4-7 issue the F fillers per MFMA slot");
    run<0>(out, ticks, seed, 512, 1, "partner idle-ish (F=1 load)");
    run<2>(out, ticks, seed, 512, 1, "role split");
    run<4>(out, ticks, seed, 512, 1, "role split");
    run<6>(out, ticks, seed, 512, 1, "role split");
    run<8>(out, ticks, seed, 512, 1, "role split");
    return 0;
}
