// Does a SIMD of gfx950 run one wave's MFMAs and another wave's VALU work concurrently?  (the bf16x6
// weight-gradient kernel pays the SUM of its MFMA and conversion phases.)  One 8-wave workgroup per CU: waves 0-3 (one
// per SIMD) take role A, waves 4-7 role B.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/probe/mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: role A runs MFMAs; bit 1: role B runs VALU FMAs; bit 2: role B runs MFMAs too; bit 3: role A also VALU
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool roleA = wave < 4;
    const bool do_mfma = roleA ? (mode & 1) : (mode & 4);
    const bool do_valu = roleA ? (mode & 8) : (mode & 2);
    float r = 0.f;
    if (do_mfma && !do_valu) {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) r += acc[t][0];
    } else if (do_valu && !do_mfma) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
        const float c = 1.0001f, d = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 48; ++u) v[u & 15] = __builtin_fmaf(v[u & 15], c, d);     // 48 VALU per 8 "MFMA slots" = 6 per MFMA
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    } else if (do_mfma && do_valu) {      // same wave: interleaved
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
        const float c = 1.0001f, d = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
                for (int w = 0; w < 6; ++w) v[(u * 6 + w) & 15] = __builtin_fmaf(v[(u * 6 + w) & 15], c, d);
            }
        }
        for (int t = 0; t < 4; ++t) r += acc[t][0];
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    if (r == 123456.789f) out[threadIdx.x] = r;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[] = {"A: MFMA          B: idle", "A: idle          B: VALU", "A: MFMA          B: VALU",
                           "A: MFMA          B: MFMA", "A: MFMA+VALU (one wave, interleaved)  B: idle", "A: MFMA+VALU  B: MFMA+VALU (two waves per SIMD, interleaved)",
                           "A: VALU          B: VALU"};
    const int modes[] = {1, 2, 3, 5, 9, 15, 10};
    for (int m = 0; m < 7; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, modes[m]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = 8.0 * iters, va = 48.0 * iters;
        printf("%-62s %8.3f ms   (%.1f cycles@2.4GHz per MFMA slot; %d MFMA, %d VALU per wave)\n", names[m], ms,
               ms * 1e-3 * 2.4e9 / mf, (int)mf, (int)va);
    }
    return 0;
}
