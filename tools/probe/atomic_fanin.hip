// Round 6: what does the range-slot publish cost?  N workgroups each end with ONE agent-scope atomicMax on word (b % W) * S of a
// slot (S = stride in words: 1 = all words in one 128-byte line, 32 = one line per word), after streaming a little memory.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_fanin tools/probe/atomic_fanin.hip && ./atomic_fanin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* __restrict__ x, float4* __restrict__ y, unsigned* slot, int W, int S, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float4 v = x[i];
    v.x += 1.f;
    y[i] = v;
    if (mode == 0) return;
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (mode == 1) { if ((threadIdx.x & 63) == 0) atomicMax(slot + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % W) * S, __float_as_uint(m)); return; }
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (mode == 2) atomicMax(slot + (blockIdx.x % W) * S, __float_as_uint(m));
        else slot[blockIdx.x] = __float_as_uint(m);                  // mode 3: plain per-workgroup store
    }
}
int main() {
    const int NMAX = 16384;
    float4 *x, *y; unsigned* slot;
    hipMalloc(&x, (size_t)NMAX * 256 * 16); hipMalloc(&y, (size_t)NMAX * 256 * 16); hipMalloc(&slot, 1 << 20);
    hipMemset(x, 0, (size_t)NMAX * 256 * 16); hipMemset(slot, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int Ns[] = {512, 1024, 4096, 16384};
    for (int N : Ns) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int cfg = 0; cfg < (mode == 1 || mode == 2 ? 4 : 1); ++cfg) {
                const int W = cfg < 2 ? 32 : 8, S = (cfg & 1) ? 32 : 1;
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k, dim3(N), dim3(256), 0, 0, x, y, slot, W, S, mode);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("N=%5d mode=%d (%s) W=%2d S=%2d: %.2f us per launch\n", N, mode,
                       mode == 0 ? "no publish" : mode == 1 ? "atomic per wave" : mode == 2 ? "atomic per workgroup" : "store per workgroup", W, S, best * 100.f);
            }
        }
    }
    return 0;
}
