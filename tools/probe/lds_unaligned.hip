// Does ds_read_b128 / ds_read_b64 on gfx950 accept addresses that are only 2-byte (or 4-byte) aligned, does it return the
// right bytes, and what does it cost?  (Design question of the bf16x6 Conv2d weight gradient: its B operand is a tap-shifted
// window of bf16 values, i.e. 16-byte fragments at arbitrary ELEMENT offsets.)
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k(unsigned* out, long long* cyc, int shift_bytes, int iters, int width) {
    __shared__ unsigned short lds[16384 + 64];
    for (int i = threadIdx.x; i < 16384 + 64; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    // lane t reads the 16 bytes starting at element 8 t (+ shift): lanes hit consecutive 16-byte slots
    unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 16 + shift_bytes;
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (width == 16) {
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc += v;
        } else {
            u32x2 a, b;
            asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(addr) : "memory");
            acc[0] += a[0]; acc[1] += a[1]; acc[2] += b[0]; acc[3] += b[1];
        }
        addr ^= (it & 1) ? 0u : 0u;
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = acc[i];
}

int main() {
    unsigned* d; long long* c;
    hipMalloc(&d, 256 * 4 * 4); hipMalloc(&c, 8);
    for (int width : {16, 8})
    for (int shift : {0, 2, 4, 6, 8, 10, 12, 14}) {
        const int iters = 1;
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, c, shift, iters, width);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> h(1024);
        hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t)
            for (int i = 0; i < 4; ++i) {
                const unsigned e0 = t * 8 + shift / 2 + 2 * i, want = (e0 & 0xffff) | ((e0 + 1) << 16);
                if (h[t * 4 + i] != want) ++bad;
            }
        // timing: many iterations
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, c, shift, 4096, width);
        hipDeviceSynchronize();
        long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("width %2d  shift %2d bytes: err=%d wrong dwords=%d of 1024   %.1f cycles per read round (4 waves)\n", width, shift, (int)e, bad, cy / 4096.0);
    }
    return 0;
}
