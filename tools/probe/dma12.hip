// probe: does buffer_load_dwordx3 ... lds (12-byte LDS-DMA) work on gfx950, and how does it lay data out?
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* in, float* out, int n, int misalign) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = -1.f;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    unsigned off = (unsigned)(lane * 5 + misalign) * 4u;          // arbitrary 4-byte aligned sources
    if (lane == 7) off = 0x80000000u;                             // OOB lane -> zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + 3), 12, off, 0, 0, 0);   // LDS base not 16B aligned
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
extern "C" int run(const float* in, float* out, int n, int misalign, void* s) {
    hipLaunchKernelGGL(k, 1, 64, 4096, (hipStream_t)s, in, out, n, misalign);
    return (int)hipGetLastError();
}
