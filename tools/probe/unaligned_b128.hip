// Do MUBUF dwordx4 loads / stores need more than dword alignment on gfx950, and is a raw-buffer dwordx4 access that
// straddles num_records checked per dword?   hipcc --offload-arch=gfx950 -O3 -o unaligned_b128 unaligned_b128.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* buf, int nfloats, int shift) {
    const auto r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nfloats * 4, 0x00020000);
    const unsigned off = (unsigned)((threadIdx.x * 4 + shift) * 4);          // element offset 4*t + shift
    u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(__uint_as_float(v[i]) + 1000.f);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}
int main() {
    for (int shift = 0; shift < 4; ++shift) {
        const int n = 64 * 4 + 2;              // the last lane's access straddles the end for shift >= 3; shift 1,2 fit or straddle
        std::vector<float> h(n + 8);
        for (int i = 0; i < n + 8; ++i) h[i] = (float)i;
        float* d;
        hipMalloc(&d, (n + 8) * 4);
        hipMemcpy(d, h.data(), (n + 8) * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(n + 8);
        hipMemcpy(o.data(), d, (n + 8) * 4, hipMemcpyDeviceToHost);
        int bad = 0, beyond = 0;
        for (int i = 0; i < n + 8; ++i) {
            const bool touched = i >= shift && i < 256 + shift;
            const float want = (touched && i < n) ? i + 1000.f : (float)i;
            if (o[i] != want) { if (i >= n) ++beyond; else ++bad; }
        }
        printf("shift %d: err=%d  wrong inside=%d  written beyond num_records=%d  (last 6: %g %g %g %g %g %g)\n", shift, (int)e, bad, beyond,
               o[n - 4], o[n - 3], o[n - 2], o[n - 1], o[n], o[n + 1]);
        hipFree(d);
    }
    return 0;
}
