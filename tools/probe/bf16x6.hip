// probe: exact-f32 GEMM/conv inner loop on the bf16 matrix cores ("bf16x6"): every f32 operand is split into
// three bf16 pieces (truncation splits are exact: x == h1 + h2 + h3), and a product is the six partial products
// with i + j <= 4 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) accumulated in f32 by v_mfma_f32_32x32x16_bf16.
//   C[m][n] = sum_t sum_c W[t][c][m] * X[c][n + t]        (conv-like: taps re-use the staged X tile)
// Weights are pre-split (split_weights_kernel) into [t][c/8][3][M][8] bf16; the X tile is DMA'd as f32 into LDS,
// converted once per K chunk into [c/8][3][pos][8] bf16, and then read with ds_read_b128 (no VALU in the MFMA loop).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ void split3(float x, unsigned& a, unsigned& b, unsigned& c) {
    a = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(a);
    b = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b);
    c = __float_as_uint(r2) & 0xffff0000u;
}

// W [T][C][M] f32 -> wq [T][C/8][3][M][8] bf16
__global__ void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ wq, int T, int C, int M) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)T * C * M) return;
    const int m = e % M;
    const long r = e / M;
    const int c = r % C, t = r / C;
    unsigned a, b, cc;
    split3(w[e], a, b, cc);
    const long base = (((long)(t * (C / 8) + c / 8) * 3) * M + m) * 8 + (c & 7);
    wq[base] = a >> 16;
    wq[base + (long)M * 8] = b >> 16;
    wq[base + 2l * M * 8] = cc >> 16;
}

template <int TM>
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wq,
                                                          float* __restrict__ out, int T, int C, int M, int N, int NX, int mode) {
    constexpr int BM = TM * 32, BN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int P = BN + T - 1;                 // positions staged per channel
    const int pitch = (P + 3) & ~3;
    // LDS map: [xf32 stage 0][xf32 stage 1][xb16][wq stage 0][wq stage 1]
    const int xf_floats = (16 * pitch + 255) & ~255;   // DMA instructions write whole 256-float slots
    const int xb_bytes = 2 * 3 * P * 16;
    const int w_bytes = T * 2 * 3 * BM * 16;
    float* xf = reinterpret_cast<float*>(smem_raw);
    unsigned char* xb = smem_raw + 2 * xf_floats * 4;
    unsigned char* ws = xb + xb_bytes;
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((long)C * NX * 4), 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wq), 0, (unsigned)((long)T * C * M * 6), 0x00020000);

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    const int nchunks = C / 16;
    auto issue = [&](int ch, int stage) {
        // weights: LDS image [t][cb(2)][s(3)][BM] x 16 B
        const int w16 = T * 2 * 3 * BM;
        unsigned char* wdst = ws + stage * w_bytes;
        for (int q = wave; q * 64 < w16; q += 4) {
            const int f = q * 64 + lane;
            const int m = f % BM;
            int r = f / BM;
            const int s = r % 3; r /= 3;
            const int cb = r % 2, t = r / 2;
            unsigned off = kOOB;
            if (f < w16) off = (unsigned)(((((long)t * (C / 8) + ch * 2 + cb) * 3 + s) * M + m0 + m) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wdst + q * 1024), 16, off, 0, 0, 0);
        }
        float* xdst = xf + stage * xf_floats;
        const int x4 = 16 * (pitch / 4);
        for (int q = wave; q * 64 < x4; q += 4) {
            const int f = q * 64 + lane;
            const int c = f / (pitch / 4), v = f - c * (pitch / 4);
            unsigned off = kOOB;
            if (f < x4 && n0 + 4 * v < NX) off = (unsigned)((((long)(ch * 16 + c)) * NX + n0 + 4 * v) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(xdst + q * 256), 16, off, 0, 0, 0);
        }
    };

    issue(0, 0);
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (i + 1 < nchunks && !((mode & 2) && i > 0)) issue(i + 1, (i + 1) & 1);
        // ---- convert the f32 tile of chunk i into bf16 triples [cb][s][pos][8]
        const float* xs = xf + (i & 1) * xf_floats;
        for (int e = tid; e < 2 * P && !((mode & 1) && i > 0); e += 256) {
            const int cb = e / P, pos = e - cb * P;
            unsigned h[3][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) split3(xs[(cb * 8 + k) * pitch + pos], h[0][k], h[1][k], h[2][k]);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                u32x4 pk;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk[k] = (h[s][2 * k] >> 16) | h[s][2 * k + 1];
                *reinterpret_cast<u32x4*>(xb + ((cb * 3 + s) * P + pos) * 16) = pk;
            }
        }
        __syncthreads();
        const unsigned char* wl = ws + (i & 1) * w_bytes;
        const int g = lane >> 5, j = lane & 31;
        for (int t = 0; t < T && !(mode & 4); ++t) {
            bf16x8 bfr[3], afr[TM][3];
#pragma unroll
            for (int s = 0; s < 3; ++s)
                bfr[s] = *reinterpret_cast<const bf16x8*>(xb + ((g * 3 + s) * P + wave * 32 + j + t) * 16);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    afr[tm][s] = *reinterpret_cast<const bf16x8*>(wl + (((t * 2 + g) * 3 + s) * BM + tm * 32 + j) * 16);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                // smallest terms first
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][2], bfr[0], acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][0], bfr[2], acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][1], bfr[1], acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][1], bfr[0], acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][0], bfr[1], acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][0], bfr[0], acc[tm], 0, 0, 0);
            }
        }
    }
    const int j = lane & 31, kh = lane >> 5;
    const int n = n0 + wave * 32 + j;
    if (n < N) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) out[(long)m * N + n] = acc[tm][r];
            }
    }
}

extern "C" int split_weights(const float* w, uint16_t* wq, int T, int C, int M, void* s) {
    const long n = (long)T * C * M;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, wq, T, C, M);
    return (int)hipGetLastError();
}

extern "C" int gemm(const float* x, const uint16_t* wq, float* out, int T, int C, int M, int N, int NX, void* s, int mode) {
    constexpr int TM = 3;
    const int P = 128 + T - 1, pitch = (P + 3) & ~3;
    const size_t lds = 2 * ((16 * pitch + 255) & ~255) * 4 + 2 * 3 * P * 16 + 2 * (size_t)T * 2 * 3 * (TM * 32) * 16;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x6_kernel<TM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (lds > 160 * 1024) return -1;
    dim3 grid((N + 127) / 128, (M + TM * 32 - 1) / (TM * 32));
    hipLaunchKernelGGL(gemm_bf16x6_kernel<TM>, grid, dim3(256), lds, (hipStream_t)s, x, wq, out, T, C, M, N, NX, mode);
    return (int)hipGetLastError();
}

// ---- v2: 2 workgroups per CU (77 KB LDS: single f32 stage, single bf16 tile, double-buffered weights), per-lane DMA
// offsets computed once, conversion placed between the DMA wait and the next DMA issue, MFMAs interleaved over tm.
template <int TM>
__global__ __launch_bounds__(256) void gemm_bf16x6_v2_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wq,
                                                             float* __restrict__ out, int T, int C, int M, int N, int NX,
                                                             int mode) {
    constexpr int BM = TM * 32, BN = 128;
    constexpr int kNW = 8, kNXI = 3;             // DMA slots per lane: weights (T <= 3), x tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int P = BN + T - 1;
    const int pitch = (P + 3) & ~3;
    const int xf_floats = (16 * pitch + 255) & ~255;
    const int xb_bytes = 2 * 3 * P * 16;
    const int w16 = T * 2 * 3 * BM;
    const int w_bytes = ((w16 + 63) & ~63) * 16;
    float* xf = reinterpret_cast<float*>(smem_raw);
    unsigned char* xb = smem_raw + xf_floats * 4;
    unsigned char* ws = xb + ((xb_bytes + 15) & ~15);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((long)C * NX * 4), 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wq), 0, (unsigned)((long)T * C * M * 6), 0x00020000);

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    // per-lane DMA source offsets at chunk 0; a chunk adds a scalar
    unsigned wo[kNW], xo[kNXI];
    const int nwi = (w16 + 255) / 256, x4 = 16 * (pitch / 4), nxi = (x4 + 255) / 256;
#pragma unroll
    for (int i = 0; i < kNW; ++i) {
        const int f = (wave + 4 * i) * 64 + lane;
        const int m = f % BM;
        int r = f / BM;
        const int s3 = r % 3; r /= 3;
        const int cb = r % 2, t = r / 2;
        wo[i] = (i < nwi && f < w16) ? (unsigned)(((((long)t * (C / 8) + cb) * 3 + s3) * M + m0 + m) * 16) : kOOB;
    }
#pragma unroll
    for (int i = 0; i < kNXI; ++i) {
        const int f = (wave + 4 * i) * 64 + lane;
        const int c = f / (pitch / 4), v = f - c * (pitch / 4);
        xo[i] = (i < nxi && f < x4 && n0 + 4 * v < NX) ? (unsigned)((((long)c) * NX + n0 + 4 * v) * 4) : kOOB;
    }
    const unsigned w_step = (unsigned)(2l * 3 * M * 16), x_step = (unsigned)(16l * NX * 4);
    auto issue_w = [&](int ch, int stage) {
#pragma unroll
        for (int i = 0; i < kNW; ++i)
            if (i < nwi && (wave + 4 * i) * 64 < w16) {
                const unsigned off = wo[i] == kOOB ? kOOB : wo[i] + ch * w_step;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(ws + stage * w_bytes + (wave + 4 * i) * 1024), 16, off, 0, 0, 0);
            }
    };
    auto issue_x = [&](int ch) {
#pragma unroll
        for (int i = 0; i < kNXI; ++i)
            if (i < nxi && (wave + 4 * i) * 256 < xf_floats) {
                const unsigned off = xo[i] == kOOB ? kOOB : xo[i] + ch * x_step;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(xf + (wave + 4 * i) * 256), 16, off, 0, 0, 0);
            }
    };

    const int nchunks = C / 16;
    const int g = lane >> 5, j = lane & 31;
    issue_w(0, 0);
    issue_x(0);
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // A: chunk i landed; everyone finished the MFMAs of chunk i-1
        for (int e = tid; e < 2 * P && !((mode & 1) && i > 0); e += 256) {
            const int cb = e / P, pos = e - cb * P;
            unsigned h[3][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) split3(xf[(cb * 8 + k) * pitch + pos], h[0][k], h[1][k], h[2][k]);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                u32x4 pk;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk[k] = (h[s3][2 * k] >> 16) | h[s3][2 * k + 1];
                *reinterpret_cast<u32x4*>(xb + ((cb * 3 + s3) * P + pos) * 16) = pk;
            }
        }
        __syncthreads();                                   // B: bf16 tile ready; the f32 stage is free again
        if (i + 1 < nchunks && !((mode & 2) && i > 0)) {
            issue_w(i + 1, (i + 1) & 1);
            issue_x(i + 1);
        }
        const unsigned char* wl = ws + (i & 1) * w_bytes;
        for (int t = 0; t < T && !(mode & 4); ++t) {
            bf16x8 bfr[3], afr[TM][3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
                bfr[s3] = *reinterpret_cast<const bf16x8*>(xb + ((g * 3 + s3) * P + wave * 32 + j + t) * 16);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
                    afr[tm][s3] = *reinterpret_cast<const bf16x8*>(wl + (((t * 2 + g) * 3 + s3) * BM + tm * 32 + j) * 16);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][SA[q]], bfr[SB[q]], acc[tm], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int kh = lane >> 5;
    const int n = n0 + wave * 32 + j;
    if (n < N) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) out[(long)m * N + n] = acc[tm][r];
            }
    }
}

extern "C" int gemm_v2(const float* x, const uint16_t* wq, float* out, int T, int C, int M, int N, int NX, void* s, int mode) {
    constexpr int TM = 3;
    if (T > 3) return -2;
    const int P = 128 + T - 1, pitch = (P + 3) & ~3;
    const int w16 = T * 2 * 3 * TM * 32;
    const size_t lds = ((16 * pitch + 255) & ~255) * 4 + ((2 * 3 * P * 16 + 15) & ~15) + 2 * (size_t)((w16 + 63) & ~63) * 16;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x6_v2_kernel<TM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (lds > 160 * 1024) return -1;
    dim3 grid((N + 127) / 128, (M + TM * 32 - 1) / (TM * 32));
    hipLaunchKernelGGL(gemm_bf16x6_v2_kernel<TM>, grid, dim3(256), lds, (hipStream_t)s, x, wq, out, T, C, M, N, NX, mode);
    return (int)hipGetLastError();
}
