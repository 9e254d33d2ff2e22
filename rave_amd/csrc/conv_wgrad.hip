// Weight gradient of Conv1d / ConvTranspose1d / Conv2d-(k,1) as an MFMA GEMM whose reduction
// dimension is (batch, position):
//     dW[m][c][t] = sum_{b,n} actR(R[b][m][n]) * actS(S[b][c][ibase(n) + off[t]*inner])
// Conv1d:          R = dy (rows = c_out),        S = act(x) (c = c_in)
// ConvTranspose1d: R = act(x) (rows = c_in),     S = dy (c = c_out)
// The activation of the forward input is re-applied on the fly (the reference keeps a separate
// activated tensor alive for autograd).  Split-K over (batch, position chunks) with partials in
// caller-provided scratch and an ordered second pass -> bitwise deterministic.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"

namespace {

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_kernel(const WgradP p) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* r_lds = smem;                 // [BM][pr]
    float* s_lds = smem + BM * p.pr;     // [nc_max][ps]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = blockIdx.y * BM, col0 = blockIdx.x * BN, z = blockIdx.z;
    const int ncols = p.C * p.T;
    const int inner = p.inner, is = p.is;
    const int c_lo = col0 / p.T;

    int sb[TN], ar[TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int col = col0 + (wn * TN + tn) * 32 + j;
        col = min(col, ncols - 1);
        const int c = col / p.T, t = col - c * p.T;
        sb[tn] = (c - c_lo) * p.ps + (p.off[t] - p.minoff) * inner + kh * is * inner;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) ar[tm] = ((wm * TM + tm) * 32 + j) * p.pr + kh * inner;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int r_rows = p.r_row / inner;
    const int kelems = p.rk * inner;
    const int s_width = ((p.rk - 1) * is + (p.maxoff - p.minoff) + 1) * inner;
    const int ch0 = z * p.chunks_per_z;
    const int ch1 = min(ch0 + p.chunks_per_z, p.total_chunks);
    for (int ch = ch0; ch < ch1; ++ch) {
        const int b = ch / p.chunks_per_b;
        const int q0 = (ch - b * p.chunks_per_b) * p.rk;
        const int nk = min(p.rk, r_rows - q0) * inner;
        __syncthreads();
        for (int r = wave; r < BM; r += NW) {
            const int m = m0 + r;
            float* dst = r_lds + r * p.pr;
            if (m < p.M) {
                const float* __restrict__ src = p.R + ((long)b * p.M + m) * p.r_row + (long)q0 * inner;
                const float alpha = (p.r_act == RH_ACT_SNAKE) ? p.r_alpha[m] : 0.f;
                for (int e = lane; e < kelems; e += 64)
                    dst[e] = e < nk ? rh_act_apply(src[e], p.r_act, p.r_slope, alpha) : 0.f;
            } else {
                for (int e = lane; e < kelems; e += 64) dst[e] = 0.f;
            }
        }
        const long lo = ((long)q0 * is + p.minoff) * inner;
        for (int r = wave; r < p.nc_max; r += NW) {
            const int c = c_lo + r;
            float* dst = s_lds + r * p.ps;
            if (c < p.C) {
                const float* __restrict__ src = p.S + ((long)b * p.C + c) * p.s_row;
                const float alpha = (p.s_act == RH_ACT_SNAKE) ? p.s_alpha[c] : 0.f;
                for (int e = lane; e < s_width; e += 64) {
                    const long f = lo + e;
                    const float v = (f >= 0 && f < p.s_valid) ? src[f] : 0.f;
                    dst[e] = rh_act_apply(v, p.s_act, p.s_slope, alpha);
                }
            } else {
                for (int e = lane; e < s_width; e += 64) dst[e] = 0.f;
            }
        }
        __syncthreads();
        for (int w = 0; w < inner; ++w) {
            for (int rr = 0; rr < p.rk; rr += 2) {
                float a[TM], bb[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = r_lds[ar[tm] + rr * inner + w];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bb[tn] = s_lds[sb[tn] + rr * is * inner + w];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }
    float* out = p.out + (long)z * p.M * ncols;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = col0 + (wn * TN + tn) * 32 + j;
        if (col >= ncols) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) out[(long)m * ncols + col] = acc[tm][tn][r];
            }
    }
}

__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, float* __restrict__ out, long n, int Z, long block) {
    // A workgroup sums 64 consecutive elements; its four waves take the slices z = w, w + 4, w + 8, ... (eight independent
    // running sums each keep eight loads in flight per lane), then the four wave results are added in a fixed order: the
    // result is bitwise deterministic, and a small weight tensor with hundreds of K slices (the 96-channel layers:
    // 27 k elements x 256 slices) spreads over 4 x more workgroups and loads than one thread per element did.
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long e = block * 64 + lane;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int z = w;
        for (; z + 28 < Z; z += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(long)(z + 4 * u) * n + e];
        }
        for (; z < Z; z += 4) s[0] += part[(long)z * n + e];
    }
    red[w][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (w == 0 && e < n) out[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              long n, int Z) {
    reduce_partials_body(part, out, n, Z, blockIdx.x);
}

// Few slices of a large tensor (the wide layers: 1 ... 5 M weights x 2 ... 8 slices): one thread per FOUR consecutive elements,
// 16-byte loads, all slices of a thread in flight (four running sums, fixed order -> deterministic).  The kernel above spends
// a 256-thread workgroup on 64 elements, which is right for 27 k elements x 256 slices and wrong here (47 us for 75 MB).
__device__ __forceinline__ void reduce_partials_vec4_body(const float* __restrict__ part, float* __restrict__ out, long n4, int Z,
                                                          long block) {
    const long e = block * 256 + threadIdx.x;
    if (e >= n4) return;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(part) + e;
    f32x4 s[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int z = 0;
    for (; z + 3 < Z; z += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v = src[(long)(z + u) * n4];
            s[u] += v;
        }
    }
    for (; z < Z; ++z) s[0] += src[(long)z * n4];
    reinterpret_cast<f32x4*>(out)[e] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(256) void reduce_partials_vec4_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                   long n4, int Z) {
    reduce_partials_vec4_body(part, out, n4, Z, blockIdx.x);
}

// The reductions of SEVERAL layers in one launch (round 6: the v2 step ran 56 of them, one behind every weight-gradient kernel of
// the side stream -- 0.70 ms of 9 - 15 us launches): a table BY VALUE of up to kReduceBatch items, each reduced by the workgroups
// and in the summation order its own launch would have used (same bodies: the same bits).  rave_amd.ops collects the items of
// consecutive layers (rh_defer_reduce: the weight-gradient call leaves its K-slice partials in its own scratch) and flushes
// them every few layers on the side stream, and before the weight-norm backward / a data-parallel bucket needs the gradients.
constexpr int kReduceBatch = 32;
struct ReduceTable {
    const float* part[kReduceBatch];
    float* out[kReduceBatch];
    long n[kReduceBatch];              // elements (vec4 items: groups of four)
    long blk_begin[kReduceBatch + 1];  // prefix over the items' workgroups
    int Z[kReduceBatch];
    int vec4[kReduceBatch];
    int items;
};
__global__ __launch_bounds__(256) void reduce_partials_batched_kernel(const ReduceTable tb) {
    int it = 0;
    while (it + 1 < tb.items && (long)blockIdx.x >= tb.blk_begin[it + 1]) ++it;      // (uniform: scalar loop over <= 32 entries)
    const long block = (long)blockIdx.x - tb.blk_begin[it];
    if (tb.vec4[it]) reduce_partials_vec4_body(tb.part[it], tb.out[it], tb.n[it], tb.Z[it], block);
    else reduce_partials_body(tb.part[it], tb.out[it], tb.n[it], tb.Z[it], block);
}

static bool reduce_is_vec4(const float* part, const float* out, long n, int Z) {
    return Z <= 16 && n >= (1l << 16) && (n & 3) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0;
}

static void reduce_partials_go(const float* part, float* out, long n, int Z, hipStream_t stream) {
    if (reduce_is_vec4(part, out, n, Z)) {
        const long n4 = n / 4;
        hipLaunchKernelGGL(reduce_partials_vec4_kernel, dim3((unsigned)rh_cdiv64(n4, 256)), dim3(256), 0, stream, part, out, n4, Z);
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)rh_cdiv64(n, 64)), dim3(256), 0, stream, part, out, n, Z);
    }
}

// reduce_partials + weight_norm_bwd_kernel (misc.hip) as ONE launch per layer: a workgroup owns a complete dim-0 row of the
// weight tensor -- exactly the unit torch._weight_norm's backward (rave/blocks.py:15-22: weight_norm on every conv of the
// generator) needs: it sums the row's K-slice partials in the SAME fixed order as the two kernels above (order4 = the vec4
// kernel's: four running sums over z % 4; otherwise four waves on z = w, w + 4, ... with eight running sums each), keeps
// the summed row in LDS, forms <dw, v> in weight_norm_bwd_kernel's order (256 partial sums, shuffle tree, four waves) and
// writes dv = (g/||v||)(dw - v <dw,v>/||v||^2), dg = <dw,v>/||v||.  dw itself never exists in memory; bitwise the same
// dv / dg as the two-launch path.  16 waves: 4 z groups x 4 column groups of 64 lanes x 4 consecutive elements (a pass
// covers 1024 row elements; 96-row layers with 256 K slices keep 8 x 16 B x 64 lanes in flight per active wave).
constexpr int kRwnMaxN = 8192;
__global__ __launch_bounds__(1024) void reduce_wn_bwd_kernel(const float* __restrict__ part, long zstride, int Z, int N, int order4,
                                                             const float* __restrict__ v, const float* __restrict__ g,
                                                             const float* __restrict__ norms, float* __restrict__ dv,
                                                             float* __restrict__ dg) {
    extern __shared__ __attribute__((aligned(16))) float rwn_sm[];
    f32x4* const red = reinterpret_cast<f32x4*>(rwn_sm);            // [z group][column group][lane]
    float* const dwrow = rwn_sm + 4096;                             // [N]
    f32x4* const dwrow4 = reinterpret_cast<f32x4*>(dwrow);
    __shared__ float wred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r = blockIdx.x;
    const int N4 = N >> 2;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(part + r * N);
    const long zs4 = zstride >> 2;
    if (order4) {
        for (int e4 = tid; e4 < N4; e4 += 1024) {
            f32x4 s[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            int z = 0;
            for (; z + 3 < Z; z += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 t = src[(long)(z + u) * zs4 + e4];
                    s[u] += t;
                }
            }
            for (; z < Z; ++z) s[0] += src[(long)z * zs4 + e4];
            dwrow4[e4] = (s[0] + s[1]) + (s[2] + s[3]);
        }
    } else {
        const int cg = wave & 3, zg = wave >> 2;
        for (int c0 = 0; c0 < N4; c0 += 256) {
            const int e4 = c0 + cg * 64 + lane;
            f32x4 s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e4 < N4) {
                int z = zg;
                for (; z + 28 < Z; z += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const f32x4 t = src[(long)(z + 4 * u) * zs4 + e4];
                        s[u] += t;
                    }
                }
                for (; z < Z; z += 4) s[0] += src[(long)z * zs4 + e4];
            }
            red[(zg * 4 + cg) * 64 + lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
            __syncthreads();
            if (zg == 0 && e4 < N4)
                dwrow4[e4] = (red[cg * 64 + lane] + red[(4 + cg) * 64 + lane]) + (red[(8 + cg) * 64 + lane] + red[(12 + cg) * 64 + lane]);
            __syncthreads();
        }
    }
    __syncthreads();
    const float* __restrict__ vr = v + r * N;
    float s = 0.f;
    if (tid < 256)
        for (int e = tid; e < N; e += 256) s += dwrow[e] * vr[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (tid < 256 && lane == 0) wred[wave] = s;
    __syncthreads();
    const float dot = wred[0] + wred[1] + wred[2] + wred[3];
    const float norm = norms[r];
    const float scale = g[r] / norm;
    const float coef = dot / (norm * norm);
    float* __restrict__ dvr = dv + r * N;
    for (int e = tid; e < N; e += 1024) dvr[e] = scale * (dwrow[e] - vr[e] * coef);
    if (tid == 0) dg[r] = dot / norm;
}

// dbias[m] = sum_{b,h,w} dy * act'(y): grid (M, kBiasSlices) partial sums over interleaved 1024-element segments
// of the (b, plane) index space, then an ordered pass over the slices (deterministic).
constexpr int kBiasSlices = 64;

// g_out (optional): the same pass also WRITES g = dy * act'(y) -- the operand the data- and weight-gradient kernels of a
// conv with an output activation take -- so that one sweep replaces act_bwd_kernel (dy, y -> g) + this kernel (g -> sums).
// g_range (optional, with g_out): max |g| goes to that range slot (common.hpp: the scale of the f16 matrix-core kernels that read
// g next) -- one atomic per workgroup, M x 64 of them; saves the separate rh_amax_f32 pass over g (4.8 ms of a discrete-config step).
__global__ __launch_bounds__(256) void bias_grad2d_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          float* __restrict__ part, int B, int M, long plane, int act,
                                                          float slope, float* __restrict__ g_out, unsigned* __restrict__ g_range) {
    __shared__ float red[4];
    __shared__ float red_max[4];
    float mx = 0.f;
    const int m = blockIdx.x, sl = blockIdx.y;
    const long segs_per_b = (plane + 1023) / 1024;
    const long nseg = (long)B * segs_per_b;
    float s = 0.f;
    for (long sg = sl; sg < nseg; sg += kBiasSlices) {
        const long b = sg / segs_per_b, e0 = (sg - b * segs_per_b) * 1024;
        const long base = ((long)b * M + m) * plane;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long e = e0 + u * 256 + threadIdx.x;
            if (e < plane) {
                float v = dy[base + e];
                if (y) v *= rh_act_grad(y[base + e], act, slope, 0.f);
                if (g_out) g_out[base + e] = v;
                mx = fmaxf(mx, fabsf(v));
                s += v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)m * kBiasSlices + sl] = (red[0] + red[1]) + (red[2] + red[3]);
    if (g_range) rh_range_publish(g_range, mx, blockIdx.x + blockIdx.y * 7u, red_max);      // (uniform)
}

__global__ __launch_bounds__(64) void bias_grad2d_finalize_kernel(const float* __restrict__ part, float* __restrict__ db,
                                                                  int M) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    for (int i = 0; i < kBiasSlices; ++i) s += part[(long)m * kBiasSlices + i];
    db[m] = s;
}


typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;
// exact n / d for n < 2^32 / d with magic = ceil(2^32 / d) (d >= 2); magic == 0 encodes d == 1
__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

// Same GEMM as wgrad_kernel, software-pipelined with LDS-DMA: two LDS stages, the DMA engine
// fills stage (i+1)&1 with the next (batch, position-chunk) tiles of R and S while the matrix
// cores reduce stage i&1.  Tiles are copied as FLAT arrays (every DMA lane derives its own source
// element from its LDS position), so row pitches can be odd / tight and overlapping writes are
// always identical.  Zero padding comes from out-of-range buffer offsets; LeakyReLU of the saved
// forward input is applied when the operand is read from LDS.
template <int TM, int TN, int WM, int WN, bool RLEAKY, bool SLEAKY>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_dma_kernel(const WgradP p) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = blockIdx.y * BM, col0 = blockIdx.x * BN, z = blockIdx.z;
    const int ncols = p.C * p.T;
    const int inner = p.inner, is = p.is;
    const int c_lo = col0 / p.T;

    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);
    const auto s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.S), 0, p.s_bytes, 0x00020000);

    int sb[TN], ar[TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int col = col0 + (wn * TN + tn) * 32 + j;
        col = min(col, ncols - 1);
        const int c = col / p.T, t = col - c * p.T;
        sb[tn] = (c - c_lo) * p.ps + (p.off[t] - p.minoff) * inner + kh * is * inner;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) ar[tm] = ((wm * TM + tm) * 32 + j) * p.pr + kh * inner;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int r_rows = p.r_row / inner;
    const int s_width = ((p.rk - 1) * is + (p.maxoff - p.minoff) + 1) * inner;
    const int r_instrs = p.r_floats >> 6, s_instrs = p.s_floats >> 6;

    auto issue = [&](int ch, float* stage) {
        const unsigned b = mdiv((unsigned)ch, p.magic_cpb);
        const int q0 = (ch - (int)b * p.chunks_per_b) * p.rk;
        const int nk = min(p.rk, r_rows - q0) * inner;
        const unsigned rb = (b * p.M) * (unsigned)p.r_row + (unsigned)(q0 * inner);
        for (int q = wave; q < r_instrs; q += NW) {
            const unsigned pos = (unsigned)q * 64u + lane;
            const unsigned row = mdiv(pos, p.magic_pr);
            const unsigned e = pos - row * p.pr;
            const unsigned m = m0 + row;
            unsigned off = kOOB;
            if (row < (unsigned)BM && m < (unsigned)p.M && e < (unsigned)nk) off = (rb + m * (unsigned)p.r_row + e) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(stage + q * 64), 4, off, 0, 0, 0);
        }
        const int lo = (q0 * is + p.minoff) * inner;
        const unsigned sbase = (b * p.C) * (unsigned)p.s_row;
        float* ss = stage + p.r_floats;
        for (int q = wave; q < s_instrs; q += NW) {
            const unsigned pos = (unsigned)q * 64u + lane;
            const unsigned row = mdiv(pos, p.magic_ps);
            const unsigned e = pos - row * p.ps;
            const unsigned c = c_lo + row;
            const int f = lo + (int)e;
            unsigned off = kOOB;
            if (row < (unsigned)p.nc_max && c < (unsigned)p.C && e < (unsigned)s_width && f >= 0 && f < p.s_valid)
                off = (sbase + c * (unsigned)p.s_row + (unsigned)f) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void*)(ss + q * 64), 4, off, 0, 0, 0);
        }
    };

    // ---- fast path: 16-byte DMA with per-lane source offsets precomputed once ------------------
    // Each lane owns up to kNI (row, float4) slots of the R tile and of the S tile; per chunk only a
    // scalar base is added.  Used for chunks whose whole halo lies inside the sequence (no zero
    // padding inside a float4); boundary chunks take the element-wise path above.
    constexpr int kNI = 12;
    unsigned rc[kNI], sc[kNI];
    int ni_r = 0, ni_s = 0;
    if (p.vec) {
        const int lpr_r = p.pr >> 2, lpr_s = p.ps >> 2;
        const int data_r = p.rk >> 2, data_s = (s_width + 3) >> 2;
        ni_r = (BM * lpr_r + 64 * NW - 1) / (64 * NW);
        ni_s = (p.nc_max * lpr_s + 64 * NW - 1) / (64 * NW);
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            const unsigned g = (unsigned)(wave + NW * i) * 64u + lane;
            const unsigned rrow = mdiv(g, p.magic_lpr_r);
            const unsigned rv = g - rrow * lpr_r;
            rc[i] = (rrow < (unsigned)BM && m0 + rrow < (unsigned)p.M && rv < (unsigned)data_r)
                        ? (m0 + rrow) * (unsigned)p.r_row + 4u * rv : kOOB;
            const unsigned srow = mdiv(g, p.magic_lpr_s);
            const unsigned sv = g - srow * lpr_s;
            sc[i] = (srow < (unsigned)p.nc_max && c_lo + srow < (unsigned)p.C && sv < (unsigned)data_s)
                        ? (c_lo + srow) * (unsigned)p.s_row + 4u * sv : kOOB;
        }
    }
    auto issue_any = [&](int ch, float* stage) {
        if (!p.vec) { issue(ch, stage); return; }
        const unsigned b = mdiv((unsigned)ch, p.magic_cpb);
        const int q0 = (ch - (int)b * p.chunks_per_b) * p.rk;
        const int lo = q0 * is + p.minoff;
        const bool interior = (q0 + p.rk <= r_rows) && lo >= 0 && lo + 4 * ((s_width + 3) >> 2) <= p.s_valid;
        if (!interior) { issue(ch, stage); return; }
        const unsigned rbase = (b * p.M) * (unsigned)p.r_row + (unsigned)q0;
        const unsigned sbase = (b * p.C) * (unsigned)p.s_row + (unsigned)lo;
        float* ss = stage + p.r_floats;
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            if (i < ni_r && (wave + NW * i) * 256 < p.r_floats) {   // never write past the tile's LDS region
                const unsigned off = rc[i] == kOOB ? kOOB : (rbase + rc[i]) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(stage + (wave + NW * i) * 256), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            if (i < ni_s && (wave + NW * i) * 256 < p.s_floats) {
                const unsigned off = sc[i] == kOOB ? kOOB : (sbase + sc[i]) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void*)(ss + (wave + NW * i) * 256), 16, off, 0, 0, 0);
            }
        }
    };

    const int ch0 = z * p.chunks_per_z;
    const int nch = min(p.chunks_per_z, p.total_chunks - ch0);
    if (nch > 0) issue_any(ch0, smem);
    for (int i = 0; i < nch; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (i + 1 < nch) issue_any(ch0 + i + 1, smem + ((i + 1) & 1) * p.stage_floats);
        const float* r_lds = smem + (i & 1) * p.stage_floats;
        const float* s_lds = r_lds + p.r_floats;
        for (int w = 0; w < inner; ++w) {
            const float* rl = r_lds + w;
            const float* sl = s_lds + w;
            int rr = 0;
            if (p.vec) {
                // 16-byte aligned rows (pitch = 4 mod 32 floats): one conflict-free ds_read_b128 per row
                // tile delivers the A operand of FOUR k-steps.  K pairing inside a group of 8 positions:
                // step u multiplies positions (rr+u) [lanes 0-31] and (rr+4+u) [lanes 32-63], so every
                // lane consumes its whole float4 and no component select is needed (A and B agree).
                for (; rr + 8 <= p.rk; rr += 8) {
                    f32x4 a4[TM];
                    float bb[4][TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        a4[tm] = *reinterpret_cast<const f32x4*>(rl + ar[tm] - kh + rr + 4 * kh);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) bb[u][tn] = sl[sb[tn] - kh * is + (rr + 4 * kh + u) * is];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            if (SLEAKY) bb[u][tn] = bb[u][tn] > 0.f ? bb[u][tn] : bb[u][tn] * p.s_slope;
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            float av = a4[tm][u];
                            if (RLEAKY) av = av > 0.f ? av : av * p.r_slope;
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[u][tn], acc[tm][tn], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; rr + 8 <= p.rk; rr += 8) {
                // all 4 x (TM + TN) LDS reads of the group are issued back to back (sched_barrier keeps
                // hipcc from re-serialising them into read -> wait -> MFMA), then the 4 x TM x TN MFMAs
                float a[4][TM], bb[4][TN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) a[u][tm] = rl[ar[tm] + (rr + 2 * u) * inner];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bb[u][tn] = sl[sb[tn] + (rr + 2 * u) * is * inner];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        if (RLEAKY) a[u][tm] = a[u][tm] > 0.f ? a[u][tm] : a[u][tm] * p.r_slope;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        if (SLEAKY) bb[u][tn] = bb[u][tn] > 0.f ? bb[u][tn] : bb[u][tn] * p.s_slope;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][tm], bb[u][tn], acc[tm][tn], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; rr < p.rk; rr += 2) {
                float a[TM], bb[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float v = rl[ar[tm] + rr * inner];
                    if (RLEAKY) v = v > 0.f ? v : v * p.r_slope;
                    a[tm] = v;
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float v = sl[sb[tn] + rr * is * inner];
                    if (SLEAKY) v = v > 0.f ? v : v * p.s_slope;
                    bb[tn] = v;
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }
    float* out = p.out + (long)z * p.M * ncols;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = col0 + (wn * TN + tn) * 32 + j;
        if (col >= ncols) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) out[(long)m * ncols + col] = acc[tm][tn][r];
            }
    }
}

struct WPlan {
    int bm, bn, mt, ct, Z, rk, chunks_per_b, total_chunks, chunks_per_z;
};

void fill(const rh_conv1d_desc* d, WgradP* p) {
    const int k = d->kernel;
    p->B = d->batch;
    p->T = k;
    p->inner = d->inner;
    p->is = d->stride;
    const int x_valid = d->in_valid ? d->in_valid : d->l_in * d->inner;
    if (!d->transposed) {
        p->M = d->c_out; p->C = d->c_in;
        p->r_row = d->l_out * d->inner;
        p->s_row = x_valid; p->s_valid = x_valid;
        p->r_act = RH_ACT_NONE; p->s_act = d->act;
        p->r_slope = 0.f; p->s_slope = d->act_slope;
        for (int t = 0; t < k; ++t) p->off[t] = t * d->dilation - d->pad_left;
    } else {
        p->M = d->c_in; p->C = d->c_out;
        p->r_row = d->l_in;
        p->s_row = d->l_out; p->s_valid = d->l_out;
        p->r_act = d->act; p->s_act = RH_ACT_NONE;
        p->r_slope = d->act_slope; p->s_slope = 0.f;
        for (int t = 0; t < k; ++t) p->off[t] = t - d->pad_left;
    }
    int lo = p->off[0], hi = p->off[0];
    for (int t = 1; t < k; ++t) { lo = lo < p->off[t] ? lo : p->off[t]; hi = hi > p->off[t] ? hi : p->off[t]; }
    p->minoff = lo; p->maxoff = hi;
}

void tile_of(int M, int ncols, int* bm, int* bn) {
    if (M <= 32) { *bm = 32; *bn = 256; }
    else if (M <= 64) { *bm = 64; *bn = 128; }
    else if (M % 96 == 0 || M < 96) {
        *bm = 96;
        // C*T is a multiple of 96 for every 96*2^n-channel layer of v2: 96-wide column tiles (3 waves)
        // leave no ragged last tile (288 = 3 x 96, not 2 x 128 + 32)
        *bn = (ncols % 96 == 0 && ncols % 128 != 0) ? 96 : 128;
    }
    else { *bm = 128; *bn = 128; }
}

WPlan plan(const WgradP& p) {
    WPlan w{};
    tile_of(p.M, p.C * p.T, &w.bm, &w.bn);
    w.mt = rh_cdiv(p.M, w.bm);
    w.ct = rh_cdiv(p.C * p.T, w.bn);
    static const int rk_env = [] { const char* e = getenv("RH_WGRAD_RK"); return e ? atoi(e) : 0; }();
    // reduction chunk: 32 positions (3 workgroups per CU) measured best, except pointwise convs
    // whose S tile has one row per column (64 keeps the DMA instructions full)
    static const int rk_inner_env = [] { const char* e = getenv("RH_WGRAD_RK_INNER"); return e ? atoi(e) : 0; }();
    int rk = ((rk_env > 0 ? rk_env : 32) / p.inner) & ~1;
    if (p.inner > 1) rk = ((rk_inner_env > 0 ? rk_inner_env : 32) / p.inner) & ~1;
    if (rk < 2) rk = 2;
    const int r_rows = p.r_row / p.inner;
    if (r_rows < rk) rk = (r_rows + 1) & ~1;  // short sequences: do not pad the K chunk with zeros
    if (rk < 2) rk = 2;
    w.rk = rk;
    w.chunks_per_b = rh_cdiv(r_rows, rk);
    w.total_chunks = p.B * w.chunks_per_b;
    static const int ztarget = [] { const char* e = getenv("RH_WGRAD_BLOCKS"); return e ? atoi(e) : 1024; }();
    static const int zmin_chunks = [] { const char* e = getenv("RH_WGRAD_MINCHUNKS"); return e ? atoi(e) : 16; }();
    int Z = ztarget / (w.mt * w.ct);
    const int zmax = w.total_chunks / (rk * p.inner >= 64 ? zmin_chunks / 2 : zmin_chunks);   // positions per K slice
    if (Z > zmax) Z = zmax;
    if (Z < 1) Z = 1;
    w.chunks_per_z = rh_cdiv(w.total_chunks, Z);
    w.Z = rh_cdiv(w.total_chunks, w.chunks_per_z);
    if (w.Z < 1) w.Z = 1;
    return w;
}

inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / (unsigned long long)d); }

template <int TM, int TN, int WM, int WN>
int launch_w(WgradP& p, const WPlan& w, hipStream_t stream) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    p.rk = w.rk;
    p.chunks_per_b = w.chunks_per_b;
    p.total_chunks = w.total_chunks;
    p.chunks_per_z = w.chunks_per_z;
    // ---- LDS-DMA pipelined path (LeakyReLU / no activation, tensors < 2 GiB, K chunk <= 64) ----
    {
        const unsigned long long rb = 4ull * p.B * p.M * (unsigned long long)p.r_row;
        const unsigned long long sbytes = 4ull * p.B * p.C * (unsigned long long)p.s_row;
        const bool ok = p.r_act != RH_ACT_SNAKE && p.s_act != RH_ACT_SNAKE && rb < 0x7fffffffull &&
                        sbytes < 0x7fffffffull && w.rk * p.inner <= 64 && w.total_chunks < 32768;
        if (ok) {
            p.nc_max = (BN - 1) / p.T + 2;
            if (p.nc_max > p.C) p.nc_max = p.C;
            const int s_width = ((w.rk - 1) * p.is + (p.maxoff - p.minoff) + 1) * p.inner;
            static const int novec = [] { const char* e = getenv("RH_WGRAD_NOVEC"); return e ? atoi(e) : 0; }();
            // 16-byte DMA needs 16-byte aligned rows in LDS (pitches multiple of 4 floats) and in HBM
            p.vec = !novec && p.inner == 1 && (w.rk % 4) == 0 && (p.r_row % 4) == 0 && (p.s_row % 4) == 0 &&
                    ((uintptr_t)p.R % 16) == 0 && ((uintptr_t)p.S % 16) == 0;
            if (p.vec) {
                p.pr = w.rk + 4;                       // one pad float4 per row: row stride = 4 (mod 32) banks
                p.ps = ((s_width + 3) & ~3) + 4;
                if ((p.ps & 31) == 0) p.ps += 4;
                const int ni_r = rh_cdiv(BM * (p.pr >> 2), 64 * WM * WN);
                const int ni_s = rh_cdiv(p.nc_max * (p.ps >> 2), 64 * WM * WN);
                if (ni_r > 12 || ni_s > 12) p.vec = 0;
            }
            if (!p.vec) {
                p.ps = s_width | 1;
                p.pr = (w.rk * p.inner) | 1;
            }
            p.magic_pr = magic_of(p.pr);
            p.magic_lpr_r = magic_of(p.pr >> 2);
            p.magic_lpr_s = magic_of(p.ps >> 2);
            p.r_floats = (BM * p.pr + 255) & ~255;
            p.s_floats = (p.nc_max * p.ps + 255) & ~255;
            p.stage_floats = p.r_floats + p.s_floats;
            p.magic_ps = magic_of(p.ps);
            p.magic_cpb = magic_of(w.chunks_per_b);
            p.r_bytes = (unsigned)rb;
            p.s_bytes = (unsigned)sbytes;
            const size_t lds = sizeof(float) * 2 * (size_t)p.stage_floats;
            if (lds <= 160 * 1024 && (unsigned)p.s_floats < (1u << 15)) {
                dim3 grid(w.ct, w.mt, w.Z);
                auto go = [&](auto kern) {
                    static std::once_flag once;
                    std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
                    rh_launch_main(kern, grid, dim3(WM * WN * 64), lds, stream, p);
                };
                if (p.r_act == RH_ACT_LEAKY) go(wgrad_dma_kernel<TM, TN, WM, WN, true, false>);
                else if (p.s_act == RH_ACT_LEAKY) go(wgrad_dma_kernel<TM, TN, WM, WN, false, true>);
                else go(wgrad_dma_kernel<TM, TN, WM, WN, false, false>);
                return rh_check_launch("conv1d_bwd_weight(dma)");
            }
        }
    }
    p.pr = (w.rk * p.inner) | 1;
    p.ps = (((w.rk - 1) * p.is + (p.maxoff - p.minoff) + 1) * p.inner) | 1;
    p.nc_max = (BN - 1) / p.T + 2;
    if (p.nc_max > p.C) p.nc_max = p.C;
    const size_t lds = sizeof(float) * ((size_t)BM * p.pr + (size_t)p.nc_max * p.ps);
    RH_REQUIRE(lds <= 160 * 1024, RH_ERR_UNSUPPORTED, "conv1d_bwd_weight: tile needs %zu B of LDS", lds);
    auto kern = wgrad_kernel<TM, TN, WM, WN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    dim3 grid(w.ct, w.mt, w.Z);
    rh_launch_main(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    return rh_check_launch("conv1d_bwd_weight");
}

}  // namespace

int64_t rh_wgrad_x6_workspace(const WgradP& w);
int rh_wgrad_x6_launch(const WgradP& w, float* dw, float* rsum_out, void* ws, hipStream_t stream, bool* used, int* left_z,
                       const unsigned* r_range, const unsigned* s_range);
int rh_reduce_wn_bwd_launch(const float* part, int Z, long M, long N, const float* v, const float* g, const float* norms,
                            float* dv, float* dg, hipStream_t stream, bool* used);

extern "C" int rh_conv1d_bwd_weight_kernel_family(const rh_conv1d_desc* d) {
    if (!d || d->batch <= 0) return 0;
    WgradP p{};
    fill(d, &p);
    if (rh_smallc_wgrad_eligible(d)) return 2;
    alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
    p.R = dummy; p.S = dummy;
    return d->act != RH_ACT_SNAKE && rh_wgrad_x6_workspace(p) >= 0 ? 1 : 0;
}

int64_t rh_wgrad_workspace(const rh_conv1d_desc* d) {
    WgradP p{};
    fill(d, &p);
    const int64_t bias = rh_bias_grad_workspace(d->c_out);
    if (rh_smallc_wgrad_eligible(d)) return bias + rh_smallc_wgrad_workspace(d);
    int64_t x6 = -1;
    if (d->act != RH_ACT_SNAKE) {      // exact f32 on the 16-bit matrix cores (conv_wgrad_x6.hip) when the geometry fits
        x6 = rh_wgrad_x6_workspace(p);
        if (x6 >= 0 && !RH_X6_F16) return bias + x6;
    }
    // (f16 build: a call without range slots falls back to the f32-input kernels -- the scratch covers both plans)
    const WPlan w = plan(p);
    const int64_t f32 = w.Z <= 1 ? 0 : (int64_t)w.Z * p.M * p.C * p.T * (int64_t)sizeof(float);
    return bias + (x6 > f32 ? x6 : f32);
}

extern "C" int rh_weight_norm_bwd_f32(const float* dw, const float* v, const float* g, const float* norms, int64_t rows,
                                      int64_t cols, float* dv, float* dg, rh_stream_t stream);

// tail != null: the weight is weight-normed (w = g v/||v||, dim 0 = the rows of dw in both conv directions) and the caller
// wants dv, dg instead of dw: straight from the K-slice partials where the row fits (reduce_wn_bwd_kernel), else through dw.
// rh_defer_reduce(item): the NEXT weight-gradient call of this thread may leave its K-slice partials unreduced in its scratch
// and describe the pending reduction in *item (Z > 1) instead of launching it; item->Z = 0 when nothing is pending (the call
// completed dw itself).  Consumed by that call.  The caller keeps scratch and dw alive and runs rh_reduce_partials_batched_f32.
thread_local rh_reduce_item* g_defer_item = nullptr;
extern "C" int rh_defer_reduce(rh_reduce_item* item) {
    g_defer_item = item;
    if (item) *item = rh_reduce_item{};
    return RH_OK;
}

extern "C" int rh_reduce_partials_batched_f32(const rh_reduce_item* items, int32_t n_items, rh_stream_t stream) {
    RH_REQUIRE(n_items >= 0 && (n_items == 0 || items), RH_ERR_INVALID, "reduce_partials_batched: bad arguments");
    for (int i0 = 0; i0 < n_items; i0 += kReduceBatch) {
        ReduceTable tb{};
        long blocks = 0;
        for (int i = i0; i < n_items && i < i0 + kReduceBatch; ++i) {
            const rh_reduce_item& q = items[i];
            if (q.Z <= 0 || q.n <= 0) continue;           // nothing pending for this layer
            RH_REQUIRE(q.part && q.out, RH_ERR_INVALID, "reduce_partials_batched: null pointer in item %d", i);
            const int k = tb.items++;
            tb.part[k] = q.part; tb.out[k] = q.out; tb.Z[k] = q.Z;
            tb.vec4[k] = reduce_is_vec4(q.part, q.out, q.n, q.Z) ? 1 : 0;
            tb.n[k] = tb.vec4[k] ? q.n / 4 : q.n;
            tb.blk_begin[k] = blocks;
            blocks += tb.vec4[k] ? rh_cdiv64(tb.n[k], 256) : rh_cdiv64(q.n, 64);
        }
        if (tb.items == 0) continue;
        tb.blk_begin[tb.items] = blocks;
        RH_REQUIRE(blocks < 0x7fffffffl, RH_ERR_UNSUPPORTED, "reduce_partials_batched: too many workgroups");
        hipLaunchKernelGGL(reduce_partials_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb);
        if (int e = rh_check_launch("reduce_partials_batched")) return e;
    }
    return RH_OK;
}

int rh_wgrad_run(const rh_conv1d_desc* d, const float* dy, const float* x, const float* alpha,
                 float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t stream, const RhWnTail* tail) {
    rh_reduce_item* const defer = tail ? nullptr : g_defer_item;       // (consumed by this call whatever happens below)
    g_defer_item = nullptr;
    WgradP p{};
    fill(d, &p);
    const unsigned *dy_range = nullptr, *x_range = nullptr;
    rh_take_ranges(&dy_range, &x_range, nullptr, nullptr);
    const bool have_ranges = !RH_X6_F16 || (dy_range && x_range);
    auto through_dw = [&](int e) {
        if (e || !tail) return e;
        return rh_weight_norm_bwd_f32(dw, tail->v, tail->g, tail->norms, p.M, (int64_t)p.C * p.T, tail->dv, tail->dg, (rh_stream_t)stream);
    };
    if (!d->transposed) { p.R = dy; p.S = x; p.r_alpha = nullptr; p.s_alpha = alpha; }
    else                { p.R = x; p.S = dy; p.r_alpha = alpha; p.s_alpha = nullptr; }
    const long nw = (long)p.M * p.C * p.T;
    const int64_t bias_ws = rh_bias_grad_workspace(d->c_out);
    if (rh_smallc_wgrad_eligible(d)) {       // 1- / 2-channel first layers: vector-ALU kernel, bias gradient included
        const int64_t need = rh_smallc_wgrad_workspace(d);
        RH_REQUIRE(ws && ws_bytes >= need, RH_ERR_WORKSPACE, "conv1d_bwd_weight: workspace %lld B < %lld B",
                   (long long)ws_bytes, (long long)need);
        return through_dw(rh_smallc_wgrad(d, dy, x, dw, dbias, ws, stream));
    }
    // Conv1d on the bf16x6 weight-gradient kernel: the bias gradient (row sums of dy) comes out of the same pass
    const bool x6_path = have_ranges && d->act != RH_ACT_SNAKE && p.B > 0 && p.r_row > 0 && rh_wgrad_x6_workspace(p) >= 0;
    const bool fuse_bias = x6_path && !d->transposed && dbias != nullptr;
    if (dbias && d->batch > 0 && !fuse_bias) {
        RH_REQUIRE(ws && ws_bytes >= bias_ws, RH_ERR_WORKSPACE, "conv1d_bwd_weight: workspace %lld B < %lld B",
                   (long long)ws_bytes, (long long)bias_ws);
        // the first c_out*64 floats of the workspace; the split-K partials follow
        if (int e = rh_bias_grad_launch(dy, nullptr, (float*)ws, dbias, d->batch, d->c_out,
                                        (long)d->l_out * d->inner, RH_ACT_NONE, 0.f, stream))
            return e;
    }
    ws = ws ? (void*)((char*)ws + bias_ws) : nullptr;
    ws_bytes = ws_bytes > bias_ws ? ws_bytes - bias_ws : 0;
    if (p.B <= 0 || p.r_row <= 0) {
        (void)hipMemsetAsync(dw, 0, nw * sizeof(float), stream);
        if (dbias) (void)hipMemsetAsync(dbias, 0, d->c_out * sizeof(float), stream);
        return through_dw(RH_OK);
    }
    if (d->act != RH_ACT_SNAKE && have_ranges) {
        const int64_t x6 = rh_wgrad_x6_workspace(p);
        if (x6 >= 0) {
            RH_REQUIRE(x6 == 0 || (ws && ws_bytes >= x6), RH_ERR_WORKSPACE,
                       "conv1d_bwd_weight: workspace %lld B < %lld B", (long long)ws_bytes, (long long)x6);
            bool used = false;
            int left_z = 0;      // > 0: the partials were left unreduced in ws for the fused tail
            if (int e = rh_wgrad_x6_launch(p, dw, fuse_bias ? dbias : nullptr, ws, stream, &used, (tail || defer) ? &left_z : nullptr,
                                           d->transposed ? x_range : dy_range, d->transposed ? dy_range : x_range))
                return e;
            if (used && left_z > 1 && defer) {        // the caller batches this layer's reduction with its neighbours'
                defer->part = (const float*)ws; defer->out = dw; defer->n = nw; defer->Z = left_z;
                return RH_OK;
            }
            if (used && left_z > 1) {
                bool fused = false;
                if (int e = rh_reduce_wn_bwd_launch((const float*)ws, left_z, p.M, (long)p.C * p.T, tail->v, tail->g, tail->norms,
                                                    tail->dv, tail->dg, stream, &fused))
                    return e;
                if (fused) return RH_OK;
                return through_dw(rh_reduce_partials_launch((const float*)ws, dw, nw, left_z, stream, "conv1d_bwd_weight_reduce"));
            }
            if (used) return through_dw(RH_OK);
        }
    }
    const WPlan w = plan(p);
    const int64_t need = w.Z > 1 ? (int64_t)w.Z * nw * (int64_t)sizeof(float) : 0;
    RH_REQUIRE(need == 0 || (ws && ws_bytes >= need), RH_ERR_WORKSPACE,
               "conv1d_bwd_weight: workspace %lld B < %lld B", (long long)ws_bytes, (long long)need);
    p.out = w.Z > 1 ? (float*)ws : dw;
    int e;
    if (w.bm == 32) e = launch_w<1, 2, 1, 4>(p, w, stream);
    else if (w.bm == 64) e = launch_w<2, 1, 1, 4>(p, w, stream);
    else if (w.bm == 96 && w.bn == 96) e = launch_w<3, 1, 1, 3>(p, w, stream);
    else if (w.bm == 96) e = launch_w<3, 1, 1, 4>(p, w, stream);
    else e = launch_w<2, 2, 2, 2>(p, w, stream);
    if (e) return e;
    if (w.Z > 1) {
        if (tail) {
            bool fused = false;
            if (int e2 = rh_reduce_wn_bwd_launch((const float*)ws, w.Z, p.M, nw / p.M, tail->v, tail->g, tail->norms, tail->dv,
                                                 tail->dg, stream, &fused))
                return e2;
            if (fused) return RH_OK;
        }
        if (defer) {
            defer->part = (const float*)ws; defer->out = dw; defer->n = nw; defer->Z = w.Z;
            return RH_OK;
        }
        reduce_partials_go((const float*)ws, dw, nw, w.Z, stream);
        return through_dw(rh_check_launch("conv1d_bwd_weight_reduce"));
    }
    return through_dw(RH_OK);
}

int rh_reduce_partials_launch(const float* part, float* out, long n, int Z, hipStream_t stream, const char* what) {
    if (n <= 0) return RH_OK;
    reduce_partials_go(part, out, n, Z, stream);
    return rh_check_launch(what);
}

// K-slice partials [Z][M][N] -> dv, dg of the weight-normed parameter in one launch (reduce_wn_bwd_kernel).  *used = false
// (nothing launched) when the row does not fit: the caller reduces into dw and runs weight_norm_bwd.
static std::atomic<int64_t> g_rwn_launches{0};       // diagnostics: how often the one-launch form ran (tests)
int rh_reduce_wn_bwd_launch(const float* part, int Z, long M, long N, const float* v, const float* g, const float* norms,
                            float* dv, float* dg, hipStream_t stream, bool* used) {
    *used = false;
    // OPT-IN (RH_WN_FUSED=1; read per call: tests).  Measured on the v2 step (batch 32, graph replay, same box, two runs
    // each): 10.29 / 10.27 ms with this launch against 9.99 / 10.02 ms with reduce_partials + weight_norm_bwd -- one
    // workgroup per row means 96 ... 768 workgroups pull the 2.5 GB of partials of a step where reduce_partials_kernel
    // spreads them over 4 x as many, and a 16-wave workgroup waits for a whole free CU beside the conv kernels of the other
    // stream (profiles/round4_negative_fused_reduce_weight_norm.txt).  Kept for the bit-identity test and as evidence.
    const char* e = getenv("RH_WN_FUSED");
    if (!(e && atoi(e) == 1)) return RH_OK;
    if (Z < 2 || M <= 0 || N <= 0 || (N & 3) || N > kRwnMaxN || ((uintptr_t)part & 15)) return RH_OK;
    const long n = M * N;
    // (the order reduce_partials_go would have taken for this tensor; its dw scratch is always 16-byte aligned)
    const int order4 = Z <= 16 && n >= (1l << 16);
    hipLaunchKernelGGL(reduce_wn_bwd_kernel, dim3((unsigned)M), dim3(1024), (size_t)(4096 + N) * sizeof(float), stream, part, n, Z,
                       (int)N, order4, v, g, norms, dv, dg);
    *used = true;
    ++g_rwn_launches;
    return rh_check_launch("conv1d_bwd_weight_reduce_wn");
}

extern "C" int64_t rh_conv1d_bwd_weight_wn_fused_launches(void) { return g_rwn_launches.load(); }

int64_t rh_bias_grad_workspace(int M) { return (int64_t)M * kBiasSlices * (int64_t)sizeof(float); }

int rh_bias_grad_launch(const float* dy, const float* y, float* part, float* db, int B, int M, long plane, int act,
                        float slope, hipStream_t stream, float* g_out, unsigned* g_range) {
    hipLaunchKernelGGL(bias_grad2d_kernel, dim3(M, kBiasSlices), dim3(256), 0, stream, dy, y, part, B, M, plane, act, slope, g_out,
                       g_out ? g_range : nullptr);
    if (int e = rh_check_launch("bias_grad")) return e;
    hipLaunchKernelGGL(bias_grad2d_finalize_kernel, dim3(rh_cdiv(M, 64)), dim3(64), 0, stream, (const float*)part, db, M);
    return rh_check_launch("bias_grad_finalize");
}

// g = dy * act'(y) AND dbias[m] = sum_{b, plane} g in one pass over (B, M, plane) tensors (ordered partials: deterministic):
// the backward prologue of a conv whose LeakyReLU sits on its OUTPUT (rave/discriminator.py:50,
// rave/descript_discriminator.py:27).  workspace: rh_act_bwd_bias_workspace_bytes(M).
extern "C" int64_t rh_act_bwd_bias_workspace_bytes(int32_t M) { return rh_bias_grad_workspace(M); }
extern "C" int rh_act_bwd_bias_f32(const float* dy, const float* y, int32_t act, float slope, int32_t B, int32_t M, int64_t plane,
                                   float* g, float* dbias, void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    unsigned* g_range = nullptr;                       // where max |g| goes, if the caller armed an output slot (rh_x6_set_ranges)
    rh_take_ranges(nullptr, nullptr, &g_range, nullptr);
    RH_REQUIRE(dy && y && g && dbias && B >= 0 && M > 0 && plane > 0, RH_ERR_INVALID, "act_bwd_bias: bad arguments");
    RH_REQUIRE(act == RH_ACT_NONE || act == RH_ACT_LEAKY, RH_ERR_UNSUPPORTED, "act_bwd_bias: activation must be none or leaky");
    RH_REQUIRE(workspace && workspace_bytes >= rh_bias_grad_workspace(M), RH_ERR_WORKSPACE, "act_bwd_bias: workspace too small");
    if (B == 0) return hipMemsetAsync(dbias, 0, (size_t)M * sizeof(float), (hipStream_t)stream) == hipSuccess ? RH_OK : RH_ERR_INVALID;
    return rh_bias_grad_launch(dy, y, (float*)workspace, dbias, B, M, plane, act, slope, (hipStream_t)stream, g, g_range);
}
