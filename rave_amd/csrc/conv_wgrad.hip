// Weight gradient of Conv1d / ConvTranspose1d / Conv2d-(k,1) as an MFMA GEMM whose reduction
// dimension is (batch, position):
//     dW[m][c][t] = sum_{b,n} actR(R[b][m][n]) * actS(S[b][c][ibase(n) + off[t]*inner])
// Conv1d:          R = dy (rows = c_out),        S = act(x) (c = c_in)
// ConvTranspose1d: R = act(x) (rows = c_in),     S = dy (c = c_out)
// The activation of the forward input is re-applied on the fly (the reference keeps a separate
// activated tensor alive for autograd).  Split-K over (batch, position chunks) with partials in
// caller-provided scratch and an ordered second pass -> bitwise deterministic.
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

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              long n, int Z) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int z = 0; z < Z; ++z) s += part[(long)z * n + e];
    out[e] = s;
}

// dbias[m] = sum_{b,n} dy[b][m][n]  (one block per channel, fixed reduction order)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db,
                                                        int B, int M, int row) {
    __shared__ float red[256];
    const int m = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* src = dy + ((long)b * M + m) * row;
        for (int e = threadIdx.x; e < row; e += 256) s += src[e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[m] = red[0];
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

void tile_of(int M, int* bm, int* bn) {
    if (M <= 32) { *bm = 32; *bn = 256; }
    else if (M <= 64) { *bm = 64; *bn = 128; }
    else if (M % 96 == 0 || M < 96) { *bm = 96; *bn = 128; }
    else { *bm = 128; *bn = 128; }
}

WPlan plan(const WgradP& p) {
    WPlan w{};
    tile_of(p.M, &w.bm, &w.bn);
    w.mt = rh_cdiv(p.M, w.bm);
    w.ct = rh_cdiv(p.C * p.T, w.bn);
    int rk = (64 / p.inner) & ~1;
    if (rk < 2) rk = 2;
    const int r_rows = p.r_row / p.inner;
    if (r_rows < rk) rk = (r_rows + 1) & ~1;  // short sequences: do not pad the K chunk with zeros
    if (rk < 2) rk = 2;
    w.rk = rk;
    w.chunks_per_b = rh_cdiv(r_rows, rk);
    w.total_chunks = p.B * w.chunks_per_b;
    int Z = 1024 / (w.mt * w.ct);
    const int zmax = w.total_chunks / 4;
    if (Z > zmax) Z = zmax;
    if (Z < 1) Z = 1;
    w.chunks_per_z = rh_cdiv(w.total_chunks, Z);
    w.Z = rh_cdiv(w.total_chunks, w.chunks_per_z);
    if (w.Z < 1) w.Z = 1;
    return w;
}

template <int TM, int TN, int WM, int WN>
int launch_w(WgradP& p, const WPlan& w, hipStream_t stream) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    p.rk = w.rk;
    p.chunks_per_b = w.chunks_per_b;
    p.total_chunks = w.total_chunks;
    p.chunks_per_z = w.chunks_per_z;
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
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    return rh_check_launch("conv1d_bwd_weight");
}

}  // namespace

int64_t rh_wgrad_workspace(const rh_conv1d_desc* d) {
    WgradP p{};
    fill(d, &p);
    const WPlan w = plan(p);
    if (w.Z <= 1) return 0;
    return (int64_t)w.Z * p.M * p.C * p.T * (int64_t)sizeof(float);
}

int rh_wgrad_run(const rh_conv1d_desc* d, const float* dy, const float* x, const float* alpha,
                 float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t stream) {
    WgradP p{};
    fill(d, &p);
    if (!d->transposed) { p.R = dy; p.S = x; p.r_alpha = nullptr; p.s_alpha = alpha; }
    else                { p.R = x; p.S = dy; p.r_alpha = alpha; p.s_alpha = nullptr; }
    const long nw = (long)p.M * p.C * p.T;
    if (dbias && d->batch > 0) {
        const int row = d->l_out * d->inner;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(d->c_out), dim3(256), 0, stream, dy, dbias, d->batch, d->c_out, row);
        if (int e = rh_check_launch("conv1d_bias_grad")) return e;
    }
    if (p.B <= 0 || p.r_row <= 0) {
        (void)hipMemsetAsync(dw, 0, nw * sizeof(float), stream);
        if (dbias) (void)hipMemsetAsync(dbias, 0, d->c_out * sizeof(float), stream);
        return RH_OK;
    }
    const WPlan w = plan(p);
    const int64_t need = w.Z > 1 ? (int64_t)w.Z * nw * (int64_t)sizeof(float) : 0;
    RH_REQUIRE(need == 0 || (ws && ws_bytes >= need), RH_ERR_WORKSPACE,
               "conv1d_bwd_weight: workspace %lld B < %lld B", (long long)ws_bytes, (long long)need);
    p.out = w.Z > 1 ? (float*)ws : dw;
    int e;
    if (w.bm == 32) e = launch_w<1, 2, 1, 4>(p, w, stream);
    else if (w.bm == 64) e = launch_w<2, 1, 1, 4>(p, w, stream);
    else if (w.bm == 96) e = launch_w<3, 1, 1, 4>(p, w, stream);
    else e = launch_w<2, 2, 2, 2>(p, w, stream);
    if (e) return e;
    if (w.Z > 1) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)rh_cdiv64(nw, 256)), dim3(256), 0, stream,
                           (const float*)ws, dw, nw, w.Z);
        return rh_check_launch("conv1d_bwd_weight_reduce");
    }
    return RH_OK;
}
