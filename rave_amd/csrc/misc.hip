// Small memory-bound kernels: weight normalisation, output head, activation, avg-pool.
#include "common.hpp"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    // 256 threads; fixed-order tree -> deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// w[r,:] = g[r] * v[r,:] / ||v[r,:]||   (torch._weight_norm(v, g, 0); rave/blocks.py:15-22)
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              long cols, float* __restrict__ w, float* __restrict__ norms) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const float* vr = v + r * cols;
    float s = 0.f;
    for (long e = threadIdx.x; e < cols; e += 256) { const float a = vr[e]; s += a * a; }
    const float norm = sqrtf(block_sum(s, red));
    const float scale = g[r] / norm;
    float* wr = w + r * cols;
    for (long e = threadIdx.x; e < cols; e += 256) wr[e] = vr[e] * scale;
    if (threadIdx.x == 0 && norms) norms[r] = norm;
}

// norms[r] = ||v[r,:]||, scale[r] = g[r] / norms[r]  (weight norm folded into the weight repack)
__global__ __launch_bounds__(256) void weight_norm_scales_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                 long cols, float* __restrict__ norms, float* __restrict__ scale) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const float* vr = v + r * cols;
    float s = 0.f;
    for (long e = threadIdx.x; e < cols; e += 256) { const float a = vr[e]; s += a * a; }
    const float norm = sqrtf(block_sum(s, red));
    if (threadIdx.x == 0) { norms[r] = norm; scale[r] = g[r] / norm; }
}

// dg[r] = <dw, v> / ||v|| ;  dv = (g/||v||) * (dw - v * <dw, v> / ||v||^2)
__device__ __forceinline__ void weight_norm_bwd_row(const float* __restrict__ dw, const float* __restrict__ v,
                                                    const float* __restrict__ g, const float* __restrict__ norms, long r,
                                                    long cols, float* __restrict__ dv, float* __restrict__ dg, float* red) {
    const float* vr = v + r * cols;
    const float* dwr = dw + r * cols;
    float s = 0.f;
    for (long e = threadIdx.x; e < cols; e += 256) s += dwr[e] * vr[e];
    const float dot = block_sum(s, red);
    const float norm = norms[r];
    const float scale = g[r] / norm;
    const float coef = dot / (norm * norm);
    float* dvr = dv + r * cols;
    for (long e = threadIdx.x; e < cols; e += 256) dvr[e] = scale * (dwr[e] - vr[e] * coef);
    if (threadIdx.x == 0) dg[r] = dot / norm;
}

__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ norms,
                                                              long cols, float* __restrict__ dv, float* __restrict__ dg) {
    __shared__ float red[4];
    weight_norm_bwd_row(dw, v, g, norms, blockIdx.x, cols, dv, dg, red);
}

// The same row program for MANY weight tensors in one launch: the backward pass of a v2 step runs 56 of these 4-6 us
// latency-bound launches (one per weight-normed conv, 0.35 ms) whose only consumer is the optimizer; collected and run
// once per backward pass / gradient bucket they are one bandwidth-bound pass over the weights (rave_amd/ops.py:
// _WN_PENDING).  The table travels by value in the kernel arguments (as adam.hip's): recorded with the launch by a hipGraph
// capture.  One workgroup per row, rows of all tensors concatenated; same arithmetic, same bits as the kernel above.
constexpr int kWnItems = 64;
struct WnTable {
    const float* dw[kWnItems];
    const float* v[kWnItems];
    const float* g[kWnItems];
    const float* norms[kWnItems];
    float* dv[kWnItems];
    float* dg[kWnItems];
    int row_begin[kWnItems + 1];
    int cols[kWnItems];
    int count;
};

__global__ __launch_bounds__(256) void weight_norm_bwd_batched_kernel(const WnTable tb) {
    __shared__ float red[4];
    int lo = 0, hi = tb.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tb.row_begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    weight_norm_bwd_row(tb.dw[lo], tb.v[lo], tb.g[lo], tb.norms[lo], (long)((int)blockIdx.x - tb.row_begin[lo]), (long)tb.cols[lo],
                        tb.dv[lo], tb.dg[lo], red);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// y = tanh(a * sigmoid(m)), x = [a | m] on the channel axis (rave/blocks.py:705-711)
__global__ __launch_bounds__(256) void amp_tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long cl, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long b = e / cl, r = e - b * cl;
    const float a = x[b * 2 * cl + r], m = x[b * 2 * cl + cl + r];
    y[e] = tanhf(a * sigmoidf_(m));
}
__global__ __launch_bounds__(256) void amp_tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dx, long cl, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long b = e / cl, r = e - b * cl;
    const float a = x[b * 2 * cl + r], m = x[b * 2 * cl + cl + r];
    const float s = sigmoidf_(m);
    const float t = tanhf(a * s);
    const float gz = dy[e] * (1.f - t * t);
    dx[b * 2 * cl + r] = gz * s;
    dx[b * 2 * cl + cl + r] = gz * a * s * (1.f - s);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                      int act, float slope, int c, long l, long total,
                                                      float* __restrict__ y) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float al = 0.f;
    if (act == RH_ACT_SNAKE) al = alpha[(e / l) % c];
    y[e] = rh_act_apply(x[e], act, slope, al);
}

// g = dy * act'(y) for an output LeakyReLU, 4 elements per lane where alignment allows
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act,
                                                      float slope, long n4, long total, float* __restrict__ g) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < n4) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dy)[e];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[e];
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = d[i] * rh_act_grad(v[i], act, slope, 0.f);
        reinterpret_cast<f32x4*>(g)[e] = o;
    } else {
        const long t = n4 * 4 + (e - n4);
        if (t < total) g[t] = dy[t] * rh_act_grad(y[t], act, slope, 0.f);
    }
}

// Snake backward (rave/blocks.py:852-860): f = x + sin^2(a x)/(a+eps)
//   df/dx = 1 + a sin(2 a x)/(a+eps) ;  df/da = x sin(2 a x)/(a+eps) - sin^2(a x)/(a+eps)^2
// grid (C, S): block (c, s) handles batch items s, s+S, ... of channel c; writes dx and a partial
// sum of dy*df/da; snake_alpha_reduce_kernel adds the S partials in order (deterministic).
__global__ __launch_bounds__(256) void snake_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ alpha, int B, int C, long l,
                                                        float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const float a = alpha[c];
    const float inv = 1.f / (a + 1e-9f);
    float acc = 0.f;
    for (int b = s; b < B; b += S) {
        const long base = ((long)b * C + c) * l;
        for (long e = threadIdx.x; e < l; e += 256) {
            const float xv = x[base + e], g = dy[base + e];
            const float sn = sinf(a * xv), s2 = sinf(2.f * a * xv);
            dx[base + e] = g * (1.f + a * s2 * inv);
            acc += g * (xv * s2 * inv - sn * sn * inv * inv);
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) part[c * S + s] = tot;
}
__global__ __launch_bounds__(256) void snake_alpha_reduce_kernel(const float* __restrict__ part, int C, int S,
                                                                 float* __restrict__ dalpha) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += part[c * S + i];
    dalpha[c] = s;
}

__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int l_in, int l_out, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long r = e / l_out;
    const int i = (int)(e - r * l_out);
    const float* src = x + r * l_in + 2 * i;
    y[e] = (src[0] + src[1]) * 0.5f;
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           int l_in, int l_out, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;  // over dx elements
    if (e >= total) return;
    const long r = e / l_in;
    const int i = (int)(e - r * l_in);
    const int o = i >> 1;
    dx[e] = o < l_out ? dy[r * l_out + o] * 0.5f : 0.f;
}

inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

// ---- spectral distance of AudioDistanceV1 (rave/core.py:330-344) on two complex STFTs -------------
// a = |Sx|, b = |Sy| ; sums[0] = sum (a-b)^2, sums[1] = sum a^2, sums[2] = sum |log(a+eps) - log(b+eps)|
// distance = sums[0]/sums[1] + sums[2]/n.  One pass over both spectrograms instead of ~12 elementwise
// + reduction launches; per-block partials + ordered finalize (deterministic).
constexpr int kSpecBlocks = 1024;

__global__ __launch_bounds__(256) void spectral_partials_kernel(const float2* __restrict__ sx, const float2* __restrict__ sy,
                                                                long n, float eps, float* __restrict__ part) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float2 x = sx[e], y = sy[e];
        const float a = sqrtf(x.x * x.x + x.y * x.y), b = sqrtf(y.x * y.x + y.y * y.y);
        const float d = a - b;
        s0 += d * d;
        s1 += a * a;
        s2 += fabsf(logf(a + eps) - logf(b + eps));
    }
    const float t0 = block_sum(s0, red);
    const float t1 = block_sum(s1, red);
    const float t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = t0;
        part[blockIdx.x * 3 + 1] = t1;
        part[blockIdx.x * 3 + 2] = t2;
    }
}

__global__ __launch_bounds__(256) void spectral_finalize_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
    __shared__ float red[4];
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i * 3 + k];
        const float t = block_sum(s, red);
        if (threadIdx.x == 0) sums[k] = t;
        __syncthreads();
    }
}

// gradients w.r.t. both complex spectrograms: d|z|/dz = z/|z| (0 at z = 0, as torch's abs backward)
__global__ __launch_bounds__(256) void spectral_bwd_kernel(const float2* __restrict__ sx, const float2* __restrict__ sy,
                                                           const float* __restrict__ sums, const float* __restrict__ gout,
                                                           long n, float eps, float2* __restrict__ dsx, float2* __restrict__ dsy,
                                                           int half_bins) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    // half_bins > 0: emit the operand of the C2R transform that is the adjoint of rfft -- interior bins halved
    // (dx_n = Re sum_k g_k e^{+i theta} = irfft_unnormalised(h), h_0 = g_0, h_{N/2} = g_{N/2}, h_k = g_k / 2)
    float g = gout[0];
    if (half_bins > 0) {
        const int k = (int)(e % half_bins);
        if (k != 0 && k != half_bins - 1) g *= 0.5f;
    }
    const float A = sums[0], B = sums[1];
    const float invB = 1.f / B, invN = 1.f / (float)n;
    const float2 x = sx[e], y = sy[e];
    const float a = sqrtf(x.x * x.x + x.y * x.y), b = sqrtf(y.x * y.x + y.y * y.y);
    const float d = a - b;
    const float ld = logf(a + eps) - logf(b + eps);
    const float sg = ld > 0.f ? 1.f : (ld < 0.f ? -1.f : 0.f);
    const float da = g * (2.f * d * invB - 2.f * a * A * invB * invB + sg * invN / (a + eps));
    const float db = g * (-2.f * d * invB - sg * invN / (b + eps));
    const float ra = a > 0.f ? da / a : 0.f, rb = b > 0.f ? db / b : 0.f;
    if (dsx) dsx[e] = make_float2(x.x * ra, x.y * ra);
    if (dsy) dsy[e] = make_float2(y.x * rb, y.y * rb);
}

}  // namespace

extern "C" int rh_weight_norm_fwd_f32(const float* v, const float* g, int64_t rows, int64_t cols, float* w,
                                      float* norms, rh_stream_t stream) {
    RH_REQUIRE(v && g && w, RH_ERR_INVALID, "weight_norm_fwd: null pointer");
    RH_REQUIRE(rows >= 0 && cols > 0, RH_ERR_INVALID, "weight_norm_fwd: bad shape");
    if (rows == 0) return RH_OK;
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, v, g,
                       (long)cols, w, norms);
    return rh_check_launch("weight_norm_fwd");
}

int rh_weight_norm_scales(const float* v, const float* g, int64_t rows, int64_t cols, float* norms, float* scale,
                          hipStream_t stream) {
    if (rows == 0) return RH_OK;
    hipLaunchKernelGGL(weight_norm_scales_kernel, dim3((unsigned)rows), dim3(256), 0, stream, v, g, (long)cols, norms, scale);
    return rh_check_launch("weight_norm_scales");
}

extern "C" int rh_weight_norm_bwd_f32(const float* dw, const float* v, const float* g, const float* norms,
                                      int64_t rows, int64_t cols, float* dv, float* dg, rh_stream_t stream) {
    RH_REQUIRE(dw && v && g && norms && dv && dg, RH_ERR_INVALID, "weight_norm_bwd: null pointer");
    RH_REQUIRE(rows >= 0 && cols > 0, RH_ERR_INVALID, "weight_norm_bwd: bad shape");
    if (rows == 0) return RH_OK;
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dw, v, g,
                       norms, (long)cols, dv, dg);
    return rh_check_launch("weight_norm_bwd");
}

extern "C" int rh_weight_norm_bwd_batched_f32(const rh_wn_bwd_item* items, int32_t n_items, rh_stream_t stream) {
    RH_REQUIRE(n_items >= 0 && (n_items == 0 || items), RH_ERR_INVALID, "weight_norm_bwd_batched: bad arguments");
    for (int i = 0; i < n_items;) {
        WnTable tb;
        int cnt = 0;
        long rows = 0;
        for (; i < n_items && cnt < kWnItems; ++i) {
            const rh_wn_bwd_item& it = items[i];
            RH_REQUIRE(it.dw && it.v && it.g && it.norms && it.dv && it.dg && it.rows >= 0 && it.cols > 0 && it.cols < 0x7fffffffl,
                       RH_ERR_INVALID, "weight_norm_bwd_batched: bad item %d", i);
            if (it.rows == 0) continue;
            if (rows + it.rows >= 0x7fffffffl) break;          // (next launch)
            tb.dw[cnt] = it.dw; tb.v[cnt] = it.v; tb.g[cnt] = it.g; tb.norms[cnt] = it.norms; tb.dv[cnt] = it.dv; tb.dg[cnt] = it.dg;
            tb.cols[cnt] = (int)it.cols;
            tb.row_begin[cnt] = (int)rows;
            rows += it.rows;
            ++cnt;
        }
        if (cnt == 0) {
            RH_REQUIRE(i >= n_items || items[i].rows < 0x7fffffffl, RH_ERR_UNSUPPORTED, "weight_norm_bwd_batched: too many rows");
            continue;
        }
        tb.row_begin[cnt] = (int)rows;
        tb.count = cnt;
        hipLaunchKernelGGL(weight_norm_bwd_batched_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, tb);
        if (int e = rh_check_launch("weight_norm_bwd_batched")) return e;
    }
    return RH_OK;
}

extern "C" int rh_amp_tanh_fwd_f32(const float* x, int32_t batch, int32_t c, int32_t l, float* y,
                                   rh_stream_t stream) {
    RH_REQUIRE(x && y, RH_ERR_INVALID, "amp_tanh_fwd: null pointer");
    const long cl = (long)c * l, total = cl * batch;
    if (total <= 0) return RH_OK;
    hipLaunchKernelGGL(amp_tanh_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, cl, total);
    return rh_check_launch("amp_tanh_fwd");
}

extern "C" int rh_amp_tanh_bwd_f32(const float* dy, const float* x, int32_t batch, int32_t c, int32_t l,
                                   float* dx, rh_stream_t stream) {
    RH_REQUIRE(dy && x && dx, RH_ERR_INVALID, "amp_tanh_bwd: null pointer");
    const long cl = (long)c * l, total = cl * batch;
    if (total <= 0) return RH_OK;
    hipLaunchKernelGGL(amp_tanh_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, cl, total);
    return rh_check_launch("amp_tanh_bwd");
}

extern "C" int rh_act_fwd_f32(const float* x, const float* snake_alpha, int32_t act, float slope,
                              int32_t batch, int32_t c, int32_t l, float* y, rh_stream_t stream) {
    RH_REQUIRE(x && y, RH_ERR_INVALID, "act_fwd: null pointer");
    RH_REQUIRE(act != RH_ACT_SNAKE || snake_alpha, RH_ERR_INVALID, "act_fwd: snake needs alpha");
    const long total = (long)batch * c * l;
    if (total <= 0) return RH_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, snake_alpha, act,
                       slope, c, (long)l, total, y);
    return rh_check_launch("act_fwd");
}

extern "C" int rh_act_bwd_f32(const float* dy, const float* y, int32_t act, float slope, int64_t n, float* g,
                              rh_stream_t stream) {
    RH_REQUIRE(n >= 0, RH_ERR_INVALID, "act_bwd: bad size");
    if (n == 0) return RH_OK;
    RH_REQUIRE(dy && y && g, RH_ERR_INVALID, "act_bwd: null pointer");
    RH_REQUIRE(act == RH_ACT_NONE || act == RH_ACT_LEAKY, RH_ERR_UNSUPPORTED, "act_bwd: none / leaky only");
    const bool al = (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g) & 15) == 0;
    const long n4 = al ? n / 4 : 0;
    const long threads = n4 + (n - 4 * n4);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(threads)), dim3(256), 0, (hipStream_t)stream, dy, y, act, slope,
                       n4, (long)n, g);
    return rh_check_launch("act_bwd");
}

extern "C" int64_t rh_snake_bwd_workspace_bytes(int32_t batch, int32_t c) {
    const int S = batch < 32 ? (batch > 0 ? batch : 1) : 32;
    return (int64_t)c * S * (int64_t)sizeof(float);
}

extern "C" int rh_snake_bwd_f32(const float* dy, const float* x, const float* alpha, int32_t batch, int32_t c,
                                int32_t l, float* dx, float* dalpha, void* workspace, int64_t workspace_bytes,
                                rh_stream_t stream) {
    RH_REQUIRE(batch >= 0 && c > 0 && l >= 0, RH_ERR_INVALID, "snake_bwd: bad shape");
    RH_REQUIRE(dalpha && alpha, RH_ERR_INVALID, "snake_bwd: null pointer");
    if (batch == 0 || l == 0) {
        (void)hipMemsetAsync(dalpha, 0, c * sizeof(float), (hipStream_t)stream);
        return RH_OK;
    }
    RH_REQUIRE(dy && x && dx, RH_ERR_INVALID, "snake_bwd: null pointer");
    const int S = batch < 32 ? batch : 32;
    RH_REQUIRE(workspace && workspace_bytes >= (int64_t)c * S * (int64_t)sizeof(float), RH_ERR_WORKSPACE,
               "snake_bwd: workspace too small");
    hipLaunchKernelGGL(snake_bwd_kernel, dim3(c, S), dim3(256), 0, (hipStream_t)stream, dy, x, alpha, batch, c, (long)l,
                       dx, (float*)workspace);
    if (int e = rh_check_launch("snake_bwd")) return e;
    hipLaunchKernelGGL(snake_alpha_reduce_kernel, dim3(blocks_for(c)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, c, S, dalpha);
    return rh_check_launch("snake_alpha_reduce");
}

// ---- STFT framing (centre + reflect pad + window) and its adjoint ---------------------------------
// frames[r][f][i] = window[i] * x[r][reflect(f*hop + i - n/2)]   (torch.stft(center=True, "reflect")
// as used by torchaudio.transforms.Spectrogram in rave/core.py:286-292), one pass instead of
// reflection_pad + strided copy + window multiply.  The FFT itself stays on rocFFT (torch.fft.rfft).
__device__ __forceinline__ int reflect_idx(int p, int t) { return p < 0 ? -p : (p >= t ? 2 * (t - 1) - p : p); }

__global__ __launch_bounds__(256) void stft_frame_fwd_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                             int t_len, int n, int hop, int n_frames, long total,
                                                             float* __restrict__ frames) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e % n);
    const long rf = e / n;
    const int f = (int)(rf % n_frames);
    const long r = rf / n_frames;
    const int p = reflect_idx(f * hop + i - n / 2, t_len);
    frames[e] = win[i] * x[r * t_len + p];
}

// four consecutive samples of a frame per thread (16-byte accesses; a quarter of the index divisions): n, hop, t_len
// multiples of four, 16-byte aligned tensors
__global__ __launch_bounds__(256) void stft_frame_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                              int t_len, int n, int hop, int n_frames, unsigned total4,
                                                              float* __restrict__ frames) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total4) return;
    const unsigned n4 = (unsigned)n >> 2;
    const unsigned rf = t / n4;
    const int i = (int)(t - rf * n4) * 4;
    const unsigned r = rf / (unsigned)n_frames;
    const int f = (int)(rf - r * (unsigned)n_frames);
    const int p0 = f * hop + i - n / 2;
    const float* __restrict__ xr = x + (long)r * t_len;
    const f32x4 w = *reinterpret_cast<const f32x4*>(win + i);
    f32x4 v;
    if (p0 >= 0 && p0 + 3 < t_len) v = *reinterpret_cast<const f32x4*>(xr + p0);
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = xr[reflect_idx(p0 + k, t_len)];
    }
    *reinterpret_cast<f32x4*>(frames + 4l * t) = f32x4{w[0] * v[0], w[1] * v[1], w[2] * v[2], w[3] * v[3]};
}

// dx[r][p] = sum over padded coordinates q that reflect onto p of sum_f window[q - f hop] dframes[r][f][q - f hop]
__global__ __launch_bounds__(256) void stft_frame_bwd_kernel(const float* __restrict__ dfr, const float* __restrict__ win,
                                                             int t_len, int n, int hop, int n_frames, long total,
                                                             float* __restrict__ dx, int accumulate) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % t_len);
    const long r = e / t_len;
    const float* base = dfr + r * (long)n_frames * n;
    const int half = n / 2;
    int qs[3];
    int nq = 0;
    qs[nq++] = p + half;
    if (p >= 1 && p <= half) qs[nq++] = half - p;                                   // left reflection
    if (p <= t_len - 2 && p >= t_len - 1 - half) qs[nq++] = 2 * (t_len - 1) - p + half;   // right reflection
    float s = 0.f;
    for (int k = 0; k < nq; ++k) {
        const int q = qs[k];
        int f_hi = q / hop;
        if (f_hi > n_frames - 1) f_hi = n_frames - 1;
        for (int f = f_hi; f >= 0; --f) {
            const int i = q - f * hop;
            if (i >= n) break;
            s += win[i] * base[(long)f * n + i];
        }
    }
    dx[e] = accumulate ? dx[e] + s : s;      // the scales of a multi-scale distance add up in place (no autograd add passes)
}

// four consecutive samples per thread where no reflected coordinate maps onto them (all but the n/2 samples at either
// end of a row): the same frames cover all four, 16-byte loads; same additions in the same order as the scalar kernel
__global__ __launch_bounds__(256) void stft_frame_bwd4_kernel(const float* __restrict__ dfr, const float* __restrict__ win,
                                                              int t_len, int n, int hop, int n_frames, unsigned total4,
                                                              float* __restrict__ dx, int accumulate) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total4) return;
    const unsigned t4 = (unsigned)t_len >> 2;
    const unsigned r = t / t4;
    const int p0 = (int)(t - r * t4) * 4;
    const float* base = dfr + (long)r * n_frames * n;
    const int half = n / 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (p0 > half && p0 + 3 < t_len - 1 - half) {
        const int q = p0 + half;
        int f_hi = q / hop;
        if (f_hi > n_frames - 1) f_hi = n_frames - 1;
        for (int f = f_hi; f >= 0; --f) {
            const int i = q - f * hop;
            if (i >= n) break;
            const f32x4 w = *reinterpret_cast<const f32x4*>(win + i);
            const f32x4 d = *reinterpret_cast<const f32x4*>(base + (long)f * n + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += w[k] * d[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k;
            int qs[3];
            int nq = 0;
            qs[nq++] = p + half;
            if (p >= 1 && p <= half) qs[nq++] = half - p;
            if (p <= t_len - 2 && p >= t_len - 1 - half) qs[nq++] = 2 * (t_len - 1) - p + half;
            float a = 0.f;
            for (int j = 0; j < nq; ++j) {
                const int q = qs[j];
                int f_hi = q / hop;
                if (f_hi > n_frames - 1) f_hi = n_frames - 1;
                for (int f = f_hi; f >= 0; --f) {
                    const int i = q - f * hop;
                    if (i >= n) break;
                    a += win[i] * base[(long)f * n + i];
                }
            }
            s[k] = a;
        }
    }
    f32x4* o = reinterpret_cast<f32x4*>(dx + 4l * t);
    if (accumulate) {
        const f32x4 old = *o;
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = old[k] + s[k];
    }
    *o = s;
}

extern "C" int rh_stft_frame_fwd_f32(const float* x, const float* window, int64_t rows, int32_t t_len, int32_t n_fft,
                                     int32_t hop, int32_t n_frames, float* frames, rh_stream_t stream) {
    RH_REQUIRE(x && window && frames, RH_ERR_INVALID, "stft_frame_fwd: null pointer");
    RH_REQUIRE(n_fft > 0 && hop > 0 && t_len > n_fft / 2 && n_frames > 0, RH_ERR_INVALID, "stft_frame_fwd: bad geometry");
    const long total = rows * (long)n_frames * n_fft;
    if (total <= 0) return RH_OK;
    // (the 16-byte paths index window / frames at q = p0 + n_fft / 2: n_fft / 2 must be a multiple of 4 too)
    const bool vec = (n_fft & 7) == 0 && (hop & 3) == 0 && (t_len & 3) == 0 && total / 4 < 0xffffffffl &&
                     (((uintptr_t)x | (uintptr_t)window | (uintptr_t)frames) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(stft_frame_fwd4_kernel, dim3(blocks_for(total / 4)), dim3(256), 0, (hipStream_t)stream, x, window, t_len,
                           n_fft, hop, n_frames, (unsigned)(total / 4), frames);
    else
        hipLaunchKernelGGL(stft_frame_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, window, t_len,
                           n_fft, hop, n_frames, total, frames);
    return rh_check_launch("stft_frame_fwd");
}

extern "C" int rh_stft_frame_bwd_acc_f32(const float* dframes, const float* window, int64_t rows, int32_t t_len,
                                         int32_t n_fft, int32_t hop, int32_t n_frames, float* dx, int32_t accumulate,
                                         rh_stream_t stream) {
    RH_REQUIRE(dframes && window && dx, RH_ERR_INVALID, "stft_frame_bwd: null pointer");
    RH_REQUIRE(n_fft > 0 && hop > 0 && t_len > n_fft / 2 && n_frames > 0, RH_ERR_INVALID, "stft_frame_bwd: bad geometry");
    const long total = rows * (long)t_len;
    if (total <= 0) return RH_OK;
    // (the 16-byte paths index window / frames at q = p0 + n_fft / 2: n_fft / 2 must be a multiple of 4 too)
    const bool vec = (n_fft & 7) == 0 && (hop & 3) == 0 && (t_len & 3) == 0 && total / 4 < 0xffffffffl &&
                     (((uintptr_t)dframes | (uintptr_t)window | (uintptr_t)dx) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(stft_frame_bwd4_kernel, dim3(blocks_for(total / 4)), dim3(256), 0, (hipStream_t)stream, dframes, window,
                           t_len, n_fft, hop, n_frames, (unsigned)(total / 4), dx, accumulate);
    else
        hipLaunchKernelGGL(stft_frame_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dframes, window,
                           t_len, n_fft, hop, n_frames, total, dx, accumulate);
    return rh_check_launch("stft_frame_bwd");
}

extern "C" int rh_stft_frame_bwd_f32(const float* dframes, const float* window, int64_t rows, int32_t t_len,
                                     int32_t n_fft, int32_t hop, int32_t n_frames, float* dx, rh_stream_t stream) {
    return rh_stft_frame_bwd_acc_f32(dframes, window, rows, t_len, n_fft, hop, n_frames, dx, 0, stream);
}

// distance = sum over the scales of sums[s][0] / sums[s][1] + sums[s][2] / n[s]   (rave/core.py:330-344, summed over the
// scales of MultiScaleSTFT as AudioDistanceV1.forward does): one tiny launch instead of a chain of scalar ATen kernels
__global__ void spectral_total_kernel(const float* __restrict__ sums, const float* __restrict__ inv_n, int S, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float d = 0.f;
        for (int s = 0; s < S; ++s) d += sums[3 * s] / sums[3 * s + 1] + sums[3 * s + 2] * inv_n[s];
        out[0] = d;
    }
}

// ---- the generator loss: scaled_i = w1_i * value_i (what the step logs), total = sum_i scaled_i * w2_i, as ONE one-thread
// launch instead of ~2 scalar ATen launches per term forward and backward (rave/model.py:336-344,392-412: `weights[...] * v`
// per distance, then `loss_gen_value += v * self.weights.get(k, 1.)`).  Every product is rounded on its own and the sum
// runs in term order from 0.0f -- bit for bit what the ATen chain computes (no contraction into FMAs).
constexpr int kLossItems = 16;
struct LossTable {
    const float* value[kLossItems];
    const float* w1_dev[kLossItems];      // device scalar (beta_factor of a recorded step) or null -> w1
    float w1[kLossItems];
    float w2[kLossItems];
    int count;
};

__global__ void loss_combine_fwd_kernel(const LossTable tb, float* __restrict__ scaled, float* __restrict__ total) {
#pragma clang fp contract(off)          // (plain * and + below: hip's __fmul_rn / __fadd_rn are inlined with contraction allowed)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t = 0.f;
    for (int i = 0; i < tb.count; ++i) {
        const float w = tb.w1_dev[i] ? tb.w1_dev[i][0] : tb.w1[i];
        const float sc = tb.value[i][0] * w;
        scaled[i] = sc;
        const float term = sc * tb.w2[i];
        t = t + term;
    }
    total[0] = t;
}

// d total / d value_i = (g * w2_i) * w1_i  (the order autograd's two MulBackward nodes multiply in)
__global__ void loss_combine_bwd_kernel(const LossTable tb, const float* __restrict__ g, float* __restrict__ grads) {
#pragma clang fp contract(off)
    const int i = threadIdx.x;
    if (blockIdx.x != 0 || i >= tb.count) return;
    const float w = tb.w1_dev[i] ? tb.w1_dev[i][0] : tb.w1[i];
    const float gw = g[0] * tb.w2[i];
    grads[i] = gw * w;
}

static int loss_table(const rh_loss_item* items, int32_t n, LossTable* tb, const char* what) {
    RH_REQUIRE(items && n > 0 && n <= kLossItems, RH_ERR_INVALID, "%s: 1 .. %d terms", what, kLossItems);
    for (int i = 0; i < n; ++i) {
        RH_REQUIRE(items[i].value, RH_ERR_INVALID, "%s: null value %d", what, i);
        tb->value[i] = items[i].value; tb->w1_dev[i] = items[i].w1_dev; tb->w1[i] = items[i].w1; tb->w2[i] = items[i].w2;
    }
    tb->count = n;
    return RH_OK;
}

extern "C" int rh_loss_combine_fwd_f32(const rh_loss_item* items, int32_t n_items, float* scaled, float* total, rh_stream_t stream) {
    LossTable tb;
    if (int e = loss_table(items, n_items, &tb, "loss_combine_fwd")) return e;
    RH_REQUIRE(scaled && total, RH_ERR_INVALID, "loss_combine_fwd: null output");
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tb, scaled, total);
    return rh_check_launch("loss_combine_fwd");
}

extern "C" int rh_loss_combine_bwd_f32(const rh_loss_item* items, int32_t n_items, const float* grad_total, float* grads,
                                       rh_stream_t stream) {
    LossTable tb;
    if (int e = loss_table(items, n_items, &tb, "loss_combine_bwd")) return e;
    RH_REQUIRE(grad_total && grads, RH_ERR_INVALID, "loss_combine_bwd: null pointer");
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tb, grad_total, grads);
    return rh_check_launch("loss_combine_bwd");
}

extern "C" int rh_spectral_total_f32(const float* sums, const float* inv_n, int32_t n_scales, float* out, rh_stream_t stream) {
    RH_REQUIRE(sums && inv_n && out && n_scales > 0, RH_ERR_INVALID, "spectral_total: bad arguments");
    hipLaunchKernelGGL(spectral_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, inv_n, n_scales, out);
    return rh_check_launch("spectral_total");
}

extern "C" int64_t rh_spectral_distance_workspace_bytes(void) { return (int64_t)kSpecBlocks * 3 * (int64_t)sizeof(float); }

extern "C" int rh_spectral_distance_fwd_f32(const float* sx, const float* sy, int64_t n_complex, float eps,
                                            float* sums, void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    RH_REQUIRE(sx && sy && sums && n_complex > 0, RH_ERR_INVALID, "spectral_distance_fwd: bad arguments");
    RH_REQUIRE(workspace && workspace_bytes >= rh_spectral_distance_workspace_bytes(), RH_ERR_WORKSPACE,
               "spectral_distance_fwd: workspace too small");
    long nb = (n_complex + 255) / 256;
    if (nb > kSpecBlocks) nb = kSpecBlocks;
    hipLaunchKernelGGL(spectral_partials_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)sx, (const float2*)sy, (long)n_complex, eps, (float*)workspace);
    if (int e = rh_check_launch("spectral_partials")) return e;
    hipLaunchKernelGGL(spectral_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                       (int)nb, sums);
    return rh_check_launch("spectral_finalize");
}

extern "C" int rh_spectral_distance_bwd_f32(const float* sx, const float* sy, const float* sums, const float* grad_out,
                                            int64_t n_complex, float eps, float* dsx, float* dsy, int32_t half_bins,
                                            rh_stream_t stream) {
    RH_REQUIRE(sx && sy && sums && grad_out && n_complex > 0 && half_bins >= 0, RH_ERR_INVALID,
               "spectral_distance_bwd: bad arguments");
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3(blocks_for(n_complex)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)sx, (const float2*)sy, sums, grad_out, (long)n_complex, eps, (float2*)dsx, (float2*)dsy, half_bins);
    return rh_check_launch("spectral_bwd");
}

extern "C" int rh_avgpool2_fwd_f32(const float* x, int64_t rows, int32_t l_in, float* y, rh_stream_t stream) {
    RH_REQUIRE(x && y, RH_ERR_INVALID, "avgpool2_fwd: null pointer");
    const int l_out = l_in / 2;
    const long total = rows * l_out;
    if (total <= 0) return RH_OK;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, l_in, l_out, total);
    return rh_check_launch("avgpool2_fwd");
}

extern "C" int rh_avgpool2_bwd_f32(const float* dy, int64_t rows, int32_t l_in, float* dx, rh_stream_t stream) {
    RH_REQUIRE(dy && dx, RH_ERR_INVALID, "avgpool2_bwd: null pointer");
    const int l_out = l_in / 2;
    const long total = rows * l_in;
    if (total <= 0) return RH_OK;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, l_in, l_out, total);
    return rh_check_launch("avgpool2_bwd");
}

// ---- VariationalEncoder.reparametrize (rave/blocks.py:727-745) in two launches + one for the gradient --------------------
//   mean, scale = z.chunk(2, 1);  std = softplus(scale) + 1e-4;  zs = eps * std + mean
//   kl = (mean^2 + std^2 - log(std^2) - 1).sum(1).mean()  = total / (B * L)
// (~12 ATen kernels forward and ~20 backward on a 0.5 MB tensor: launch-bound plumbing between encoder and decoder)
namespace {

constexpr int kRpBlocks = 64;

__device__ __forceinline__ float softplus_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// zs_range (round 6, optional): max |zs| goes to that range slot -- the decoder's first convolution reads its scale from there
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ z, const float* __restrict__ eps, int C, int L,
                                                          long n, float* __restrict__ zs, float* __restrict__ part,
                                                          unsigned* __restrict__ zs_range) {
    __shared__ float red[4];
    __shared__ float red_pub[4];
    float s = 0.f, amax = 0.f;
    const long CL = (long)C * L;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long b = e / CL, r = e - b * CL;
        const float mean = z[b * 2 * CL + r], scale = z[b * 2 * CL + CL + r];
        // every operation rounded separately, in ATen's order (no fused multiply-add): zs is then bit-identical to the
        // ATen chain it replaces -- the activations downstream are gated on its sign pattern
        const float sd = __fadd_rn(softplus_(scale), 1e-4f);
        const float var = __fmul_rn(sd, sd);
        const float zv = __fadd_rn(__fmul_rn(eps[e], sd), mean);
        zs[e] = zv;
        amax = fmaxf(amax, fabsf(zv));
        s += __fsub_rn(__fsub_rn(__fadd_rn(__fmul_rn(mean, mean), var), logf(var)), 1.f);
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
    if (zs_range) rh_range_publish(zs_range, amax, blockIdx.x, red_pub);      // (uniform)
}

__global__ void reparam_finalize_kernel(const float* __restrict__ part, int nblocks, float inv, float* __restrict__ kl) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nblocks; ++i) s += part[i];
        kl[0] = s * inv;
    }
}

__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ z, const float* __restrict__ eps,
                                                          const float* __restrict__ dzs, const float* __restrict__ dkl, int C, int L,
                                                          long n, float inv, float* __restrict__ dz) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const long CL = (long)C * L;
    const long b = e / CL, r = e - b * CL;
    const float mean = z[b * 2 * CL + r], scale = z[b * 2 * CL + CL + r];
    const float sd = __fadd_rn(softplus_(scale), 1e-4f);
    const float g = dkl ? dkl[0] * inv : 0.f;
    const float up = dzs ? dzs[e] : 0.f;
    const float dmean = up + g * 2.f * mean;
    const float dsd = up * eps[e] + g * (2.f * sd - 2.f / sd);
    const float sig = scale > 20.f ? 1.f : 1.f / (1.f + expf(-scale));       // d softplus (threshold 20 as torch)
    dz[b * 2 * CL + r] = dmean;
    dz[b * 2 * CL + CL + r] = dsd * sig;
}

}  // namespace

extern "C" int64_t rh_reparam_workspace_bytes(void) { return (int64_t)kRpBlocks * (int64_t)sizeof(float); }

extern "C" int rh_reparam_fwd_f32(const float* z, const float* eps, int32_t batch, int32_t c, int32_t l, float* zs, float* kl,
                                  void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    unsigned* zs_range = nullptr;              // where max |zs| goes, if the caller armed an output slot (rh_x6_set_ranges)
    rh_take_ranges(nullptr, nullptr, &zs_range, nullptr);
    RH_REQUIRE(z && eps && zs && kl && workspace && workspace_bytes >= rh_reparam_workspace_bytes(), RH_ERR_INVALID, "reparam_fwd: bad arguments");
    const long n = (long)batch * c * l;
    RH_REQUIRE(n > 0, RH_ERR_INVALID, "reparam_fwd: empty tensor");
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(kRpBlocks), dim3(256), 0, (hipStream_t)stream, z, eps, c, l, n, zs, (float*)workspace,
                       zs_range);
    if (int e = rh_check_launch("reparam_fwd")) return e;
    hipLaunchKernelGGL(reparam_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, kRpBlocks,
                       (float)(1.0 / ((double)batch * l)), kl);
    return rh_check_launch("reparam_finalize");
}

extern "C" int rh_reparam_bwd_f32(const float* z, const float* eps, const float* dzs, const float* dkl, int32_t batch, int32_t c,
                                  int32_t l, float* dz, rh_stream_t stream) {
    RH_REQUIRE(z && eps && dz, RH_ERR_INVALID, "reparam_bwd: null pointer");
    const long n = (long)batch * c * l;
    if (n <= 0) return RH_OK;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, eps, dzs, dkl, c, l, n,
                       (float)(1.0 / ((double)batch * l)), dz);
    return rh_check_launch("reparam_bwd");
}

// ---- AdaptiveInstanceNormalization, eval mode (rave/blocks.py:863-926) -----------------------------------------------
// Training mode is the identity; in eval mode the module keeps running per-(batch item, channel) statistics of the maps it
// sees (learn_x / learn_y) and, once both sides hold statistics, maps x -> (x - mean_x) / (std_x + 1e-5) * std_y + mean_y.
namespace {

// One workgroup per (batch item, channel) row of x (rows, l): mean and the UNBIASED standard deviation of the row
// (torch.std's default), two passes in a fixed order (deterministic), then the running update of rave/blocks.py:876-879,
// target[:bs] += (source - target[:bs]) / (num_updates + 1), on the first `rows` entries of the two statistic buffers.
__global__ __launch_bounds__(256) void adain_stats_kernel(const float* __restrict__ x, long l, const float* __restrict__ num_updates,
                                                          float* __restrict__ mean_buf, float* __restrict__ std_buf) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const float* xr = x + r * l;
    float s = 0.f;
    for (long e = threadIdx.x; e < l; e += 256) s += xr[e];
    const float mean = block_sum(s, red) / (float)l;
    float q = 0.f;
    for (long e = threadIdx.x; e < l; e += 256) { const float d = xr[e] - mean; q += d * d; }
    const float var = block_sum(q, red) / (float)(l - 1);          // l == 1: 0 / 0 = NaN, as torch.std
    if (threadIdx.x == 0) {
        const float den = num_updates[0] + 1.f;
        mean_buf[r] += (mean - mean_buf[r]) / den;
        std_buf[r] += (sqrtf(var) - std_buf[r]) / den;
    }
}

// y = (x - mean_x) / (std_x + 1e-5) * std_y + mean_y, statistics per row; every operation rounded on its own (the
// reference is a chain of separate ATen operations: no fused multiply-add).
__global__ __launch_bounds__(256) void adain_transfer_kernel(const float* __restrict__ x, long l, long total,
                                                             const float* __restrict__ mean_x, const float* __restrict__ std_x,
                                                             const float* __restrict__ mean_y, const float* __restrict__ std_y,
                                                             float* __restrict__ y) {
#pragma clang fp contract(off)
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long r = e / l;
    const float n = (x[e] - mean_x[r]) / (std_x[r] + 1e-5f);
    y[e] = n * std_y[r] + mean_y[r];
}

}  // namespace

extern "C" int rh_adain_stats_update_f32(const float* x, int64_t rows, int32_t l, const float* num_updates, float* mean_buf,
                                         float* std_buf, rh_stream_t stream) {
    RH_REQUIRE(rows >= 0 && l > 0, RH_ERR_INVALID, "adain_stats_update: bad shape");
    if (rows == 0) return RH_OK;
    RH_REQUIRE(x && num_updates && mean_buf && std_buf, RH_ERR_INVALID, "adain_stats_update: null pointer");
    RH_REQUIRE(rows < (1ll << 31), RH_ERR_UNSUPPORTED, "adain_stats_update: too many rows");
    hipLaunchKernelGGL(adain_stats_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (long)l, num_updates,
                       mean_buf, std_buf);
    return rh_check_launch("adain_stats_update");
}

extern "C" int rh_adain_transfer_f32(const float* x, int64_t rows, int32_t l, const float* mean_x, const float* std_x,
                                     const float* mean_y, const float* std_y, float* y, rh_stream_t stream) {
    RH_REQUIRE(rows >= 0 && l > 0, RH_ERR_INVALID, "adain_transfer: bad shape");
    const long total = (long)rows * l;
    if (total == 0) return RH_OK;
    RH_REQUIRE(x && mean_x && std_x && mean_y && std_y && y, RH_ERR_INVALID, "adain_transfer: null pointer");
    hipLaunchKernelGGL(adain_transfer_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long)l, total,
                       mean_x, std_x, mean_y, std_y, y);
    return rh_check_launch("adain_transfer");
}

// ---- range slot of a tensor no kernel of this library produced (common.hpp: RH_X6_F16; include/rave_hip.h: rh_amax_f32) ----
namespace {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, unsigned* slot) {
    float m = 0.f;
    const long n4 = ((uintptr_t)x & 15) == 0 ? n >> 2 : 0;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[4];
    rh_range_publish(slot, m, blockIdx.x, red);
}
}  // namespace

extern "C" int rh_amax_f32(const float* x, int64_t n, uint32_t* slot, rh_stream_t stream) {
    RH_REQUIRE(slot && (x || n == 0) && n >= 0, RH_ERR_INVALID, "amax: bad arguments");
    if (n == 0) return RH_OK;
    long blocks = (n / 4 + 1023) / 1024;          // >= 4 16-byte loads per thread
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, slot);
    return rh_check_launch("amax");
}
