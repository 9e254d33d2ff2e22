// Feature-matching distance of the GAN phase (rave/model.py:359-372: for every discriminator, the mean over its feature
// maps of mean_difference(real, fake, "L1"[, relative]) -- rave/core.py:236-252 --, averaged over the discriminators) on the
// UNSPLIT feature maps: a feature map holds the real half of the batch followed by the fake half (the discriminators run on
// cat([x, y])), so  A_i = sum |r - f|,  B_i = sum |r|  come from one pass over it, and
//     distance = sum_i w_i * (relative ? A_i / B_i : A_i)
// with the 1 / (features x discriminators [x elements]) factors folded into w_i by the host.  The backward writes both
// halves of d distance / d feature in one pass:  d/dr = w (sign(r-f)/B - A sign(r)/B^2),  d/df = -w sign(r-f)/B
// (non-relative: +-w sign(r-f)).  As ATen ops this was ~10 elementwise / reduction / copy passes per feature map (sub, abs,
// mean, and in backward sign, div, neg, expand and the cat of the two half gradients): ~10 % of a GAN-phase step.
// All feature maps of a step travel in ONE table in the kernel arguments (capturable); sums are per-block partials +
// an ordered finalize: deterministic.
#include "common.hpp"

namespace {

constexpr int kFm = 80;
constexpr int kFmChunk = 4096;            // elements of one half per workgroup

struct FmTable {
    const float* f[kFm];
    float* df[kFm];
    long half[kFm];
    float w[kFm];
    int blk_begin[kFm + 1];
    int count;
    int relative;
};

__device__ __forceinline__ float fm_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ int fm_find(const FmTable& tb, int blk) {
    int lo = 0, hi = tb.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tb.blk_begin[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void fm_partials_kernel(const FmTable tb, float* __restrict__ part) {
    __shared__ float red[4];
    const int it = fm_find(tb, blockIdx.x);
    const long half = tb.half[it];
    const long e0 = (long)(blockIdx.x - tb.blk_begin[it]) * kFmChunk;
    const float* __restrict__ r = tb.f[it];
    const float* __restrict__ f = r + half;
    float sa = 0.f, sb = 0.f;
    const bool vec = (half & 3) == 0 && (((uintptr_t)r) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int u = 0; u < kFmChunk / 1024; ++u) {
            const long e = e0 + u * 1024 + 4 * threadIdx.x;
            if (e < half) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(r + e), b = *reinterpret_cast<const f32x4*>(f + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) { sa += fabsf(a[k] - b[k]); sb += fabsf(a[k]); }
            }
        }
    } else {
        for (int u = 0; u < kFmChunk / 256; ++u) {
            const long e = e0 + u * 256 + threadIdx.x;
            if (e < half) { sa += fabsf(r[e] - f[e]); sb += fabsf(r[e]); }
        }
    }
    const float ta = fm_block_sum(sa, red), tbb = fm_block_sum(sb, red);
    if (threadIdx.x == 0) { part[2l * blockIdx.x] = ta; part[2l * blockIdx.x + 1] = tbb; }
}

// one workgroup per feature map: its partials in order -> sums[i] = (A, B)
__global__ __launch_bounds__(256) void fm_finalize_kernel(const FmTable tb, const float* __restrict__ part, float* __restrict__ sums) {
    __shared__ float red[4];
    const int it = blockIdx.x;
    const int b0 = tb.blk_begin[it], b1 = tb.blk_begin[it + 1];
    float sa = 0.f, sb = 0.f;
    for (int b = b0 + threadIdx.x; b < b1; b += 256) { sa += part[2l * b]; sb += part[2l * b + 1]; }
    const float ta = fm_block_sum(sa, red), tbb = fm_block_sum(sb, red);
    if (threadIdx.x == 0) { sums[2 * it] = ta; sums[2 * it + 1] = tbb; }
}

__global__ void fm_total_kernel(const FmTable tb, const float* __restrict__ sums, float* __restrict__ out, int accumulate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float d = accumulate ? out[0] : 0.f;
        for (int i = 0; i < tb.count; ++i) d += tb.w[i] * (tb.relative ? sums[2 * i] / sums[2 * i + 1] : sums[2 * i]);
        out[0] = d;
    }
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void fm_bwd_kernel(const FmTable tb, const float* __restrict__ sums, const float* __restrict__ gout) {
    const int it = fm_find(tb, blockIdx.x);
    const long half = tb.half[it];
    const long e0 = (long)(blockIdx.x - tb.blk_begin[it]) * kFmChunk;
    const float* __restrict__ r = tb.f[it];
    const float* __restrict__ f = r + half;
    float* __restrict__ dr = tb.df[it];
    float* __restrict__ dff = dr + half;
    const float gw = gout[0] * tb.w[it];
    float c1, c2;                               // d/dr = c1 sign(r - f) - c2 sign(r);  d/df = -c1 sign(r - f)
    if (tb.relative) {
        const float A = sums[2 * it], B = sums[2 * it + 1];
        c1 = gw / B;
        c2 = gw * A / (B * B);
    } else {
        c1 = gw;
        c2 = 0.f;
    }
    const bool vec = (half & 3) == 0 && ((((uintptr_t)r) | ((uintptr_t)dr)) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int u = 0; u < kFmChunk / 1024; ++u) {
            const long e = e0 + u * 1024 + 4 * threadIdx.x;
            if (e < half) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(r + e), b = *reinterpret_cast<const f32x4*>(f + e);
                f32x4 ga, gb;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s = c1 * sgn(a[k] - b[k]);
                    ga[k] = s - c2 * sgn(a[k]);
                    gb[k] = -s;
                }
                *reinterpret_cast<f32x4*>(dr + e) = ga;
                *reinterpret_cast<f32x4*>(dff + e) = gb;
            }
        }
    } else {
        for (int u = 0; u < kFmChunk / 256; ++u) {
            const long e = e0 + u * 256 + threadIdx.x;
            if (e < half) {
                const float s = c1 * sgn(r[e] - f[e]);
                dr[e] = s - c2 * sgn(r[e]);
                dff[e] = -s;
            }
        }
    }
}

int fill(const rh_fm_item* items, int n, int relative, bool need_df, FmTable* tb) {
    RH_REQUIRE(n > 0 && n <= kFm && items, RH_ERR_UNSUPPORTED, "feature_matching: 1 ... %d feature maps per call", kFm);
    long blk = 0;
    for (int i = 0; i < n; ++i) {
        RH_REQUIRE(items[i].f && items[i].half > 0 && (!need_df || items[i].df), RH_ERR_INVALID, "feature_matching: bad item %d", i);
        tb->f[i] = items[i].f; tb->df[i] = items[i].df; tb->half[i] = items[i].half; tb->w[i] = items[i].w;
        tb->blk_begin[i] = (int)blk;
        blk += (items[i].half + kFmChunk - 1) / kFmChunk;
        RH_REQUIRE(blk < 0x7fffffffl, RH_ERR_UNSUPPORTED, "feature_matching: too many elements");
    }
    tb->blk_begin[n] = (int)blk;
    tb->count = n;
    tb->relative = relative;
    return RH_OK;
}

}  // namespace

extern "C" int64_t rh_feature_matching_workspace_bytes(const rh_fm_item* items, int32_t n_items) {
    if (!items || n_items <= 0 || n_items > kFm) return -1;
    long blk = 0;
    for (int i = 0; i < n_items; ++i) blk += (items[i].half + kFmChunk - 1) / kFmChunk;
    return (int64_t)blk * 2 * (int64_t)sizeof(float);
}

extern "C" int rh_feature_matching_fwd_f32(const rh_fm_item* items, int32_t n_items, int32_t relative, void* workspace,
                                           int64_t workspace_bytes, float* sums, float* out, rh_stream_t stream) {
    FmTable tb;
    if (int e = fill(items, n_items, relative, false, &tb)) return e;
    RH_REQUIRE(workspace && sums && out && workspace_bytes >= rh_feature_matching_workspace_bytes(items, n_items), RH_ERR_WORKSPACE,
               "feature_matching_fwd: workspace too small");
    hipLaunchKernelGGL(fm_partials_kernel, dim3((unsigned)tb.blk_begin[n_items]), dim3(256), 0, (hipStream_t)stream, tb, (float*)workspace);
    if (int e = rh_check_launch("feature_matching_partials")) return e;
    hipLaunchKernelGGL(fm_finalize_kernel, dim3((unsigned)n_items), dim3(256), 0, (hipStream_t)stream, tb, (const float*)workspace, sums);
    if (int e = rh_check_launch("feature_matching_finalize")) return e;
    hipLaunchKernelGGL(fm_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tb, (const float*)sums, out, 0);
    return rh_check_launch("feature_matching_total");
}

extern "C" int rh_feature_matching_bwd_f32(const rh_fm_item* items, int32_t n_items, int32_t relative, const float* sums,
                                           const float* grad_out, rh_stream_t stream) {
    FmTable tb;
    if (int e = fill(items, n_items, relative, true, &tb)) return e;
    RH_REQUIRE(sums && grad_out, RH_ERR_INVALID, "feature_matching_bwd: null pointer");
    hipLaunchKernelGGL(fm_bwd_kernel, dim3((unsigned)tb.blk_begin[n_items]), dim3(256), 0, (hipStream_t)stream, tb, sums, grad_out);
    return rh_check_launch("feature_matching_bwd");
}
