// Adam update of many parameter tensors in a few launches (torch.optim.Adam semantics: no weight decay, no amsgrad;
// rave/model.py:226-233 -- Adam(lr, betas = (0.5, 0.9)) for the generator and the discriminator).  Plumbing beside the
// hot path: torch's fused multi-tensor kernel covers the 15.8 M parameters of the v2 generator with ~240 blocks of 64 K
// elements and reaches 1.6 TB/s (0.28 ms per step); here every workgroup takes 2048 elements of one tensor
// (16-byte accesses), ~7700 workgroups per step.
//
//   m = lerp(m, g, 1 - beta1)                      (ATen's lerp: the branch on the weight included)
//   v = beta2 v + (1 - beta2) g g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// The tensor table travels BY VALUE in the kernel arguments (<= 64 tensors per launch), so a hipGraph capture records
// it with the launch -- no host-to-device copy node, nothing to keep alive.  Step counters and the learning rate are
// device scalars (a recorded graph reads their current values).  EVERY parameter has its own counter, as in
// torch.optim.Adam (a parameter that receives its first gradient later -- the noise branch of the v1 generator only runs
// once the model is warmed up, rave/blocks.py:418 -- starts its bias corrections at t = 1, not at the age of the optimizer;
// one that stops receiving gradients -- the encoder after the warm-up, :743 -- keeps its count): a first launch advances every DISTINCT counter of the call by one and leaves its two bias corrections
// (computed in double) in `aux`, the update launches read the pair of their tensor's counter.
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "common.hpp"

namespace {

constexpr int kAdamItems = 64;
constexpr int kAdamElems = 2048;          // elements per workgroup
constexpr int kAdamTicks = 256;           // counters per tick launch

struct AdamTable {
    float* p[kAdamItems];
    const float* g[kAdamItems];
    float* m[kAdamItems];
    float* v[kAdamItems];
    int blk_begin[kAdamItems + 1];        // prefix of workgroups
    int n[kAdamItems];
    int slot[kAdamItems];                 // which pair of `aux` (= which distinct step counter) the tensor reads
    int count;
};

struct AdamTicks {
    float* step[kAdamTicks];
    int count;
};

__global__ void adam_tick_kernel(const AdamTicks ts, float* __restrict__ aux, double beta1, double beta2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ts.count) return;
    const float t = ts.step[i][0] + 1.f;
    ts.step[i][0] = t;
    aux[2 * i] = (float)(1.0 - pow(beta1, (double)t));            // bias_correction1
    aux[2 * i + 1] = (float)sqrt(1.0 - pow(beta2, (double)t));    // sqrt(bias_correction2)
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float step_size,
                                         float inv_bc2s, float eps) {
    const float diff = g - m;
    m = w1 < 0.5f ? m + w1 * diff : g - diff * (1.f - w1);
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) * inv_bc2s + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_update_kernel(const AdamTable tb, const float* __restrict__ lr,
                                                          const float* __restrict__ aux, float beta1, float beta2, float eps) {
    int lo = 0, hi = tb.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tb.blk_begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int n = tb.n[lo];
    const int off = ((int)blockIdx.x - tb.blk_begin[lo]) * kAdamElems;
    float* __restrict__ p = tb.p[lo];
    const float* __restrict__ g = tb.g[lo];
    float* __restrict__ m = tb.m[lo];
    float* __restrict__ v = tb.v[lo];
    const float w1 = 1.f - beta1;
    const int slot = tb.slot[lo];
    const float step_size = lr[0] / aux[2 * slot];
    const float inv_bc2s = 1.f / aux[2 * slot + 1];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#pragma unroll
    for (int j = 0; j < kAdamElems / 1024; ++j) {
        const int i = off + j * 1024 + 4 * threadIdx.x;
        if (vec && i + 3 < n) {
            f32x4 pp = *reinterpret_cast<const f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 mm = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = pp[k], b = mm[k], c = vv[k];
                adam_one(a, gg[k], b, c, w1, beta2, step_size, inv_bc2s, eps);
                pp[k] = a; mm[k] = b; vv[k] = c;
            }
            *reinterpret_cast<f32x4*>(p + i) = pp;
            *reinterpret_cast<f32x4*>(m + i) = mm;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) adam_one(p[i + k], g[i + k], m[i + k], v[i + k], w1, beta2, step_size, inv_bc2s, eps);
        }
    }
}

}  // namespace

extern "C" int rh_adam_step_f32(const rh_adam_item* items, int32_t n_items, const float* lr, float beta1, float beta2, float eps,
                                float* aux, rh_stream_t stream) {
    RH_REQUIRE(n_items >= 0 && lr && aux, RH_ERR_INVALID, "adam_step: bad arguments");
    RH_REQUIRE(n_items == 0 || items, RH_ERR_INVALID, "adam_step: null table");
    // ---- the distinct step counters of the call (items may share one), each advanced exactly once
    std::vector<float*> uniq;
    std::vector<int> slot((size_t)n_items, -1);
    {
        std::unordered_map<const float*, int> seen;
        for (int i = 0; i < n_items; ++i) {
            const rh_adam_item& it = items[i];
            RH_REQUIRE(it.p && it.g && it.m && it.v && it.step && it.n >= 0 && it.n < 0x7fffffffl, RH_ERR_INVALID, "adam_step: bad item %d", i);
            if (it.n == 0) continue;
            auto f = seen.find(it.step);
            if (f == seen.end()) {
                f = seen.emplace(it.step, (int)uniq.size()).first;
                uniq.push_back(it.step);
            }
            slot[(size_t)i] = f->second;
        }
    }
    for (size_t u0 = 0; u0 < uniq.size(); u0 += kAdamTicks) {
        AdamTicks ts;
        ts.count = (int)std::min<size_t>(kAdamTicks, uniq.size() - u0);
        for (int k = 0; k < ts.count; ++k) ts.step[k] = uniq[u0 + k];
        hipLaunchKernelGGL(adam_tick_kernel, dim3((ts.count + 63) / 64), dim3(64), 0, (hipStream_t)stream, ts, aux + 2 * u0, (double)beta1,
                           (double)beta2);
        if (int e = rh_check_launch("adam_tick")) return e;
    }
    // `i` is the consumed index: empty tensors are skipped without taking a table slot, so a chunk may span more than
    // kAdamItems items -- the next chunk continues where this one stopped (never re-processing an item)
    for (int i = 0; i < n_items;) {
        AdamTable tb;
        int cnt = 0, blk = 0;
        for (; i < n_items && cnt < kAdamItems; ++i) {
            const rh_adam_item& it = items[i];
            if (it.n == 0) continue;
            tb.p[cnt] = it.p; tb.g[cnt] = it.g; tb.m[cnt] = it.m; tb.v[cnt] = it.v;
            tb.n[cnt] = (int)it.n;
            tb.slot[cnt] = slot[(size_t)i];
            tb.blk_begin[cnt] = blk;
            blk += (int)((it.n + kAdamElems - 1) / kAdamElems);
            ++cnt;
        }
        if (cnt == 0) continue;
        tb.blk_begin[cnt] = blk;
        tb.count = cnt;
        hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, tb, lr, (const float*)aux, beta1,
                           beta2, eps);
        if (int e = rh_check_launch("adam_update")) return e;
    }
    return RH_OK;
}
