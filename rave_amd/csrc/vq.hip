// Residual vector quantisation bottleneck (rave/quantization.py:59-181,283-300; DiscreteEncoder,
// rave/blocks.py:794-830): one EuclideanCodebook step as two kernels.
//
//   vq_assign_kernel : nearest code of every vector under the reference's expanded distance
//                      d(n,k) = (|x_n|^2 - 2 x_n.e_k) + |e_k|^2   (quantization.py:131-136),
//                      first index on ties (torch.max semantics); writes the index, the residual
//                      x - e_ind for the next quantiser, accumulates e_ind into the running sum of
//                      quantised vectors, and emits per-block partial sums of |e_ind - x|^2 (commit loss).
//   vq_ema_kernel    : training-time codebook update (quantization.py:165-179): per-code counts and
//                      vector sums in vector order (deterministic; no float atomics), EMA of
//                      cluster_size / embed_avg;  vq_normalize_kernel: Laplace smoothing + embed = avg/size.
// Data is tiny (2048 x 128 vectors, 1024 codes per layer for discrete.gin): these kernels are
// latency-bound; the point is 3 launches per quantiser instead of ~25 ATen ops and no (N, K) one-hot.
#include "common.hpp"

namespace {

constexpr int kVecPerBlock = 16;

__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ x, const float* __restrict__ embed,
                                                        int N, int D, int K, long long* __restrict__ ind,
                                                        float* __restrict__ residual, float* __restrict__ qsum,
                                                        float* __restrict__ loss_part) {
    extern __shared__ float sm[];
    float* xs = sm;                              // [16][D]
    float* x2 = sm + kVecPerBlock * D;           // [16]
    float* bestv = x2 + kVecPerBlock;            // [4 waves][16]
    int* besti = reinterpret_cast<int*>(bestv + 4 * kVecPerBlock);
    int* sel = besti + 4 * kVecPerBlock;         // [16]
    float* lred = reinterpret_cast<float*>(sel + kVecPerBlock);   // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * kVecPerBlock;
    for (int e = tid; e < kVecPerBlock * D; e += 256) {
        const int v = e / D, d = e - v * D;
        xs[e] = (n0 + v < N) ? x[(long)(n0 + v) * D + d] : 0.f;
    }
    __syncthreads();
    if (tid < kVecPerBlock) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += xs[tid * D + d] * xs[tid * D + d];
        x2[tid] = s;
    }
    __syncthreads();

    float bv[kVecPerBlock];
    int bi[kVecPerBlock];
#pragma unroll
    for (int v = 0; v < kVecPerBlock; ++v) { bv[v] = INFINITY; bi[v] = 0x7fffffff; }
    for (int k = tid; k < K; k += 256) {
        const float* __restrict__ er = embed + (long)k * D;
        float dot[kVecPerBlock];
#pragma unroll
        for (int v = 0; v < kVecPerBlock; ++v) dot[v] = 0.f;
        float e2 = 0.f;
        for (int d = 0; d < D; ++d) {
            const float ev = er[d];
            e2 += ev * ev;
#pragma unroll
            for (int v = 0; v < kVecPerBlock; ++v) dot[v] += xs[v * D + d] * ev;
        }
#pragma unroll
        for (int v = 0; v < kVecPerBlock; ++v) {
            const float dist = (x2[v] - 2.f * dot[v]) + e2;
            if (dist < bv[v]) { bv[v] = dist; bi[v] = k; }      // k ascending per thread: strict < keeps the first
        }
    }
    // reduce (min distance, then min index) across the block
#pragma unroll
    for (int v = 0; v < kVecPerBlock; ++v) {
        float val = bv[v];
        int idx = bi[v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(val, o, 64);
            const int oi = __shfl_down(idx, o, 64);
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (lane == 0) { bestv[wave * kVecPerBlock + v] = val; besti[wave * kVecPerBlock + v] = idx; }
    }
    __syncthreads();
    if (tid < kVecPerBlock) {
        float val = bestv[tid];
        int idx = besti[tid];
        for (int w = 1; w < 4; ++w) {
            const float ov = bestv[w * kVecPerBlock + tid];
            const int oi = besti[w * kVecPerBlock + tid];
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        sel[tid] = idx;
        if (n0 + tid < N) ind[n0 + tid] = idx;
    }
    __syncthreads();
    float ls = 0.f;
    for (int e = tid; e < kVecPerBlock * D; e += 256) {
        const int v = e / D, d = e - v * D;
        if (n0 + v < N) {
            const float q = embed[(long)sel[v] * D + d];
            const float xv = xs[e];
            const long o = (long)(n0 + v) * D + d;
            if (residual) residual[o] = xv - q;
            if (qsum) qsum[o] += q;
            const float df = q - xv;
            ls += df * df;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_down(ls, o, 64);
    if (lane == 0) lred[wave] = ls;
    __syncthreads();
    if (tid == 0 && loss_part) loss_part[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
}

// one block per code: count and ordered vector sum of the vectors assigned to it, then the EMAs.
// The scan over the N indices goes 64 at a time (one coalesced load + a ballot per wave) and only the hits -- N / K = 1 on
// average -- touch x, in ascending n: the same sums in the same order as a serial scan.  (The serial scan, one dependent
// 8-byte load per vector, took 395 us per quantiser: 6.3 ms of the 131 ms discrete step for 1024 vectors.)
__global__ __launch_bounds__(128) void vq_ema_kernel(const float* __restrict__ x, const long long* __restrict__ ind,
                                                     int N, int D, float decay, float* __restrict__ cluster_size,
                                                     float* __restrict__ embed_avg) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63;
    int count = 0;
    for (int d0 = 0; d0 < D; d0 += 128) {
        const int d = d0 + threadIdx.x;
        float s = 0.f;
        int c = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int n = n0 + lane;
            const bool hit = n < N && ind[n] == k;
            unsigned long long mask = __ballot(hit);
            c += __popcll(mask);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (d < D) s += x[(long)(n0 + b) * D + d];
            }
        }
        count = c;
        if (d < D) {
            const long o = (long)k * D + d;
            embed_avg[o] = embed_avg[o] * decay + s * (1.f - decay);
        }
    }
    if (threadIdx.x == 0) cluster_size[k] = cluster_size[k] * decay + (float)count * (1.f - decay);
}

// embed = embed_avg / (laplace_smoothing(cluster_size) * sum(cluster_size))
__global__ __launch_bounds__(128) void vq_normalize_kernel(const float* __restrict__ cluster_size,
                                                           const float* __restrict__ embed_avg, int D, int K, float eps,
                                                           float* __restrict__ embed) {
    __shared__ float red[128];
    float s = 0.f;
    for (int i = threadIdx.x; i < K; i += 128) s += cluster_size[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float total = red[0];
    const int k = blockIdx.x;
    const float cs = (cluster_size[k] + eps) / (total + (float)K * eps) * total;
    for (int d = threadIdx.x; d < D; d += 128) embed[(long)k * D + d] = embed_avg[(long)k * D + d] / cs;
}

}  // namespace

extern "C" int64_t rh_vq_loss_partials(int64_t n_vectors) { return (n_vectors + kVecPerBlock - 1) / kVecPerBlock; }

extern "C" int rh_vq_assign_f32(const float* x, const float* embed, int64_t n_vectors, int32_t dim,
                                int32_t codebook_size, int64_t* indices, float* residual, float* quantized_sum,
                                float* loss_partials, rh_stream_t stream) {
    RH_REQUIRE(n_vectors >= 0 && dim > 0 && codebook_size > 0, RH_ERR_INVALID, "vq_assign: bad sizes");
    RH_REQUIRE(dim <= 1024 && n_vectors < (1ll << 31), RH_ERR_UNSUPPORTED, "vq_assign: dim > 1024 or too many vectors");
    if (n_vectors == 0) return RH_OK;
    RH_REQUIRE(x && embed && indices, RH_ERR_INVALID, "vq_assign: null pointer");
    const unsigned blocks = (unsigned)rh_vq_loss_partials(n_vectors);
    const size_t lds = sizeof(float) * ((size_t)kVecPerBlock * dim + kVecPerBlock + 4 * kVecPerBlock) +
                       sizeof(int) * (4 * kVecPerBlock + kVecPerBlock) + sizeof(float) * 4;
    hipLaunchKernelGGL(vq_assign_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, embed, (int)n_vectors, dim,
                       codebook_size, (long long*)indices, residual, quantized_sum, loss_partials);
    return rh_check_launch("vq_assign");
}

extern "C" int rh_vq_ema_update_f32(const float* x, const int64_t* indices, int64_t n_vectors, int32_t dim,
                                    int32_t codebook_size, float decay, float epsilon, float* cluster_size,
                                    float* embed_avg, float* embed, rh_stream_t stream) {
    RH_REQUIRE(n_vectors >= 0 && dim > 0 && codebook_size > 0 && n_vectors < (1ll << 31), RH_ERR_INVALID,
               "vq_ema_update: bad sizes");
    RH_REQUIRE(x && indices && cluster_size && embed_avg && embed, RH_ERR_INVALID, "vq_ema_update: null pointer");
    hipLaunchKernelGGL(vq_ema_kernel, dim3(codebook_size), dim3(128), 0, (hipStream_t)stream, x,
                       (const long long*)indices, (int)n_vectors, dim, decay, cluster_size, embed_avg);
    if (int e = rh_check_launch("vq_ema")) return e;
    hipLaunchKernelGGL(vq_normalize_kernel, dim3(codebook_size), dim3(128), 0, (hipStream_t)stream, cluster_size,
                       embed_avg, dim, codebook_size, epsilon, embed);
    return rh_check_launch("vq_normalize");
}
