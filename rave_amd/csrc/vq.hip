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
#include <cstdlib>
#include <mutex>

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


// ---- round 5: the distance search on the f32 matrix cores ------------------------------------------------------------------
// vq_assign_kernel above is a vector-ALU kernel on 128 workgroups whose threads walk one code row each (uncoalesced): 77 us per
// quantiser at 2048 x 1024 x 128, 62 launches = 4.8 ms of a discrete training step.  Here the 2 N K D flops of
// x_n . e_k run on v_mfma_f32_32x32x2_f32 -- an f32 MFMA accumulates its two k products as chained fmas in k order, so
// dot(n, k) is the SAME fmaf chain over d = 0 .. D-1 the vector kernel computes, |x|^2 and |e|^2 are the same sequential
// chains, the distance the same expression: same bits, same indices.
//   vq_partial_kernel : grid (N / 32, K / (128 kVqIters)); a workgroup stages its 32 vectors and, per iteration, 128 code rows
//                       in LDS (coalesced 16-byte loads, row pitch D + 1: conflict-free column reads), every wave multiplies
//                       its 32 codes x 32 vectors over D (D / 2 MFMAs), and keeps per vector the best (distance, index) of
//                       its codes -- strict < in ascending index order, ties across lanes / waves by the smaller index = the
//                       first index, torch.max semantics; one candidate per (vector, code group) goes to scratch.
//   vq_pick_kernel    : merges the candidates of a vector (smaller distance, then smaller index) and does the rest of
//                       vq_assign_kernel: index, residual, running sum, loss partials -- same block shape (16 vectors, 256
//                       threads), same summation order, so the loss partials are the same bits too.
constexpr int kVqIters = 2;            // code tiles (128 codes) per workgroup
constexpr int kVqTileCodes = 128;

__global__ __launch_bounds__(256) void vq_partial_kernel(const float* __restrict__ x, const float* __restrict__ embed, int N, int D,
                                                         int K, float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    extern __shared__ float sm[];
    const int P = D + 1;
    float* xs = sm;                          // [32][P]
    float* es = xs + 32 * P;                 // [128][P]
    float* x2s = es + kVqTileCodes * P;      // [32]
    float* e2s = x2s + 32;                   // [128]
    float* wv = e2s + kVqTileCodes;          // [4][32]
    int* wi = reinterpret_cast<int*>(wv + 4 * 32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int q = blockIdx.y, nq = gridDim.y;
    const int d4n = D >> 2;                  // float4s per row
    for (int e = tid; e < 32 * d4n; e += 256) {
        const int r = e / d4n, c = e - r * d4n;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n0 + r < N) v = *reinterpret_cast<const f32x4*>(x + (long)(n0 + r) * D + 4 * c);
        float* dst = xs + r * P + 4 * c;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += xs[tid * P + d] * xs[tid * P + d];
        x2s[tid] = s;
    }
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int it = 0; it < kVqIters; ++it) {
        const int c0 = (q * kVqIters + it) * kVqTileCodes;
        __syncthreads();                     // the previous tile's reads are done (and x2s is written)
        for (int e = tid; e < kVqTileCodes * d4n; e += 256) {
            const int r = e / d4n, c = e - r * d4n;
            const f32x4 v = *reinterpret_cast<const f32x4*>(embed + (long)(c0 + r) * D + 4 * c);
            float* dst = es + r * P + 4 * c;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        __syncthreads();
        if (tid < kVqTileCodes) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += es[tid * P + d] * es[tid * P + d];
            e2s[tid] = s;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ea = es + (wave * 32 + j) * P + g;
        const float* xb = xs + j * P + g;
        for (int kk = 0; kk < D; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[kk], xb[kk], acc, 0, 0, 0);
        __syncthreads();                     // e2s
        const float x2 = x2s[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {       // rows ascend with r: strict < keeps the first index
            const int m = wave * 32 + 4 * g + (r & 3) + 8 * (r >> 2);
            const float dist = (x2 - 2.f * acc[r]) + e2s[m];
            if (dist < bv) { bv = dist; bi = c0 + m; }
        }
    }
    {   // the two half-waves hold different codes of the same vector
        const float ov = __shfl_xor(bv, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (g == 0) { wv[wave * 32 + j] = bv; wi[wave * 32 + j] = bi; }
    __syncthreads();
    if (tid < 32 && n0 + tid < N) {
        float val = wv[tid];
        int idx = wi[tid];
        for (int w = 1; w < 4; ++w) {
            const float ov = wv[w * 32 + tid];
            const int oi = wi[w * 32 + tid];
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        cand_val[(long)(n0 + tid) * nq + q] = val;
        cand_idx[(long)(n0 + tid) * nq + q] = idx;
    }
}

// (x and residual may be the same buffer -- the eval-mode forward quantises in place: no __restrict__ on them; every element is
// read and then written by the same thread)
__global__ __launch_bounds__(256) void vq_pick_kernel(const float* x, const float* __restrict__ embed,
                                                      const float* __restrict__ cand_val, const int* __restrict__ cand_idx, int nq,
                                                      int N, int D, long long* __restrict__ ind, float* residual,
                                                      float* __restrict__ qsum, float* __restrict__ loss_part) {
    __shared__ int sel[kVecPerBlock];
    __shared__ float lred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * kVecPerBlock;
    if (tid < kVecPerBlock && n0 + tid < N) {
        const long o = (long)(n0 + tid) * nq;
        float val = cand_val[o];
        int idx = cand_idx[o];
        for (int c = 1; c < nq; ++c) {
            const float ov = cand_val[o + c];
            const int oi = cand_idx[o + c];
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        sel[tid] = idx;
        ind[n0 + tid] = idx;
    }
    __syncthreads();
    float ls = 0.f;
    for (int e = tid; e < kVecPerBlock * D; e += 256) {      // (the tail of vq_assign_kernel, same order of operations)
        const int v = e / D, d = e - v * D;
        if (n0 + v < N) {
            const float qv = embed[(long)sel[v] * D + d];
            const long o = (long)(n0 + v) * D + d;
            const float xv = x[o];
            if (residual) residual[o] = xv - qv;
            if (qsum) qsum[o] += qv;
            const float df = qv - xv;
            ls += df * df;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_down(ls, o, 64);
    if (lane == 0) lred[wave] = ls;
    __syncthreads();
    if (tid == 0 && loss_part) loss_part[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
}

bool vq_mfma_ok(int64_t n, int dim, int K) {
    const char* e = getenv("RH_VQ_MFMA");            // read per call (the tests compare both kernels)
    if (e && e[0] == '0') return false;
    return dim % 4 == 0 && dim >= 8 && dim <= 128 && K % (kVqTileCodes * kVqIters) == 0 && n >= 256;
}

}  // namespace

extern "C" int64_t rh_vq_loss_partials(int64_t n_vectors) { return (n_vectors + kVecPerBlock - 1) / kVecPerBlock; }

// Scratch of rh_vq_assign_ws_f32: one (distance, index) candidate per vector and group of 256 codes when the distance search
// runs on the matrix cores; 0 when the one-launch vector kernel takes the geometry.
extern "C" int64_t rh_vq_assign_workspace_bytes(int64_t n_vectors, int32_t dim, int32_t codebook_size) {
    if (n_vectors <= 0 || dim <= 0 || codebook_size <= 0 || !vq_mfma_ok(n_vectors, dim, codebook_size)) return 0;
    return n_vectors * (codebook_size / (kVqTileCodes * kVqIters)) * (int64_t)(sizeof(float) + sizeof(int));
}

static int vq_assign_impl(const float* x, const float* embed, int64_t n_vectors, int32_t dim, int32_t codebook_size,
                          int64_t* indices, float* residual, float* quantized_sum, float* loss_partials, void* workspace,
                          int64_t workspace_bytes, rh_stream_t stream) {
    RH_REQUIRE(n_vectors >= 0 && dim > 0 && codebook_size > 0, RH_ERR_INVALID, "vq_assign: bad sizes");
    RH_REQUIRE(dim <= 1024 && n_vectors < (1ll << 31), RH_ERR_UNSUPPORTED, "vq_assign: dim > 1024 or too many vectors");
    if (n_vectors == 0) return RH_OK;
    RH_REQUIRE(x && embed && indices, RH_ERR_INVALID, "vq_assign: null pointer");
    const unsigned blocks = (unsigned)rh_vq_loss_partials(n_vectors);
    const int64_t need = rh_vq_assign_workspace_bytes(n_vectors, dim, codebook_size);
    if (need > 0 && workspace && workspace_bytes >= need && ((uintptr_t)x & 15) == 0 && ((uintptr_t)embed & 15) == 0) {
        const int nq = codebook_size / (kVqTileCodes * kVqIters);
        float* cv = static_cast<float*>(workspace);
        int* ci = reinterpret_cast<int*>(cv + n_vectors * nq);
        const size_t lds = sizeof(float) * ((size_t)(32 + kVqTileCodes) * (dim + 1) + 32 + kVqTileCodes + 4 * 32) + sizeof(int) * 4 * 32;
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
        });
        hipLaunchKernelGGL(vq_partial_kernel, dim3((unsigned)((n_vectors + 31) / 32), (unsigned)nq), dim3(256), lds,
                           (hipStream_t)stream, x, embed, (int)n_vectors, dim, codebook_size, cv, ci);
        if (int e = rh_check_launch("vq_partial")) return e;
        hipLaunchKernelGGL(vq_pick_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, embed, (const float*)cv,
                           (const int*)ci, nq, (int)n_vectors, dim, (long long*)indices, residual, quantized_sum, loss_partials);
        return rh_check_launch("vq_pick");
    }
    const size_t lds = sizeof(float) * ((size_t)kVecPerBlock * dim + kVecPerBlock + 4 * kVecPerBlock) +
                       sizeof(int) * (4 * kVecPerBlock + kVecPerBlock) + sizeof(float) * 4;
    hipLaunchKernelGGL(vq_assign_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, embed, (int)n_vectors, dim,
                       codebook_size, (long long*)indices, residual, quantized_sum, loss_partials);
    return rh_check_launch("vq_assign");
}

extern "C" int rh_vq_assign_f32(const float* x, const float* embed, int64_t n_vectors, int32_t dim,
                                int32_t codebook_size, int64_t* indices, float* residual, float* quantized_sum,
                                float* loss_partials, rh_stream_t stream) {
    return vq_assign_impl(x, embed, n_vectors, dim, codebook_size, indices, residual, quantized_sum, loss_partials, nullptr, 0, stream);
}

extern "C" int rh_vq_assign_ws_f32(const float* x, const float* embed, int64_t n_vectors, int32_t dim, int32_t codebook_size,
                                   int64_t* indices, float* residual, float* quantized_sum, float* loss_partials, void* workspace,
                                   int64_t workspace_bytes, rh_stream_t stream) {
    return vq_assign_impl(x, embed, n_vectors, dim, codebook_size, indices, residual, quantized_sum, loss_partials, workspace,
                          workspace_bytes, stream);
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
