// 16-band PQMF analysis / synthesis in the FOLDED fast form (rave/pqmf.py:32-52,245-294).
//
// The bank is cosine modulated: hk[k, lpad + tau] = 2 h[tau] cos((2k+1) pi/32 (tau - 188) + (-1)^k pi/4), and
// cos(theta_k(m + 32 j)) = (-1)^j cos(theta_k(m)).  With hs[tau] = (-1)^(tau/32) h[tau] (the "signed prototype") and
// Cm[k][m] = 2 cos(theta_k(m)) (a 16 x 32 matrix) both directions split into a 384-tap fold shared by all bands and a
// 16 x 32 matrix product per frame: 889 MAC per frame of 16 samples = 55.6 MAC/sample instead of 513 -- the transform
// becomes what it algorithmically is, one pass over contiguous waveform frames (8 bytes per sample: HBM-bound).
//
//   fold -> matrix  (analysis forward, synthesis backward):
//       out[k][n] = s(k,n) * scale * sum_m Cm[k][m] * w[n][m],   w[n][m] = sum_j hs[m + 32 j] * in[16 n + m + 32 j + o0]
//   matrix -> overlap-add  (synthesis forward, analysis backward):
//       out[q] = scale * sum_{n'} hs[tau] * g[n'][tau % 32],  tau = q - 16 n' + dp in [0, 384),
//       g[n'][m] = sum_c Cm[c][m] * s(c,n') * in[c][n']
//   s(k,n) = -1 iff k odd and n even  (reverse_half, rave/pqmf.py:13-17: bit-exact sign pattern)
//
// Lanes are FRAMES (64 frames per wave): the fold reads a skewed LDS image of the waveform (pitch 17 words per hop of
// 16: conflict-free), the prototype and the cosine matrix are wave-uniform scalars, so neither the fold nor the matrix
// product needs a cross-lane reduction; stores are coalesced along the frame axis.  The direct-form MFMA kernels
// (pqmf.hip) stay as the bit-level reference for arbitrary `forward_conv.weight` contents; the host side only selects
// this form when the stored bank equals the closed form to 5e-6 (rave_amd/pqmf.py).
#include <cstdlib>
#include "common.hpp"

namespace {

constexpr int kTaps = 384;          // prototype taps, zero padded (377 for 100 dB / 16 bands)
constexpr int kFr1 = 256;           // frames per workgroup of the fold -> matrix kernel (one per lane; two per lane halves
                                    // the workgroup count to one per CU and measured 1.6x slower)

// tab: hs[384] followed by Cm[16][32]
__global__ __launch_bounds__(256) void pqmf_fold_k1_kernel(const float* __restrict__ in, const float* __restrict__ tab,
                                                           float* __restrict__ out, int t_len, int n_frames, int o0,
                                                           float scale) {
    __shared__ float xs[(16 * kFr1 + kTaps) / 16 * 17 + 17];
    const int tid = threadIdx.x;
    const int row = blockIdx.y;
    const int n0 = blockIdx.x * kFr1;
    const float* __restrict__ src = in + (long)row * t_len;
    const int g0 = 16 * n0 + o0;
    // all 18 loads of a thread in flight before the first LDS store (a rolled loop waits for each load in turn: the
    // kernel then runs at 17 memory latencies, 36 us in the training step against 14 us with the waveform in cache)
    constexpr int kSpan = 16 * kFr1 + kTaps, kLoads = (kSpan + 255) / 256;
    float stage[kLoads];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const int s = tid + 256 * i;
        const int gi = g0 + s;
        stage[i] = (s < kSpan && gi >= 0 && gi < t_len) ? src[gi] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const int s = tid + 256 * i;
        if (s < kSpan) xs[s + (s >> 4)] = stage[i];
    }
    __syncthreads();
    const float* __restrict__ hs = tab;
    const float* __restrict__ cm = tab + kTaps;
    const float* xl = xs + 17 * tid;
    float w[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) w[m] = 0.f;
#pragma unroll
    for (int t = 0; t < kTaps; ++t) w[t & 31] = fmaf(hs[t], xl[t + (t >> 4)], w[t & 31]);
    const int n = n0 + tid;
    if (n >= n_frames) return;
    float* __restrict__ dst = out + (long)row * 16 * n_frames + n;
    const float neg = (n & 1) ? scale : -scale;       // odd band, even frame: -1
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float y = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) y = fmaf(cm[k * 32 + m], w[m], y);
        dst[(long)k * n_frames] = y * ((k & 1) ? neg : scale);
    }
}

constexpr int kFr2 = 224;           // output frames (x16 samples) per workgroup of the matrix -> overlap-add kernel
constexpr int kNF = 256;            // frames whose contributions reach them: 224 + 24 (+ alignment) <= 256 = one per thread

__global__ __launch_bounds__(256) void pqmf_fold_k2_kernel(const float* __restrict__ in, const float* __restrict__ tab,
                                                           float* __restrict__ out, int n_frames, int n_out, int dp,
                                                           float scale) {
    __shared__ float gl[kNF * 33];
    const int tid = threadIdx.x;
    const int row = blockIdx.y;
    const int n0 = blockIdx.x * kFr2;
    const float* __restrict__ hs = tab;
    const float* __restrict__ cm = tab + kTaps;
    // frames n' with tau = q - 16 n' + dp in [0, 384) for q in [16 n0, 16 (n0 + 224)): n' >= n0 + floor((dp - 383) / 16)
    const int a = dp - (kTaps - 1);
    const int fl = (a >= 0 ? a : a - 15) / 16;                    // floor(a / 16)
    const int nlo = n0 + fl;
    const float* __restrict__ src = in + (long)row * 16 * n_frames;
    {   // phase 1: lane = frame; g[m] = sum_c Cm[c][m] * s(c,n') * in[c][n']  (Cm: wave-uniform scalars)
        const int np = nlo + tid;
        float g[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) g[m] = 0.f;
        if (np >= 0 && np < n_frames) {
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                v[c] = src[(long)c * n_frames + np];
                if ((c & 1) && !(np & 1)) v[c] = -v[c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int m = 0; m < 32; ++m) g[m] = fmaf(cm[c * 32 + m], v[c], g[m]);
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) gl[tid * 33 + m] = g[m];
    }
    // phase 2: lane = (residue r = sample % 16, hop block); tau = r + e - 16 (jj + f0) takes the same 24 values for
    // every sample of a residue class, so the 24 prototype taps live in registers and the g column alternates between
    // two fixed values: 24 LDS reads (immediate offsets) + 24 FMAs per output sample
    const int e = dp - 16 * fl;                                   // tau = ql + e - 16 f,  e in [383, 398]
    const int r = tid & 15;
    const int top0 = r + e;                                       // tau at hop block 0, frame 0
    const int f0 = top0 >= kTaps ? (top0 - (kTaps - 1) + 15) >> 4 : 0;
    float hreg[24];
#pragma unroll
    for (int jj = 0; jj < 24; ++jj) {
        const int tau = top0 - 16 * (f0 + jj);
        hreg[jj] = tau >= 0 ? hs[tau] * scale : 0.f;
    }
    const int ma = (top0 - 16 * f0) & 31, mb = (top0 - 16 * f0 - 16) & 31;
    __syncthreads();
    float* __restrict__ dst = out + (long)row * n_out;
#pragma unroll 2
    for (int i = 0; i < kFr2 / 16; ++i) {
        const int blk = (tid >> 4) + 16 * i;                      // hop block inside the workgroup
        const int q = 16 * (n0 + blk) + r;
        const float* ga = gl + (blk + f0) * 33 + ma;              // frame blk + f0 + jj, jj even
        const float* gb = gl + (blk + f0) * 33 + mb;              // jj odd
        float acc = 0.f;
#pragma unroll
        for (int jj = 0; jj < 24; ++jj) acc = fmaf(hreg[jj], (jj & 1) ? gb[jj * 33] : ga[jj * 33], acc);
        if (q < n_out) dst[q] = acc;
    }
}

bool fold_v2() {
    const char* e = getenv("RH_PQMF_V2");       // read per call (tests): 0 = the first-generation kernels of this file
    return !(e && atoi(e) == 0);
}

}  // namespace

// second generation (pqmf_fold2.hip): matrix on the MFMA, sliding-window fold / overlap-add -- bit-identical outputs
int rh_pqmf_fold_k1v2_launch(const float* in, const float* tab, int rows, int t_len, int n_frames, int o0, float scale, float* out,
                             hipStream_t stream, unsigned* out_range);
int rh_pqmf_fold_k2v2_launch(const float* in, const float* tab, int rows, int n_frames, int n_out, int dp, float scale, float* out,
                             hipStream_t stream);

extern "C" int rh_pqmf_fold_k1_f32(const float* in, const float* tab, int32_t rows, int32_t t_len, int32_t n_frames,
                                   int32_t o0, float scale, float* out, rh_stream_t stream) {
    unsigned* out_range = nullptr;             // where max |out| goes, if the caller armed an output slot (rh_x6_set_ranges)
    rh_take_ranges(nullptr, nullptr, &out_range, nullptr);
    RH_REQUIRE(rows >= 0 && t_len >= 0 && n_frames >= 0, RH_ERR_INVALID, "pqmf_fold_k1: bad sizes");
    if (rows == 0 || n_frames == 0) return RH_OK;
    RH_REQUIRE(in && tab && out, RH_ERR_INVALID, "pqmf_fold_k1: null pointer");
    if (fold_v2() && (long)t_len * 4 < 0x7fffffffl && (long)n_frames * 64 < 0x7fffffffl)
        return rh_pqmf_fold_k1v2_launch(in, tab, rows, t_len, n_frames, o0, scale, out, (hipStream_t)stream, out_range);
    hipLaunchKernelGGL(pqmf_fold_k1_kernel, dim3(rh_cdiv(n_frames, kFr1), rows), dim3(256), 0, (hipStream_t)stream, in, tab,
                       out, t_len, n_frames, o0, scale);
    if (int e = rh_check_launch("pqmf_fold_k1")) return e;
    // (the first-generation kernel does not publish: a requested slot is filled by a pass over the output)
    return out_range ? rh_amax_f32(out, (int64_t)rows * 16 * n_frames, out_range, stream) : RH_OK;
}

extern "C" int rh_pqmf_fold_k2_f32(const float* in, const float* tab, int32_t rows, int32_t n_frames, int32_t n_out,
                                   int32_t dp, float scale, float* out, rh_stream_t stream) {
    RH_REQUIRE(rows >= 0 && n_frames >= 0 && n_out >= 0, RH_ERR_INVALID, "pqmf_fold_k2: bad sizes");
    if (rows == 0 || n_out == 0) return RH_OK;
    RH_REQUIRE(in && tab && out, RH_ERR_INVALID, "pqmf_fold_k2: null pointer");
    if (fold_v2() && (long)n_out * 4 < 0x7fffffffl && (long)n_frames * 64 < 0x7fffffffl)
        return rh_pqmf_fold_k2v2_launch(in, tab, rows, n_frames, n_out, dp, scale, out, (hipStream_t)stream);
    hipLaunchKernelGGL(pqmf_fold_k2_kernel, dim3(rh_cdiv(n_out, 16 * kFr2), rows), dim3(256), 0, (hipStream_t)stream, in, tab,
                       out, n_frames, n_out, dp, scale);
    return rh_check_launch("pqmf_fold_k2");
}
