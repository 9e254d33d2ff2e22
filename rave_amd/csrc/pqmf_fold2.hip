// 16-band PQMF in the folded fast form, second generation (round 3): the same two kernels as pqmf_fold.hip -- same
// arithmetic, same fmaf chains in the same order, BIT-IDENTICAL outputs (tests compare with torch.equal) -- restructured
// for the machine instead of for one frame per lane:
//
//   * the 16 x 32 cosine matrix runs on the matrix cores: v_mfma_f32_16x16x4_f32 is an exact f32 fmaf chain in k order
//     (cdna_hip_programming.md section 3), 8 instructions per 16 frames, the matrix a per-lane CONSTANT operand (8
//     registers) -- pqmf_fold.hip issued 512 vector FMAs per frame with 512 scalar operands;
//   * the 384-tap fold is 32 x 2 independent 12-tap FIRs (one per fold column m and frame parity: frames n and n + 2
//     read the same samples one tap apart).  A lane owns one (m, parity) pair: 27 LDS reads feed 16 frames x 12 FMAs
//     (pqmf_fold.hip: 384 reads per frame, every sample re-read by 24 lanes), conflict-free (m is the lane index);
//   * the overlap-add of the synthesis direction slides the same way: a lane owns a residue class and 7 consecutive hops,
//     60 LDS reads for 7 x 24 FMAs (was 24 reads per sample);
//   * results go through an LDS staging tile so that every global store instruction writes whole 512-byte band rows /
//     16-byte aligned sample runs.
// Algorithmic traffic 8 bytes per sample (16.8 MB per direction at batch 32 x 65536): HBM-bound by construction; what is
// left besides the two passes is one workgroup-level load -> compute -> store latency chain.
#include "common.hpp"

namespace {

constexpr int kTaps = 384;
constexpr unsigned kOOB = 0x80000000u;

// ---------------------------------------------------------------------------------------------------------------- K1
//   out[k][n] = s(k,n) * scale * sum_m Cm[k][m] * w[n][m],   w[n][m] = sum_j hs[m + 32 j] * in[16 n + m + 32 j + o0]
constexpr int kF1 = 128;                                  // frames per workgroup (4 waves x 32)
constexpr int kSeg1 = 16 * (kF1 - 1) + kTaps;             // samples a workgroup folds: 2416
constexpr int kWP = 34;                                   // pitch of a frame's 32 fold values in LDS (conflict-free MFMA reads)
constexpr int kOP = kF1 + 4;                              // pitch of a band row in the output staging tile

// out_range (round 6, optional): max |out| goes to that range slot (common.hpp) -- the first convolution of the encoder reads its
// scale from there instead of from a separate rh_amax_f32 pass over the band tensor.
__global__ __launch_bounds__(256) void pqmf_fold_k1v2_kernel(const float* __restrict__ in, const float* __restrict__ tab,
                                                             float* __restrict__ out, int t_len, int n_frames, int o0,
                                                             float scale, unsigned* __restrict__ out_range) {
    __shared__ float red_pub[4];
    float amax = 0.f;
    __shared__ __attribute__((aligned(16))) float xs[(kSeg1 + 255) / 256 * 256];
    __shared__ __attribute__((aligned(16))) float wb[kF1 * kWP];
    __shared__ __attribute__((aligned(16))) float ob[16 * kOP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.y;
    const int n0 = blockIdx.x * kF1;
    const int g0 = 16 * n0 + o0;
    {   // ---- the waveform segment, coalesced; all loads of a thread in flight before the first LDS store
        const auto src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (long)row * t_len), 0, (unsigned)t_len * 4u, 0x00020000);
        constexpr int kLoads = (kSeg1 + 255) / 256;
        float stage[kLoads];
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int s = tid + 256 * i;
            const int gi = g0 + s;
            stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, (s < kSeg1 && gi >= 0) ? (unsigned)gi * 4u : kOOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < kLoads; ++i) xs[tid + 256 * i] = stage[i];
    }
    // per-lane constants (L2-resident table): 12 prototype taps of fold column m, 8 matrix entries of the B operand
    const int m = lane & 31, par = lane >> 5;
    float hsr[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) hsr[j] = tab[m + 32 * j];
    const int l16 = lane & 15, lg = lane >> 4;
    float cmr[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) cmr[s] = tab[kTaps + l16 * 32 + 4 * s + lg];
    __syncthreads();
    {   // ---- fold: lane = (column m, frame parity), wave = 32 frames; u[i] = sample of frame 2 i + par at tap group 0
        const float* ub = xs + 512 * wave + 16 * par + m;
        float u[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) u[i] = ub[32 * i];
        float* wd = wb + (32 * wave + par) * kWP + m;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) acc = fmaf(hsr[j], u[f + j], acc);
            wd[2 * f * kWP] = acc;
        }
    }
    __syncthreads();
    // ---- matrix on the matrix cores: D[frame][band] = sum_m w[frame][m] Cm[band][m]  (exact f32, m ascending)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float* wa = wb + (32 * wave + 16 * nb + l16) * kWP + lg;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * s], cmr[s], acc, 0, 0, 0);
        // lane: band l16, frames 32 wave + 16 nb + 4 lg + r;  reverse_half: -1 iff band odd and frame even
        const float se = (l16 & 1) ? -scale : scale;
        f32x4 o = {acc[0] * se, acc[1] * scale, acc[2] * se, acc[3] * scale};
        *reinterpret_cast<f32x4*>(ob + l16 * kOP + 32 * wave + 16 * nb + 4 * lg) = o;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            if (n0 + 32 * wave + 16 * nb + 4 * lg + r4 < n_frames) amax = fmaxf(amax, fabsf(o[r4]));
    }
    __syncthreads();
    // ---- whole band rows out: 16 bands x 128 frames, 16 bytes per lane
    const auto dst = __builtin_amdgcn_make_buffer_rsrc(out + (long)row * 16 * n_frames, 0, (unsigned)(16 * n_frames) * 4u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + 256 * it;
        const int band = q >> 5, fq = (q & 31) * 4;
        const int n = n0 + fq;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ob + band * kOP + fq);
        const unsigned base = (unsigned)(band * n_frames + n) * 4u;
        if (n + 3 < n_frames) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), dst, base, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < n_frames) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), dst, base + 4u * e, 0, 0);
        }
    }
    if (out_range) rh_range_publish(out_range, amax, blockIdx.x + blockIdx.y * 7u, red_pub);      // (uniform)
}

// ---------------------------------------------------------------------------------------------------------------- K2
//   out[q] = scale * sum_{n'} hs[tau] * g[n'][tau % 32], tau = q - 16 n' + dp in [0, 384),
//   g[n'][m] = sum_c Cm[c][m] * s(c,n') * in[c][n']
constexpr int kFr2 = 224;           // output hops (x16 samples) per workgroup
constexpr int kNF = 256;            // input frames whose contributions reach them (4 waves x 4 blocks of 16)
constexpr int kGP = 34;             // LDS pitch of a frame's 32 g values

__global__ __launch_bounds__(256) void pqmf_fold_k2v2_kernel(const float* __restrict__ in, const float* __restrict__ tab,
                                                             float* __restrict__ out, int n_frames, int n_out, int dp,
                                                             float scale) {
    __shared__ __attribute__((aligned(16))) float gl[kNF * kGP];
    __shared__ __attribute__((aligned(16))) float ob[16 * kFr2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.y;
    const int n0 = blockIdx.x * kFr2;
    const float* __restrict__ hs = tab;
    const int a = dp - (kTaps - 1);
    const int fl = (a >= 0 ? a : a - 15) / 16;                    // floor(a / 16)
    const int nlo = n0 + fl;
    const int l16 = lane & 15, lg = lane >> 4;
    {   // ---- phase 1 on the matrix cores: G[frame][m] = sum_c (s(c,frame) in[c][frame]) Cm[c][m]  (c ascending, exact f32)
        const auto src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (long)row * 16 * n_frames), 0, (unsigned)(16 * n_frames) * 4u, 0x00020000);
        float cmr[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) cmr[s][h] = tab[kTaps + (4 * s + lg) * 32 + l16 + 16 * h];
        float av[4][4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int np = nlo + 64 * wave + 16 * nb + l16;
            const bool ok = np >= 0 && np < n_frames;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                av[nb][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, ok ? (unsigned)((4 * s + lg) * n_frames + np) * 4u : kOOB, 0, 0));
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int np = nlo + 64 * wave + 16 * nb + l16;
            const bool neg = (lg & 1) && !(np & 1);               // band c = 4 s + lg odd, frame even
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v = neg ? -av[nb][s] : av[nb][s];
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, cmr[s][h], acc[h], 0, 0, 0);
            }
            float* gd = gl + (64 * wave + 16 * nb + 4 * lg) * kGP + l16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gd[r * kGP] = acc[0][r];
                gd[r * kGP + 16] = acc[1][r];
            }
        }
    }
    // ---- phase 2: a lane owns a residue r (sample % 16) and 7 consecutive hops.  tau = r + e - 16 (jj + f0) takes the same
    // 24 values for every sample of a residue class; the g column alternates between two fixed values ma / mb.
    const int e = dp - 16 * fl;                                   // tau = ql + e - 16 f,  e in [383, 398]
    const int r = tid & 15;
    const int top0 = r + e;
    const int f0 = top0 >= kTaps ? (top0 - (kTaps - 1) + 15) >> 4 : 0;
    float hreg[24];
#pragma unroll
    for (int jj = 0; jj < 24; ++jj) {
        const int tau = top0 - 16 * (f0 + jj);
        hreg[jj] = tau >= 0 ? hs[tau] * scale : 0.f;
    }
    const int ma = (top0 - 16 * f0) & 31, mb = (top0 - 16 * f0 - 16) & 31;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int seg = (tid >> 4) + 16 * it;                     // 32 segments of 7 hops
        const float* gb0 = gl + (7 * seg + f0) * kGP;
        float ga[30], gb[30];
#pragma unroll
        for (int i = 0; i < 30; ++i) {
            ga[i] = gb0[i * kGP + ma];
            gb[i] = gb0[i * kGP + mb];
        }
#pragma unroll
        for (int uu = 0; uu < 7; ++uu) {
            float acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < 24; ++jj) acc = fmaf(hreg[jj], (jj & 1) ? gb[uu + jj] : ga[uu + jj], acc);
            ob[(7 * seg + uu) * 16 + r] = acc;
        }
    }
    __syncthreads();
    // ---- 3584 contiguous samples out, 16 bytes per lane
    const auto dst = __builtin_amdgcn_make_buffer_rsrc(out + (long)row * n_out, 0, (unsigned)n_out * 4u, 0x00020000);
    const int q0 = 16 * n0;
#pragma unroll
    for (int it = 0; it < (16 * kFr2 / 4 + 255) / 256; ++it) {
        const int qd = tid + 256 * it;
        if (qd >= 16 * kFr2 / 4) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ob + 4 * qd);
        const int q = q0 + 4 * qd;
        if (q + 3 < n_out) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), dst, (unsigned)q * 4u, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q + k < n_out) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), dst, (unsigned)(q + k) * 4u, 0, 0);
        }
    }
}

}  // namespace

int rh_pqmf_fold_k1v2_launch(const float* in, const float* tab, int rows, int t_len, int n_frames, int o0, float scale, float* out,
                             hipStream_t stream, unsigned* out_range) {
    hipLaunchKernelGGL(pqmf_fold_k1v2_kernel, dim3(rh_cdiv(n_frames, kF1), rows), dim3(256), 0, stream, in, tab, out, t_len,
                       n_frames, o0, scale, out_range);
    return rh_check_launch("pqmf_fold_k1");
}

int rh_pqmf_fold_k2v2_launch(const float* in, const float* tab, int rows, int n_frames, int n_out, int dp, float scale, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(pqmf_fold_k2v2_kernel, dim3(rh_cdiv(n_out, 16 * kFr2), rows), dim3(256), 0, stream, in, tab, out, n_frames,
                       n_out, dp, scale);
    return rh_check_launch("pqmf_fold_k2");
}
