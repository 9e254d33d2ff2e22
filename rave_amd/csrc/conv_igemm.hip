// Implicit-GEMM 1-D convolution on the f32-input matrix cores (v_mfma_f32_32x32x2_f32):
// forward of Conv1d / ConvTranspose1d / Conv2d-(k,1) and the data gradient of all three, with
//   * the pre-activation (LeakyReLU / Snake) fused into the LDS staging of the input tile,
//   * asymmetric / causal zero padding, dilation, stride and MPD "fold" padding done in-kernel
//     (the reference materialises an F.pad copy per conv -- SURVEY.md section 2.2),
//   * bias, residual add and the activation derivative fused into the epilogue.
//
// GEMM view:  M = output channels, N = output positions (batch folded in for short sequences),
// K = (tap, input channel).  A = packed weights [tap][c][M] (M innermost -> conflict-free,
// 16-byte staging), B = input tile [c][position] staged once per channel chunk and re-read for
// every tap (the im2col matrix is never materialised).  Exact f32: results are k-ordered fmaf
// chains, same numerics class as the reference's CPU path.
#include <mutex>
#include "conv_params.hpp"

namespace {

__device__ __forceinline__ int ibase_of(int n, int inner, int is) {
    if (inner == 1) return n * is;
    const int r = n / inner;
    return r * is * inner + (n - r * inner);
}
__device__ __forceinline__ int oidx_of(int n, int inner, int os, int oph) {
    if (inner == 1) return n * os + oph;
    const int r = n / inner;
    return (r * os + oph) * inner + (n - r * inner);
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvP p) {
    constexpr int BM = TM * WM * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w_lds = smem;
    float* x_lds = smem + p.wlds_floats;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    const int phase = blockIdx.z;
    const int ntaps = p.ph_ntaps[phase];
    const int tap0 = p.ph_tap0[phase];
    const int minoff = p.ph_minoff[phase];
    const int maxoff = p.ph_maxoff[phase];
    const int oph = p.ph_oph[phase];
    const float* __restrict__ wp = p.wp + p.ph_wofs[phase];

    const int bt = blockIdx.x / p.tiles_per_b;
    const int nt = blockIdx.x - bt * p.tiles_per_b;
    const int b0 = bt * p.nb;
    const int n0 = nt * p.bnl;
    const int m0 = blockIdx.y * BM;
    const int inner = p.inner, is = p.is;

    const int nlast = min(n0 + p.bnl, p.ncols) - 1;
    const int ib0 = ibase_of(n0, inner, is);
    const int lo = ib0 + minoff * inner;
    const int width = ibase_of(nlast, inner, is) - ib0 + (maxoff - minoff) * inner + 1;

    int xb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = min(n0 + nl, p.ncols - 1);
        xb[tn] = (bl * p.ck + kh) * p.pitch + ibase_of(n, inner, is) - ib0;
    }
    const int arow = wm * TM * 32 + j + kh * BM;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    for (int c0 = 0; c0 < p.C; c0 += p.ck) {
        __syncthreads();
        // ---- stage the input tile (activation fused) ----
        const int nrows = p.nb * p.ck;
        for (int r = wave; r < nrows; r += NW) {
            const int bl = r / p.ck, c = r - bl * p.ck;
            const int b = b0 + bl, ch = c0 + c;
            float* dst = x_lds + r * p.pitch;
            if (b < p.B && ch < p.C) {
                const float* __restrict__ src = p.in + ((long)b * p.C + ch) * p.in_row;
                const float alpha = (p.in_act == RH_ACT_SNAKE) ? p.in_alpha[ch] : 0.f;
                for (int w = lane; w < width; w += 64) {
                    const int f = lo + w;
                    const float v = (f >= 0 && f < p.in_valid) ? src[f] : 0.f;
                    dst[w] = rh_act_apply(v, p.in_act, p.in_slope, alpha);
                }
            } else {
                for (int w = lane; w < width; w += 64) dst[w] = 0.f;
            }
        }
        // ---- stage the weight tile [tap][c][BM] ----
        constexpr int V = BM / 4;
        const int wrows = ntaps * p.ck;
        for (int e = tid; e < wrows * V; e += NT) {
            const int kr = e / V, v4 = e - kr * V;
            const int t = kr / p.ck, c = kr - t * p.ck;
            const int ch = c0 + c;
            const int m = m0 + v4 * 4;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (ch < p.C && m < p.Mp)
                val = *reinterpret_cast<const f32x4*>(wp + ((long)t * p.C + ch) * p.Mp + m);
            *reinterpret_cast<f32x4*>(w_lds + kr * BM + v4 * 4) = val;
        }
        __syncthreads();
        // ---- MFMA over (tap, channel pair) ----
        for (int t = 0; t < ntaps; ++t) {
            const int toff = (p.off[tap0 + t] - minoff) * inner;
            const float* wl = w_lds + t * p.ck * BM + arow;
            const float* xl = x_lds + toff;
            for (int c = 0; c < p.ck; c += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = wl[c * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = xl[xb[tn] + c * p.pitch];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, activation derivative, residual / gradient add ----
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = n0 + nl, b = b0 + bl;
        if (n >= p.ncols || b >= p.B) continue;
        const int oi = oidx_of(n, inner, p.os, oph);
        if (oi >= p.out_valid) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) {
                    const long idx = ((long)b * p.M + m) * p.out_row + oi;
                    float v = acc[tm][tn][r];
                    if (p.bias) v += p.bias[m];
                    if (p.mul_src) {
                        const float al = (p.epi_act == RH_ACT_SNAKE) ? p.mul_alpha[m] : 0.f;
                        v *= rh_act_grad(p.mul_src[idx], p.epi_act, p.epi_slope, al);
                    }
                    if (p.add) v += p.add[idx];
                    if (p.out_act == RH_ACT_LEAKY) v = v > 0.f ? v : v * p.out_slope;
                    p.out[idx] = v;
                }
            }
        }
    }
}

template <int TM, int TN, int WM, int WN>
int launch_cfg(ConvP& p, hipStream_t stream, const char* what) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    // column fold: nb batch items x bnl columns per tile
    int bnl = BN;
    if (p.ncols < BN) {
        bnl = 32;
        while (bnl < p.ncols) bnl <<= 1;
    }
    p.bnl = bnl;
    p.bnl_shift = __builtin_ctz(bnl);
    p.nb = BN / bnl;
    p.tiles_per_b = rh_cdiv(p.ncols, bnl);
    int span = 0, maxtaps = 0;
    for (int i = 0; i < p.nphase; ++i) {
        span = span > p.ph_maxoff[i] - p.ph_minoff[i] ? span : p.ph_maxoff[i] - p.ph_minoff[i];
        maxtaps = maxtaps > p.ph_ntaps[i] ? maxtaps : p.ph_ntaps[i];
    }
    if (p.inner == 1)
        p.pitch = (bnl - 1) * p.is + span + 1;
    else
        p.pitch = ((bnl - 1) / p.inner + 1) * p.is * p.inner + p.inner + span * p.inner + 1;
    p.pitch |= 1;  // odd pitch: the two k-halves of a wave never share a bank row start
    const int budget = 15 * 1024;  // floats (60 KiB -> two workgroups per CU)
    const int per_ch = (maxtaps > 0 ? maxtaps : 1) * BM + p.nb * p.pitch;
    int ck = budget / per_ch;
    ck &= ~1;
    if (ck > 32) ck = 32;
    if (ck < 2) ck = 2;
    const int cmax = (p.C + 1) & ~1;
    if (ck > cmax) ck = cmax;
    p.ck = ck;
    p.wlds_floats = (maxtaps > 0 ? maxtaps : 1) * ck * BM;
    const size_t lds = sizeof(float) * ((size_t)p.wlds_floats + (size_t)p.nb * ck * p.pitch);
    RH_REQUIRE(lds <= 160 * 1024, RH_ERR_UNSUPPORTED, "%s: tile needs %zu B of LDS", what, lds);
    auto kern = conv_igemm_kernel<TM, TN, WM, WN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    dim3 grid(rh_cdiv(p.B, p.nb) * p.tiles_per_b, rh_cdiv(p.M, BM), p.nphase);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    return rh_check_launch(what);
}

}  // namespace

int rh_conv_launch(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes) {
    if (p.B <= 0 || p.ncols <= 0 || p.M <= 0) return RH_OK;
    if (rh_conv_dma_eligible(p)) return rh_conv_launch_dma(p, stream, what, ws, ws_bytes);
    return rh_conv_launch_sync(p, stream, what);
}

int rh_conv_launch_sync(ConvP& p, hipStream_t stream, const char* what) {
    if (p.B <= 0 || p.ncols <= 0 || p.M <= 0) return RH_OK;
    int rc;
    if (p.M <= 32) rc = launch_cfg<1, 2, 1, 4>(p, stream, what);
    else if (p.M <= 64) rc = launch_cfg<2, 1, 1, 4>(p, stream, what);
    else if (p.M % 96 == 0 || p.M < 96) rc = launch_cfg<3, 1, 1, 4>(p, stream, what);
    else rc = launch_cfg<2, 2, 2, 2>(p, stream, what);
    return rc ? rc : rh_range_after(p, stream);
}

// A launch of a kernel family that does not publish the output's range slot (everything but conv_x6_kernel and its finalize
// pass): fill a requested slot with one pass over the output.
int rh_range_after(const ConvP& p, hipStream_t stream) {
    if (!p.out_range) return RH_OK;
    return rh_amax_f32(p.out, (int64_t)p.B * (p.Mr > 0 && p.vs > 1 ? p.Mr : p.M) * p.out_row, p.out_range, (rh_stream_t)stream);
}
