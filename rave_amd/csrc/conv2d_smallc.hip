// General Conv2d FORWARD with few INPUT channels (C <= 4), stride 1, up to 32 output channels: the first conv of the
// spectral discriminators' stacks -- rave/discriminator.py:60 (2 n_channels -> capacity, (9,3)) and descript MRD
// (rave/descript_discriminator.py:137: 2 n_channels -> 32, (3,9)) -- whose "GEMM" has K = C * 27 <= 108: the 32-row f32
// MFMA tile ran it at 25-40 TFLOP/s (0.34-0.59 ms per scale, 2.6 ms of an Encodec pass), while it is one pass over a
// C_out x H x W tensor (525 MB written at BASELINE sizes).
// Vector-ALU kernel in the shape of conv2d_smallm.hip / conv_smallc.hip: lanes run along W (every store is a full 256-byte
// line per wave and output row), a thread owns 4 output rows x ALL output channels in registers (128 accumulators), the
// whole input patch (all C channels: 14 KB) and the weights [c][tw][th][m] (7 KB) sit in LDS.  Per (channel, tap column) a
// thread reads its 4 + KH - 1 strip values once and spends KH x 4 x 32 FMAs on them; the weights come as 16-byte LDS
// broadcasts (4 output channels per read).  Bias + LeakyReLU in the epilogue.
#include <cstdlib>
#include "conv_params.hpp"

namespace {

struct SmallC2P {
    const float* in;       // [B][C][H][W]
    const float* wp;       // f32 packed forward operand [slot = th * kw + tw][c][Mp = 32]
    const float* bias;     // [M] or null
    float* out;            // [B][M][out_h][out_w]
    int B, C, M;
    int in_h, in_w, out_h, out_w;
    int kw, dw, ph, pw;
    int out_act;
    float out_slope;
    int tiles_w, tiles_h;
    int PW;                // patch pitch = 64 + (kw - 1) * dw
};

constexpr int kScR = 4, kScTH = 16, kScTW = 64;

template <int KH>
__global__ __launch_bounds__(256) void conv2d_smallc_fwd_kernel(const SmallC2P p) {
    constexpr int PHt = kScTH + KH - 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const wsm = sm;                               // [C][kw][KH][32]
    float* const patch = sm + p.C * p.kw * KH * 32;     // [C][PHt][PW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x;
    const int tw_i = bx % p.tiles_w;
    bx /= p.tiles_w;
    const int th_i = bx % p.tiles_h;
    const int b = bx / p.tiles_h;
    const int h0 = th_i * kScTH, w0 = tw_i * kScTW;
    const int PW = p.PW;
    const float* __restrict__ in = p.in;
    // ---- weights: wsm[((c * kw + tw) * KH + th) * 32 + m] = wp[((th * kw + tw) * C + c) * 32 + m]
    for (int e = tid; e < p.C * p.kw * KH * 32; e += 256) {
        const int m = e & 31;
        int r = e >> 5;
        const int th = r % KH;
        r /= KH;
        const int tw = r % p.kw, c = r / p.kw;
        wsm[e] = p.wp[((long)(th * p.kw + tw) * p.C + c) * 32 + m];
    }
    // ---- the patch (all channels), rows round-robin over the waves, U rows x 2 column passes in flight
    {
        constexpr int U = 8;
        const int rows = p.C * PHt;
        for (int row0 = wave; row0 < rows; row0 += 4 * U) {
            float v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 4 * u;
                const int c = row / PHt, pr = row - c * PHt;
                const int gh = h0 - p.ph + pr;
                const bool rok = row < rows && gh >= 0 && gh < p.in_h;
                const float* src = in + (((long)b * p.C + c) * p.in_h + gh) * p.in_w;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pc = lane + 64 * k;
                    const int gw = w0 - p.pw + pc;
                    v[u][k] = (rok && pc < PW && gw >= 0 && gw < p.in_w) ? src[gw] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 4 * u;
                if (row >= rows) break;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pc = lane + 64 * k;
                    if (pc < PW) patch[row * PW + pc] = v[u][k];
                }
            }
        }
    }
    __syncthreads();
    float acc[kScR][32];
#pragma unroll
    for (int i = 0; i < kScR; ++i)
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[i][m] = 0.f;
    for (int c = 0; c < p.C; ++c) {
        for (int tw = 0; tw < p.kw; ++tw) {
            const float* col = patch + (c * PHt + wave * kScR) * PW + lane + tw * p.dw;
            float s[kScR + KH - 1];
#pragma unroll
            for (int k = 0; k < kScR + KH - 1; ++k) s[k] = col[k * PW];
            const f32x4* wl = reinterpret_cast<const f32x4*>(wsm + (c * p.kw + tw) * KH * 32);     // wave-uniform: LDS broadcasts
#pragma unroll
            for (int th = 0; th < KH; ++th) {
#pragma unroll
                for (int mq = 0; mq < 8; ++mq) {
                    const f32x4 w4 = wl[th * 8 + mq];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < kScR; ++i) acc[i][4 * mq + k] = fmaf(w4[k], s[i + th], acc[i][4 * mq + k]);
                }
            }
        }
    }
    const int w = w0 + lane;
    if (w >= p.out_w) return;
    const bool leaky = p.out_act == RH_ACT_LEAKY;
    // (fully unrolled, no early exit: a loop the compiler cannot unroll would index the accumulators dynamically = scratch)
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const bool mok = m < p.M;
        const float bv = (mok && p.bias) ? p.bias[m] : 0.f;
#pragma unroll
        for (int i = 0; i < kScR; ++i) {
            const int r = h0 + wave * kScR + i;
            float v = acc[i][m] + bv;
            if (leaky) v = v > 0.f ? v : v * p.out_slope;
            if (mok && r < p.out_h) p.out[(((long)b * p.M + m) * p.out_h + r) * p.out_w + w] = v;
        }
    }
}

bool smallc2_enabled() {
    const char* e = getenv("RH_CONV2D_SMALLM");    // (one switch for both vector-ALU Conv2d kernels; read per call: tests)
    return !(e && atoi(e) == 0);
}

}  // namespace

bool rh_conv2d_smallc_fwd_eligible(const rh_conv2d_desc* d) {
    if (!smallc2_enabled() || d->batch <= 0) return false;
    if (d->c_in > 4 || d->c_out > 32 || d->c_out <= 4 || d->sh != 1 || d->sw != 1 || d->dh != 1 || (d->kh != 3 && d->kh != 9)) return false;
    if (64 + (d->kw - 1) * d->dw > 128) return false;
    const long tiles = (long)d->batch * rh_cdiv(d->h_out, kScTH) * rh_cdiv(d->w_out, kScTW);
    const size_t lds = (size_t)d->c_in * (d->kw * d->kh * 32 + (kScTH + d->kh - 1) * (64 + (d->kw - 1) * d->dw)) * 4;
    return tiles < 0x7fffffffl && lds <= 64 * 1024;
}

int rh_conv2d_smallc_fwd_launch(const rh_conv2d_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y,
                                hipStream_t stream) {
    SmallC2P p{};
    p.in = x; p.wp = wp_fwd; p.bias = bias; p.out = y;
    p.B = d->batch; p.C = d->c_in; p.M = d->c_out;
    p.in_h = d->h_in; p.in_w = d->w_in; p.out_h = d->h_out; p.out_w = d->w_out;
    p.kw = d->kw; p.dw = d->dw; p.ph = d->ph; p.pw = d->pw;
    p.out_act = d->act; p.out_slope = d->act_slope;
    p.tiles_w = rh_cdiv(p.out_w, kScTW); p.tiles_h = rh_cdiv(p.out_h, kScTH);
    p.PW = kScTW + (d->kw - 1) * d->dw;
    const size_t lds = (size_t)p.C * (d->kw * d->kh * 32 + (kScTH + d->kh - 1) * p.PW) * 4;
    const dim3 grid((unsigned)((long)p.B * p.tiles_h * p.tiles_w));
    if (d->kh == 9) rh_launch_main(conv2d_smallc_fwd_kernel<9>, grid, dim3(256), lds, stream, p);
    else rh_launch_main(conv2d_smallc_fwd_kernel<3>, grid, dim3(256), lds, stream, p);
    return rh_check_launch("conv2d_smallc_fwd");
}
