// General Conv2d with FEW OUTPUT ROWS (M <= 4), stride 1: the two ends of the spectral discriminators' Conv2d stacks
//   * data gradient of the first conv (rave/discriminator.py:60: 2 n_channels -> capacity, (9,3); descript MRD
//     rave/descript_discriminator.py:137: (3,9)): dx has 2 .. 4 channels, K = capacity x 27 taps;
//   * forward of the scoring conv (rave/discriminator.py:65-67, descript conv_post :146: capacity -> 1, (3,3)).
// A 32-row MFMA tile is 3 - 12 % used there (round 3: 4 TFLOP/s, 3.5 ms per data gradient -- 17.7 of the 108 ms Encodec
// pass).  This is vector-ALU code shaped like conv_smallc.hip: lanes run along W (every global access is a full line per
// wave), a thread owns a strip of 8 output rows x M channels in registers, the input patch of a channel chunk sits in LDS,
// and the weights of the chunk -- identical for every lane -- next to it (read as LDS broadcasts: as scalar loads from the
// packed operand, one 128-byte line per (tap, channel), they missed the scalar cache on every tap).  Per (channel, tap
// column) a thread reads its 8 + KH - 1 strip values once and spends KH x 8 x M FMAs on them.
// FLOPs: 2 B M C kh kw H W, priced against the f32 vector peak (157.3 TFLOP/s with packed FMA, 78.6 plain).
#include <cstdlib>
#include "conv_params.hpp"

namespace {

struct SmallM2P {
    const float* in;       // [B][C][in_h][in_w]
    const float* wp;       // f32 packed operand [slot = th * kw + tw][c][Mp]  (rows m >= M are zero)
    const float* bias;     // [M] or null
    float* out;            // [B][M][out_h][out_w]
    int B, C, M, Mp;
    int in_h, in_w, out_h, out_w;
    int kw, dw;
    int oh_min, ow_min;    // input coordinates of the patch origin relative to the tile origin
    int out_act;
    float out_slope;
    int tiles_w, tiles_h;
    int PW;                // patch pitch = 64 + (kw - 1) * dw
    int ck;                // channels per LDS chunk
};

constexpr int kSmR = 8, kSmTH = 32, kSmTW = 64;

// FLIP = 1: data gradient (tap th reads strip element i + KH-1-th, tap tw column lane + (kw-1-tw) dw);
// FLIP = 0: forward (i + th, lane + tw dw).
template <int KH, int MM, int FLIP>
__global__ __launch_bounds__(256) void conv2d_smallm_kernel(const SmallM2P p) {
    constexpr int PHt = kSmTH + KH - 1;
    extern __shared__ float patch[];                    // [ck][PHt][PW], then the chunk's weights [ck][kw][KH][MM]
    float* const wsm = patch + p.ck * PHt * p.PW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x;
    const int tw_i = bx % p.tiles_w;
    bx /= p.tiles_w;
    const int th_i = bx % p.tiles_h;
    const int b = bx / p.tiles_h;
    const int h0 = th_i * kSmTH, w0 = tw_i * kSmTW;
    const int PW = p.PW;
    float acc[kSmR][MM];
#pragma unroll
    for (int i = 0; i < kSmR; ++i)
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[i][m] = 0.f;
    const float* __restrict__ wp = p.wp;
    const float* __restrict__ in = p.in;
    for (int c0 = 0; c0 < p.C; c0 += p.ck) {
        const int nc = min(p.ck, p.C - c0);
        __syncthreads();
        // ---- the patch of this chunk: rows round-robin over the waves, lanes along W (coalesced, zero outside the plane).
        // U rows x 2 column passes are loaded before the first LDS store: one global-memory latency per 2 U elements (one
        // element at a time the fill was latency-bound: 1.0 ms per first-layer data gradient, 7x both of its rooflines)
        constexpr int U = 8;
        const int rows = nc * PHt;
        for (int row0 = wave; row0 < rows; row0 += 4 * U) {
            float v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 4 * u;
                const int c = row / PHt, pr = row - c * PHt;
                const int gh = h0 + p.oh_min + pr;
                const bool rok = row < rows && gh >= 0 && gh < p.in_h;
                const float* src = in + (((long)b * p.C + c0 + c) * p.in_h + gh) * p.in_w;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pc = lane + 64 * k;
                    const int gw = w0 + p.ow_min + pc;
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
        // ---- and its weights, compact: the scalar loads this replaces (one 128-byte line per (tap, channel)) missed the
        // scalar cache on every tap -- 14 TFLOP/s; as LDS broadcasts the same values cost one read per 16 FMAs
        for (int e = tid; e < nc * p.kw * KH * MM; e += 256) {
            const int m = e % MM;
            int r = e / MM;
            const int th = r % KH;
            r /= KH;
            const int tw = r % p.kw, c = r / p.kw;
            wsm[e] = wp[((long)(th * p.kw + tw) * p.C + c0 + c) * p.Mp + m];
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            for (int tw = 0; tw < p.kw; ++tw) {
                const int cs = (FLIP ? p.kw - 1 - tw : tw) * p.dw;
                const float* col = patch + (c * PHt + wave * kSmR) * PW + lane + cs;
                float s[kSmR + KH - 1];
#pragma unroll
                for (int k = 0; k < kSmR + KH - 1; ++k) s[k] = col[k * PW];
                const float* wl = wsm + (c * p.kw + tw) * KH * MM;          // wave-uniform address: LDS broadcast
#pragma unroll
                for (int th = 0; th < KH; ++th) {
#pragma unroll
                    for (int m = 0; m < MM; ++m) {
                        const float wm = wl[th * MM + m];
#pragma unroll
                        for (int i = 0; i < kSmR; ++i) acc[i][m] = fmaf(wm, s[i + (FLIP ? KH - 1 - th : th)], acc[i][m]);
                    }
                }
            }
        }
    }
    const int w = w0 + lane;
    if (w >= p.out_w) return;
    const bool leaky = p.out_act == RH_ACT_LEAKY;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m >= p.M) break;
        const float bv = p.bias ? p.bias[m] : 0.f;
#pragma unroll
        for (int i = 0; i < kSmR; ++i) {
            const int r = h0 + wave * kSmR + i;
            if (r >= p.out_h) continue;
            float v = acc[i][m] + bv;
            if (leaky) v = v > 0.f ? v : v * p.out_slope;
            p.out[(((long)b * p.M + m) * p.out_h + r) * p.out_w + w] = v;
        }
    }
}

bool smallm_enabled() {
    const char* e = getenv("RH_CONV2D_SMALLM");    // read per call (tests): 0 = the MFMA tiles
    return !(e && atoi(e) == 0);
}

template <int KH, int FLIP>
void smallm_go(const SmallM2P& p, int mm, dim3 grid, size_t lds, hipStream_t stream) {
    if (mm == 1) rh_launch_main(conv2d_smallm_kernel<KH, 1, FLIP>, grid, dim3(256), lds, stream, p);
    else if (mm == 2) rh_launch_main(conv2d_smallm_kernel<KH, 2, FLIP>, grid, dim3(256), lds, stream, p);
    else rh_launch_main(conv2d_smallm_kernel<KH, 4, FLIP>, grid, dim3(256), lds, stream, p);
}

}  // namespace

// which = 0 forward, 1 data gradient (of a conv whose output activation the caller has already folded into dy).
bool rh_conv2d_smallm_eligible(const rh_conv2d_desc* d, int which) {
    if (!smallm_enabled() || d->batch <= 0) return false;
    const int M = which == 0 ? d->c_out : d->c_in;
    if (M > 4 || d->sh != 1 || d->sw != 1 || d->dh != 1 || (d->kh != 3 && d->kh != 9)) return false;
    if (which == 1 && d->act != RH_ACT_NONE) return false;
    const long in_el = (long)d->batch * (which == 0 ? d->c_in : d->c_out) * (which == 0 ? (long)d->h_in * d->w_in : (long)d->h_out * d->w_out);
    const long out_el = (long)d->batch * M * (which == 0 ? (long)d->h_out * d->w_out : (long)d->h_in * d->w_in);
    const long tiles = (long)d->batch * rh_cdiv(which == 0 ? d->h_out : d->h_in, kSmTH) * rh_cdiv(which == 0 ? d->w_out : d->w_in, kSmTW);
    return in_el < (1l << 40) && out_el < (1l << 40) && tiles < 0x7fffffffl && 64 + (d->kw - 1) * d->dw <= 128;
}

// wp: the f32 section of the packed operand of that direction ([slot = th * kw + tw][c][Mp], Mp = 32)
int rh_conv2d_smallm_launch(const rh_conv2d_desc* d, int which, const float* in, const float* wp, const float* bias, float* out,
                            hipStream_t stream) {
    SmallM2P p{};
    p.in = in; p.wp = wp; p.bias = which == 0 ? bias : nullptr; p.out = out;
    p.B = d->batch;
    p.kw = d->kw; p.dw = d->dw;
    if (which == 0) {
        p.C = d->c_in; p.M = d->c_out;
        p.in_h = d->h_in; p.in_w = d->w_in; p.out_h = d->h_out; p.out_w = d->w_out;
        p.oh_min = -d->ph; p.ow_min = -d->pw;
        p.out_act = d->act; p.out_slope = d->act_slope;
    } else {
        p.C = d->c_out; p.M = d->c_in;
        p.in_h = d->h_out; p.in_w = d->w_out; p.out_h = d->h_in; p.out_w = d->w_in;
        p.oh_min = d->ph - (d->kh - 1); p.ow_min = d->pw - (d->kw - 1) * d->dw;
        p.out_act = RH_ACT_NONE; p.out_slope = 0.f;
    }
    p.Mp = (p.M + 31) & ~31;
    p.tiles_w = rh_cdiv(p.out_w, kSmTW); p.tiles_h = rh_cdiv(p.out_h, kSmTH);
    p.PW = kSmTW + (d->kw - 1) * d->dw;
    const int PHt = kSmTH + d->kh - 1;
    p.ck = 4;
    while (p.ck > 1 && (size_t)p.ck * PHt * p.PW * 4 > 48 * 1024) p.ck >>= 1;
    const int mm = p.M == 1 ? 1 : (p.M == 2 ? 2 : 4);
    const size_t lds = (size_t)p.ck * (PHt * p.PW + d->kw * d->kh * mm) * 4;
    const dim3 grid((unsigned)((long)p.B * p.tiles_h * p.tiles_w));
    if (d->kh == 9) { if (which) smallm_go<9, 1>(p, mm, grid, lds, stream); else smallm_go<9, 0>(p, mm, grid, lds, stream); }
    else            { if (which) smallm_go<3, 1>(p, mm, grid, lds, stream); else smallm_go<3, 0>(p, mm, grid, lds, stream); }
    return rh_check_launch(which ? "conv2d_smallm_dgrad" : "conv2d_smallm_fwd");
}
