// General 2-D convolution (zero padding, stride, dilation in both dims, groups == 1) on the
// f32-input matrix cores, for the Conv2d stacks of the spectral discriminators:
//   rave/discriminator.py:23-74 (EncodecConvNet), rave/descript_discriminator.py:30-66,118-184 (MPD, MRD).
//
// Same implicit-GEMM view as the 1-D kernels (M = out channels, N = output positions, K = (tap,
// in channel)) but the column tile is a 2-D block of TR x TQ output positions (x nb batch items when the
// plane is small), and the staged B operand is the 2-D input patch that block needs:
//   PH = (TR-1)*is_h + tap span_h + 1 rows,  PW = (TQ-1)*is_w + tap span_w + 1 columns,
// zero-filled by (h, w) coordinates -- all four paddings come out of the staging, the im2col matrix is
// never materialised and a tap is just a constant LDS offset.  The data gradient is the same kernel run
// per output phase (alpha_h < sh, alpha_w < sw), with dy * act'(y) formed while staging.
// The weight gradient reduces over 2-D chunks (rk x wk positions of dy) per batch item; split-K partials
// + ordered reduction keep it bitwise deterministic.
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"
#include "conv2d_x6.hpp"

// wgrad2d_x6.hip: weight gradient on the bf16 matrix cores (<= 32 output channels, stride 1 along W)
int64_t rh_wgrad2d_x6_workspace(const rh_conv2d_desc* d);
int rh_wgrad2d_x6_launch(const rh_conv2d_desc* d, const float* dy, const float* x, float* dw, void* ws, int64_t ws_bytes,
                         hipStream_t stream, bool* used, const unsigned* dy_range, const unsigned* x_range);
// conv2d_smallm.hip: vector-ALU kernels for convolutions with <= 4 output rows (first-layer data gradient, scoring conv)
bool rh_conv2d_smallm_eligible(const rh_conv2d_desc* d, int which);
int rh_conv2d_smallm_launch(const rh_conv2d_desc* d, int which, const float* in, const float* wp, const float* bias, float* out,
                            hipStream_t stream);

// conv2d_smallc.hip: vector-ALU forward for <= 4 input channels (the first conv of the spectral discriminators' stacks)
bool rh_conv2d_smallc_fwd_eligible(const rh_conv2d_desc* d);
int rh_conv2d_smallc_fwd_launch(const rh_conv2d_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y,
                                hipStream_t stream);

namespace {

constexpr int kPh2 = 8;   // max sh*sw

struct Conv2P {
    const float* in;
    const float* wp;
    float* out;
    const float* bias;
    const float* in_mul;            // staged value *= act'(in_mul[same index])   (data gradient) or null
    int B, C, M, Mp;
    int in_h, in_w, out_h, out_w;
    int rows, qcols;                // per-phase output grid
    int is_h, is_w, os_h, os_w;
    int TQ, TR, tq_shift, tr_shift, nb, tiles_q, tiles_r;
    int PH, PW, PWp, chp;           // staged patch per (batch item, channel): PH x PW, row pitch PWp, plane pitch chp
    int ck, wlds_floats;
    unsigned magic_pw, magic_ph, magic_ck;   // ceil(2^32/d) reciprocals (0 encodes d == 1); operands stay < 2^20
    int ni, stage_floats;                    // LDS-DMA pipeline: 64-float DMA slots per plane, floats per stage
    unsigned in_bytes, w_bytes;
    int mul_act, epi_act;
    float mul_slope, epi_slope;
    int nphase;
    int ph_oph_h[kPh2], ph_oph_w[kPh2], ph_ntaps[kPh2], ph_tap0[kPh2], ph_minh[kPh2], ph_minw[kPh2];
    long ph_wofs[kPh2];
    int offh[kMaxTaps], offw[kMaxTaps];
};

struct Wgrad2P {
    const float* R;                 // dy  [B][M][r_h][r_w]
    const float* Rmul;              // y (same shape) or null: R *= act'(y)
    const float* S;                 // x   [B][C][s_h][s_w]
    float* out;                     // [Z][M][C*T]
    int B, M, C, T;
    int r_h, r_w, s_h, s_w, is_h, is_w;
    int rk, wk, wk_shift;           // dy chunk: rk rows x wk columns (wk = power of two >= 2)
    int chunks_r, chunks_w, total_chunks, chunks_per_z;
    int pr, PH, PW, PWp, chp, nc_max;
    unsigned magic_pw, magic_ph;
    int nr, ns, r_floats, stage_floats;   // LDS-DMA pipeline: 64-float slots per R row / S plane, floats per stage
    unsigned r_bytes, s_bytes;
    int minh, minw;
    int r_act;
    float r_slope;
    int offh[kMaxTaps], offw[kMaxTaps];
};

// n / d for n * d < 2^32 with magic = ceil(2^32 / d); magic == 0 encodes d == 1
__device__ __forceinline__ int mdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv2d_igemm_kernel(const Conv2P p) {
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
    const int minh = p.ph_minh[phase], minw = p.ph_minw[phase];
    const float* __restrict__ wp = p.wp + p.ph_wofs[phase];

    int bx = blockIdx.x;
    const int tq = bx % p.tiles_q;
    bx /= p.tiles_q;
    const int tr = bx % p.tiles_r;
    const int bt = bx / p.tiles_r;
    const int b0 = bt * p.nb, r0 = tr * p.TR, q0 = tq * p.TQ;
    const int m0 = blockIdx.y * BM;
    const int h0 = r0 * p.is_h + minh, w0 = q0 * p.is_w + minw;

    int xb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        xb[tn] = (bl * p.ck + kh) * p.chp + rl * p.is_h * p.PWp + ql * p.is_w;
    }
    const int arow = wm * TM * 32 + j + kh * BM;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    for (int c0 = 0; c0 < p.C && ntaps > 0; c0 += p.ck) {
        __syncthreads();
        // ---- stage the 2-D input patches (zero fill by coordinates = all four paddings) ----
        // flat element index -> (batch item, channel, patch row, column); U loads are issued before the first
        // LDS store so that one global-memory latency is paid per U elements, not per element
        constexpr int U = 8;
        const int total = p.nb * p.ck * p.PH * p.PW;
        for (int e0 = tid; e0 < total; e0 += NT * U) {
            float v[U];
            int dsto[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * NT;
                v[u] = 0.f;
                dsto[u] = -1;
                if (e < total) {
                    const int row = mdiv(e, p.magic_pw), w = e - row * p.PW;
                    const int bc = mdiv(row, p.magic_ph), phh = row - bc * p.PH;
                    const int bl = mdiv(bc, p.magic_ck), c = bc - bl * p.ck;
                    const int b = b0 + bl, ch = c0 + c, h = h0 + phh, gw = w0 + w;
                    dsto[u] = bc * p.chp + phh * p.PWp + w;
                    if (b < p.B && ch < p.C && h >= 0 && h < p.in_h && gw >= 0 && gw < p.in_w) {
                        const long idx = (((long)b * p.C + ch) * p.in_h + h) * p.in_w + gw;
                        v[u] = p.in[idx];
                        if (p.in_mul) v[u] *= rh_act_grad(p.in_mul[idx], p.mul_act, p.mul_slope, 0.f);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dsto[u] >= 0) x_lds[dsto[u]] = v[u];
        }
        // ---- stage the weight tile [tap][c][BM] ----
        constexpr int V = BM / 4;
        constexpr int UW = 4;
        const int wtotal = ntaps * p.ck * V;
        for (int e0 = tid; e0 < wtotal; e0 += NT * UW) {
            f32x4 val[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int e = e0 + u * NT;
                val[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < wtotal) {
                    const int kr = e / V, v4 = e - kr * V;
                    const int t = mdiv(kr, p.magic_ck), c = kr - t * p.ck;
                    const int ch = c0 + c;
                    const int m = m0 + v4 * 4;
                    if (ch < p.C && m < p.Mp)
                        val[u] = *reinterpret_cast<const f32x4*>(wp + ((long)t * p.C + ch) * p.Mp + m);
                }
            }
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int e = e0 + u * NT;
                if (e < wtotal) *reinterpret_cast<f32x4*>(w_lds + e * 4) = val[u];
            }
        }
        __syncthreads();
        // ---- MFMA over (tap, channel pair) ----
        for (int t = 0; t < ntaps; ++t) {
            const int toff = (p.offh[tap0 + t] - minh) * p.PWp + (p.offw[tap0 + t] - minw);
            const float* wl = w_lds + t * p.ck * BM + arow;
            const float* xl = x_lds + toff;
            for (int c = 0; c < p.ck; c += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = wl[c * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = xl[xb[tn] + c * p.chp];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + output activation ----
    const int oph_h = p.ph_oph_h[phase], oph_w = p.ph_oph_w[phase];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        const int r = r0 + rl, q = q0 + ql, b = b0 + bl;
        if (r >= p.rows || q >= p.qcols || b >= p.B) continue;
        const int orow = r * p.os_h + oph_h, ocol = q * p.os_w + oph_w;
        if (orow >= p.out_h || ocol >= p.out_w) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + (wm * TM + tm) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
                if (m < p.M) {
                    const long idx = (((long)b * p.M + m) * p.out_h + orow) * p.out_w + ocol;
                    float v = acc[tm][tn][rr];
                    if (p.bias) v += p.bias[m];
                    p.out[idx] = rh_act_apply(v, p.epi_act, p.epi_slope, 0.f);
                }
            }
        }
    }
}


typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;   // >= any descriptor size we accept -> the DMA writes zeros
constexpr int kNI = 20;                  // max 64-float DMA slots per staged plane (PH*PW <= 1280)

// Same GEMM as conv2d_igemm_kernel, software-pipelined with asynchronous global->LDS DMA
// (buffer_load ... lds): two LDS stages, one barrier per K chunk, no staging registers.  A staged plane is
// the compact PH x PW patch; lane l of DMA slot i always fetches patch element 64 i + l, so its in-plane
// source offset (or the out-of-range marker that makes the DMA write 0.0: all four zero paddings) is
// computed ONCE per workgroup -- per K chunk only a scalar plane base changes.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv2d_dma_kernel(const Conv2P p) {
    constexpr int BM = TM * WM * 32;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    const int phase = blockIdx.z;
    const int ntaps = p.ph_ntaps[phase];
    const int tap0 = p.ph_tap0[phase];
    const int minh = p.ph_minh[phase], minw = p.ph_minw[phase];
    const unsigned wofs = (unsigned)p.ph_wofs[phase];

    int bx = blockIdx.x;
    const int tq = bx % p.tiles_q;
    bx /= p.tiles_q;
    const int tr = bx % p.tiles_r;
    const int bt = bx / p.tiles_r;
    const int b0 = bt * p.nb, r0 = tr * p.TR, q0 = tq * p.TQ;
    const int m0 = blockIdx.y * BM;
    const int h0 = r0 * p.is_h + minh, w0 = q0 * p.is_w + minw;

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.w_bytes, 0x00020000);

    int xb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        xb[tn] = (bl * p.ck + kh) * p.chp + rl * p.is_h * p.PW + ql * p.is_w;
    }
    const int arow = wm * TM * 32 + j + kh * BM;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    unsigned xo[kNI];
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
        const int e = lane + 64 * i;
        const int row = mdiv(e, p.magic_pw), w = e - row * p.PW;
        const int h = h0 + row, gw = w0 + w;
        const bool ok = i < p.ni && row < p.PH && h >= 0 && h < p.in_h && gw >= 0 && gw < p.in_w;
        xo[i] = ok ? (unsigned)(h * p.in_w + gw) * 4u : kOOB;
    }
    const int planes = p.nb * p.ck;
    const unsigned plane_bytes = (unsigned)p.in_h * (unsigned)p.in_w * 4u;
    const int wrows = ntaps * p.ck;
    const int w_instrs = (wrows * BM + 255) >> 8;   // 256 floats (64 lanes x 16 B) per DMA instruction

    auto issue = [&](int c0, float* stage) {
        for (int q = wave; q < w_instrs; q += NW) {   // weights: LDS image [tap][c][BM], flat
            const int f = q * 256 + lane * 4;
            const int kr = f / BM, col = f - kr * BM;
            const int t = mdiv(kr, p.magic_ck), c = kr - t * p.ck;
            const int ch = c0 + c, m = m0 + col;
            unsigned off = kOOB;
            if (kr < wrows && ch < p.C && m < p.Mp) off = (wofs + (unsigned)(t * p.C + ch) * p.Mp + m) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(stage + q * 256), 16, off, 0, 0, 0);
        }
        float* xs = stage + p.wlds_floats;
        for (int pl = wave; pl < planes; pl += NW) {
            const int bl = mdiv(pl, p.magic_ck), c = pl - bl * p.ck;
            const int b = b0 + bl, ch = c0 + c;
            const bool dead = b >= p.B || ch >= p.C;
            const unsigned base = (unsigned)(b * p.C + ch) * plane_bytes;
            float* dst = xs + pl * p.chp;
#pragma unroll
            for (int i = 0; i < kNI; ++i) {
                if (i < p.ni) {
                    const unsigned off = (dead || xo[i] == kOOB) ? kOOB : base + xo[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)(dst + i * 64), 4, off, 0, 0, 0);
                }
            }
        }
    };

    const int nchunks = ntaps > 0 ? (p.C + p.ck - 1) / p.ck : 0;
    if (nchunks > 0) issue(0, smem);
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // chunk i landed for every wave; everyone is done with the other stage
        if (i + 1 < nchunks) issue((i + 1) * p.ck, smem + ((i + 1) & 1) * p.stage_floats);
        const float* w_lds = smem + (i & 1) * p.stage_floats;
        const float* x_lds = w_lds + p.wlds_floats;
        for (int t = 0; t < ntaps; ++t) {
            const int toff = (p.offh[tap0 + t] - minh) * p.PW + (p.offw[tap0 + t] - minw);
            const float* wl = w_lds + t * p.ck * BM + arow;
            const float* xl = x_lds + toff;
            int c = 0;
            for (; c + 8 <= p.ck; c += 8) {     // 4 k-steps per trip, all LDS reads in flight before the MFMAs
                float a[4][TM], b[4][TN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) a[u][tm] = wl[(c + 2 * u) * BM + tm * 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) b[u][tn] = xl[xb[tn] + (c + 2 * u) * p.chp];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][tm], b[u][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; c < p.ck; c += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = wl[c * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = xl[xb[tn] + c * p.chp];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + output activation ----
    const int oph_h = p.ph_oph_h[phase], oph_w = p.ph_oph_w[phase];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        const int r = r0 + rl, q = q0 + ql, b = b0 + bl;
        if (r >= p.rows || q >= p.qcols || b >= p.B) continue;
        const int orow = r * p.os_h + oph_h, ocol = q * p.os_w + oph_w;
        if (orow >= p.out_h || ocol >= p.out_w) continue;
        const long cbase = ((long)b * p.M * p.out_h + orow) * p.out_w + ocol;
        const int plane = p.out_h * p.out_w;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + (wm * TM + tm) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
                if (m < p.M) {
                    float v = acc[tm][tn][rr];
                    if (p.bias) v += p.bias[m];
                    p.out[cbase + (long)m * plane] = rh_act_apply(v, p.epi_act, p.epi_slope, 0.f);
                }
            }
        }
    }
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void wgrad2d_kernel(const Wgrad2P p) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* r_lds = smem;                 // [BM][pr]
    float* s_lds = smem + BM * p.pr;     // [nc_max][chp]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = blockIdx.y * BM, col0 = blockIdx.x * BN, z = blockIdx.z;
    const int ncols = p.C * p.T;
    const int c_lo = col0 / p.T;

    int sb[TN], ar[TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int col = col0 + (wn * TN + tn) * 32 + j;
        col = min(col, ncols - 1);
        const int c = col / p.T, t = col - c * p.T;
        sb[tn] = (c - c_lo) * p.chp + (p.offh[t] - p.minh) * p.PWp + (p.offw[t] - p.minw) + kh * p.is_w;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) ar[tm] = ((wm * TM + tm) * 32 + j) * p.pr + kh;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int kelems = p.rk * p.wk;
    const int ch0 = z * p.chunks_per_z;
    const int ch1 = min(ch0 + p.chunks_per_z, p.total_chunks);
    for (int ch = ch0; ch < ch1; ++ch) {
        const int cw = ch % p.chunks_w;
        const int t2 = ch / p.chunks_w;
        const int cr = t2 % p.chunks_r;
        const int b = t2 / p.chunks_r;
        const int hr0 = cr * p.rk, wc0 = cw * p.wk;
        __syncthreads();
        constexpr int U = 8;
        constexpr int NT = NW * 64;
        {   // R tile: [BM][rk*wk] (kelems is a power of two)
            const int ksh = __builtin_ctz(kelems);
            const int total = BM << ksh;
            for (int e0 = tid; e0 < total; e0 += NT * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + u * NT;
                    v[u] = 0.f;
                    if (e < total) {
                        const int r = e >> ksh, k = e & (kelems - 1);
                        const int rr = k >> p.wk_shift, ww = k & (p.wk - 1);
                        const int m = m0 + r, h = hr0 + rr, w = wc0 + ww;
                        if (m < p.M && h < p.r_h && w < p.r_w) {
                            const long idx = (((long)b * p.M + m) * p.r_h + h) * p.r_w + w;
                            v[u] = p.R[idx];
                            if (p.Rmul) v[u] *= rh_act_grad(p.Rmul[idx], p.r_act, p.r_slope, 0.f);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + u * NT;
                    if (e < total) r_lds[(e >> ksh) * p.pr + (e & (kelems - 1))] = v[u];
                }
            }
        }
        {   // S patches: [nc_max][PH][PW]
            const int h0 = hr0 * p.is_h + p.minh, w0 = wc0 * p.is_w + p.minw;
            const int total = p.nc_max * p.PH * p.PW;
            for (int e0 = tid; e0 < total; e0 += NT * U) {
                float v[U];
                int dsto[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + u * NT;
                    v[u] = 0.f;
                    dsto[u] = -1;
                    if (e < total) {
                        const int row = mdiv(e, p.magic_pw), w = e - row * p.PW;
                        const int cc = mdiv(row, p.magic_ph), phh = row - cc * p.PH;
                        const int c = c_lo + cc, h = h0 + phh, gw = w0 + w;
                        dsto[u] = cc * p.chp + phh * p.PWp + w;
                        if (c < p.C && h >= 0 && h < p.s_h && gw >= 0 && gw < p.s_w)
                            v[u] = p.S[(((long)b * p.C + c) * p.s_h + h) * p.s_w + gw];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (dsto[u] >= 0) s_lds[dsto[u]] = v[u];
            }
        }
        __syncthreads();
        for (int rr = 0; rr < p.rk; ++rr) {
            const float* rl = r_lds + rr * p.wk;
            const float* sl = s_lds + rr * p.is_h * p.PWp;
            for (int ww = 0; ww < p.wk; ww += 2) {
                float a[TM], bb[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = rl[ar[tm] + ww];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bb[tn] = sl[sb[tn] + ww * p.is_w];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }
    float* out = p.out + (long)z * p.M * ncols;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = col0 + (wn * TN + tn) * 32 + j;
        if (col >= ncols) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) out[(long)m * ncols + col] = acc[tm][tn][r];
            }
    }
}


constexpr int kNR = 4;    // 64-float DMA slots per R row   (rk*wk <= 256)
constexpr int kNS = 16;   // 64-float DMA slots per S plane (PH*PW <= 1024)

// Weight gradient for M <= 32 (the spectral discriminators' 32-channel stacks), LDS-DMA double-buffered:
// a workgroup owns 32 x (TN*128) of dW (a range of (c, tap) columns) for its K slice; per chunk of rk x wk
// positions of dy it stages the 32 dy rows and the input patches of its channels with asynchronous DMA
// (per-lane offsets inside a row / plane precomputed once; only validity bits and scalar bases per chunk).
template <int TN>
__global__ __launch_bounds__(256) void wgrad2d_dma_kernel(const Wgrad2P p) {
    constexpr int BN = TN * 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 32, col0 = blockIdx.x * BN, z = blockIdx.z;
    const int ncols = p.C * p.T;
    const int c_lo = col0 / p.T;
    const int nc = min(p.nc_max, p.C - c_lo);

    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);
    const auto s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.S), 0, p.s_bytes, 0x00020000);

    int sb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int col = col0 + (wave * TN + tn) * 32 + j;
        col = min(col, ncols - 1);
        const int c = col / p.T, t = col - c * p.T;
        sb[tn] = (c - c_lo) * p.chp + (p.offh[t] - p.minh) * p.PW + (p.offw[t] - p.minw) + kh * p.is_w;
    }
    const int ar = j * p.pr + kh;

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

    // per-lane slot geometry: (row, col) inside the dy chunk / the input patch, packed, and the element offset
    unsigned rp[kNR], ro[kNR], sp[kNS], so[kNS];
#pragma unroll
    for (int i = 0; i < kNR; ++i) {
        const int k = lane + 64 * i;
        const int rr = k >> p.wk_shift, ww = k & (p.wk - 1);
        rp[i] = ((unsigned)rr << 16) | (unsigned)ww;
        ro[i] = (unsigned)(rr * p.r_w + ww) * 4u;
    }
#pragma unroll
    for (int i = 0; i < kNS; ++i) {
        const int e = lane + 64 * i;
        const int row = mdiv(e, p.magic_pw), w = e - row * p.PW;
        sp[i] = (i < p.ns && row < p.PH) ? (((unsigned)row << 16) | (unsigned)w) : 0xffffffffu;
        so[i] = (unsigned)(row * p.s_w + w) * 4u;
    }

    auto issue = [&](int ch, float* stage) {
        const int cw = ch % p.chunks_w;
        const int t2 = ch / p.chunks_w;
        const int cr = t2 % p.chunks_r;
        const int b = t2 / p.chunks_r;
        const int hr0 = cr * p.rk, wc0 = cw * p.wk;
        unsigned rmask = 0;
#pragma unroll
        for (int i = 0; i < kNR; ++i) {
            const int rr = rp[i] >> 16, ww = rp[i] & 0xffff;
            if (i < p.nr && hr0 + rr < p.r_h && wc0 + ww < p.r_w) rmask |= 1u << i;
        }
        for (int r = wave; r < 32; r += 4) {
            const int m = m0 + r;
            const bool dead = m >= p.M;
            const unsigned base = (unsigned)(((b * p.M + m) * p.r_h + hr0) * p.r_w + wc0) * 4u;
            float* dst = stage + r * p.pr;
#pragma unroll
            for (int i = 0; i < kNR; ++i) {
                if (i < p.nr) {
                    const unsigned off = (dead || !((rmask >> i) & 1u)) ? kOOB : base + ro[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(dst + i * 64), 4, off, 0, 0, 0);
                }
            }
        }
        const int h0 = hr0 * p.is_h + p.minh, w0 = wc0 * p.is_w + p.minw;
        unsigned smask = 0;
#pragma unroll
        for (int i = 0; i < kNS; ++i) {
            const int h = h0 + (int)(sp[i] >> 16), gw = w0 + (int)(sp[i] & 0xffff);
            if (sp[i] != 0xffffffffu && h >= 0 && h < p.s_h && gw >= 0 && gw < p.s_w) smask |= 1u << i;
        }
        float* xs = stage + p.r_floats;
        const unsigned sbase0 = (unsigned)((h0 * p.s_w + w0) * 4);      // may wrap; only used where the sum is valid
        const unsigned plane_bytes = (unsigned)p.s_h * (unsigned)p.s_w * 4u;
        for (int cc = wave; cc < p.nc_max; cc += 4) {
            const bool dead = cc >= nc;
            const unsigned base = (unsigned)(b * p.C + c_lo + cc) * plane_bytes + sbase0;
            float* dst = xs + cc * p.chp;
#pragma unroll
            for (int i = 0; i < kNS; ++i) {
                if (i < p.ns) {
                    const unsigned off = (dead || !((smask >> i) & 1u)) ? kOOB : base + so[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void*)(dst + i * 64), 4, off, 0, 0, 0);
                }
            }
        }
    };

    const int ch0 = z * p.chunks_per_z;
    const int ch1 = min(ch0 + p.chunks_per_z, p.total_chunks);
    if (ch0 < ch1) issue(ch0, smem);
    for (int ch = ch0; ch < ch1; ++ch) {
        const int i = ch - ch0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + 1 < ch1) issue(ch + 1, smem + ((i + 1) & 1) * p.stage_floats);
        const float* r_lds = smem + (i & 1) * p.stage_floats + ar;
        const float* s_lds = smem + (i & 1) * p.stage_floats + p.r_floats;
        for (int rr = 0; rr < p.rk; ++rr) {
            const float* rl = r_lds + rr * p.wk;
            const float* sl = s_lds + rr * p.is_h * p.PW;
            int ww = 0;
            for (; ww + 8 <= p.wk; ww += 8) {
                float a[4], bb[4][TN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = rl[ww + 2 * u];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bb[u][tn] = sl[sb[tn] + (ww + 2 * u) * p.is_w];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bb[u][tn], acc[tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; ww < p.wk; ww += 2) {
                const float a = rl[ww];
                float bb[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bb[tn] = sl[sb[tn] + ww * p.is_w];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[tn], acc[tn], 0, 0, 0);
            }
        }
    }
    float* out = p.out + (long)z * p.M * ncols;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = col0 + (wave * TN + tn) * 32 + j;
        if (col >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < p.M) out[(long)m * ncols + col] = acc[tn][r];
        }
    }
}

inline int round32(int m) { return (m + 31) & ~31; }
inline unsigned magic32(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }
inline int pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int validate2(const rh_conv2d_desc* d) {
    RH_REQUIRE(d, RH_ERR_INVALID, "conv2d: null descriptor");
    RH_REQUIRE(d->batch >= 0 && d->c_in > 0 && d->c_out > 0 && d->h_in > 0 && d->w_in > 0 && d->h_out > 0 &&
                   d->w_out > 0,
               RH_ERR_INVALID, "conv2d: bad sizes");
    RH_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->sh >= 1 && d->sw >= 1 && d->dh >= 1 && d->dw >= 1 && d->ph >= 0 &&
                   d->pw >= 0,
               RH_ERR_INVALID, "conv2d: kernel/stride/dilation must be >= 1, padding >= 0");
    RH_REQUIRE(d->kh * d->kw <= kMaxTaps, RH_ERR_UNSUPPORTED, "conv2d: %d x %d taps > %d", d->kh, d->kw, kMaxTaps);
    RH_REQUIRE(d->sh * d->sw <= kPh2, RH_ERR_UNSUPPORTED, "conv2d: stride %d x %d > %d phases", d->sh, d->sw, kPh2);
    RH_REQUIRE(d->act == RH_ACT_NONE || d->act == RH_ACT_LEAKY, RH_ERR_UNSUPPORTED,
               "conv2d: output activation must be none or leaky");
    const int ho = (d->h_in + 2 * d->ph - d->dh * (d->kh - 1) - 1) / d->sh + 1;
    const int wo = (d->w_in + 2 * d->pw - d->dw * (d->kw - 1) - 1) / d->sw + 1;
    RH_REQUIRE(d->h_in + 2 * d->ph - d->dh * (d->kh - 1) - 1 >= 0 && d->w_in + 2 * d->pw - d->dw * (d->kw - 1) - 1 >= 0,
               RH_ERR_INVALID, "conv2d: kernel larger than the padded input");
    RH_REQUIRE(ho == d->h_out && wo == d->w_out, RH_ERR_INVALID, "conv2d: (h_out, w_out) = (%d, %d), geometry gives (%d, %d)",
               d->h_out, d->w_out, ho, wo);
    RH_REQUIRE((int64_t)d->h_in * d->w_in < (1ll << 30) && (int64_t)d->h_out * d->w_out < (1ll << 30), RH_ERR_UNSUPPORTED,
               "conv2d: plane too large");
    return RH_OK;
}

struct Plan2 {
    int nphase = 0;
    int oph_h[kPh2], oph_w[kPh2], ntaps[kPh2], tap0[kPh2], minh[kPh2], minw[kPh2], maxh[kPh2], maxw[kPh2];
    int nslots = 0;
    int kk[kMaxTaps], offh[kMaxTaps], offw[kMaxTaps];
};

// which: 0 = forward operand, 1 = data-gradient operand
void build_plan2(const rh_conv2d_desc* d, int which, Plan2* pl) {
    Plan2& t = *pl;
    if (which == 0) {
        t.nphase = 1;
        t.oph_h[0] = t.oph_w[0] = 0;
        t.tap0[0] = 0;
        for (int th = 0; th < d->kh; ++th)
            for (int tw = 0; tw < d->kw; ++tw) {
                t.kk[t.nslots] = th * d->kw + tw;
                t.offh[t.nslots] = th * d->dh - d->ph;
                t.offw[t.nslots] = tw * d->dw - d->pw;
                ++t.nslots;
            }
        t.ntaps[0] = t.nslots;
    } else {
        for (int ah = 0; ah < d->sh; ++ah)
            for (int aw = 0; aw < d->sw; ++aw) {
                const int ph = t.nphase++;
                t.oph_h[ph] = ah;
                t.oph_w[ph] = aw;
                t.tap0[ph] = t.nslots;
                int n = 0;
                for (int th = 0; th < d->kh; ++th) {
                    const int nh = ah + d->ph - th * d->dh;
                    if (nh % d->sh != 0) continue;
                    for (int tw = 0; tw < d->kw; ++tw) {
                        const int nw = aw + d->pw - tw * d->dw;
                        if (nw % d->sw != 0) continue;
                        t.kk[t.nslots] = th * d->kw + tw;
                        t.offh[t.nslots] = nh / d->sh;
                        t.offw[t.nslots] = nw / d->sw;
                        ++t.nslots;
                        ++n;
                    }
                }
                t.ntaps[ph] = n;
            }
    }
    for (int ph = 0; ph < t.nphase; ++ph) {
        t.minh[ph] = t.maxh[ph] = t.minw[ph] = t.maxw[ph] = 0;
        for (int i = 0; i < t.ntaps[ph]; ++i) {
            const int oh = t.offh[t.tap0[ph] + i], ow = t.offw[t.tap0[ph] + i];
            if (i == 0 || oh < t.minh[ph]) t.minh[ph] = oh;
            if (i == 0 || oh > t.maxh[ph]) t.maxh[ph] = oh;
            if (i == 0 || ow < t.minw[ph]) t.minw[ph] = ow;
            if (i == 0 || ow > t.maxw[ph]) t.maxw[ph] = ow;
        }
    }
}

void fill_common(const Plan2& t, Conv2P* p, int C, int M) {
    p->C = C;
    p->M = M;
    p->Mp = round32(M);
    p->nphase = t.nphase;
    long wofs = 0;
    for (int ph = 0; ph < t.nphase; ++ph) {
        p->ph_oph_h[ph] = t.oph_h[ph];
        p->ph_oph_w[ph] = t.oph_w[ph];
        p->ph_ntaps[ph] = t.ntaps[ph];
        p->ph_tap0[ph] = t.tap0[ph];
        p->ph_minh[ph] = t.minh[ph];
        p->ph_minw[ph] = t.minw[ph];
        p->ph_wofs[ph] = wofs;
        wofs += (long)t.ntaps[ph] * C * p->Mp;
    }
    for (int i = 0; i < t.nslots; ++i) {
        p->offh[i] = t.offh[i];
        p->offw[i] = t.offw[i];
    }
}

template <int TM, int TN, int WM, int WN>
int launch2(Conv2P& p, const Plan2& t, hipStream_t stream, const char* what) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    p.TQ = pow2ceil(p.qcols) < BN ? pow2ceil(p.qcols) : BN;
    const int tr_cap = BN / p.TQ;
    p.TR = pow2ceil(p.rows) < tr_cap ? pow2ceil(p.rows) : tr_cap;
    p.nb = BN / (p.TQ * p.TR);
    p.tq_shift = __builtin_ctz(p.TQ);
    p.tr_shift = __builtin_ctz(p.TR);
    p.tiles_q = rh_cdiv(p.qcols, p.TQ);
    p.tiles_r = rh_cdiv(p.rows, p.TR);
    int span_h = 0, span_w = 0, maxtaps = 1;
    for (int i = 0; i < t.nphase; ++i) {
        span_h = span_h > t.maxh[i] - t.minh[i] ? span_h : t.maxh[i] - t.minh[i];
        span_w = span_w > t.maxw[i] - t.minw[i] ? span_w : t.maxw[i] - t.minw[i];
        maxtaps = maxtaps > t.ntaps[i] ? maxtaps : t.ntaps[i];
    }
    p.PH = (p.TR - 1) * p.is_h + span_h + 1;
    p.PW = (p.TQ - 1) * p.is_w + span_w + 1;
    p.PWp = p.PW | 1;
    p.chp = (p.PH * p.PWp) | 1;
    const int per_ch = maxtaps * BM + p.nb * p.chp;
    static const int budget = [] {
        const char* e = getenv("RH_CONV2D_STAGE_FLOATS");
        return e ? atoi(e) : 15 * 1024;
    }();
    int ck = budget / per_ch;
    ck &= ~1;
    if (ck > 32) ck = 32;
    if (ck < 2) ck = 2;
    const int cmax = (p.C + 1) & ~1;
    if (ck > cmax) ck = cmax;
    p.ck = ck;
    p.magic_pw = magic32(p.PW);
    p.magic_ph = magic32(p.PH);
    p.magic_ck = magic32(ck);
    p.wlds_floats = maxtaps * ck * BM;
    const size_t lds = sizeof(float) * ((size_t)p.wlds_floats + (size_t)p.nb * ck * p.chp);
    RH_REQUIRE((long)p.nb * ck * p.PH * p.PW < (1l << 20) && (long)maxtaps * ck * BM < (1l << 20), RH_ERR_UNSUPPORTED,
               "%s: tile too large", what);
    RH_REQUIRE(lds <= 160 * 1024, RH_ERR_UNSUPPORTED, "%s: tile needs %zu B of LDS", what, lds);
    auto kern = conv2d_igemm_kernel<TM, TN, WM, WN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    });
    const long gx = (long)rh_cdiv(p.B, p.nb) * p.tiles_r * p.tiles_q;
    RH_REQUIRE(gx < (1l << 31), RH_ERR_UNSUPPORTED, "%s: grid too large", what);
    dim3 grid((unsigned)gx, rh_cdiv(p.M, BM), p.nphase);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    return rh_check_launch(what);
}


// LDS-DMA variant: returns RH_ERR_UNSUPPORTED (without setting a message the caller shows) when the geometry
// does not fit its staging scheme; the caller then uses the register-staged kernel.
template <int TM, int TN, int WM, int WN>
int launch2_dma(Conv2P& p, const Plan2& t, hipStream_t stream, const char* what, bool* done) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    *done = false;
    if (p.in_mul) return RH_OK;
    const long in_b = (long)p.B * p.C * p.in_h * p.in_w * 4, w_b = 0;
    (void)w_b;
    if (in_b >= (1l << 31)) return RH_OK;
    int span_h = 0, span_w = 0, maxtaps = 1;
    long wfloats = 0;
    for (int i = 0; i < t.nphase; ++i) {
        span_h = span_h > t.maxh[i] - t.minh[i] ? span_h : t.maxh[i] - t.minh[i];
        span_w = span_w > t.maxw[i] - t.minw[i] ? span_w : t.maxw[i] - t.minw[i];
        maxtaps = maxtaps > t.ntaps[i] ? maxtaps : t.ntaps[i];
        wfloats += (long)t.ntaps[i] * p.C * p.Mp;
    }
    if (wfloats * 4 >= (1l << 31)) return RH_OK;
    int TQ = pow2ceil(p.qcols) < BN ? pow2ceil(p.qcols) : BN;
    int TR = pow2ceil(p.rows) < BN / TQ ? pow2ceil(p.rows) : BN / TQ;
    auto patch = [&](int tq_, int tr_) { return (long)((tr_ - 1) * p.is_h + span_h + 1) * ((tq_ - 1) * p.is_w + span_w + 1); };
    while (patch(TQ, TR) > 64 * kNI) {
        if (TR > 1) TR >>= 1;
        else if (TQ > 16) TQ >>= 1;
        else return RH_OK;
    }
    p.TQ = TQ; p.TR = TR;
    p.nb = BN / (TQ * TR);
    p.tq_shift = __builtin_ctz(TQ);
    p.tr_shift = __builtin_ctz(TR);
    p.tiles_q = rh_cdiv(p.qcols, TQ);
    p.tiles_r = rh_cdiv(p.rows, TR);
    p.PH = (TR - 1) * p.is_h + span_h + 1;
    p.PW = (TQ - 1) * p.is_w + span_w + 1;
    p.PWp = p.PW;
    p.ni = (p.PH * p.PW + 63) / 64;
    p.chp = p.ni * 64 + 32;                       // % 64 == 32: the two k-halves of a wave hit disjoint banks
    static const int budget = [] {
        const char* e = getenv("RH_CONV2D_DMA_STAGE_FLOATS");
        return e ? atoi(e) : 8 * 1024;
    }();
    const int per_ch = maxtaps * BM + p.nb * p.chp;
    int ck = budget / per_ch;
    if (ck >= 8) ck &= ~7;
    else ck &= ~1;
    if (ck > 32) ck = 32;
    if (ck < 2) ck = 2;
    const int cmax = (p.C + 1) & ~1;
    if (ck > cmax) ck = cmax;
    p.ck = ck;
    p.magic_pw = magic32(p.PW);
    p.magic_ph = magic32(p.PH);
    p.magic_ck = magic32(ck);
    p.wlds_floats = (maxtaps * ck * BM + 255) & ~255;
    p.stage_floats = p.wlds_floats + p.nb * ck * p.chp;
    const size_t lds = sizeof(float) * 2 * (size_t)p.stage_floats;
    if (lds > 160 * 1024) return RH_OK;
    p.in_bytes = (unsigned)in_b;
    p.w_bytes = (unsigned)(wfloats * 4);
    auto kern = conv2d_dma_kernel<TM, TN, WM, WN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    });
    const long gx = (long)rh_cdiv(p.B, p.nb) * p.tiles_r * p.tiles_q;
    if (gx >= (1l << 31)) return RH_OK;
    dim3 grid((unsigned)gx, rh_cdiv(p.M, BM), p.nphase);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    *done = true;
    return rh_check_launch(what);
}

template <int TM, int TN, int WM, int WN>
int launch2_any(Conv2P& p, const Plan2& t, hipStream_t stream, const char* what) {
    static const bool no_dma = getenv("RH_CONV2D_NODMA") != nullptr;
    if (!no_dma) {
        bool done = false;
        Conv2P q = p;
        if (int e = launch2_dma<TM, TN, WM, WN>(q, t, stream, what, &done)) return e;
        if (done) return RH_OK;
    }
    return launch2<TM, TN, WM, WN>(p, t, stream, what);
}

int launch_conv2(Conv2P& p, const Plan2& t, hipStream_t stream, const char* what) {
    if (p.B <= 0) return RH_OK;
    if (p.M <= 32) return launch2_any<1, 2, 1, 4>(p, t, stream, what);
    if (p.M <= 64) return launch2_any<2, 1, 1, 4>(p, t, stream, what);
    if (p.M % 96 == 0) return launch2_any<3, 1, 1, 4>(p, t, stream, what);
    return launch2_any<2, 2, 2, 2>(p, t, stream, what);
}

struct W2Plan {
    int bm, bn, Z;
    size_t lds;
};

W2Plan plan_w2(Wgrad2P& p) {
    W2Plan w{};
    if (p.M <= 32) { w.bm = 32; w.bn = 256; }
    else if (p.M <= 64) { w.bm = 64; w.bn = 128; }
    else if (p.M % 96 == 0) { w.bm = 96; w.bn = 128; }
    else { w.bm = 128; w.bn = 128; }
    const int ncols = p.C * p.T;
    p.nc_max = (w.bn - 1) / p.T + 2;
    if (p.nc_max > p.C) p.nc_max = p.C;
    p.wk = pow2ceil(p.r_w) < 64 ? pow2ceil(p.r_w) : 64;
    if (p.wk < 2) p.wk = 2;
    p.wk_shift = __builtin_ctz(p.wk);
    int span_h = 0, span_w = 0, minh = 0, minw = 0;
    for (int t = 0; t < p.T; ++t) {
        if (t == 0 || p.offh[t] < minh) minh = p.offh[t];
        if (t == 0 || p.offw[t] < minw) minw = p.offw[t];
    }
    for (int t = 0; t < p.T; ++t) {
        span_h = span_h > p.offh[t] - minh ? span_h : p.offh[t] - minh;
        span_w = span_w > p.offw[t] - minw ? span_w : p.offw[t] - minw;
    }
    p.minh = minh;
    p.minw = minw;
    p.PW = (p.wk - 1) * p.is_w + span_w + 1;
    p.PWp = p.PW | 1;
    int rk = 256 / p.wk > 4 ? 256 / p.wk : 4;       // aim at ~256 reduction elements per chunk
    if (rk > pow2ceil(p.r_h)) rk = pow2ceil(p.r_h);
    for (;; rk >>= 1) {
        p.rk = rk;
        p.PH = (rk - 1) * p.is_h + span_h + 1;
        p.chp = (p.PH * p.PWp) | 1;
        p.pr = rk * p.wk + 2;
        w.lds = sizeof(float) * ((size_t)w.bm * p.pr + (size_t)p.nc_max * p.chp);
        static const size_t cap = [] {
            const char* e = getenv("RH_WGRAD2D_LDS_BYTES");
            return e ? (size_t)atol(e) : (size_t)80 * 1024;
        }();
        if (w.lds <= cap || rk == 1) break;
    }
    p.magic_pw = magic32(p.PW);
    p.magic_ph = magic32(p.PH);
    p.chunks_r = rh_cdiv(p.r_h, p.rk);
    p.chunks_w = rh_cdiv(p.r_w, p.wk);
    p.total_chunks = p.B * p.chunks_r * p.chunks_w;
    const int bxy = rh_cdiv(ncols, w.bn) * rh_cdiv(p.M, w.bm);
    int Z = 1024 / (bxy > 0 ? bxy : 1);
    if (Z > p.total_chunks / 4) Z = p.total_chunks / 4;     // at least 4 chunks per slice
    if (Z < 1) Z = 1;
    p.chunks_per_z = rh_cdiv(p.total_chunks, Z);
    w.Z = rh_cdiv(p.total_chunks, p.chunks_per_z);
    return w;
}

template <int TM, int TN, int WM, int WN>
int launch_w2(const Wgrad2P& p, const W2Plan& w, hipStream_t stream) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    RH_REQUIRE(w.lds <= 160 * 1024, RH_ERR_UNSUPPORTED, "conv2d_bwd_weight: tile needs %zu B of LDS", w.lds);
    auto kern = wgrad2d_kernel<TM, TN, WM, WN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    });
    dim3 grid(rh_cdiv(p.C * p.T, BN), rh_cdiv(p.M, BM), w.Z);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), w.lds, stream, p);
    return rh_check_launch("conv2d_bwd_weight");
}


// LDS-DMA weight-gradient plan (M <= 32, no fused activation derivative); false = use the generic kernel
bool plan_w2_dma(Wgrad2P& p, W2Plan* w, int* tn_out) {
    static const bool off = getenv("RH_WGRAD2D_NODMA") != nullptr;
    if (off || p.M > 32 || p.Rmul) return false;
    const long rb = (long)p.B * p.M * p.r_h * p.r_w * 4, sbytes = (long)p.B * p.C * p.s_h * p.s_w * 4;
    if (rb >= (1l << 31) || sbytes >= (1l << 31)) return false;
    const int ncols = p.C * p.T;
    const int TN = ncols <= 128 ? 1 : (ncols <= 256 ? 2 : 4);
    const int BN = TN * 128;
    p.nc_max = (BN - 1) / p.T + 2;
    if (p.nc_max > p.C) p.nc_max = p.C;
    int span_h = 0, span_w = 0, minh = 0, minw = 0;
    for (int t = 0; t < p.T; ++t) {
        if (t == 0 || p.offh[t] < minh) minh = p.offh[t];
        if (t == 0 || p.offw[t] < minw) minw = p.offw[t];
    }
    for (int t = 0; t < p.T; ++t) {
        span_h = span_h > p.offh[t] - minh ? span_h : p.offh[t] - minh;
        span_w = span_w > p.offw[t] - minw ? span_w : p.offw[t] - minw;
    }
    p.minh = minh;
    p.minw = minw;
    static const int stage_cap = [] {
        const char* e = getenv("RH_WGRAD2D_DMA_STAGE_FLOATS");
        return e ? atoi(e) : 10 * 1024;
    }();
    int wk = pow2ceil(p.r_w) < 64 ? pow2ceil(p.r_w) : 64;
    if (wk < 2) wk = 2;
    int rk = 128 / wk > 1 ? 128 / wk : 1;                       // ~128 positions of dy per chunk
    for (;;) {
        if (rk * wk < 64) rk = 64 / wk;                          // a DMA slot is 64 floats
        p.wk = wk; p.rk = rk;
        p.wk_shift = __builtin_ctz(wk);
        p.PH = (rk - 1) * p.is_h + span_h + 1;
        p.PW = (wk - 1) * p.is_w + span_w + 1;
        p.PWp = p.PW;
        p.nr = rk * wk / 64;
        p.ns = (p.PH * p.PW + 63) / 64;
        p.pr = rk * wk + 2;
        p.chp = p.ns * 64 + 1;
        p.r_floats = 32 * p.pr;
        p.stage_floats = p.r_floats + p.nc_max * p.chp;
        const bool fits = p.nr <= kNR && p.ns <= kNS && p.stage_floats <= stage_cap;
        if (fits) break;
        if (rk * wk > 64 && rk > 1) rk >>= 1;                    // shrink the chunk, rows first
        else if (wk > 8) { wk >>= 1; rk = 64 / wk > 1 ? 64 / wk : 1; }
        else if (p.nr <= kNR && p.ns <= kNS && 2 * p.stage_floats * 4 <= 160 * 1024) break;   // over the soft cap only
        else return false;
    }
    p.magic_pw = magic32(p.PW);
    p.magic_ph = magic32(p.PH);
    p.chunks_r = rh_cdiv(p.r_h, p.rk);
    p.chunks_w = rh_cdiv(p.r_w, p.wk);
    p.total_chunks = p.B * p.chunks_r * p.chunks_w;
    p.r_bytes = (unsigned)rb;
    p.s_bytes = (unsigned)sbytes;
    const int bxy = rh_cdiv(ncols, BN);
    int Z = 1024 / bxy;
    if (Z > p.total_chunks / 8) Z = p.total_chunks / 8;          // at least 8 chunks per slice
    if (Z < 1) Z = 1;
    p.chunks_per_z = rh_cdiv(p.total_chunks, Z);
    w->Z = rh_cdiv(p.total_chunks, p.chunks_per_z);
    w->bm = 32;
    w->bn = BN;
    w->lds = sizeof(float) * 2 * (size_t)p.stage_floats;
    *tn_out = TN;
    return true;
}

template <int TN>
int launch_w2_dma(const Wgrad2P& p, const W2Plan& w, hipStream_t stream) {
    auto kern = wgrad2d_dma_kernel<TN>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    });
    dim3 grid(rh_cdiv(p.C * p.T, TN * 128), 1, w.Z);
    hipLaunchKernelGGL(kern, grid, dim3(256), w.lds, stream, p);
    return rh_check_launch("conv2d_bwd_weight");
}

void fill_w2(const rh_conv2d_desc* d, Wgrad2P* p) {
    *p = Wgrad2P{};
    p->B = d->batch; p->M = d->c_out; p->C = d->c_in; p->T = d->kh * d->kw;
    p->r_h = d->h_out; p->r_w = d->w_out; p->s_h = d->h_in; p->s_w = d->w_in;
    p->is_h = d->sh; p->is_w = d->sw;
    p->r_act = d->act; p->r_slope = d->act_slope;
    for (int th = 0; th < d->kh; ++th)
        for (int tw = 0; tw < d->kw; ++tw) {
            p->offh[th * d->kw + tw] = th * d->dh - d->ph;
            p->offw[th * d->kw + tw] = tw * d->dw - d->pw;
        }
}

int fill_pack2(const rh_conv2d_desc* d, int which, const float* w, float* wp, PackP* p) {
    *p = PackP{};
    if (!wp) return RH_OK;
    Plan2 t;
    build_plan2(d, which, &t);
    p->w = w; p->scale = nullptr; p->wp = wp;
    p->C = which == 0 ? d->c_in : d->c_out;
    p->M = which == 0 ? d->c_out : d->c_in;
    p->Mp = round32(p->M);
    p->k = d->kh * d->kw;
    p->m_major = which == 0 ? 1 : 0;      // w[co][ci][kk]: forward m = co (m-major), data gradient m = ci (c-major)
    p->total = (long)t.nslots * p->C * p->Mp;
    p->nslots = t.nslots;
    for (int i = 0; i < t.nslots; ++i) p->kk[i] = t.kk[i];
    // bf16x6 section behind the f32 one (conv2d_x6.hip): fragments [phase][chunk][tap][g][piece][Mp] -- the mode-1 layout
    // of the 1-D packer with (kh, kw) taps as its taps
    long ph_ofs[kPh2];
    const long units = rh_conv2d_x6_units(p->C, p->M, t.nphase, t.ntaps, ph_ofs);
    if (units > 0 && units * 4 < 0x7fffffffl) {
        p->wq = reinterpret_cast<unsigned*>(wp + p->total);
        p->x6_mode = 1;
        p->range = p->wq + units * 4;          // {max |w|, max row sum |w|, 0, 0} behind the fragments (conv_host.hip)
        for (int ph = 0; ph < t.nphase; ++ph)
            for (int tl = 0; tl < t.ntaps[ph]; ++tl) {
                p->q2a[t.tap0[ph] + tl] = (int)(ph_ofs[ph] + (long)tl * 2 * kX6P * p->Mp);
                p->q2n[t.tap0[ph] + tl] = t.ntaps[ph] * 2 * kX6P * p->Mp;
            }
    }
    return RH_OK;
}

// Parameters of the bf16x6 launch (conv2d_x6.hip) from the tap plan; the tile plan is made there.
void fill_c2x(const Plan2& t, int C, int M, const float* wp, C2X* q) {
    *q = C2X{};
    q->C = C; q->M = M; q->Mp = round32(M);
    q->nphase = t.nphase;
    const long total = (long)t.nslots * C * q->Mp;
    const long units = rh_conv2d_x6_units(C, M, t.nphase, t.ntaps, q->ph_q2ofs);
    q->wq = (units > 0 && units * 4 < 0x7fffffffl) ? reinterpret_cast<const unsigned*>(wp + total) : nullptr;
    q->wq_bytes = (unsigned)(units * 16);
    for (int ph = 0; ph < t.nphase; ++ph) {
        q->ph_oph_h[ph] = t.oph_h[ph]; q->ph_oph_w[ph] = t.oph_w[ph];
        q->ph_ntaps[ph] = t.ntaps[ph]; q->ph_tap0[ph] = t.tap0[ph];
        q->ph_minh[ph] = t.minh[ph]; q->ph_minw[ph] = t.minw[ph];
        q->ph_maxh[ph] = t.maxh[ph]; q->ph_maxw[ph] = t.maxw[ph];
    }
    for (int i = 0; i < t.nslots; ++i) { q->offh[i] = t.offh[i]; q->offw[i] = t.offw[i]; }
}

}  // namespace

extern "C" int64_t rh_conv2d_packed_floats(const rh_conv2d_desc* d, int which) {
    if (validate2(d)) return -1;
    const int64_t M = which == 0 ? d->c_out : d->c_in;
    const int64_t C = which == 0 ? d->c_in : d->c_out;
    const int64_t base = (int64_t)d->kh * d->kw * C * round32((int)M);
    // + the bf16x6 fragments of the same weights (6 bytes per weight: conv2d_x6.hip), for channel counts in blocks of 16
    const int T = d->kh * d->kw;
    const long units = rh_conv2d_x6_units((int)C, (int)M, 1, &T, nullptr);
    return base + ((units > 0 && units * 4 < 0x7fffffffl) ? units * 4 + 4 : 0);      // (+ the range record)
}

// Diagnostics (tests): which kernel family a forward (which = 0) / data-gradient (1) launch of this geometry takes and, for
// the bf16x6 kernels, its tile plan.  out[16] = {family (0 = f32-input MFMA, 1 = bf16x6, 2 = vector ALU: conv2d_smallm.hip), tm, tn, tasks per thread, TR, TQ,
// nb, LDS bytes, workgroups, PH, PW, P, largest tap offset inside the patch, phases, row tiles, column tiles}.
extern "C" int rh_conv2d_plan_info(const rh_conv2d_desc* d, int32_t which, int64_t* out) {
    if (int e = validate2(d)) return e;
    RH_REQUIRE(out && (which == 0 || which == 1), RH_ERR_INVALID, "conv2d_plan_info: bad arguments");
    for (int i = 0; i < 16; ++i) out[i] = 0;
    rh_conv2d_desc dd = *d;
    if (which == 1) dd.act = RH_ACT_NONE;              // (the data gradient sees dy with act'(y) folded in)
    if (rh_conv2d_smallm_eligible(&dd, which) || (which == 0 && rh_conv2d_smallc_fwd_eligible(d))) { out[0] = 2; return RH_OK; }
    Plan2 t;
    build_plan2(d, which, &t);
    C2X q;
    static float dummy_w[4] __attribute__((aligned(16)));
    const int C = which == 0 ? d->c_in : d->c_out, M = which == 0 ? d->c_out : d->c_in;
    fill_c2x(t, C, M, dummy_w, &q);
    if (!q.wq) return RH_OK;
    q.wq = reinterpret_cast<const unsigned*>(dummy_w);      // (alignment test only: nothing is launched)
    q.in = dummy_w; q.out = dummy_w;
    q.B = d->batch;
    if (which == 0) {
        q.in_h = d->h_in; q.in_w = d->w_in; q.out_h = d->h_out; q.out_w = d->w_out; q.rows = d->h_out; q.qcols = d->w_out;
        q.is_h = d->sh; q.is_w = d->sw; q.os_h = 1; q.os_w = 1;
    } else {
        q.in_h = d->h_out; q.in_w = d->w_out; q.out_h = d->h_in; q.out_w = d->w_in;
        q.rows = rh_cdiv(d->h_in, d->sh); q.qcols = rh_cdiv(d->w_in, d->sw);
        q.is_h = 1; q.is_w = 1; q.os_h = d->sh; q.os_w = d->sw;
    }
    long v[8];
    if (!rh_conv2d_x6_plan_query(q, v)) return RH_OK;
    out[0] = 1;
    for (int i = 0; i < 8; ++i) out[1 + i] = v[i];
    // (the query works on a copy: redo the geometry-only part for the patch figures)
    int span_h = 0, span_w = 0;
    for (int ph = 0; ph < t.nphase; ++ph) {
        if (t.ntaps[ph] < 1) continue;
        span_h = span_h > t.maxh[ph] - t.minh[ph] ? span_h : t.maxh[ph] - t.minh[ph];
        span_w = span_w > t.maxw[ph] - t.minw[ph] ? span_w : t.maxw[ph] - t.minw[ph];
    }
    const long TR = v[3], TQ = v[4], nb = v[5];
    const long PH = (TR - 1) * q.is_h + span_h + 1, PW = (TQ - 1) * q.is_w + span_w + 1;
    out[9] = PH; out[10] = PW; out[11] = nb * PH * PW;
    out[12] = (long)span_h * PW + span_w;
    out[13] = t.nphase;
    out[14] = rh_cdiv(q.rows, (int)TR); out[15] = rh_cdiv(q.qcols, (int)TQ);
    return RH_OK;
}

extern "C" int rh_conv2d_pack_f32(const rh_conv2d_desc* d, const float* w, float* wp_fwd, float* wp_bwd,
                                  rh_stream_t stream) {
    if (int e = validate2(d)) return e;
    RH_REQUIRE(w, RH_ERR_INVALID, "conv2d_pack: null weight");
    PackP a, b;
    if (int e = fill_pack2(d, 0, w, wp_fwd, &a)) return e;
    if (int e = fill_pack2(d, 1, w, wp_bwd, &b)) return e;
    // (rows = output channels = dim 0 of w, columns = c_in * kh * kw: the range records of both copies, then the pack)
    return rh_pack_launch(a, b, (hipStream_t)stream, "conv2d_pack", w, d->c_out, (long)d->c_in * d->kh * d->kw);
}

extern "C" int rh_conv2d_fwd_f32(const rh_conv2d_desc* d, const float* x, const float* wp_fwd, const float* bias,
                                 float* y, rh_stream_t stream) {
    const unsigned* in_range = nullptr;
    unsigned* out_range = nullptr;
    rh_take_ranges(nullptr, &in_range, &out_range, nullptr);       // consumed by this call whatever happens below
    if (int e = validate2(d)) return e;
    if (d->batch == 0) return RH_OK;
    RH_REQUIRE(x && wp_fwd && y, RH_ERR_INVALID, "conv2d_fwd: null pointer");
    // kernels other than conv2d_x6_kernel do not publish the output's range: a requested slot is filled by a pass over y
    auto range_after = [&](int e) {
        return (e || !out_range) ? e : rh_amax_f32(y, (int64_t)d->batch * d->c_out * d->h_out * d->w_out, out_range, stream);
    };
    Plan2 t;
    build_plan2(d, 0, &t);
    Conv2P p{};
    fill_common(t, &p, d->c_in, d->c_out);
    p.in = x; p.wp = wp_fwd; p.out = y; p.bias = bias; p.in_mul = nullptr;
    p.B = d->batch;
    p.in_h = d->h_in; p.in_w = d->w_in; p.out_h = d->h_out; p.out_w = d->w_out;
    p.rows = d->h_out; p.qcols = d->w_out;
    p.is_h = d->sh; p.is_w = d->sw; p.os_h = 1; p.os_w = 1;
    p.mul_act = RH_ACT_NONE; p.mul_slope = 0.f;
    p.epi_act = d->act; p.epi_slope = d->act_slope;
    if (rh_conv2d_smallm_eligible(d, 0)) return range_after(rh_conv2d_smallm_launch(d, 0, x, wp_fwd, bias, y, (hipStream_t)stream));
    if (rh_conv2d_smallc_fwd_eligible(d)) return range_after(rh_conv2d_smallc_fwd_launch(d, x, wp_fwd, bias, y, (hipStream_t)stream));
    {   // exact f32 on the bf16 matrix cores where the geometry allows (C in blocks of 16): conv2d_x6.hip
        C2X q;
        fill_c2x(t, d->c_in, d->c_out, wp_fwd, &q);
        if (q.wq) {
            q.in = x; q.out = y; q.bias = bias;
            q.B = d->batch;
            q.in_h = d->h_in; q.in_w = d->w_in; q.out_h = d->h_out; q.out_w = d->w_out;
            q.rows = d->h_out; q.qcols = d->w_out;
            q.is_h = d->sh; q.is_w = d->sw; q.os_h = 1; q.os_w = 1;
            q.out_act = d->act; q.out_slope = d->act_slope;
            q.in_range = in_range; q.out_range = out_range;
            bool used = false;
            if (int e = rh_conv2d_x6_launch(q, (hipStream_t)stream, "conv2d_fwd_x6", &used)) return e;
            if (used) return RH_OK;
        }
    }
    return range_after(launch_conv2(p, t, (hipStream_t)stream, "conv2d_fwd"));
}

extern "C" int rh_conv2d_bwd_data_f32(const rh_conv2d_desc* d, const float* dy, const float* y, const float* wp_bwd,
                                      float* dx, rh_stream_t stream) {
    const unsigned* in_range = nullptr;
    unsigned* out_range = nullptr;
    rh_take_ranges(nullptr, &in_range, &out_range, nullptr);
    if (int e = validate2(d)) return e;
    if (d->batch == 0) return RH_OK;
    RH_REQUIRE(dy && wp_bwd && dx, RH_ERR_INVALID, "conv2d_bwd_data: null pointer");
    auto range_after = [&](int e) {
        return (e || !out_range) ? e : rh_amax_f32(dx, (int64_t)d->batch * d->c_in * d->h_in * d->w_in, out_range, stream);
    };
    RH_REQUIRE(d->act == RH_ACT_NONE || y, RH_ERR_INVALID, "conv2d_bwd_data: the output activation needs y");
    Plan2 t;
    build_plan2(d, 1, &t);
    Conv2P p{};
    fill_common(t, &p, d->c_out, d->c_in);
    p.in = dy; p.wp = wp_bwd; p.out = dx; p.bias = nullptr;
    p.in_mul = d->act == RH_ACT_NONE ? nullptr : y;
    p.B = d->batch;
    p.in_h = d->h_out; p.in_w = d->w_out; p.out_h = d->h_in; p.out_w = d->w_in;
    p.rows = rh_cdiv(d->h_in, d->sh); p.qcols = rh_cdiv(d->w_in, d->sw);
    p.is_h = 1; p.is_w = 1; p.os_h = d->sh; p.os_w = d->sw;
    p.mul_act = d->act; p.mul_slope = d->act_slope;
    p.epi_act = RH_ACT_NONE; p.epi_slope = 0.f;
    if (rh_conv2d_smallm_eligible(d, 1)) return range_after(rh_conv2d_smallm_launch(d, 1, dy, wp_bwd, nullptr, dx, (hipStream_t)stream));
    if (d->act == RH_ACT_NONE) {   // (the caller has folded act'(y) into dy: rave_amd.ops._Conv2dFn.backward)
        C2X q;
        fill_c2x(t, d->c_out, d->c_in, wp_bwd, &q);
        if (q.wq) {
            q.in = dy; q.out = dx; q.bias = nullptr;
            q.B = d->batch;
            q.in_h = d->h_out; q.in_w = d->w_out; q.out_h = d->h_in; q.out_w = d->w_in;
            q.rows = rh_cdiv(d->h_in, d->sh); q.qcols = rh_cdiv(d->w_in, d->sw);
            q.is_h = 1; q.is_w = 1; q.os_h = d->sh; q.os_w = d->sw;
            q.out_act = RH_ACT_NONE; q.out_slope = 0.f;
            q.in_range = in_range; q.out_range = out_range;
            bool used = false;
            if (int e = rh_conv2d_x6_launch(q, (hipStream_t)stream, "conv2d_bwd_data_x6", &used)) return e;
            if (used) return RH_OK;
        }
    }
    return range_after(launch_conv2(p, t, (hipStream_t)stream, "conv2d_bwd_data"));
}

// 1 = the weight gradient of this geometry runs on the bf16 matrix cores (wgrad2d_x6.hip), 0 = f32-input MFMA kernels
extern "C" int rh_conv2d_bwd_weight_kernel_family(const rh_conv2d_desc* d) {
    if (validate2(d) || d->act != RH_ACT_NONE) return 0;
    return rh_wgrad2d_x6_workspace(d) >= 0 ? 1 : 0;
}

extern "C" int64_t rh_conv2d_workspace_bytes(const rh_conv2d_desc* d) {
    if (validate2(d)) return -1;
    if (d->batch == 0) return 0;
    Wgrad2P p;
    fill_w2(d, &p);
    W2Plan w{};
    int tn = 0;
    // the fused activation derivative (act != NONE) takes the generic kernel; size for the larger of the two
    Wgrad2P q = p;
    int64_t need = 0;
    if (plan_w2_dma(q, &w, &tn)) need = w.Z > 1 ? (int64_t)w.Z * p.M * p.C * p.T * (int64_t)sizeof(float) : 0;
    const W2Plan g = plan_w2(p);
    const int64_t need_g = g.Z > 1 ? (int64_t)g.Z * p.M * p.C * p.T * (int64_t)sizeof(float) : 0;
    int64_t best = need > need_g ? need : need_g;
    const int64_t need_x6 = rh_wgrad2d_x6_workspace(d);
    if (need_x6 > best) best = need_x6;
    return best + rh_bias_grad_workspace(d->c_out);
}

extern "C" int rh_conv2d_bwd_weight_f32(const rh_conv2d_desc* d, const float* dy, const float* y, const float* x,
                                        float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                                        rh_stream_t stream_) {
    const unsigned *dy_range = nullptr, *x_range = nullptr;
    rh_take_ranges(&dy_range, &x_range, nullptr, nullptr);
    if (int e = validate2(d)) return e;
    hipStream_t stream = (hipStream_t)stream_;
    RH_REQUIRE(dw && (d->batch == 0 || (dy && x)), RH_ERR_INVALID, "conv2d_bwd_weight: null pointer");
    RH_REQUIRE(d->act == RH_ACT_NONE || d->batch == 0 || y, RH_ERR_INVALID,
               "conv2d_bwd_weight: the output activation needs y");
    const long nw = (long)d->c_out * d->c_in * d->kh * d->kw;
    if (d->batch == 0) {
        if (hipError_t e = hipMemsetAsync(dw, 0, nw * sizeof(float), stream)) return (int)e;
        if (dbias)
            if (hipError_t e = hipMemsetAsync(dbias, 0, d->c_out * sizeof(float), stream)) return (int)e;
        return RH_OK;
    }
    const float* ymul = d->act == RH_ACT_NONE ? nullptr : y;
    const int64_t bias_ws = rh_bias_grad_workspace(d->c_out);
    if (dbias) {
        RH_REQUIRE(workspace && workspace_bytes >= bias_ws, RH_ERR_WORKSPACE,
                   "conv2d_bwd_weight: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)bias_ws);
        // the first c_out*64 floats of the workspace; the split-K partials follow
        if (int e = rh_bias_grad_launch(dy, ymul, (float*)workspace, dbias, d->batch, d->c_out,
                                        (long)d->h_out * d->w_out, d->act, d->act_slope, stream))
            return e;
    }
    workspace = workspace ? (void*)((char*)workspace + bias_ws) : nullptr;
    workspace_bytes = workspace_bytes > bias_ws ? workspace_bytes - bias_ws : 0;
    if (d->act == RH_ACT_NONE) {      // (the caller has folded act'(y) into dy) -- bf16x6 kernel where the geometry allows
        bool used = false;
        if (int e = rh_wgrad2d_x6_launch(d, dy, x, dw, workspace, workspace_bytes, stream, &used, dy_range, x_range)) return e;
        if (used) return RH_OK;
    }
    Wgrad2P p;
    fill_w2(d, &p);
    p.R = dy; p.Rmul = ymul; p.S = x;
    W2Plan w{};
    int tn = 0;
    const bool dma = plan_w2_dma(p, &w, &tn);
    if (!dma) w = plan_w2(p);
    const int64_t need = w.Z > 1 ? (int64_t)w.Z * nw * (int64_t)sizeof(float) : 0;
    RH_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), RH_ERR_WORKSPACE,
               "conv2d_bwd_weight: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    p.out = w.Z > 1 ? (float*)workspace : dw;
    int e;
    if (dma) e = tn == 1 ? launch_w2_dma<1>(p, w, stream) : (tn == 2 ? launch_w2_dma<2>(p, w, stream) : launch_w2_dma<4>(p, w, stream));
    else if (w.bm == 32) e = launch_w2<1, 2, 1, 4>(p, w, stream);
    else if (w.bm == 64) e = launch_w2<2, 1, 1, 4>(p, w, stream);
    else if (w.bm == 96) e = launch_w2<3, 1, 1, 4>(p, w, stream);
    else e = launch_w2<2, 2, 2, 2>(p, w, stream);
    if (e) return e;
    if (w.Z > 1) return rh_reduce_partials_launch((const float*)workspace, dw, nw, w.Z, stream, "conv2d_bwd_weight_reduce");
    return RH_OK;
}
