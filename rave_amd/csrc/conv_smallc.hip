// First layers of the discriminators (rave/discriminator.py:77-100: ConvNet's first conv reads the 1- or 2-channel
// waveform, k = 15 / 5, stride 4 / period fold): C_in <= 2, so the "GEMM" has K = k * C_in <= 30 and is pure data
// movement -- one tensor of C_out x L floats (400 MB at the BASELINE sizes) written (forward) or read (both gradients)
// once.  A 32-row MFMA tile is 3-6 % used here and the generic kernels reach 0.5 - 2 TB/s; these three kernels are
// vector-ALU code shaped for the memory system instead: lanes run along the position axis (every load / store of the
// big tensor is a full 256-byte line per wave), the tiny operand (the waveform window, the weights) lives in
// registers / LDS.
//   forward  : thread = output position; its k*C input samples sit in registers, the weights are LDS broadcasts,
//              one coalesced store per output channel.
//   dgrad    : thread = dy position; accumulates its k*C contributions over all output channels, the overlapping
//              windows are summed through LDS in a fixed order (deterministic), coalesced store of dx.
//   wgrad    : thread = position, workgroup = 8 ... 24 output channels x a strided set of position blocks; per-thread
//              accumulators [rows][k*C], ordered wave / workgroup reduction, partial tiles + reduce_partials_kernel.
// Algorithmic bytes: 4 * B * C_out * L_out (+ the waveform): the HBM roofline of each launch.
#include <cstdlib>
#include <type_traits>
#include "conv_params.hpp"

namespace {

struct SmallP {
    const float* x;        // forward / wgrad: input [B][C][l_in];   dgrad: unused
    const float* big;      // forward: unused;  dgrad / wgrad: dy [B][M][l_out]
    const float* w;        // packed f32 operand (forward: wp_fwd [t][c][Mp];  dgrad: wp_bwd [slot][m][32])
    const float* bias;
    float* out;            // forward: y;  dgrad: dx;  wgrad: partial tiles [chunks][M][C*K]
    int B, M, Mp, l_in, l_out, s, pad;
    int out_act;
    float out_slope;
    int nblk;              // position blocks per batch item
    int total_blocks;      // wgrad: B * nblk
    int slot[16];          // dgrad: packed slot of tap t
    unsigned* out_range;   // forward: range slot that receives max |y| (consumers on the f16 kernels, common.hpp) or null
};

template <int K, int C>
__global__ __launch_bounds__(256) void smallc_fwd_kernel(const SmallP p) {
    constexpr int KC = K * C, PITCH = (KC + 3) & ~3;
    extern __shared__ float wl[];                       // [M][PITCH]: w[m][c][t]
    const int tid = threadIdx.x;
    for (int e = tid; e < p.M * KC; e += 256) {
        const int m = e / KC, i = e - m * KC;
        const int c = i / K, t = i - c * K;
        wl[m * PITCH + i] = p.w[((long)t * C + c) * p.Mp + m];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + tid;
    float amax = 0.f;
    if (n < p.l_out) {
        float xv[KC];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int pos = n * p.s - p.pad + t;
                xv[c * K + t] = (pos >= 0 && pos < p.l_in) ? p.x[((long)b * C + c) * p.l_in + pos] : 0.f;
            }
        float* __restrict__ dst = p.out + (long)b * p.M * p.l_out + n;
        const bool leaky = p.out_act == RH_ACT_LEAKY;
#pragma unroll 2
        for (int m = 0; m < p.M; ++m) {
            float acc = p.bias ? p.bias[m] : 0.f;
#pragma unroll
            for (int i = 0; i < KC; ++i) acc = fmaf(wl[m * PITCH + i], xv[i], acc);
            if (leaky) acc = acc > 0.f ? acc : acc * p.out_slope;
            dst[(long)m * p.l_out] = acc;
            amax = fmaxf(amax, fabsf(acc));
        }
    }
    if (p.out_range) {           // (uniform; every thread of the workgroup gets here)
        __shared__ float red[4];
        rh_range_publish(p.out_range, amax, blockIdx.x + blockIdx.y * 7u, red);
    }
}

template <int K, int C>
__global__ __launch_bounds__(256) void smallc_dgrad_kernel(const SmallP p) {
    constexpr int KC = K * C, PITCH = (KC + 3) & ~3, VP = KC + 1;
    extern __shared__ float sm[];
    float* wl = sm;                                      // [M][PITCH]: w[m][c][t]
    float* vals = sm + p.M * PITCH;                      // [256][VP]
    const int tid = threadIdx.x;
    for (int e = tid; e < p.M * KC; e += 256) {
        const int m = e / KC, i = e - m * KC;
        const int c = i / K, t = i - c * K;
        wl[m * PITCH + i] = p.w[((long)p.slot[t] * p.M + m) * 32 + c];
    }
    __syncthreads();
    const int H = (K - 1 + p.s - 1) / p.s;               // dy positions a dx sample can reach on either side
    const int NB = 256 - 2 * H;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * NB;
    const int n = n0 - H + tid;
    const bool valid = n >= 0 && n < p.l_out;
    float o[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) o[i] = 0.f;
    const float* __restrict__ src = p.big + (long)b * p.M * p.l_out + (valid ? n : 0);
#pragma unroll 2
    for (int m = 0; m < p.M; ++m) {
        const float g = valid ? src[(long)m * p.l_out] : 0.f;
#pragma unroll
        for (int i = 0; i < KC; ++i) o[i] = fmaf(wl[m * PITCH + i], g, o[i]);
    }
#pragma unroll
    for (int i = 0; i < KC; ++i) vals[tid * VP + i] = o[i];
    __syncthreads();
    // dx[p] = sum over taps t = (p + pad) mod s, + s, ... of the contribution of dy position (p + pad - t) / s
    const int np = NB * p.s;
    for (int j = tid; j < np; j += 256) {
        const int pp = n0 * p.s + j;
        if (pp >= p.l_in) break;
        const int q = pp + p.pad;
        const int r0 = q % p.s;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
            for (int t = r0; t < K; t += p.s) {
                const int nl = (q - t) / p.s - (n0 - H);
                if (nl >= 0 && nl < 256) acc += vals[nl * VP + c * K + t];
            }
            p.out[((long)b * C + c) * p.l_in + pp] = acc;
        }
    }
}

template <int K, int C, int R>
__global__ __launch_bounds__(256) void smallc_wgrad_kernel(const SmallP p) {
    constexpr int KC = K * C;
    __shared__ float red[4][R * (KC + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * R;
    float acc[R][KC], accb[R];                           // accb: the bias gradient, sum of dy over (batch, position)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        accb[r] = 0.f;
#pragma unroll
        for (int i = 0; i < KC; ++i) acc[r][i] = 0.f;
    }
    // no barrier in the loop and the next block's operands are requested before this block's FMAs: the loop is a
    // stream of loads with two blocks in flight per thread (with one it ran at the memory latency: 1.2 TB/s)
    float g[2][R], xv[2][KC];
    auto fetch = [&](int blk, float (&gg)[R], float (&xx)[KC]) {
        const bool live = blk < p.total_blocks;
        const int b = live ? blk / p.nblk : 0;
        const int n = (live ? blk - b * p.nblk : 0) * 256 + tid;
        const bool valid = live && n < p.l_out;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int pos = n * p.s - p.pad + t;
                xx[c * K + t] = (valid && pos >= 0 && pos < p.l_in) ? p.x[((long)b * C + c) * p.l_in + pos] : 0.f;
            }
        const float* __restrict__ src = p.big + ((long)b * p.M + m0) * p.l_out + (valid ? n : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) gg[r] = (valid && m0 + r < p.M) ? src[(long)r * p.l_out] : 0.f;
    };
    auto fma_block = [&](const float (&gg)[R], const float (&xx)[KC]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            accb[r] += gg[r];
#pragma unroll
            for (int i = 0; i < KC; ++i) acc[r][i] = fmaf(gg[r], xx[i], acc[r][i]);
        }
    };
    const int stride = gridDim.x;
    fetch(blockIdx.x, g[0], xv[0]);
    for (int blk = blockIdx.x; blk < p.total_blocks; blk += 2 * stride) {
        fetch(blk + stride, g[1], xv[1]);
        fma_block(g[0], xv[0]);
        fetch(blk + 2 * stride, g[0], xv[0]);
        fma_block(g[1], xv[1]);
    }
    // ordered reduction: butterfly inside the wave, then the four waves in index order
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i <= KC; ++i) {
            float v = i < KC ? acc[r][i < KC ? i : 0] : accb[r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wave][r * (KC + 1) + i] = v;
        }
    __syncthreads();
    // partial tiles: [chunks][M][KC] weight gradients, then [chunks][M] bias gradients
    float* const part_b = p.out + (long)gridDim.x * p.M * KC;
    for (int e = tid; e < R * (KC + 1); e += 256) {
        const int r = e / (KC + 1), i = e - r * (KC + 1);
        if (m0 + r >= p.M) continue;
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (i < KC) p.out[((long)blockIdx.x * p.M + m0 + r) * KC + i] = v;
        else part_b[(long)blockIdx.x * p.M + m0 + r] = v;
    }
}

bool shape_ok(const rh_conv1d_desc* d) {
    return !d->transposed && d->inner == 1 && d->dilation == 1 && d->groups == 1 && d->in_valid == 0 &&
           (d->c_in == 1 || d->c_in == 2) && (d->kernel == 5 || d->kernel == 15) && d->stride >= 1 && d->stride <= 8 &&
           d->c_out >= 8 && d->c_out <= 256 && d->act == RH_ACT_NONE && d->batch > 0 && d->batch <= 65535 &&
           d->l_out > 0 && d->l_in > 0 &&
           // the data gradient's halo ceil((K-1)/s) and its `nl < 256` guard assume the window starts inside the taps
           d->pad_left >= 0 && d->pad_left < d->kernel;
}

SmallP base(const rh_conv1d_desc* d) {
    SmallP p{};
    p.B = d->batch; p.M = d->c_out; p.Mp = (d->c_out + 31) & ~31;
    p.l_in = d->l_in; p.l_out = d->l_out; p.s = d->stride; p.pad = d->pad_left;
    p.out_act = d->out_act; p.out_slope = d->out_slope;
    return p;
}

template <typename F>
void by_shape(const rh_conv1d_desc* d, F&& f) {
    if (d->kernel == 5 && d->c_in == 1) f(std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{});
    else if (d->kernel == 5) f(std::integral_constant<int, 5>{}, std::integral_constant<int, 2>{});
    else if (d->c_in == 1) f(std::integral_constant<int, 15>{}, std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 15>{}, std::integral_constant<int, 2>{});
}

constexpr int rows_of(int kc) { return kc <= 5 ? 24 : (kc <= 10 ? 12 : (kc <= 15 ? 8 : 4)); }

}  // namespace

static bool smallc_enabled() {
    const char* e = getenv("RH_SMALLC");          // read per call: the parity tests flip it at run time
    return !(e && atoi(e) == 0);
}

bool rh_smallc_fwd_eligible(const rh_conv1d_desc* d, bool has_residual) {
    return smallc_enabled() && shape_ok(d) && !has_residual && (d->out_act == RH_ACT_NONE || d->out_act == RH_ACT_LEAKY);
}

int rh_smallc_fwd(const rh_conv1d_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y,
                  hipStream_t stream, unsigned* out_range) {
    SmallP p = base(d);
    p.x = x; p.w = wp_fwd; p.bias = bias; p.out = y;
    p.out_range = out_range;
    by_shape(d, [&](auto kc, auto cc) {
        constexpr int K = decltype(kc)::value, C = decltype(cc)::value;
        const size_t lds = (size_t)p.M * ((K * C + 3) & ~3) * sizeof(float);
        hipLaunchKernelGGL((smallc_fwd_kernel<K, C>), dim3(rh_cdiv(p.l_out, 256), p.B), dim3(256), lds, stream, p);
    });
    return rh_check_launch("conv1d_fwd_smallc");
}

bool rh_smallc_dgrad_eligible(const rh_conv1d_desc* d, bool has_add) {
    if (!(smallc_enabled() && shape_ok(d) && !has_add && (d->kernel - 1 + d->stride - 1) / d->stride <= 16)) return false;
    const int kc = d->kernel * d->c_in;
    // weights + the per-position contributions must fit the default dynamic LDS limit
    return ((size_t)d->c_out * ((kc + 3) & ~3) + 256 * (size_t)(kc + 1)) * sizeof(float) <= 64 * 1024;
}

int rh_smallc_dgrad(const rh_conv1d_desc* d, const float* dy, const float* wp_bwd, const int* slot_of_tap, float* dx,
                    hipStream_t stream) {
    SmallP p = base(d);
    p.big = dy; p.w = wp_bwd; p.out = dx;
    for (int t = 0; t < d->kernel; ++t) p.slot[t] = slot_of_tap[t];
    const int H = (d->kernel - 1 + d->stride - 1) / d->stride;
    const int NB = 256 - 2 * H;
    by_shape(d, [&](auto kc, auto cc) {
        constexpr int K = decltype(kc)::value, C = decltype(cc)::value;
        const size_t lds = ((size_t)p.M * ((K * C + 3) & ~3) + 256 * (K * C + 1)) * sizeof(float);
        hipLaunchKernelGGL((smallc_dgrad_kernel<K, C>), dim3(rh_cdiv(p.l_in, NB * p.s), p.B), dim3(256), lds, stream, p);
    });
    return rh_check_launch("conv1d_bwd_data_smallc");
}

bool rh_smallc_wgrad_eligible(const rh_conv1d_desc* d) { return smallc_enabled() && shape_ok(d); }

static int wgrad_chunks(const rh_conv1d_desc* d) {
    const int R = rows_of(d->kernel * d->c_in);
    const long blocks = (long)d->batch * rh_cdiv(d->l_out, 256);
    // one resident round of workgroups (3 per CU at ~160 VGPRs) over the row groups; at least 16 position blocks per
    // workgroup (the final reduction of the per-thread accumulators costs about as much as 4 blocks)
    static const int target = [] { const char* e = getenv("RH_SMALLC_WGS"); return e ? atoi(e) : 768; }();
    long chunks = rh_cdiv(target, rh_cdiv(d->c_out, R));
    if (chunks > blocks / 16) chunks = blocks / 16;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

int64_t rh_smallc_wgrad_workspace(const rh_conv1d_desc* d) {
    return (int64_t)wgrad_chunks(d) * d->c_out * (d->c_in * d->kernel + 1) * (int64_t)sizeof(float);
}

// dw and (when dbias != null) the bias gradient from ONE pass over dy
int rh_smallc_wgrad(const rh_conv1d_desc* d, const float* dy, const float* x, float* dw, float* dbias, void* ws,
                    hipStream_t stream) {
    SmallP p = base(d);
    p.x = x; p.big = dy;
    p.nblk = rh_cdiv(p.l_out, 256);
    p.total_blocks = p.B * p.nblk;
    const int chunks = wgrad_chunks(d);
    const long nw = (long)d->c_out * d->c_in * d->kernel;
    p.out = (float*)ws;
    by_shape(d, [&](auto kc, auto cc) {
        constexpr int K = decltype(kc)::value, C = decltype(cc)::value, R = rows_of(K * C);
        hipLaunchKernelGGL((smallc_wgrad_kernel<K, C, R>), dim3(chunks, rh_cdiv(p.M, R)), dim3(256), 0, stream, p);
    });
    if (int e = rh_check_launch("conv1d_bwd_weight_smallc")) return e;
    if (int e = rh_reduce_partials_launch((const float*)ws, dw, nw, chunks, stream, "conv1d_bwd_weight_smallc_reduce")) return e;
    if (dbias) return rh_reduce_partials_launch((const float*)ws + (long)chunks * nw, dbias, d->c_out, chunks, stream, "conv1d_bwd_bias_smallc_reduce");
    return RH_OK;
}
