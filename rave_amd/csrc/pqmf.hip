// PQMF analysis / synthesis (rave/pqmf.py CachedPQMF, 16 bands) and their input gradients.
//
// All four transforms are the same GEMM in polyphase form,  M = 16, K = A*16, N = frames,
// computed with v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fmaf chain):
//
//   W2B (waveform -> bands):  out[b,k,n]   = sgn(k,n) * sum_{a,r} F[a][r][k] * in[b, 16(n+a) + r - shift]
//        analysis forward (rave/pqmf.py:279-283)   and   synthesis backward
//   B2W (bands -> waveform):  out[b,16n+i] =            sum_{a,c} F[a][c][i] * sgn(c,m) in[b,c,m],  m = n+a-sh
//        synthesis forward (rave/pqmf.py:285-294)  and   analysis backward
//
// sgn(k,n) = -1 iff k odd and n even == reverse_half (rave/pqmf.py:13-17); it is pure index
// arithmetic and therefore bit-exact.  F is gathered on the fly from the module's own weight
// tensor (forward_conv.weight (16,1,513) / inverse_conv.weight (16,16,33)) while staging the
// A operand into LDS, so a checkpoint with different filter taps is honoured.
//
// Data movement per block (256 frames x 16 bands): one coalesced pass over the contiguous
// waveform frames / band rows into LDS (bank-conflict-free pitches), filter taps in LDS,
// MFMA accumulators in registers, 16-byte coalesced stores.  HBM traffic = 8 B/sample.
#include "common.hpp"

namespace {

constexpr int kBands = 16;
constexpr int kFramesPerBlock = 256;  // 4 waves x 64 frames
constexpr int kMaxA = 40;

enum FilterMode { F_ANALYSIS_FWD = 0, F_SYNTH_BWD = 1, F_SYNTH_FWD = 2, F_ANALYSIS_BWD = 3 };

struct PqmfP {
    const float* in;
    const float* w;
    float* out;
    int rows;      // B * channels
    int in_len;    // W2B: samples per row ; B2W: frames per band row
    int out_len;   // W2B: frames per band row ; B2W: samples per row
    int n_tiles;   // column tiles per row
    int A;         // polyphase taps
    int shift;     // W2B: sample shift ; B2W: frame shift `sh`
    int K;         // kernel length of w's last dim
    int pad_left;
    int mode;
};

// A operand element F[a][x][i]  (kidx = a*16 + x), see file header + include/rave_hip.h.
__device__ __forceinline__ float filter_elem(const PqmfP& p, int a, int x, int i) {
    switch (p.mode) {
        case F_ANALYSIS_FWD: {  // F[a][r][k] = Wf[k, 16a + r]
            const int t = 16 * a + x;
            return t < p.K ? p.w[i * p.K + t] : 0.f;
        }
        case F_SYNTH_BWD: {  // F[a][r][c] = 16 * Wi[15 - r, c, K2 - 1 - a]
            const int t = p.K - 1 - a;
            return t >= 0 ? 16.f * p.w[((15 - x) * kBands + i) * p.K + t] : 0.f;
        }
        case F_SYNTH_FWD: {  // F[a][c][i] = 16 * Wi[15 - i, c, a]
            return a < p.K ? 16.f * p.w[((15 - i) * kBands + x) * p.K + a] : 0.f;
        }
        default: {  // F_ANALYSIS_BWD: F[a][k][r] = Wf[k, r + pad_left + 16 (sh - a)]
            const int t = i + p.pad_left + 16 * (p.shift - a);
            return (t >= 0 && t < p.K) ? p.w[x * p.K + t] : 0.f;
        }
    }
}

// ---- waveform -> bands ------------------------------------------------------------------
__global__ __launch_bounds__(256) void pqmf_w2b_kernel(PqmfP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a_lds = smem;                          // [A*16][16]
    float* x_lds = smem + p.A * 16 * kBands;      // [(256 + A) rows][17]
    const int tid = threadIdx.x;
    const int row = blockIdx.x / p.n_tiles;
    const int n0 = (blockIdx.x % p.n_tiles) * kFramesPerBlock;

    for (int e = tid; e < p.A * 16 * kBands; e += 256) {
        const int i = e & 15, kidx = e >> 4;
        a_lds[e] = filter_elem(p, kidx >> 4, kidx & 15, i);
    }
    const int n_samp = 16 * (kFramesPerBlock + p.A);
    const long base = 16l * n0 - p.shift;
    const float* in = p.in + (long)row * p.in_len;
    for (int e = tid; e < n_samp; e += 256) {
        const long s = base + e;
        const float v = (s >= 0 && s < p.in_len) ? in[s] : 0.f;
        x_lds[(e >> 4) * 17 + (e & 15)] = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = a_lds + kq * 16 + j;                 // + ks*64
    const float* bp = x_lds + (wave * 64 + j) * 17 + kq;   // + tn*16*17 + (ks/4)*17 + (ks%4)*4
    const int ksteps = p.A * 4;
    for (int ks = 0; ks < ksteps; ++ks) {
        const float a = ap[ks * 64];
        const int boff = (ks >> 2) * 17 + (ks & 3) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float b = bp[t * 16 * 17 + boff];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
    float* out = p.out + (long)row * kBands * p.out_len;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + wave * 64 + t * 16 + j;
        if (n < p.out_len) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int band = kq * 4 + r;
                float v = acc[t][r];
                if ((band & 1) && !(n & 1)) v = -v;
                out[(long)band * p.out_len + n] = v;
            }
        }
    }
}

// ---- bands -> waveform ------------------------------------------------------------------
__global__ __launch_bounds__(256) void pqmf_b2w_kernel(PqmfP p, int pitch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a_lds = smem;                       // [A*16][16]
    float* u_lds = smem + p.A * 16 * kBands;   // [16][pitch]
    const int tid = threadIdx.x;
    const int row = blockIdx.x / p.n_tiles;
    const int n0 = (blockIdx.x % p.n_tiles) * kFramesPerBlock;

    for (int e = tid; e < p.A * 16 * kBands; e += 256) {
        const int i = e & 15, kidx = e >> 4;
        a_lds[e] = filter_elem(p, kidx >> 4, kidx & 15, i);
    }
    const int width = kFramesPerBlock + p.A;
    const float* in = p.in + (long)row * kBands * p.in_len;
    for (int c = tid >> 6; c < kBands; c += 4) {
        for (int e = tid & 63; e < width; e += 64) {
            const int m = n0 + e - p.shift;
            float v = 0.f;
            if (m >= 0 && m < p.in_len) {
                v = in[(long)c * p.in_len + m];
                if ((c & 1) && !(m & 1)) v = -v;
            }
            u_lds[c * pitch + e] = v;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = a_lds + kq * 16 + j;
    const float* bp = u_lds + kq * pitch + wave * 64 + j;  // + tn*16 + a + (ks%4)*4*pitch
    const int ksteps = p.A * 4;
    for (int ks = 0; ks < ksteps; ++ks) {
        const float a = ap[ks * 64];
        const int boff = (ks >> 2) + (ks & 3) * 4 * pitch;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float b = bp[t * 16 + boff];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
    float* out = p.out + (long)row * p.out_len;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + wave * 64 + t * 16 + j;
        const long o = 16l * n + kq * 4;
        if (o + 3 < p.out_len && (p.out_len & 3) == 0) {
            *reinterpret_cast<f32x4*>(out + o) = acc[t];  // 16-byte aligned: rows start at multiples of 4 floats
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (o + r < p.out_len) out[o + r] = acc[t][r];
        }
    }
}

int launch(const PqmfP& p, bool w2b, hipStream_t stream) {
    const int grid = p.rows * p.n_tiles;
    if (grid <= 0) return RH_OK;
    if (w2b) {
        const size_t lds = sizeof(float) * (size_t)(p.A * 16 * kBands + (kFramesPerBlock + p.A) * 17);
        hipLaunchKernelGGL(pqmf_w2b_kernel, dim3(grid), dim3(256), lds, stream, p);
    } else {
        int pitch = kFramesPerBlock + p.A;
        pitch += ((16 - (pitch & 31)) + 32) & 31;  // pitch % 32 == 16: the two band rows read by a
                                                   // 32-lane group land on disjoint bank halves
        const size_t lds = sizeof(float) * (size_t)(p.A * 16 * kBands + kBands * pitch);
        hipLaunchKernelGGL(pqmf_b2w_kernel, dim3(grid), dim3(256), lds, stream, p, pitch);
    }
    return rh_check_launch(w2b ? "pqmf_w2b" : "pqmf_b2w");
}

int check_common(const void* a, const void* b, const void* c, int rows, int n_band) {
    RH_REQUIRE(rows >= 0, RH_ERR_INVALID, "pqmf: rows < 0");
    RH_REQUIRE(rows == 0 || (a && b && c), RH_ERR_INVALID, "pqmf: null pointer");
    RH_REQUIRE(n_band == kBands, RH_ERR_UNSUPPORTED, "pqmf: only n_band == 16 is implemented (got %d)", n_band);
    return RH_OK;
}

}  // namespace

extern "C" int rh_pqmf_analysis_fwd_f32(const float* x, const float* w, int32_t rows, int32_t t_len,
                                        int32_t n_band, int32_t kernel, int32_t pad_left,
                                        int32_t n_frames, float* y, rh_stream_t stream) {
    if (int e = check_common(x, w, y, rows, n_band)) return e;
    PqmfP p{};
    p.in = x; p.w = w; p.out = y; p.rows = rows; p.in_len = t_len; p.out_len = n_frames;
    p.n_tiles = rh_cdiv(n_frames, kFramesPerBlock);
    p.A = rh_cdiv(kernel, 16); p.shift = pad_left; p.K = kernel; p.pad_left = pad_left;
    p.mode = F_ANALYSIS_FWD;
    RH_REQUIRE(p.A <= kMaxA && kernel > 0, RH_ERR_UNSUPPORTED, "pqmf analysis: kernel %d too long", kernel);
    return launch(p, true, (hipStream_t)stream);
}

extern "C" int rh_pqmf_analysis_bwd_f32(const float* dy, const float* w, int32_t rows, int32_t t_len,
                                        int32_t n_band, int32_t kernel, int32_t pad_left,
                                        int32_t n_frames, float* dx, rh_stream_t stream) {
    if (int e = check_common(dy, w, dx, rows, n_band)) return e;
    PqmfP p{};
    p.in = dy; p.w = w; p.out = dx; p.rows = rows; p.in_len = n_frames; p.out_len = t_len;
    p.n_tiles = rh_cdiv(rh_cdiv(t_len, 16), kFramesPerBlock);
    int sh = kernel - 1 - pad_left;
    sh = sh > 0 ? rh_cdiv(sh, 16) : 0;
    p.shift = sh; p.K = kernel; p.pad_left = pad_left;
    p.A = (15 + pad_left + 16 * sh) / 16 + 1;
    p.mode = F_ANALYSIS_BWD;
    RH_REQUIRE(p.A <= kMaxA && pad_left >= 0, RH_ERR_UNSUPPORTED, "pqmf analysis bwd: unsupported geometry");
    return launch(p, false, (hipStream_t)stream);
}

extern "C" int rh_pqmf_synthesis_fwd_f32(const float* y, const float* w, int32_t rows,
                                         int32_t n_frames, int32_t n_band, int32_t kernel,
                                         int32_t pad_left, int32_t n_out, float* x,
                                         rh_stream_t stream) {
    if (int e = check_common(y, w, x, rows, n_band)) return e;
    PqmfP p{};
    p.in = y; p.w = w; p.out = x; p.rows = rows; p.in_len = n_frames; p.out_len = n_out * 16;
    p.n_tiles = rh_cdiv(n_out, kFramesPerBlock);
    p.A = kernel; p.shift = pad_left; p.K = kernel; p.pad_left = pad_left;
    p.mode = F_SYNTH_FWD;
    RH_REQUIRE(p.A <= kMaxA && kernel > 0, RH_ERR_UNSUPPORTED, "pqmf synthesis: kernel %d too long", kernel);
    return launch(p, false, (hipStream_t)stream);
}

extern "C" int rh_pqmf_synthesis_bwd_f32(const float* dx, const float* w, int32_t rows,
                                         int32_t n_frames, int32_t n_band, int32_t kernel,
                                         int32_t pad_left, int32_t n_out, float* dy,
                                         rh_stream_t stream) {
    if (int e = check_common(dx, w, dy, rows, n_band)) return e;
    PqmfP p{};
    p.in = dx; p.w = w; p.out = dy; p.rows = rows; p.in_len = n_out * 16; p.out_len = n_frames;
    p.n_tiles = rh_cdiv(n_frames, kFramesPerBlock);
    p.A = kernel; p.shift = 16 * (kernel - 1 - pad_left); p.K = kernel; p.pad_left = pad_left;
    p.mode = F_SYNTH_BWD;
    RH_REQUIRE(p.A <= kMaxA && kernel > 0, RH_ERR_UNSUPPORTED, "pqmf synthesis bwd: kernel %d too long", kernel);
    return launch(p, true, (hipStream_t)stream);
}
