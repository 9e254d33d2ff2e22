// Spectral distance of AudioDistanceV1 for one STFT scale with the transform IN the kernel (rave/core.py:269-344:
// torchaudio Spectrogram(n_fft, hop = n_fft/4, center, reflect, normalized window, power = None) -> |.| -> relative L2 +
// L1 of logs), forward and backward.  The unfused path (misc.hip framing kernels + rocFFT + spectral_*_kernel) moves the
// 4x-overlapped frames and both complex spectrograms through HBM four times per direction (270 MB per scale for 16 MB of
// signal; 1.9 ms of a 12 ms training step over the 2 x 5 scales); here a frame lives in LDS from the windowed load to the
// partial sums (forward) or to the overlap-added gradient (backward): HBM traffic = the two signals, read once (forward)
// or read + their two gradients (backward).
//
//   * one complex n-point transform per frame PAIR: z = w (x + i y); X_k = (Z_k + conj Z_{n-k}) / 2,
//     Y_k = (Z_k - conj Z_{n-k}) / 2i.  Stockham autosort, radix 8 (+ one radix 2 / 4 pass), n / 8 threads per transform
//     holding 8 points each, 256 / (n / 8) transforms per workgroup side by side; LDS image padded by one slot per 8 so
//     that the stride-8 stores of the first pass are conflict free; twiddles from a host-made table (f64 -> f32).
//   * the two signals of a pair are EQUALISED first (round 5): y enters as s y with s = 2^(exponent of max |x| over the frame -
//     exponent of max |y|), exact in floating point, and Y = Y' / s after the transform.  A transform's rounding error is
//     ~ 1e-7 of the norm of its WHOLE input, so without this the quieter signal inherits the louder one's noise: at the start
//     of training the decoder's output is ~ 50x below the target and d log(|Y| + 1e-7) turned that noise into a multiband
//     gradient 60 % off the f64 value, where torch's separate f32 transforms are 0.3 % off
//     (profiles/round5_stft_pair_equalisation.txt).  In the backward the gradient operand is scaled the other way
//     (Wy / s, result x s), for the same reason: |Wy| ~ s |Wx|.
//   * backward: the same transform again (the spectra are not stored), the gradient w.r.t. both magnitudes turned into
//     the Hermitian-extended operand W = Wx + i Wy in place, the unnormalised inverse as swap(FFT(swap(W))) -> real part
//     = d frame_x, imaginary part = d frame_y, window, overlap-add into an LDS image of the workgroup's span of the padded
//     row.  A workgroup owns whole hop blocks and recomputes the 3 frames that reach in from the left (no atomics: the
//     result is bit-reproducible); frames are taken in four phases by (frame mod 4) -- frames of one phase do not overlap
//     -- so concurrent transforms add to disjoint samples.  The first / last workgroup of a row owns the reflected
//     margins together with the samples they fold onto, folds in LDS, and the span is written (or added: the scales of a
//     multi-scale distance accumulate in place) to dx / dy in one coalesced pass.
//   * sums[0] = sum (a-b)^2, sums[1] = sum a^2, sums[2] = sum |log(a+eps) - log(b+eps)| over all bins (a = |X|, b = |Y|):
//     per-workgroup partials + ordered finalize, as spectral_partials_kernel; same gradient formula as spectral_bwd_kernel.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float c32 __attribute__((ext_vector_type(2)));       // complex as a packed pair: adds / twiddle products are v_pk_* ops

__device__ __forceinline__ c32 mk(float a, float b) { c32 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) {
    c32 r = a.xx * w;
    return __builtin_elementwise_fma(a.yy, mk(-w.y, w.x), r);
}
__device__ __forceinline__ c32 mul_mi(c32 a) { return mk(a.y, -a.x); }          // a * (-i)

// forward DFTs (e^{-i}) of 2 / 4 / 8 points in natural order, in place on v[0], v[S], v[2S] ...
template <int S>
__device__ __forceinline__ void dft2(c32* v) {
    const c32 a = v[0], b = v[S];
    v[0] = a + b;
    v[S] = a - b;
}
template <int S>
__device__ __forceinline__ void dft4(c32* v) {
    const c32 b0 = v[0] + v[2 * S], b1 = v[0] - v[2 * S];
    const c32 b2 = v[S] + v[3 * S], b3 = mul_mi(v[S] - v[3 * S]);
    v[0] = b0 + b2;
    v[S] = b1 + b3;
    v[2 * S] = b0 - b2;
    v[3 * S] = b1 - b3;
}
__device__ __forceinline__ void dft8(c32* v) {
    dft4<2>(v);          // even samples -> E_k at v[2k]
    dft4<2>(v + 1);      // odd samples  -> O_k at v[2k+1]
    constexpr float kR = 0.70710678118654752440f;
    const c32 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const c32 o0 = v[1];
    const c32 o1 = mk(v[3].x + v[3].y, v[3].y - v[3].x) * kR;         // * (1 - i) / sqrt 2
    const c32 o2 = mul_mi(v[5]);
    const c32 o3 = mk(v[7].y - v[7].x, -(v[7].x + v[7].y)) * kR;      // * (-1 - i) / sqrt 2
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}
template <int R>
__device__ __forceinline__ void dft(c32* v) {
    if (R == 8) dft8(v);
    else if (R == 4) dft4<1>(v);
    else dft2<1>(v);
}

__device__ __forceinline__ int zpad(int i) { return i + (i >> 3); }

template <int N>
struct Fft {
    static constexpr int TPF = N / 8;                       // threads per transform
    static constexpr int G = 256 / TPF;                     // transforms per workgroup
    static constexpr int ZP = N + N / 8;                    // padded LDS image
    static constexpr int P8 = N >= 512 ? 3 : 2;             // radix-8 passes
    static constexpr int LAST = N / (P8 == 3 ? 512 : 64);   // then one pass of radix 2 / 4 (1: none)
    static constexpr int RF = LAST > 1 ? LAST : 8;          // radix of the final pass
    static constexpr int NSF = N / RF;
    static constexpr int NSL = LAST > 1 ? (P8 == 3 ? 512 : 64) : (P8 == 3 ? 64 : 8);    // NS of the final pass
    static constexpr bool MID2 = P8 == 3 && LAST > 1;       // a second middle pass (NS = 64)
    static constexpr int NTF = (RF - 1) * (8 / RF);

    // A transform's threads sit in ONE wave when TPF <= 64: LDS traffic of a wave is executed in order, no barrier needed
    static __device__ __forceinline__ void sync() {
        if (TPF > 64) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
    }

    // every twiddle a thread ever needs depends on its index alone: fetched once, kept in registers across the frames
    struct Tw {
        c32 a[7];            // pass NS = 8
        c32 b[MID2 ? 7 : 1]; // pass NS = 64 (1024 / 2048 only)
        c32 f[NTF];          // final pass
        __device__ __forceinline__ void init(int t, const c32* __restrict__ tw) {
#pragma unroll
            for (int r = 1; r < 8; ++r) a[r - 1] = tw[r * (t & 7) * (N / 64)];
            if (MID2) {
#pragma unroll
                for (int r = 1; r < 8; ++r) b[r - 1] = tw[r * (t & 63) * (N / 512)];
            }
#pragma unroll
            for (int u = 0; u < 8 / RF; ++u) {
                const int k = (t + u * TPF) & (NSL - 1);
#pragma unroll
                for (int r = 1; r < RF; ++r) f[u * (RF - 1) + r - 1] = tw[r * k * (N / (NSL * RF))];
            }
        }
    };

    // one Stockham pass: butterfly j = t + u TPF (u < 8 / R) takes in[j + r N/R], twiddles by e^{-2 pi i r k / (NS R)},
    // k = j mod NS, and its output r goes to (j - k) R + k + r NS.  With the padded index i + i/8 every address is a
    // per-thread base plus a compile-time constant (n/8 and NS >= 8 are multiples of 8): immediate offsets, no index math
    // per access.
    template <int R>
    static __device__ __forceinline__ void load(c32 (&v)[8], const c32* z, int t, const c32* w) {
        const c32* zt = z + zpad(t);
#pragma unroll
        for (int u = 0; u < 8 / R; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c32 a = zt[(u * TPF + r * (N / R)) / 8 * 9];
                if (r > 0) a = cmul(a, w[u * (R - 1) + r - 1]);
                v[u * R + r] = a;
            }
        }
    }
    template <int R>
    static __device__ __forceinline__ void bfly(c32 (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8 / R; ++u) dft<R>(&v[u * R]);
    }
    // radix-8 passes only (R = 8, one butterfly per thread: j = t)
    template <int NS>
    static __device__ __forceinline__ void store8(const c32 (&v)[8], c32* z, int t) {
        const int k = t & (NS - 1);
        // NS = 1: 9 t + r;  NS = 8: 9 (t - k) + k + 9 r;  NS = 64: 9 (t - k) + k + k / 8 + 72 r
        c32* zt = z + 9 * (t - k) + k + (k >> 3);
#pragma unroll
        for (int r = 0; r < 8; ++r) zt[NS == 1 ? r : r * NS / 8 * 9] = v[r];
    }
    template <int NS>
    static __device__ __forceinline__ void pass8(c32 (&v)[8], c32* z, int t, const c32* w) {
        load<8>(v, z, t, w);
        bfly<8>(v);
        sync();
        store8<NS>(v, z, t);
        sync();
    }
    // first pass with the 8 inputs v[r] = in[t + r TPF] already in registers; leaves everything but the final pass done
    // and returns with the final pass's butterflies in v: output r of butterfly u is sample (t + u TPF) + r NSF.
    // Every thread of the workgroup must call it (barriers); the caller guarantees nobody still reads z.
    static __device__ __forceinline__ void run(c32 (&v)[8], c32* z, int t, const Tw& tw) {
        bfly<8>(v);
        store8<1>(v, z, t);
        sync();
        if (P8 == 3 || LAST > 1) pass8<8>(v, z, t, tw.a);
        if (MID2) pass8<64>(v, z, t, tw.b);
        load<RF>(v, z, t, tw.f);
        bfly<RF>(v);
        sync();                                             // all reads of z done: the caller may overwrite it
    }
    static __device__ __forceinline__ int out_index(int t, int u, int r) { return t + u * TPF + r * NSF; }
    // natural-order spectrum into z (barrier at the end)
    static __device__ __forceinline__ void store_natural(const c32 (&v)[8], c32* z, int t) {
        c32* zt = z + zpad(t);
#pragma unroll
        for (int u = 0; u < 8 / RF; ++u)
#pragma unroll
            for (int r = 0; r < RF; ++r) zt[(u * TPF + r * NSF) / 8 * 9] = v[u * RF + r];
        sync();
    }
};

struct StftP {
    const float* x;
    const float* y;
    const float* win;
    const c32* tw;
    int rows, t_len, n_frames;
    float eps;
    // forward
    int fpw;                 // frames per workgroup (multiple of G)
    float* part;             // [workgroups][3]
    // backward
    int cb, n_blocks;        // hop blocks per workgroup (the last one takes the remainder), blocks of the padded row
    const float* sums;
    const float* gout;
    float inv_n;
    float* dx;
    float* dy;
    int accumulate;
    int equalise;            // 0: RH_STFT_EQUALISE=0 (diagnostics: the pair transform without the power-of-two equaliser)
};

__device__ __forceinline__ int reflect_at(int p, int t) { return p < 0 ? -p : (p >= t ? 2 * (t - 1) - p : p); }

// the 8 samples n = t + r n/8 of frame f of both signals (no window yet)
template <int N>
__device__ __forceinline__ void load_frames(c32 (&v)[8], int t_len, const float* __restrict__ xr, const float* __restrict__ yr,
                                            int f, int t, bool valid) {
    constexpr int TPF = N / 8, H = N / 4;
    const int p0 = f * H - N / 2;
    if (valid && p0 >= 0 && p0 + N <= t_len) {              // all but the two frames at either end of a row
        const float* xa = xr + p0 + t;
        const float* ya = yr + p0 + t;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = mk(xa[r * TPF], ya[r * TPF]);
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float a = 0.f, b = 0.f;
        if (valid) {
            const int q = reflect_at(p0 + t + r * TPF, t_len);
            a = xr[q];
            b = yr[q];
        }
        v[r] = mk(a, b);
    }
}

// Equaliser of a frame pair: s = 2^(e_x - e_y) from the largest |x| and |y| of the frame (all N samples: the TPF threads of
// the transform; across waves through `fmx`, one slot per wave -- the barriers inside the transform that follows separate this
// read from the next frame's write), 1 when either frame is all zero.  Every thread of the workgroup must call it.
template <int TPF>
__device__ __forceinline__ void frame_scale(const c32 (&vn)[8], c32* fmx, float& s, float& inv_s, int on) {
    float mx = 0.f, my = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        mx = fmaxf(mx, fabsf(vn[r].x));
        my = fmaxf(my, fabsf(vn[r].y));
    }
    constexpr int W = TPF < 64 ? TPF : 64;
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        my = fmaxf(my, __shfl_xor(my, o, 64));
    }
    if (TPF > 64) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) fmx[wave] = mk(mx, my);
        __syncthreads();
        const int w0 = (threadIdx.x / TPF) * (TPF / 64);
        mx = my = 0.f;
#pragma unroll
        for (int i = 0; i < TPF / 64; ++i) {
            const c32 q = fmx[w0 + i];
            mx = fmaxf(mx, q.x);
            my = fmaxf(my, q.y);
        }
    }
    int d = (mx > 0.f && my > 0.f) ? (int)(__float_as_uint(mx) >> 23) - (int)(__float_as_uint(my) >> 23) : 0;
    d = d < -100 ? -100 : (d > 100 ? 100 : d);
    if (!on) d = 0;
    if (TPF >= 64) d = __builtin_amdgcn_readfirstlane(d);      // a whole wave works on one transform: the scale lives in an SGPR
    s = __uint_as_float((unsigned)(127 + d) << 23);
    inv_s = __uint_as_float((unsigned)(127 - d) << 23);
}

__device__ __forceinline__ float wg_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <int N>
__global__ __launch_bounds__(256) void stft_loss_fwd_kernel(const StftP p) {
    typedef Fft<N> F;
    constexpr int TPF = F::TPF, G = F::G;
    __shared__ c32 zb[G * F::ZP];
    __shared__ float red[4];
    __shared__ c32 fmx[4];
    const int tid = threadIdx.x;
    const int gi = tid / TPF, t = tid - gi * TPF;
    const int row = blockIdx.y;
    const int f0 = blockIdx.x * p.fpw;
    const int f1 = min(f0 + p.fpw, p.n_frames);
    const float* __restrict__ xr = p.x + (long)row * p.t_len;
    const float* __restrict__ yr = p.y + (long)row * p.t_len;
    c32* z = zb + gi * F::ZP;
    typename F::Tw tw;
    tw.init(t, p.tw);
    float wn[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wn[r] = p.win[t + r * TPF];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const float eps = p.eps;
    float sc = 1.f, isc = 1.f;       // equaliser of the current frame pair and its inverse (frame_scale)
    auto bin = [&](c32 z1, c32 z2) {
        const float xr_ = 0.5f * (z1.x + z2.x), xi = 0.5f * (z1.y - z2.y);
        const float yr_ = (0.5f * isc) * (z1.y + z2.y), yi = (0.5f * isc) * (z2.x - z1.x);
        const float a2 = xr_ * xr_ + xi * xi, b2 = yr_ * yr_ + yi * yi;
        const float a = __builtin_amdgcn_sqrtf(a2), b = __builtin_amdgcn_sqrtf(b2);
        const float d = a - b;
        s0 += d * d;
        s1 += a2;
        // |log(a + eps) - log(b + eps)| as one logarithm of the ratio (hardware log2 / reciprocal, 1 ulp each)
        s2 += fabsf(__builtin_amdgcn_logf((a + eps) * __builtin_amdgcn_rcpf(b + eps)));
    };
    c32 vn[8];
    load_frames<N>(vn, p.t_len, xr, yr, f0 + gi, t, f0 + gi < f1);
    for (int fb = f0; fb < f1; fb += G) {
        const bool valid = fb + gi < f1;
        c32 v[8];
        frame_scale<TPF>(vn, fmx, sc, isc, p.equalise);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = mk(vn[r].x, vn[r].y * sc) * wn[r];
        load_frames<N>(vn, p.t_len, xr, yr, fb + G + gi, t, fb + G + gi < f1);   // the next frame's samples fly under this one's passes
        F::run(v, z, t, tw);
        F::store_natural(v, z, t);
        if (valid) {
            // bins k = t + m n/8 and their mirrors n - k = (n/8 - t) + (7 - m) n/8 (t > 0; bin 0 pairs with itself)
            const c32* zk = z + zpad(t);
            const c32* zm = z + (t ? zpad(TPF - t) : 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) bin(zk[m * TPF / 8 * 9], t ? zm[(7 - m) * TPF / 8 * 9] : (m ? z[zpad(N - m * TPF)] : z[0]));
            if (t == 0) bin(z[zpad(N / 2)], z[zpad(N / 2)]);
        }
        F::sync();
    }
    constexpr float kLn2 = 0.69314718055994530942f;
    const float t0 = wg_sum(s0, red), t1 = wg_sum(s1, red), t2 = wg_sum(s2, red) * kLn2;
    if (tid == 0) {
        float* o = p.part + 3l * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = t0; o[1] = t1; o[2] = t2;
    }
}

__global__ __launch_bounds__(256) void stft_loss_finalize_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
    __shared__ float red[4];
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i * 3 + k];
        const float tt = wg_sum(s, red);
        if (threadIdx.x == 0) sums[k] = tt;
        __syncthreads();
    }
}

// The ordered finalize of ALL scales of one distance + the sum over the scales (misc.hip: spectral_total_kernel's formula) in
// one launch: per distance 5 finalize launches + 1 total on the critical path between forward and backward become 1.  One
// workgroup walks the scales with the same reduction as the kernel above (same bits).
constexpr int kStftMaxScales = 8;
struct StftFinTable {
    const float* part[kStftMaxScales];
    int nblocks[kStftMaxScales];
    int count;
};
__global__ __launch_bounds__(256) void stft_loss_finalize_all_kernel(const StftFinTable tb, const float* __restrict__ inv_n,
                                                                     float* __restrict__ sums, float* __restrict__ out) {
    __shared__ float red[4];
    float d = 0.f;
    for (int sc = 0; sc < tb.count; ++sc) {
        const float* __restrict__ part = tb.part[sc];
        const int nblocks = tb.nblocks[sc];
        float t[3];
        for (int k = 0; k < 3; ++k) {
            float s = 0.f;
            for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i * 3 + k];
            t[k] = wg_sum(s, red);
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            sums[3 * sc] = t[0]; sums[3 * sc + 1] = t[1]; sums[3 * sc + 2] = t[2];
            d += t[0] / t[1] + t[2] * inv_n[sc];
        }
    }
    if (threadIdx.x == 0) out[0] = d;
}

constexpr int kFmxBytes = 32;      // head of the backward kernel's dynamic LDS: 4 (max |x|, max |y|) slots

// Workgroups per CU the backward kernel is compiled for: 3 (<= 168 VGPRs), or -- RH_STFT_BWD_OCC2 builds, an experiment -- 2 for
// the transforms that span several waves, whose equaliser (cross-wave maximum) pushed them over 168 registers (5-13 spilled)
#ifdef RH_STFT_BWD_OCC2
constexpr int stft_bwd_occ(int n) { return n >= 1024 ? 2 : 3; }
#else
constexpr int stft_bwd_occ(int) { return 3; }
#endif

template <int N>
__global__ __launch_bounds__(256, stft_bwd_occ(N)) void stft_loss_bwd_kernel(const StftP p) {
    typedef Fft<N> F;
    constexpr int TPF = F::TPF, G = F::G, H = N / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* const fmx = reinterpret_cast<c32*>(smem);           // kFmxBytes: one slot per wave (frame_scale)
    c32* const zb = fmx + kFmxBytes / sizeof(c32);
    float* const accx = reinterpret_cast<float*>(zb + G * F::ZP);
    const int tid = threadIdx.x;
    const int gi = tid / TPF, t = tid - gi * TPF;
    const int row = blockIdx.y;
    const int b0 = blockIdx.x * p.cb;
    const int b1 = blockIdx.x == gridDim.x - 1 ? p.n_blocks : b0 + p.cb;
    const int span = (b1 - b0) * H, q0 = b0 * H;
    float* const accy = accx + span;
    const float* __restrict__ xr = p.x + (long)row * p.t_len;
    const float* __restrict__ yr = p.y + (long)row * p.t_len;
    c32* z = zb + gi * F::ZP;
    for (int i = tid; i < 2 * span; i += 256) accx[i] = 0.f;
    const int flo = max(0, b0 - 3), fhi = min(p.n_frames - 1, b1 - 1);
    typename F::Tw tw;
    tw.init(t, p.tw);
    float wn[8], wo[8];                                      // window at the samples this thread loads / at those it ends up with
#pragma unroll
    for (int r = 0; r < 8; ++r) wn[r] = p.win[t + r * TPF];
#pragma unroll
    for (int u = 0; u < 8 / F::RF; ++u)
#pragma unroll
        for (int r = 0; r < F::RF; ++r) wo[u * F::RF + r] = p.win[F::out_index(t, u, r)];
    const float g = p.gout[0], A = p.sums[0], B = p.sums[1];
    const float invB = 1.f / B, eps = p.eps;
    const float c1 = 2.f * invB, c2 = 2.f * A * invB * invB, invN = p.inv_n;
    // gradient w.r.t. the complex bins of X and Y for the pair (Z_k, Z_{n-k}); `gh` = grad_out x 0.5 for interior bins
    // (their Hermitian mirror carries the other half), x 1 for k = 0 and n/2.  sign(log(a+eps) - log(b+eps)) = sign(a - b).
    // The frame pair is equalised (frame_scale): the transform carried s y, so Y = Y' / s; the gradient operand takes Wy / s
    // (|Wy| ~ s |Wx| otherwise) and the imaginary part of the inverse transform is multiplied by s where it is added up.
    float sc = 1.f, isc = 1.f;
    auto grads = [&](c32 z1, c32 z2, float gh, c32& gx, c32& gy) {
        const c32 X = mk(0.5f * (z1.x + z2.x), 0.5f * (z1.y - z2.y));
        const c32 Y = mk((0.5f * isc) * (z1.y + z2.y), (0.5f * isc) * (z2.x - z1.x));
        const float a = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y), b = __builtin_amdgcn_sqrtf(Y.x * Y.x + Y.y * Y.y);
        const float d = a - b;
        const float sg = d > 0.f ? invN : (d < 0.f ? -invN : 0.f);
        const float da = gh * (c1 * d - c2 * a + sg * __builtin_amdgcn_rcpf(a + eps));
        const float db = gh * (-c1 * d - sg * __builtin_amdgcn_rcpf(b + eps));
        const float ra = a > 0.f ? da * __builtin_amdgcn_rcpf(a) : 0.f, rb = b > 0.f ? db * __builtin_amdgcn_rcpf(b) : 0.f;
        gx = X * ra;
        gy = Y * (rb * isc);
    };
    __syncthreads();
    for (int ph = 0; ph < 4; ++ph) {
        const int fs = flo + ((ph - flo) & 3);
        c32 vn[8];
        load_frames<N>(vn, p.t_len, xr, yr, fs + 4 * gi, t, fs + 4 * gi <= fhi);
        for (int fb = fs; fb <= fhi; fb += 4 * G) {
            const int f = fb + 4 * gi;
            const bool valid = f <= fhi;
            c32 v[8];
            frame_scale<TPF>(vn, fmx, sc, isc, p.equalise);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = mk(vn[r].x, vn[r].y * sc) * wn[r];
            load_frames<N>(vn, p.t_len, xr, yr, f + 4 * G, t, f + 4 * G <= fhi);
            F::run(v, z, t, tw);
            F::store_natural(v, z, t);
            // W = Wx + i Wy (Hermitian extensions of the half-weighted gradients), stored SWAPPED (im, re): the inverse
            // transform is swap(FFT(swap(W)))
            {
                c32* zk = z + zpad(t);
                c32* zm = z + (t ? zpad(TPF - t) : 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    c32* pk = zk + m * TPF / 8 * 9;                                        // bin k = t + m n/8
                    c32* pm = t ? zm + (7 - m) * TPF / 8 * 9 : z + zpad(N - m * TPF);      // bin n - k (unused for k = 0)
                    c32 gx, gy;
                    if (m == 0 && t == 0) {
                        grads(z[0], z[0], g, gx, gy);
                        z[0] = mk(gy.x, gx.x);
                    } else {
                        grads(*pk, *pm, 0.5f * g, gx, gy);
                        *pk = mk(gx.y + gy.x, gx.x - gy.y);          // swap of (gx + i gy)
                        *pm = mk(gy.x - gx.y, gx.x + gy.y);          // swap of (conj gx + i conj gy)
                    }
                }
            }
            if (t == 0) {
                c32 gx, gy;
                grads(z[zpad(N / 2)], z[zpad(N / 2)], g, gx, gy);
                z[zpad(N / 2)] = mk(gy.x, gx.x);
            }
            F::sync();
            {
                const c32* zt = z + zpad(t);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = zt[r * TPF / 8 * 9];
            }
            F::sync();
            F::run(v, z, t, tw);
            if (valid) {
#pragma unroll
                for (int u = 0; u < 8 / F::RF; ++u)
#pragma unroll
                    for (int r = 0; r < F::RF; ++r) {
                        const int loc = f * H + F::out_index(t, u, r) - q0;
                        if (loc >= 0 && loc < span) {
                            const c32 o = v[u * F::RF + r] * wo[u * F::RF + r];   // swapped: (d frame_y, d frame_x)
                            accx[loc] += o.y;
                            accy[loc] += o.x * sc;
                        }
                    }
            }
        }
        __syncthreads();
    }
    const int T = p.t_len;
    if (b0 == 0) {                                                                // left margin: padded q < n/2 folds onto n/2 + (n/2 - q)
        for (int i = tid + 1; i <= N / 2; i += 256) {
            accx[N / 2 + i] += accx[N / 2 - i];
            accy[N / 2 + i] += accy[N / 2 - i];
        }
        __syncthreads();
    }
    if (b1 == p.n_blocks) {                                                       // right margin: q = T + n/2 + i folds onto T + n/2 - 2 - i
        for (int i = tid; i < N / 2; i += 256) {
            const int qs = T + N / 2 + i - q0, qt = T + N / 2 - 2 - i - q0;
            if (qs < span && qt >= 0) {
                accx[qt] += accx[qs];
                accy[qt] += accy[qs];
            }
        }
        __syncthreads();
    }
    float* __restrict__ dxr = p.dx ? p.dx + (long)row * T : nullptr;
    float* __restrict__ dyr = p.dy ? p.dy + (long)row * T : nullptr;
    for (int i = tid; i < span; i += 256) {
        const int pq = q0 + i - N / 2;
        if (pq < 0 || pq >= T) continue;
        if (dxr) dxr[pq] = p.accumulate ? dxr[pq] + accx[i] : accx[i];
        if (dyr) dyr[pq] = p.accumulate ? dyr[pq] + accy[i] : accy[i];
    }
}

constexpr int kTailMin = 6;       // a remainder shorter than this joins the previous workgroup (the last one must own the
                                  // whole right margin and the samples it folds onto: n + 1 samples = 4 blocks + 1)

int equalise_on() {       // (read per call: the tests compare both settings in one process)
    const char* e = getenv("RH_STFT_EQUALISE");
    return (e && e[0] == '0') ? 0 : 1;
}

bool shape_ok(int n_fft, int hop, int t_len, long rows) {
    return (n_fft == 128 || n_fft == 256 || n_fft == 512 || n_fft == 1024 || n_fft == 2048) && hop * 4 == n_fft &&
           t_len > n_fft / 2 && (long)t_len + 2l * n_fft < 0x7fffffffl && rows > 0 && rows < 65536;
}

int chunks_of(int n_blocks, int cb) {
    int c = n_blocks / cb;
    const int rem = n_blocks - c * cb;
    if (c == 0 || rem >= kTailMin) ++c;
    return c;
}

// Hop blocks per workgroup of the backward kernel.  A workgroup of nb blocks transforms nb + 3 frames (3 reach in from the
// left), in four phases of ceil(frames of the phase / G) rounds of G transforms side by side; its LDS image grows with nb
// and bounds the workgroups per CU.  Candidates are the sizes whose phases are whole rounds (nb + 3 = 4 G m) plus "the
// whole row"; the estimate is (residency waves of the grid) x (rounds of the longest workgroup).
int rounds_of(int blocks, int first_frame_cut, int G) {
    int r = 0;
    const int frames = blocks + 3 - first_frame_cut;
    for (int ph = 0; ph < 4; ++ph) r += ((frames - ph + 3) / 4 + G - 1) / G;
    return r;
}
int choose_cb(int n_fft, int n_blocks, long rows) {
    const int G = 256 / (n_fft / 8), H = n_fft / 4;
    const size_t zb = (size_t)G * (n_fft + n_fft / 8) * 8 + kFmxBytes;
    int best = 0;
    double best_cost = 1e30;
    for (int m = 1; m <= 64; ++m) {
        int cb = 4 * G * m - 3;
        if (cb < kTailMin) continue;
        if (cb > n_blocks) cb = n_blocks;
        const int nch = chunks_of(n_blocks, cb);
        const int last = n_blocks - (nch - 1) * cb;
        const int big = last > cb ? last : cb;
        const size_t lds = zb + 2ul * big * H * 4;
        if (lds > 150 * 1024) break;
        int per_cu = (int)(160 * 1024 / lds);
        const int occ = stft_bwd_occ(n_fft);
        per_cu = per_cu > occ ? occ : per_cu;                           // __launch_bounds__(256, occ)
        const double wgs = (double)rows * nch, slots = 256.0 * per_cu;
        const int r_first = rounds_of(nch == 1 ? n_blocks : cb, 3, G);   // the first workgroup has no frames to its left
        const int r_mid = nch > 2 ? rounds_of(cb, 0, G) : 0;
        const int r_last = nch > 1 ? rounds_of(last, 0, G) : 0;
        const int r_max = r_first > r_mid ? (r_first > r_last ? r_first : r_last) : (r_mid > r_last ? r_mid : r_last);
        // whole residency waves at the longest workgroup's length, + a per-round cost that falls with the occupancy
        const double waves = wgs <= slots ? 1.0 : wgs / slots + 0.5;
        const double cost = waves * r_max * (1.0 + 0.5 / per_cu) + 0.15 * nch;      // + set-up / write-out per workgroup
        if (cost < best_cost) { best_cost = cost; best = cb; }
        if (cb == n_blocks) break;
    }
    return best ? best : n_blocks;
}

template <int N>
int launch_fwd(StftP p, int wgs_x, hipStream_t stream) {
    hipLaunchKernelGGL(stft_loss_fwd_kernel<N>, dim3(wgs_x, p.rows), dim3(256), 0, stream, p);
    return rh_check_launch("stft_loss_fwd");
}

template <int N>
int launch_bwd(StftP p, hipStream_t stream) {
    typedef Fft<N> F;
    const int nch = chunks_of(p.n_blocks, p.cb);
    const int last = p.n_blocks - (nch - 1) * p.cb;                    // blocks of the last workgroup (>= all others)
    const size_t lds = kFmxBytes + (size_t)F::G * F::ZP * sizeof(c32) + 2ul * (size_t)(last > p.cb ? last : p.cb) * (N / 4) * sizeof(float);
    auto kern = stft_loss_bwd_kernel<N>;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (lds > 160 * 1024) {
        rh_set_error("stft_loss_bwd: span does not fit in LDS");
        return RH_ERR_INVALID;
    }
    hipLaunchKernelGGL(kern, dim3(nch, p.rows), dim3(256), lds, stream, p);
    return rh_check_launch("stft_loss_bwd");
}

int fpw_of(int n_fft, int n_frames, long rows) {
    const int G = 256 / (n_fft / 8);
    int iters = (n_frames + G - 1) / G;
    // enough workgroups to fill the chip twice over, as few as that allows (per-workgroup set-up + partial sums)
    while (iters > 1 && rows * ((n_frames + G * iters - 1) / (G * iters)) < 1024) iters = (iters + 1) / 2;
    return G * iters;
}

}  // namespace

extern "C" int rh_stft_loss_supported(int32_t n_fft, int32_t hop, int32_t t_len, int64_t rows) {
    return shape_ok(n_fft, hop, t_len, rows) ? 1 : 0;
}

// Diagnostics / tests (no GPU needed): out = {frames per forward workgroup, forward workgroups per row, hop blocks per backward
// workgroup, backward workgroups per row, hop blocks of the LAST backward workgroup, dynamic LDS bytes of the backward launch}
extern "C" int rh_stft_loss_plan_info(int32_t n_fft, int32_t t_len, int64_t rows, int64_t* out6) {
    RH_REQUIRE(out6, RH_ERR_INVALID, "stft_loss_plan_info: null output");
    RH_REQUIRE(shape_ok(n_fft, n_fft / 4, t_len, rows), RH_ERR_UNSUPPORTED, "stft_loss_plan_info: unsupported geometry");
    const int H = n_fft / 4, nf = t_len / H + 1;
    const int fpw = fpw_of(n_fft, nf, rows);
    const int n_blocks = (t_len + n_fft + H - 1) / H;
    const int cb = choose_cb(n_fft, n_blocks, rows);
    const int nch = chunks_of(n_blocks, cb);
    const int last = n_blocks - (nch - 1) * cb;
    const int G = 256 / (n_fft / 8);
    out6[0] = fpw; out6[1] = (nf + fpw - 1) / fpw; out6[2] = cb; out6[3] = nch; out6[4] = last;
    out6[5] = kFmxBytes + (int64_t)G * (n_fft + n_fft / 8) * 8 + 2l * (last > cb ? last : cb) * H * 4;
    return RH_OK;
}

extern "C" int64_t rh_stft_loss_workspace_bytes(int32_t n_fft, int32_t t_len, int64_t rows) {
    if (!shape_ok(n_fft, n_fft / 4, t_len, rows)) return 0;
    const int nf = t_len / (n_fft / 4) + 1;
    const int fpw = fpw_of(n_fft, nf, rows);
    return (int64_t)rows * ((nf + fpw - 1) / fpw) * 3 * (int64_t)sizeof(float);
}

extern "C" int rh_stft_loss_fwd_f32(const float* x, const float* y, const float* window, const float* twiddle, int64_t rows,
                                    int32_t t_len, int32_t n_fft, float eps, float* sums, void* workspace,
                                    int64_t workspace_bytes, rh_stream_t stream) {
    RH_REQUIRE(x && y && window && twiddle && workspace, RH_ERR_INVALID, "stft_loss_fwd: null pointer");
    RH_REQUIRE(shape_ok(n_fft, n_fft / 4, t_len, rows), RH_ERR_UNSUPPORTED, "stft_loss_fwd: unsupported geometry (n_fft %d, t %d)", n_fft, t_len);
    RH_REQUIRE(workspace_bytes >= rh_stft_loss_workspace_bytes(n_fft, t_len, rows), RH_ERR_WORKSPACE, "stft_loss_fwd: workspace too small");
    StftP p = {};
    p.x = x; p.y = y; p.win = window; p.tw = reinterpret_cast<const c32*>(twiddle);
    p.rows = (int)rows; p.t_len = t_len; p.n_frames = t_len / (n_fft / 4) + 1; p.eps = eps;
    p.fpw = fpw_of(n_fft, p.n_frames, rows);
    p.equalise = equalise_on();
    p.part = static_cast<float*>(workspace);
    const int wx = (p.n_frames + p.fpw - 1) / p.fpw;
    int rc;
    switch (n_fft) {
        case 128: rc = launch_fwd<128>(p, wx, (hipStream_t)stream); break;
        case 256: rc = launch_fwd<256>(p, wx, (hipStream_t)stream); break;
        case 512: rc = launch_fwd<512>(p, wx, (hipStream_t)stream); break;
        case 1024: rc = launch_fwd<1024>(p, wx, (hipStream_t)stream); break;
        default: rc = launch_fwd<2048>(p, wx, (hipStream_t)stream); break;
    }
    if (rc || !sums) return rc;         // sums == NULL: the partials stay in `workspace` for rh_stft_loss_finalize_all_f32
    hipLaunchKernelGGL(stft_loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                       (int)(rows * wx), sums);
    return rh_check_launch("stft_loss_finalize");
}

extern "C" int rh_stft_loss_finalize_all_f32(const float* const* partials, const int64_t* partial_bytes, int32_t n_scales,
                                             const float* inv_n, float* sums, float* total, rh_stream_t stream) {
    RH_REQUIRE(partials && partial_bytes && inv_n && sums && total && n_scales > 0 && n_scales <= kStftMaxScales, RH_ERR_INVALID,
               "stft_loss_finalize_all: bad arguments (1 .. %d scales)", kStftMaxScales);
    StftFinTable tb;
    for (int i = 0; i < n_scales; ++i) {
        RH_REQUIRE(partials[i] && partial_bytes[i] > 0 && partial_bytes[i] % 12 == 0 && partial_bytes[i] / 12 < 0x7fffffffl, RH_ERR_INVALID,
                   "stft_loss_finalize_all: bad partials of scale %d", i);
        tb.part[i] = partials[i];
        tb.nblocks[i] = (int)(partial_bytes[i] / 12);
    }
    tb.count = n_scales;
    hipLaunchKernelGGL(stft_loss_finalize_all_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tb, inv_n, sums, total);
    return rh_check_launch("stft_loss_finalize_all");
}

extern "C" int rh_stft_loss_bwd_f32(const float* x, const float* y, const float* window, const float* twiddle, int64_t rows,
                                    int32_t t_len, int32_t n_fft, float eps, const float* sums, const float* grad_out,
                                    float* dx, float* dy, int32_t accumulate, rh_stream_t stream) {
    RH_REQUIRE(x && y && window && twiddle && sums && grad_out, RH_ERR_INVALID, "stft_loss_bwd: null pointer");
    RH_REQUIRE(shape_ok(n_fft, n_fft / 4, t_len, rows), RH_ERR_UNSUPPORTED, "stft_loss_bwd: unsupported geometry (n_fft %d, t %d)", n_fft, t_len);
    if (!dx && !dy) return RH_OK;
    StftP p = {};
    p.x = x; p.y = y; p.win = window; p.tw = reinterpret_cast<const c32*>(twiddle);
    p.rows = (int)rows; p.t_len = t_len; p.n_frames = t_len / (n_fft / 4) + 1; p.eps = eps;
    const int H = n_fft / 4;
    p.n_blocks = (t_len + n_fft + H - 1) / H;
    p.cb = choose_cb(n_fft, p.n_blocks, rows);
    p.sums = sums; p.gout = grad_out;
    p.inv_n = (float)(1.0 / ((double)rows * p.n_frames * (n_fft / 2 + 1)));
    p.dx = dx; p.dy = dy; p.accumulate = accumulate;
    p.equalise = equalise_on();
    switch (n_fft) {
        case 128: return launch_bwd<128>(p, (hipStream_t)stream);
        case 256: return launch_bwd<256>(p, (hipStream_t)stream);
        case 512: return launch_bwd<512>(p, (hipStream_t)stream);
        case 1024: return launch_bwd<1024>(p, (hipStream_t)stream);
        default: return launch_bwd<2048>(p, (hipStream_t)stream);
    }
}
