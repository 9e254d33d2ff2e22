// Implicit-GEMM convolution, software-pipelined with asynchronous global->LDS DMA
// (buffer_load ... lds, gfx950): while the matrix cores work on K-chunk i out of LDS stage i&1,
// the DMA engine fills stage (i+1)&1 with the next weight and input tiles -- no staging registers,
// one workgroup barrier per chunk.  Zero padding (left/right/causal pads, MPD fold pad, ragged
// channel/batch tails) comes from the buffer descriptor's bounds check: invalid lanes are given
// an out-of-range offset and the hardware writes 0.0 into LDS.
//
// Because DMA cannot transform data in flight, the fused pre-activation (LeakyReLU) is applied
// when the B operand is read from LDS (2 VALU per 64-cycle MFMA group -- free); Snake keeps the
// register-staged kernel of conv_igemm.hip.  Same math, same accumulation order per output as the
// synchronous kernel: exact f32, k-ordered fmaf chains.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "conv_params.hpp"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

#ifndef RH_READS_FIRST
#define RH_READS_FIRST 1
#endif
constexpr bool READS_FIRST = RH_READS_FIRST != 0;

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
__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned magic) { return (n * magic) >> 20; }

constexpr unsigned kOOB = 0x80000000u;  // >= any descriptor size we accept -> DMA writes zeros

template <int TM, int TN, int WM, int WN, bool LEAKY, bool VEC, bool CTAIL>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_dma_kernel(const ConvP p) {
    constexpr int BM = TM * WM * 32;
    constexpr int NW = WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    const int zsl = blockIdx.z / p.nphase;          // K slice
    const int phase = blockIdx.z - zsl * p.nphase;
    const int ntaps = p.ph_ntaps[phase];
    const int tap0 = p.ph_tap0[phase];
    const int minoff = p.ph_minoff[phase];
    const int oph = p.ph_oph[phase];
    const unsigned wofs = (unsigned)p.ph_wofs[phase];

    const int bt = blockIdx.x / p.tiles_per_b;
    const int nt = blockIdx.x - bt * p.tiles_per_b;
    const int b0 = bt * p.nb;
    const int n0 = nt * p.bnl;
    const int m0 = blockIdx.y * BM;
    const int inner = p.inner, is = p.is;
    const int pitch = VEC ? p.pitch : p.segs * 64;
    const int ib0 = ibase_of(n0, inner, is);
    const int lo_raw = ib0 + minoff * inner;
    // VEC: the tile origin is moved down to a multiple of 4 elements so that (rows being multiples of
    // 4 elements too) no 16-byte load straddles the start / end of a sequence row or of the buffer
    const int lo = VEC ? lo_raw - ((lo_raw % 4 + 4) % 4) : lo_raw;

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.w_bytes, 0x00020000);

    int xb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = min(n0 + nl, p.ncols - 1);
        xb[tn] = (bl * p.ck + kh) * pitch + ibase_of(n, inner, is) - ib0 + (lo_raw - lo);
    }
    const int arow = wm * TM * 32 + j + kh * BM;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int wrows = ntaps * p.ck;
    const int w_instrs = (wrows * BM + 255) >> 8;   // 256 floats (64 lanes x 16 B) per DMA instruction
    const int xrows = p.nb * p.ck;

    // ---- VEC: 16-byte DMA, per-lane source offsets precomputed once per workgroup ----------------
    // Each lane owns up to kNX float4 slots of the input tile and kNWS of the weight tile; per K chunk
    // only a scalar base is added (the tile's positions never change, only the channel chunk).
    // Zero padding cannot be produced at float4 granularity, so boundary tiles load whatever lies
    // next to the sequence (or 0 outside the tensor) and the B operand is masked at read time with
    // a per-lane bitmask over the taps -- padded positions never reach the matrix cores.
    constexpr int kNX = 8, kNWS = 10;
    // CTAIL: the last K chunk is partial (C % ck != 0) -> every slot also remembers its channel so that
    // it can be switched off; otherwise those registers are not even allocated
    unsigned xo[VEC ? kNX : 1], xc[VEC && CTAIL ? kNX : 1], wo[VEC ? kNWS : 1], wc[VEC && CTAIL ? kNWS : 1];
    unsigned long long vm[TN];
    bool bnd = false;
    int nx = 0, nws = 0, x_floats = 0;
    if constexpr (VEC) {
        const int lpr = pitch >> 2;
        x_floats = xrows * pitch;
        nx = (xrows * lpr + 64 * NW - 1) / (64 * NW);
        nws = (w_instrs + NW - 1) / NW;
        bnd = lo < 0 || lo + 4 * lpr > p.in_valid;
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
            const unsigned g = (unsigned)(wave + NW * i) * 64u + lane;
            const unsigned row = mdiv(g, p.magic_lpr);
            const unsigned v = g - row * lpr;
            const unsigned bl = mdiv(row, p.magic_ck);
            const unsigned c = row - bl * p.ck;
            const bool ok = row < (unsigned)xrows && b0 + bl < (unsigned)p.B;
            xo[i] = ok ? (bl * p.C + c) * (unsigned)p.in_row + 4u * v : kOOB;
            if (CTAIL) xc[i] = c;
        }
#pragma unroll
        for (int i = 0; i < kNWS; ++i) {
            const unsigned f = ((unsigned)(wave + NW * i) * 64u + lane) * 4u;
            const unsigned kr = f / (unsigned)BM;
            const unsigned col = f - kr * BM;
            const unsigned t = mdiv(kr, p.magic_ck);
            const unsigned c = kr - t * p.ck;
            const bool ok = kr < (unsigned)wrows && m0 + col < (unsigned)p.Mp;
            wo[i] = ok ? (t * p.C + c) * p.Mp + m0 + col : kOOB;
            if (CTAIL) wc[i] = c;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (wn * TN + tn) * 32 + j;
            const int nl = col & (p.bnl - 1);
            const int n = min(n0 + nl, p.ncols - 1);
            unsigned long long m = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int f = n * is + p.off[tap0 + t];
                if (f >= 0 && f < p.in_valid) m |= 1ull << t;
            }
            vm[tn] = m;
        }
    }

    auto issue = [&](int c0, float* stage) {
        if constexpr (VEC) {
            const unsigned wbase = wofs + (unsigned)c0 * p.Mp;
#pragma unroll
            for (int i = 0; i < kNWS; ++i) {
                if (i < nws && (wave + NW * i) < w_instrs) {
                    bool dead = wo[i] == kOOB;
                    if (CTAIL) dead = dead || c0 + wc[i] >= (unsigned)p.C;
                    const unsigned off = dead ? kOOB : (wbase + wo[i]) * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(stage + (wave + NW * i) * 256), 16, off, 0, 0, 0);
                }
            }
            float* xs = stage + p.wlds_floats;
            const unsigned xbase = ((unsigned)b0 * p.C + (unsigned)c0) * (unsigned)p.in_row + (unsigned)lo;
#pragma unroll
            for (int i = 0; i < kNX; ++i) {
                if (i < nx && (wave + NW * i) * 256 < x_floats) {
                    bool dead = xo[i] == kOOB;
                    if (CTAIL) dead = dead || c0 + xc[i] >= (unsigned)p.C;
                    const unsigned off = dead ? kOOB : (xbase + xo[i]) * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)(xs + (wave + NW * i) * 256), 16, off, 0, 0, 0);
                }
            }
            return;
        }
        // weights: LDS image [tap][c][BM], flat
        for (int q = wave; q < w_instrs; q += NW) {
            const unsigned f = (unsigned)q * 256u + (unsigned)lane * 4u;
            const unsigned kr = f / (unsigned)BM;
            const unsigned col = f - kr * BM;
            const unsigned t = mdiv(kr, p.magic_ck);
            const unsigned c = kr - t * p.ck;
            const unsigned ch = c0 + c, m = m0 + col;
            unsigned off = kOOB;
            if (kr < (unsigned)wrows && ch < (unsigned)p.C && m < (unsigned)p.Mp)
                off = (wofs + (t * p.C + ch) * p.Mp + m) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(stage + q * 256), 16, off, 0, 0, 0);
        }
        // input tile: rows (batch item, channel), 64-float segments
        float* xs = stage + p.wlds_floats;
        for (int row = wave; row < xrows; row += NW) {
            const unsigned bl = mdiv((unsigned)row, p.magic_ck);
            const unsigned c = row - bl * p.ck;
            const unsigned b = b0 + bl, ch = c0 + c;
            const bool rv = b < (unsigned)p.B && ch < (unsigned)p.C;
            const unsigned rbase = (b * p.C + ch) * (unsigned)p.in_row;
            float* dst = xs + row * pitch;
            for (int seg = 0; seg < p.segs; ++seg) {
                const int fpos = lo + seg * 64 + lane;
                unsigned off = kOOB;
                if (rv && fpos >= 0 && fpos < p.in_valid) off = (rbase + (unsigned)fpos) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)(dst + seg * 64), 4, off, 0, 0, 0);
            }
        }
    };

    const int total_chunks = (p.C + p.ck - 1) / p.ck;
    const int chunk0 = zsl * p.chunks_per_split;
    const int nchunks = min(p.chunks_per_split, total_chunks - chunk0);
    issue(chunk0 * p.ck, smem);
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // chunk i landed for every wave; everyone is done with the other stage
        if (i + 1 < nchunks) issue((chunk0 + i + 1) * p.ck, smem + ((i + 1) & 1) * p.stage_floats);
        const float* w_lds = smem + (i & 1) * p.stage_floats;
        const float* x_lds = w_lds + p.wlds_floats;
        // NOTE (measured, round 1): an explicit register ping-pong of the LDS reads (two operand sets,
        // sched_barrier) cost two waves/SIMD of occupancy and ran 24 % slower (71 -> 54 TFLOP/s over
        // the v2 layers); four resident waves per SIMD hide the LDS latency better than one wave's ILP.
        for (int t = 0; t < ntaps; ++t) {
            const int toff = (p.off[tap0 + t] - minoff) * inner;
            const float* wl = w_lds + t * p.ck * BM + arow;
            const float* xl = x_lds + toff;
            bool keep[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) keep[tn] = !(VEC && bnd) || ((vm[tn] >> t) & 1ull);
            // 4 k-steps (8 channels) per trip
            int c = 0;
            for (; c + 8 <= p.ck; c += 8) {
                float a[4][TM], b[4][TN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) a[u][tm] = wl[(c + 2 * u) * BM + tm * 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) b[u][tn] = xl[xb[tn] + (c + 2 * u) * pitch];
                }
                if (READS_FIRST) __builtin_amdgcn_sched_barrier(0);   // all 16 LDS reads in flight before the MFMAs
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        float v = b[u][tn];
                        if (VEC) v = keep[tn] ? v : 0.f;
                        if (LEAKY) v = v > 0.f ? v : v * p.in_slope;
                        b[u][tn] = v;
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][tm], b[u][tn], acc[tm][tn], 0, 0, 0);
                }
                if (READS_FIRST) __builtin_amdgcn_sched_barrier(0);
            }
            for (; c < p.ck; c += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = wl[c * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float v = xl[xb[tn] + c * pitch];
                    if (VEC) v = keep[tn] ? v : 0.f;
                    if (LEAKY) v = v > 0.f ? v : v * p.in_slope;
                    b[tn] = v;
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, activation derivative, residual / gradient add ----
    // Row offsets inside one batch item fit 32 bits (M * out_row < 2^31 is checked on the host);
    // the per-column base is the only 64-bit quantity.  `full` tiles skip the per-row bound check.
    const bool full = m0 + BM <= p.M;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = n0 + nl, b = b0 + bl;
        if (n >= p.ncols || b >= p.B) continue;
        const int oi = oidx_of(n, inner, p.os, oph);
        if (oi >= p.out_valid) continue;
        const long cbase = (long)b * p.M * p.out_row + oi;
        float* __restrict__ outp = (p.ksplit > 1 ? p.part + (long)zsl * p.part_stride : p.out) + cbase;
        const float* __restrict__ mulp = p.mul_src ? p.mul_src + cbase : nullptr;
        const float* __restrict__ addp = p.add ? p.add + cbase : nullptr;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mb = m0 + (wm * TM + tm) * 32 + 4 * kh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (full || m < p.M) {
                    const int ro = m * p.out_row;
                    float v = acc[tm][tn][r];
                    if (p.ksplit > 1) {
                        outp[ro] = v;
                        continue;
                    }
                    if (p.bias) v += p.bias[m];
                    if (mulp) {
                        const float al = (p.epi_act == RH_ACT_SNAKE) ? p.mul_alpha[m] : 0.f;
                        v *= rh_act_grad(mulp[ro], p.epi_act, p.epi_slope, al);
                    }
                    if (addp) v += addp[ro];
                    if (p.out_act == RH_ACT_LEAKY) v = v > 0.f ? v : v * p.out_slope;
                    outp[ro] = v;
                }
            }
        }
    }
}

unsigned magic_of(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }

// out = epi( sum_z part[z] ) for split-K launches (ordered sum -> deterministic).  With p.out_range set (the range slot of the
// f16 kernels' consumers, common.hpp) the launch also leaves max |out| there: ONE atomic per workgroup, so the grid is capped
// and strided (every element is still summed by one thread in slice order).
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const ConvP p, long total) {
    float mx = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / p.out_valid;
        const int oi = (int)(e - row * p.out_valid);
        const int m = (int)(row % p.M);
        const long idx = row * p.out_row + oi;
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.part[(long)z * p.part_stride + idx];
        if (p.bias) v += p.bias[m];
        if (p.mul_src) {
            const float al = (p.epi_act == RH_ACT_SNAKE) ? p.mul_alpha[m] : 0.f;
            v *= rh_act_grad(p.mul_src[idx], p.epi_act, p.epi_slope, al);
        }
        if (p.add) v += p.add[idx];
        if (p.out_act == RH_ACT_LEAKY) v = v > 0.f ? v : v * p.out_slope;
        p.out[idx] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (p.out_range) {
        __shared__ float red[4];
        rh_range_publish(p.out_range, mx, blockIdx.x, red);
    }
}

// the same, four consecutive outputs of one row per thread (16-byte loads / stores): the launch is a pure stream over
// ksplit + 1 ... ksplit + 3 tensors, and one element per thread kept too few bytes in flight per lane (2-3.5 TB/s)
__global__ __launch_bounds__(256) void splitk_finalize4_kernel(const ConvP p, long total4) {
    float mx = 0.f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const long e = 4 * t;
        const long row = e / p.out_valid;
        const int oi = (int)(e - row * p.out_valid);
        const int m = (int)(row % p.M);
        const long idx = row * p.out_row + oi;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.ksplit; ++z) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p.part + (long)z * p.part_stride + idx);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += q[i];
        }
        if (p.bias) {
            const float b = p.bias[m];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += b;
        }
        if (p.mul_src) {
            const float al = (p.epi_act == RH_ACT_SNAKE) ? p.mul_alpha[m] : 0.f;
            const f32x4 q = *reinterpret_cast<const f32x4*>(p.mul_src + idx);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= rh_act_grad(q[i], p.epi_act, p.epi_slope, al);
        }
        if (p.add) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p.add + idx);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += q[i];
        }
        if (p.out_act == RH_ACT_LEAKY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.out_slope;
        }
        *reinterpret_cast<f32x4*>(p.out + idx) = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (p.out_range) {
        __shared__ float red[4];
        rh_range_publish(p.out_range, mx, blockIdx.x, red);
    }
}

int finalize_launch(const ConvP& p, hipStream_t stream) {
    const long total = (long)p.B * p.M * p.out_valid;
    const bool vec = (p.out_valid & 3) == 0 && (p.out_row & 3) == 0 && (p.part_stride & 3) == 0 &&
                     (((uintptr_t)p.part | (uintptr_t)p.out | (uintptr_t)p.mul_src | (uintptr_t)p.add) & 15) == 0;
    const long items = vec ? total / 4 : total;
    long blocks = rh_cdiv64(items, 256);
    if (p.out_range && blocks > 4096) blocks = 4096;       // one range atomic per workgroup (see the kernels)
    if (vec) hipLaunchKernelGGL(splitk_finalize4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, items);
    else hipLaunchKernelGGL(splitk_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, items);
    return rh_check_launch("conv_splitk_finalize");
}

struct Plan { int blocks, total_chunks, ksplit, chunks_per_split; };

Plan plan_split(const ConvP& p, int BM, int col_tiles) {
    Plan pl{};
    pl.blocks = col_tiles * rh_cdiv(p.M, BM) * p.nphase;
    pl.total_chunks = rh_cdiv(p.C, p.ck);
    static const int split_below = [] { const char* e = getenv("RH_CONV_SPLIT_BELOW"); return e ? atoi(e) : 512; }();
    static const int split_target = [] { const char* e = getenv("RH_CONV_SPLIT_TARGET"); return e ? atoi(e) : 768; }();
    int z = 1;
    if (pl.blocks < split_below) {
        z = rh_cdiv(split_target, pl.blocks);
        const int zmax = pl.total_chunks / 2;
        if (z > zmax) z = zmax;
        if (z > 16) z = 16;
        if (z < 1) z = 1;
    }
    pl.chunks_per_split = rh_cdiv(pl.total_chunks, z);
    pl.ksplit = rh_cdiv(pl.total_chunks, pl.chunks_per_split);
    return pl;
}

template <int TM, int TN, int WM, int WN>
int launch_dma(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes, bool plan_only = false,
               int64_t* want = nullptr) {
    constexpr int BM = TM * WM * 32, BN = TN * WN * 32;
    int bnl = BN;
    if (p.ncols < BN) {
        bnl = 32;
        while (bnl < p.ncols) bnl <<= 1;
    }
    p.bnl = bnl;
    p.bnl_shift = __builtin_ctz(bnl);
    p.nb = BN / bnl;
    p.tiles_per_b = rh_cdiv(p.ncols, bnl);
    int span = 0, maxtaps = 1;
    for (int i = 0; i < p.nphase; ++i) {
        span = span > p.ph_maxoff[i] - p.ph_minoff[i] ? span : p.ph_maxoff[i] - p.ph_minoff[i];
        maxtaps = maxtaps > p.ph_ntaps[i] ? maxtaps : p.ph_ntaps[i];
    }
    int width;
    if (p.inner == 1)
        width = (bnl - 1) * p.is + span + 1;
    else
        width = ((bnl - 1) / p.inner + 1) * p.is * p.inner + p.inner + span * p.inner + 1;
    static const int novec = [] { const char* e = getenv("RH_CONV_NOVEC"); return e ? atoi(e) : 0; }();
    bool vec = !novec && p.inner == 1 && maxtaps <= 64 && (p.in_row % 4) == 0 && ((uintptr_t)p.in % 16) == 0 &&
               ((uintptr_t)p.wp % 16) == 0;
    static const int budget_env = [] { const char* e = getenv("RH_CONV_STAGE_FLOATS"); return e ? atoi(e) : 0; }();
    int ck = 2;
    for (int attempt = 0; attempt < 2; ++attempt) {
        p.segs = rh_cdiv(width, 64);
        p.pitch = vec ? ((width + 3 + 3) & ~3) : p.segs * 64;   // +3: origin aligned down to 4 elements
        const int per_ch = maxtaps * BM + p.nb * p.pitch;
        // floats per pipeline stage (2 stages per workgroup): 20 KiB stages give 4 workgroups per CU and
        // measured best when they still hold >= 8 channels per chunk; otherwise 40 KiB stages (2 per CU)
        // measured per layer (profiles/round1_layer_table_b32.txt sweep): 20 KiB stages (4 workgroups/CU)
        // for the small-channel / long-sequence layers, 24 KiB (3/CU) from 256 channels up
        int budget = budget_env > 0 ? budget_env : (p.C >= 192 ? 6 * 1024 : 5 * 1024);
        ck = ((budget - 512) / per_ch) & ~1;
        if (budget_env <= 0 && ck < (p.C >= 192 ? 4 : 8)) {
            budget = 10 * 1024;
            ck = ((budget - 512) / per_ch) & ~1;
        }
        if (ck > 32) ck = 32;
        if (ck < 2) ck = 2;
        const int cmax = (p.C + 1) & ~1;
        if (ck > cmax) ck = cmax;
        if (!vec) break;
        const int nx = rh_cdiv(p.nb * ck * (p.pitch >> 2), 64 * WM * WN);
        const int nws = rh_cdiv(rh_cdiv(maxtaps * ck * BM, 256), WM * WN);
        if (nx <= 8 && nws <= 10) break;
        vec = false;   // too many DMA slots per lane: use the element-wise variant
    }
    p.ck = ck;
    p.wlds_floats = (maxtaps * ck * BM + 255) & ~255;
    p.stage_floats = p.wlds_floats + ((p.nb * ck * p.pitch + 255) & ~255);
    p.magic_segs = magic_of(p.segs);
    p.magic_ck = magic_of(ck);
    p.magic_lpr = magic_of(p.pitch >> 2);
    const size_t lds = sizeof(float) * 2 * (size_t)p.stage_floats;
    const int col_tiles = rh_cdiv(p.B, p.nb) * p.tiles_per_b;
    Plan pl = plan_split(p, BM, col_tiles);
    p.part_stride = (long)p.B * p.M * p.out_row;
    const int64_t need = pl.ksplit > 1 ? (int64_t)pl.ksplit * p.part_stride * (int64_t)sizeof(float) : 0;
    if (plan_only) {
        *want = lds > 160 * 1024 ? 0 : need;
        return RH_OK;
    }
    if (lds > 160 * 1024) return rh_conv_launch_sync(p, stream, what);
    if (need > 0 && (!ws || ws_bytes < need)) {   // no scratch offered: run unsplit (slower, same result class)
        pl.ksplit = 1;
        pl.chunks_per_split = pl.total_chunks;
    }
    p.ksplit = pl.ksplit;
    p.chunks_per_split = pl.chunks_per_split;
    p.part = (float*)ws;
    auto launch = [&](auto kern) {
        dim3 grid(col_tiles, rh_cdiv(p.M, BM), p.nphase * p.ksplit);
        rh_launch_main(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    };
    auto go = [&](auto kern) {
        static std::once_flag once;
        std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        launch(kern);
    };
    const bool leaky = p.in_act == RH_ACT_LEAKY;
    const bool ctail = (p.C % p.ck) != 0;
    if (vec && !ctail) {
        if (leaky) go(conv_igemm_dma_kernel<TM, TN, WM, WN, true, true, false>);
        else go(conv_igemm_dma_kernel<TM, TN, WM, WN, false, true, false>);
    } else if (vec) {
        if (leaky) go(conv_igemm_dma_kernel<TM, TN, WM, WN, true, true, true>);
        else go(conv_igemm_dma_kernel<TM, TN, WM, WN, false, true, true>);
    } else {
        if (leaky) go(conv_igemm_dma_kernel<TM, TN, WM, WN, true, false, true>);
        else go(conv_igemm_dma_kernel<TM, TN, WM, WN, false, false, true>);
    }
    if (int e = rh_check_launch(what)) return e;
    if (p.ksplit > 1) return finalize_launch(p, stream);
    return RH_OK;
}

void fill_sizes(ConvP& p) {
    unsigned long long wtaps = 0;
    for (int i = 0; i < p.nphase; ++i) wtaps += p.ph_ntaps[i];
    p.in_bytes = (unsigned)(4ull * p.B * p.C * (unsigned long long)p.in_row);
    p.w_bytes = (unsigned)(4ull * wtaps * p.C * p.Mp);
}

}  // namespace

bool rh_conv_dma_eligible(const ConvP& p) {
    if (p.in_act == RH_ACT_SNAKE) return false;
    const unsigned long long in_b = 4ull * p.B * p.C * (unsigned long long)p.in_row;
    unsigned long long wtaps = 0;
    for (int i = 0; i < p.nphase; ++i) wtaps += p.ph_ntaps[i];
    const unsigned long long w_b = 4ull * wtaps * p.C * p.Mp;
    const unsigned long long row_span = (unsigned long long)p.M * (unsigned long long)p.out_row;
    return in_b < 0x7fffffffull && w_b < 0x7fffffffull && row_span < 0x7fffffffull;
}

int rh_splitk_finalize_launch(ConvP& p, hipStream_t stream) { return finalize_launch(p, stream); }

int rh_conv_launch_dma(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes) {
    fill_sizes(p);
    {   // stride-1 convolutions of the residual units: exact f32 on the 16-bit matrix cores (conv_x6.hip)
        bool used = false;
        if (int e = rh_conv_launch_x6(p, stream, what, ws, ws_bytes, &used)) return e;
        if (used) return RH_OK;
        if (int e = rh_conv_launch_c2x(p, stream, what, &used)) return e;      // stride-3 gathers: 2-D kernel, W = 1
        if (used) return rh_range_after(p, stream);
    }
    // f32-input MFMA kernels: they do not publish the output's range; a requested slot is filled by a pass over the output
    unsigned* const orange = p.out_range;
    p.out_range = nullptr;
    int rc;
    if (p.M <= 32) rc = launch_dma<1, 2, 1, 4>(p, stream, what, ws, ws_bytes);
    else if (p.M <= 64) rc = launch_dma<2, 1, 1, 4>(p, stream, what, ws, ws_bytes);
    else if (p.M % 96 == 0 || p.M < 96) rc = launch_dma<3, 1, 1, 4>(p, stream, what, ws, ws_bytes);
    else rc = launch_dma<2, 2, 2, 2>(p, stream, what, ws, ws_bytes);
    p.out_range = orange;
    return rc ? rc : rh_range_after(p, stream);
}

int64_t rh_conv_splitk_workspace(ConvP p) {
    if (p.B <= 0 || p.ncols <= 0 || p.M <= 0 || !rh_conv_dma_eligible(p)) return 0;
    fill_sizes(p);
    int64_t want = 0;
    const int64_t x6 = rh_conv_x6_workspace(p);
    if (p.M <= 32) launch_dma<1, 2, 1, 4>(p, nullptr, "", nullptr, 0, true, &want);
    else if (p.M <= 64) launch_dma<2, 1, 1, 4>(p, nullptr, "", nullptr, 0, true, &want);
    else if (p.M % 96 == 0 || p.M < 96) launch_dma<3, 1, 1, 4>(p, nullptr, "", nullptr, 0, true, &want);
    else launch_dma<2, 2, 2, 2>(p, nullptr, "", nullptr, 0, true, &want);
    return want > x6 ? want : x6;
}
