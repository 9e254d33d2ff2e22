// Exact-f32 convolution on the bf16 matrix cores ("x6"), for the stride-1 convolutions of the residual units
// (forward and data gradient: 44 of the 56 generator-side layers of v2).
//
// Every f32 value splits EXACTLY into three bf16 pieces (truncation splits: x == h1 + h2 + h3), and a product
// a*b is taken as the six partial products with i + j <= 4 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1), accumulated in
// f32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are <= 2^-23 |a||b|: measured error 1.5x the rounding error
// of an f32 fmaf chain (2.6e-7 ... 7.5e-7 relative on K = 288 ... 2304), i.e. the same numerics class as the
// v_mfma_f32_32x32x2_f32 kernels -- at 6 x 32 cycles per 32x32x16 block instead of 8 x 64.
//
// Pipeline per K chunk of 16 channels (same GEMM view, tile 96 x 128, 4 waves, 2 workgroups per CU):
//   DMA (16 B/lane) f32 input tile + pre-split bf16 weight triples -> LDS | barrier |
//   convert the f32 tile once (zero padding mask + LeakyReLU + 3-way split) into [c/8][piece][position][8] |
//   barrier | issue the next chunk's DMA | 18 MFMAs per tap from ds_read_b128 fragments (no VALU in the loop).
// The bf16 weight triples are written by the weight repack (conv_host.hip: pack_tile) right behind the f32 operand.
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned kOOB = 0x80000000u;
constexpr int kBM = 96, kBN = 128;
constexpr int kNXS = 8, kNWS = 8;     // 16-byte DMA slots per lane: input tile, weight tile

__device__ __forceinline__ void split3(float x, unsigned& a, unsigned& b, unsigned& c) {
    a = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(a);
    b = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b);
    c = __float_as_uint(r2) & 0xffff0000u;
}

template <bool LEAKY>
__global__ __launch_bounds__(256) void conv_x6_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xf = reinterpret_cast<float*>(smem_raw);
    unsigned char* xb = smem_raw + (size_t)p.x6_xf_floats * 4;
    unsigned char* ws = xb + p.x6_xb_bytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int zsl = blockIdx.z;
    const int ntaps = p.ph_ntaps[0];
    const int minoff = p.ph_minoff[0];

    const int bt = blockIdx.x / p.tiles_per_b;
    const int nt = blockIdx.x - bt * p.tiles_per_b;
    const int b0 = bt * p.nb;
    const int n0 = nt * p.bnl;
    const int m0 = blockIdx.y * kBM;
    const int pitch = p.pitch;
    const int lo_raw = n0 + minoff;
    const int lo = lo_raw - ((lo_raw % 4 + 4) % 4);      // 16-byte DMA: tile origin aligned down to 4 elements

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wq), 0, p.wq_bytes, 0x00020000);

    const int col = wave * 32 + j;
    const int bl_c = col >> p.bnl_shift;
    const int nl_c = col & (p.bnl - 1);
    const int n_c = min(n0 + nl_c, p.ncols - 1);
    const int bbase = ((bl_c * 2 + g) * 3) * pitch + (n_c - n0) + (lo_raw - lo);     // 16-byte units

    f32x16 acc[3];
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    // ---- per-lane DMA source offsets (chunk 0), elements / bytes; a chunk only adds a scalar
    const int lpr = pitch >> 2;
    const int xrows = p.nb * 16;
    const int x_slots = xrows * lpr;                       // float4 slots of the input tile
    const int w16 = ntaps * 2 * 3 * kBM;                   // 16-byte slots of the weight tile
    unsigned xo[kNXS], wo[kNWS];
#pragma unroll
    for (int i = 0; i < kNXS; ++i) {
        const int s = (wave + 4 * i) * 64 + lane;
        const int row = s / lpr, v = s - row * lpr;
        const int bl = row >> 4, c = row & 15;
        xo[i] = (s < x_slots && b0 + bl < p.B) ? (unsigned)((bl * p.C + c) * p.in_row + 4 * v) : kOOB;
    }
#pragma unroll
    for (int i = 0; i < kNWS; ++i) {
        const int f = (wave + 4 * i) * 64 + lane;
        const int m = f % kBM;
        int r = f / kBM;
        const int s3 = r % 3;
        r /= 3;
        const int cb = r & 1, t = r >> 1;
        wo[i] = (f < w16 && m0 + m < p.Mp) ? (unsigned)(((((long)t * (p.C >> 3) + cb) * 3 + s3) * p.Mp + m0 + m) * 16) : kOOB;
    }
    const unsigned w_step = (unsigned)(2l * 3 * p.Mp * 16);
    auto issue = [&](int chunk, int stage) {
        unsigned char* wdst = ws + stage * p.x6_w_bytes;
#pragma unroll
        for (int i = 0; i < kNWS; ++i)
            if ((wave + 4 * i) * 64 < w16) {
                const unsigned off = wo[i] == kOOB ? kOOB : wo[i] + (unsigned)chunk * w_step;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wdst + (wave + 4 * i) * 1024), 16, off, 0, 0, 0);
            }
        const unsigned xbase = ((unsigned)b0 * p.C + (unsigned)chunk * 16u) * (unsigned)p.in_row + (unsigned)lo;
#pragma unroll
        for (int i = 0; i < kNXS; ++i)
            if ((wave + 4 * i) * 256 < p.x6_xf_floats) {
                const unsigned off = xo[i] == kOOB ? kOOB : (xbase + xo[i]) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)(xf + (wave + 4 * i) * 256), 16, off, 0, 0, 0);
            }
    };

    const int total_chunks = p.C >> 4;
    const int chunk0 = zsl * p.chunks_per_split;
    const int nchunks = min(p.chunks_per_split, total_chunks - chunk0);
    // conversion work list of this thread: fragment (item, channel block, position) -> source / destination offsets
    // and validity are chunk-invariant, computed once (nfrag <= 3 x 256)
    constexpr int kNF = 3;
    const int nfrag = p.nb * 2 * pitch;
    int fsrc[kNF], fdst[kNF];
#pragma unroll
    for (int q = 0; q < kNF; ++q) {
        const int e = tid + 256 * q;
        const int bl = e / (2 * pitch);
        const int r = e - bl * 2 * pitch;
        const int cb = r >= pitch ? 1 : 0;
        const int pos = r - cb * pitch;
        const int f = lo + pos;
        const bool valid = f >= 0 && f < p.in_valid;
        fsrc[q] = e < nfrag ? (valid ? (bl * 16 + cb * 8) * pitch + pos : -1) : -2;
        fdst[q] = ((bl * 2 + cb) * 3) * pitch + pos;
    }
    if (nchunks > 0) issue(chunk0, 0);
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // chunk i landed; every wave is done with the bf16 tile of chunk i-1
        // ---- convert: zero the positions outside the sequence (padding / neighbouring rows fetched by the
        //      16-byte DMA), fused LeakyReLU, exact 3-way bf16 split; one fragment = 8 channels of one position
#pragma unroll
        for (int q = 0; q < kNF; ++q) {
            if (fsrc[q] == -2) continue;
            unsigned h[3][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = fsrc[q] >= 0 ? xf[fsrc[q] + k * pitch] : 0.f;
                if (LEAKY) v = v > 0.f ? v : v * p.in_slope;
                split3(v, h[0][k], h[1][k], h[2][k]);
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                u32x4 pk;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk[k] = (h[s3][2 * k] >> 16) | h[s3][2 * k + 1];
                *reinterpret_cast<u32x4*>(xb + (size_t)(fdst[q] + s3 * pitch) * 16) = pk;
            }
        }
        __syncthreads();      // bf16 tile ready; the f32 stage is free again
        if (i + 1 < nchunks) issue(chunk0 + i + 1, (i + 1) & 1);
        const unsigned char* wl = ws + (i & 1) * p.x6_w_bytes;
        for (int t = 0; t < ntaps; ++t) {
            const int toff = p.off[t] - minoff;
            bf16x8 bfr[3], afr[3][3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
                bfr[s3] = *reinterpret_cast<const bf16x8*>(xb + (size_t)(bbase + s3 * pitch + toff) * 16);
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
                    afr[tm][s3] = *reinterpret_cast<const bf16x8*>(wl + (size_t)((((t * 2 + g) * 3 + s3) * kBM) + tm * 32 + j) * 16);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tm][SA[q]], bfr[SB[q]], acc[tm], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue (same as the f32 kernels): bias, activation derivative, residual / gradient add
    const int n = n0 + nl_c, b = b0 + bl_c;
    if (n >= p.ncols || b >= p.B || n >= p.out_valid) return;
    const long cbase = (long)b * p.M * p.out_row + n;
    float* __restrict__ outp = (p.ksplit > 1 ? p.part + (long)zsl * p.part_stride : p.out) + cbase;
    const float* __restrict__ mulp = p.mul_src ? p.mul_src + cbase : nullptr;
    const float* __restrict__ addp = p.add ? p.add + cbase : nullptr;
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
        const int mb = m0 + tm * 32 + 4 * g;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < p.M) {
                const int ro = m * p.out_row;
                float v = acc[tm][r];
                if (p.ksplit > 1) {
                    outp[ro] = v;
                    continue;
                }
                if (p.bias) v += p.bias[m];
                if (mulp) v *= rh_act_grad(mulp[ro], p.epi_act, p.epi_slope, 0.f);
                if (addp) v += addp[ro];
                if (p.out_act == RH_ACT_LEAKY) v = v > 0.f ? v : v * p.out_slope;
                outp[ro] = v;
            }
        }
    }
}

struct X6Plan {
    size_t lds;
    int64_t wq_bytes, part_bytes;
    int ksplit, chunks_per_split, col_tiles;
};

bool plan_x6(ConvP& p, X6Plan* pl) {
    static const bool off = [] { const char* e = getenv("RH_CONV_X6"); return e && atoi(e) == 0; }();
    if (off) return false;
    if (p.inner != 1 || p.is != 1 || p.os != 1 || p.nphase != 1 || p.ph_oph[0] != 0) return false;
    if (p.in_act == RH_ACT_SNAKE || p.epi_act == RH_ACT_SNAKE) return false;
    if (!rh_x6_weights(p.M, p.C, p.ph_ntaps[0], p.nphase, p.is, p.os, p.inner) || p.x6_packed == 0) return false;
    if ((p.in_row & 3) || ((uintptr_t)p.in & 15) || ((uintptr_t)p.wp & 15)) return false;
    // pointwise convs on few channels: 18 MFMAs per 16-channel chunk do not amortise the conversion pass and its two
    // barriers (measured at C = 96: 67 us vs 54 us for the f32 kernel on the data gradient)
    if (p.ph_ntaps[0] == 1 && p.C < 192) return false;
    int bnl = kBN;
    if (p.ncols < kBN) {
        bnl = 32;
        while (bnl < p.ncols) bnl <<= 1;
    }
    p.bnl = bnl;
    p.bnl_shift = __builtin_ctz(bnl);
    p.nb = kBN / bnl;
    p.tiles_per_b = rh_cdiv(p.ncols, bnl);
    const int span = p.ph_maxoff[0] - p.ph_minoff[0];
    const int width = (bnl - 1) + span + 1;
    p.pitch = (width + 3 + 3) & ~3;
    const int x_slots = p.nb * 16 * (p.pitch >> 2);
    const int w16 = p.ph_ntaps[0] * 2 * 3 * kBM;
    if (rh_cdiv(x_slots, 256) > kNXS || rh_cdiv(w16, 256) > kNWS || p.nb * 2 * p.pitch > 3 * 256) return false;
    p.x6_xf_floats = (p.nb * 16 * p.pitch + 255) & ~255;
    p.x6_xb_bytes = p.nb * 2 * 3 * p.pitch * 16;
    p.x6_w_bytes = ((w16 + 63) & ~63) * 16;
    pl->lds = (size_t)p.x6_xf_floats * 4 + p.x6_xb_bytes + 2 * (size_t)p.x6_w_bytes;
    if (pl->lds > 160 * 1024) return false;
    pl->col_tiles = rh_cdiv(p.B, p.nb) * p.tiles_per_b;
    const int blocks = pl->col_tiles * (p.M / kBM);
    const int total_chunks = p.C >> 4;
    int z = 1;
    if (blocks < 512) {
        z = rh_cdiv(768, blocks);
        if (z > total_chunks) z = total_chunks;
        if (z > 16) z = 16;
        if (z < 1) z = 1;
    }
    pl->chunks_per_split = rh_cdiv(total_chunks, z);
    pl->ksplit = rh_cdiv(total_chunks, pl->chunks_per_split);
    p.part_stride = (long)p.B * p.M * p.out_row;
    pl->part_bytes = pl->ksplit > 1 ? (int64_t)pl->ksplit * p.part_stride * (int64_t)sizeof(float) : 0;
    pl->wq_bytes = (int64_t)p.ph_ntaps[0] * (p.C >> 3) * 3 * p.Mp * 16;
    const unsigned long long in_b = 4ull * p.B * p.C * (unsigned long long)p.in_row;
    const unsigned long long row_span = (unsigned long long)p.M * (unsigned long long)p.out_row;
    return in_b < 0x7fffffffull && pl->wq_bytes < 0x7fffffffll && row_span < 0x7fffffffull;
}

}  // namespace

int rh_splitk_finalize_launch(ConvP& p, hipStream_t stream);

int64_t rh_conv_x6_workspace(ConvP p) {
    X6Plan pl{};
    if (!plan_x6(p, &pl)) return -1;
    return pl.part_bytes;
}

// Returns RH_OK with *used = false when the geometry (or the scratch offered) does not fit this path.
int rh_conv_launch_x6(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes, bool* used) {
    *used = false;
    ConvP q = p;
    X6Plan pl{};
    if (!plan_x6(q, &pl)) return RH_OK;
    if (pl.part_bytes > 0 && (!ws || ws_bytes < pl.part_bytes)) return RH_OK;
    q.wq = reinterpret_cast<const unsigned short*>(q.wp + (long)q.ph_ntaps[0] * q.C * q.Mp);
    q.wq_bytes = (unsigned)pl.wq_bytes;
    q.in_bytes = (unsigned)(4ull * q.B * q.C * (unsigned long long)q.in_row);
    q.part = (float*)ws;
    q.ksplit = pl.ksplit;
    q.chunks_per_split = pl.chunks_per_split;
    dim3 grid(pl.col_tiles, q.M / kBM, q.ksplit);
    auto go = [&](auto kern) {
        static std::once_flag once;
        std::call_once(once, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, stream, q);
    };
    if (q.in_act == RH_ACT_LEAKY) go(conv_x6_kernel<true>);
    else go(conv_x6_kernel<false>);
    if (int e = rh_check_launch(what)) return e;
    *used = true;
    if (q.ksplit > 1) return rh_splitk_finalize_launch(q, stream);
    return RH_OK;
}
