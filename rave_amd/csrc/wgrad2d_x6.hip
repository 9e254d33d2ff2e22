// Weight gradient of the general Conv2d as exact f32 on the bf16 matrix cores (the "x6" numerics of conv_x6_kernel.inc:
// every f32 split exactly into three bf16 pieces, six partial products per block), for the 32 -> 32-channel, 27-tap
// layers of the spectral discriminators (rave/discriminator.py:23-74, rave/descript_discriminator.py:118-184), whose
// weight gradients ran at 25-43 TFLOP/s on the f32-input MFMA (conv2d.hip wgrad2d_dma_kernel) and were 35 of the 77 ms of
// an Encodec pass once forward and data gradient had moved to conv2d_x6.hip.
//
//   dW[m][c][th][tw] = sum_{b, r, q} G[b][m][r][q] * X[b][c][r*sh + th*dh - ph][q + tw*dw - pw]        (stride 1 along W)
//
// GEMM view: rows = m (<= 32), columns = (tap, channel), K = output positions.  Both operands are activations, so both
// are converted in the kernel -- ONCE per position: a workgroup (8 waves, one per CU: 143 KB of LDS) stages, per tile of
// TR x 32 output positions of one batch item,
//   * G: 32 rows x TR x 32 positions and
//   * X: 32 channels x the (TR-1)*sh + span_h + 1 by 32 + span_w patch the tile's taps touch,
// as bf16 ELEMENT planes [piece][row or channel][position] (position-contiguous, 2 bytes per element and piece).  A K block
// is 16 consecutive output positions of one output row; the A fragment of a lane is 8 consecutive bf16 of a G row
// (16-byte aligned), the B fragment of column (tap, c) the 8 consecutive bf16 of channel c's patch row r*sh + th*dh
// starting at the element offset of the tap: a 16-byte LDS read at an arbitrary 2-byte alignment, which gfx950's
// ds_read_b128 serves correctly at 1.4x the aligned cost (tools/probe/lds_unaligned.hip) -- so a tap is an address
// offset and NOTHING is converted per tap (the 1-D wgrad_x6_kernel converts the input once per tap: 10 VALU instructions
// per MFMA at M = 32; here ~1.5).
//   * GEMM column n = tap * Cg + channel (Cg = the workgroup's channels, <= 32): at C = 32 a column tile is one tap x 32
//     channels -- 27 tiles over 8 waves (4 or 3 each, 64 accumulator registers); at C = 2 (the first layer: 54 columns)
//     two tiles of 16 taps x 2 channels, every lane with its own (tap, channel) offset.  Every wave walks all K blocks of
//     the tile with its own column tiles.
//   * stride 2 along W (descript MRD, (3,9) kernels): the patch is staged as two PARITY planes (even / odd input columns),
//     so that the 8 K positions of a fragment are again 8 consecutive elements; a conversion task takes 16 consecutive
//     input columns and writes one fragment into each plane.
//   * the next tile's samples are loaded (8 dwords per task, zero outside the plane) before the current tile's MFMAs and
//     converted after them; two barriers per tile.
//   * K is split over persistent workgroups (one per CU) into ordered partial tiles + rh_reduce_partials_launch:
//     deterministic.
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;
constexpr int kW2Waves = 8, kW2Threads = 512;
constexpr int kW2Pos = 128;        // output positions per tile = 8 K blocks: TR rows x TQ columns, TQ in {8, 16, 32}
constexpr int kW2MaxTaps = 32;     // 4 column tiles per wave x 8 waves

struct W2X {
    const float* G;                // dy * act'(y)  [B][M][r_h][r_w]
    const float* X;                // x             [B][C][s_h][s_w]
    float* out;                    // partials [Z][M][C*T]
    int B, M, C, T;
    int r_h, r_w, s_h, s_w;
    int sh;                        // stride along H (1 along W)
    int TR, tr_shift;              // output rows per tile (power of two)
    int TQ, tq_shift;              // output columns per tile: 32, or 16 / 8 on narrow planes (more rows instead)
    int tiles_r, tiles_q, tiles_total;
    int PH, PWp;                   // staged patch rows, row pitch in elements (multiple of 8; per parity plane at stride 2)
    int sw;                        // stride along W: 1, or 2 (two parity planes per channel image)
    int gp, xp;                    // element pitch of one G row image / one X channel image (16-byte slots: odd count)
    int minh, minw;
    int ngt, nxt;                  // conversion tasks of one tile: G, X
    unsigned g_bytes, x_bytes;
    const unsigned* g_range;       // range slots of G and X (f16 build: common.hpp)
    const unsigned* x_range;
    int toff[kW2MaxTaps];          // element offset of a tap inside a channel's patch image
};

__device__ __forceinline__ u32x4 lds_read_b128_any(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int NQX, int SW>
__global__ __launch_bounds__(kW2Threads, 1) void wgrad2d_x6_kernel(const W2X p) {
    constexpr int XL = 8 * SW;                        // input columns one X task loads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* const g_img = reinterpret_cast<unsigned short*>(smem_raw);               // [piece][32][gp]
    unsigned short* const x_img = g_img + kX6P * 32 * p.gp;                                   // [piece][32][xp]
    int* const toff_s = reinterpret_cast<int*>(x_img + kX6P * 32 * p.xp);                     // [kW2MaxTaps]
    const unsigned g_base = (unsigned)(size_t)g_img, x_base = (unsigned)(size_t)x_img;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int c0 = blockIdx.y * 32;
    const int Cg = min(32, p.C - c0);                 // channels of this workgroup
    const int nxt = Cg * p.PH * (p.PWp >> 3);         // its X conversion tasks per tile
    const int ncol = p.T * Cg;                        // its GEMM columns: n = tap * Cg + channel
    const int TR = p.TR, PWp = p.PWp;
    if (tid < kW2MaxTaps) toff_s[tid] = tid < p.T ? p.toff[tid] : 0;
    const int gplane = p.r_h * p.r_w, xplane = p.s_h * p.s_w;

    const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
#if RH_X6_F16
    // scales (common.hpp): both operands are activations -- from their producers' range slots
    int inv_g, inv_x;
    const float gsc = __uint_as_float(rh_x6_scale_bits(rh_range_max(p.g_range), &inv_g));
    const float xsc = __uint_as_float(rh_x6_scale_bits(rh_range_max(p.x_range), &inv_x));
    const float osc = __uint_as_float(rh_x6_unscale_bits(inv_g, inv_x));
#else
    const float gsc = 1.f, xsc = 1.f, osc = 1.f;
#endif

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- conversion tasks (tile-invariant part).  G: (row m, tile row, 8-column fragment) -- one per thread at TR = 4;
    // X: (channel, patch row, 8-column fragment), NQX per thread
    const int TQ = p.TQ;
    const int gfr = TQ >> 3;                         // fragments per G tile row
    const int xfr = PWp >> 3;                        // fragments per patch row
    int gt_m, gt_r, gt_f;
    {
        const int e = tid;
        gt_f = e % gfr;
        const int rr = e / gfr;
        gt_r = rr & (TR - 1);
        gt_m = rr >> p.tr_shift;
    }
    const bool gt_task = tid < p.ngt;
    int xt_c[NQX], xt_r[NQX], xt_f[NQX];
#pragma unroll
    for (int q = 0; q < NQX; ++q) {
        const int e = tid + kW2Threads * q;
        xt_f[q] = e % xfr;
        const int rr = e / xfr;
        xt_r[q] = rr % p.PH;
        xt_c[q] = e < nxt ? rr / p.PH : -1;
    }
    float gr[8], xr[NQX][XL];
    auto load_tile = [&](int tile) {
        const int tq = tile % p.tiles_q;
        const int rest = tile / p.tiles_q;
        const int tr = rest % p.tiles_r;
        const int b = rest / p.tiles_r;
        const int r0 = tr * TR, q0 = tq * TQ;
        {   // G fragment: row r0 + gt_r, columns q0 + 8 f .. + 7, zero beyond the output plane
            const int r = r0 + gt_r, q = q0 + 8 * gt_f;
            const bool ok = gt_task && gt_m < p.M && r < p.r_h;
            const unsigned base = (unsigned)(((b * p.M + gt_m) * gplane + r * p.r_w + q) * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                gr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (ok && q + i < p.r_w) ? base + 4u * i : kOOB, 0, 0));
        }
        const int h0 = r0 * p.sh + p.minh, w0 = q0 * SW + p.minw;
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            const int h = h0 + xt_r[q], w = w0 + XL * xt_f[q];
            const bool ok = xt_c[q] >= 0 && h >= 0 && h < p.s_h;
            const unsigned base = (unsigned)(((b * p.C + c0 + xt_c[q]) * xplane + h * p.s_w + w) * 4);
#pragma unroll
            for (int i = 0; i < XL; ++i)
                xr[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    x_rsrc, (ok && w + i >= 0 && w + i < p.s_w) ? base + 4u * i : kOOB, 0, 0));
        }
    };
    auto split_store = [&](const float (&v)[8], float sc, unsigned short* dst, int piece_stride) {
#if RH_X6_F16
        u32x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const rh_h2 h = rh_h2_split(v[2 * k] * sc, v[2 * k + 1] * sc);
            hi[k] = h.hi; lo[k] = h.lo;
        }
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + piece_stride) = lo;
#else
        (void)sc;
        unsigned h[3][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[0][i] = __float_as_uint(v[i]);
            const float r1 = v[i] - __uint_as_float(h[0][i] & 0xffff0000u);
            h[1][i] = __float_as_uint(r1);
            h[2][i] = __float_as_uint(r1 - __uint_as_float(h[1][i] & 0xffff0000u));
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            u32x4 pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[k] = __builtin_amdgcn_perm(h[s3][2 * k + 1], h[s3][2 * k], 0x07060302u);
            *reinterpret_cast<u32x4*>(dst + s3 * piece_stride) = pk;
        }
#endif
    };
    auto convert_tile = [&]() {
        if (gt_task) split_store(gr, gsc, g_img + gt_m * p.gp + gt_r * TQ + 8 * gt_f, 32 * p.gp);
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            if (xt_c[q] < 0) continue;
            unsigned short* dst = x_img + xt_c[q] * p.xp + xt_r[q] * PWp + 8 * xt_f[q];
            if constexpr (SW == 1) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xr[q][i];
                split_store(v, xsc, dst, 32 * p.xp);
            } else {      // even input columns -> parity plane 0, odd -> plane 1 (PH * PWp elements further)
                float ve[8], vo[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { ve[i] = xr[q][2 * i]; vo[i] = xr[q][2 * i + 1]; }
                split_store(ve, xsc, dst, 32 * p.xp);
                split_store(vo, xsc, dst + p.PH * PWp, 32 * p.xp);
            }
        }
    };

    // this wave's column tiles: wave, wave + 8, ...; column n = tile * 32 + j <-> (tap n / Cg, channel n % Cg)
    __syncthreads();                                   // toff_s
    const int ntile = (ncol + 31) >> 5;
    unsigned boff[4];
    int col_t[4], col_c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = (wave + kW2Waves * i) * 32 + j;
        const bool ok = n < ncol;
        col_t[i] = ok ? n / Cg : -1;
        col_c[i] = ok ? n - col_t[i] * Cg : 0;
        boff[i] = ok ? 2u * (unsigned)(col_c[i] * p.xp + toff_s[col_t[i]]) : 0u;
    }
    const unsigned a_lane = g_base + 2u * (unsigned)(j * p.gp + 8 * g);
    const int nkb = (TR * TQ) >> 4;
    const unsigned a_piece = 2u * 32u * (unsigned)p.gp, b_piece = 2u * 32u * (unsigned)p.xp;

    const int z = blockIdx.x, nz = gridDim.x;
    int tile = z;
    if (tile < p.tiles_total) load_tile(tile);
    for (; tile < p.tiles_total; tile += nz) {
        __syncthreads();                       // every wave is done reading the previous tile's images
        convert_tile();
        if (tile + nz < p.tiles_total) load_tile(tile + nz);
        __syncthreads();
        if (wave < ntile)                   // (a wave without a column tile must not leave LDS reads in flight)
        for (int kb = 0; kb < nkb; ++kb) {
            // K block kb = tile positions 16 kb .. 16 kb + 15 (row-major over TR x TQ); this lane's fragment starts at
            // position 16 kb + 8 g = (row, column) of the tile -- a fragment never straddles a row (TQ is a multiple of 8)
            const unsigned ao = a_lane + 2u * (unsigned)(16 * kb);
            u32x4 av[kX6P];
#pragma unroll
            for (int s3 = 0; s3 < kX6P; ++s3) av[s3] = lds_read_b128_any(ao + s3 * a_piece);
            const int pos = 16 * kb + 8 * g;
            const unsigned bo = x_base + 2u * (unsigned)((pos >> p.tq_shift) * p.sh * PWp + (pos & (TQ - 1)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (wave + kW2Waves * i >= ntile) break;               // wave-uniform
                u32x4 bv[kX6P];
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) bv[s3] = lds_read_b128_any(bo + boff[i] + s3 * b_piece);
#if RH_X6_F16
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]));
#else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]));
#endif
                rh_x6_frag af[kX6P], bf[kX6P];
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) {
                    af[s3] = __builtin_bit_cast(rh_x6_frag, av[s3]);
                    bf[s3] = __builtin_bit_cast(rh_x6_frag, bv[s3]);
                }
                constexpr int SA[RH_X6_NPROD] = RH_X6_SA, SB[RH_X6_NPROD] = RH_X6_SB;     // smallest terms first
#pragma unroll
                for (int q = 0; q < RH_X6_NPROD; ++q) acc[i] = RH_X6_MFMA(af[SA[q]], bf[SB[q]], acc[i]);
            }
        }
    }

    // ---- partial tile of this slice: out[z][m][(c0 + c) * T + t]
    const long CT = (long)p.C * p.T;
    float* const part = p.out + (long)z * p.M * CT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (col_t[i] < 0) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 4 * g + (r & 3) + 8 * (r >> 2);
            if (m < p.M) part[m * CT + (long)(c0 + col_c[i]) * p.T + col_t[i]] = RH_X6_F16 ? acc[i][r] * osc : acc[i][r];
        }
    }
}

bool w2x_enabled() {
    for (const char* k : {"RH_WGRAD2D_X6", "RH_CONV2D_X6", "RH_CONV_X6", "RH_WGRAD_X6"}) {     // read per call (tests)
        const char* e = getenv(k);
        if (e && atoi(e) == 0) return false;
    }
    return true;
}

struct W2XPlan {
    size_t lds;
    int Z, nqx, groups;
};

bool plan_w2x(const rh_conv2d_desc* d, W2X* p, W2XPlan* pl) {
    if (!w2x_enabled() || d->batch <= 0) return false;
    const int T = d->kh * d->kw;
    if (d->c_out > 32 || (d->sw != 1 && d->sw != 2) || T > kW2MaxTaps) return false;
    const unsigned long long g_b = 4ull * d->batch * d->c_out * (unsigned long long)d->h_out * d->w_out;
    const unsigned long long x_b = 4ull * d->batch * d->c_in * (unsigned long long)d->h_in * d->w_in;
    if (!(g_b < 0x7fffffffull && x_b < 0x7fffffffull)) return false;
    *p = W2X{};
    p->B = d->batch; p->M = d->c_out; p->C = d->c_in; p->T = T;
    p->r_h = d->h_out; p->r_w = d->w_out; p->s_h = d->h_in; p->s_w = d->w_in;
    p->sh = d->sh; p->sw = d->sw;
    const int span_h = (d->kh - 1) * d->dh, span_w = (d->kw - 1) * d->dw;
    p->minh = -d->ph; p->minw = -d->pw;
    const int cg = d->c_in < 32 ? d->c_in : 32;          // channels of the fullest workgroup
    // tile = TR x TQ output positions, 128 where the plane allows: TQ = 32 columns, or 16 / 8 on narrow planes (descript
    // MRD bands shrink to 8 columns) with correspondingly more rows
    int TQ = 32;
    while (TQ > 8 && TQ / 2 >= d->w_out) TQ >>= 1;
    int TR = kW2Pos / TQ;
    while (TR > 1 && TR / 2 >= d->h_out) TR >>= 1;
    for (;; TR >>= 1) {
        p->TR = TR; p->TQ = TQ;
        p->tr_shift = __builtin_ctz(TR); p->tq_shift = __builtin_ctz(TQ);
        p->PH = (TR - 1) * d->sh + span_h + 1;
        // row pitch of a (parity) plane: the farthest fragment starts at column TQ - 8 + span_w / sw
        p->PWp = (TQ + span_w / d->sw + 7) & ~7;
        p->gp = (TR * TQ) | 8;                          // 16-byte slots per row image: odd -> lanes (rows) spread over the banks
        p->xp = d->sw * p->PH * p->PWp;
        if (((p->xp >> 3) & 1) == 0) p->xp += 8;
        pl->lds = (size_t)kX6P * 32 * ((size_t)p->gp + p->xp) * 2 + kW2MaxTaps * 4;
        p->ngt = 32 * TR * (TQ / 8);
        p->nxt = cg * p->PH * (p->PWp >> 3);
        if (pl->lds <= 160 * 1024 && p->ngt <= kW2Threads && p->nxt <= kW2Threads * (d->sw == 1 ? 6 : 3)) break;
        if (TR == 1) return false;
    }
    pl->nqx = p->nxt <= kW2Threads * 3 ? 3 : 6;
    // element offset of a tap inside a channel image: patch row th*dh, patch column tw*dw (stride 2: parity plane
    // (tw*dw) & 1, plane column (tw*dw) >> 1)
    for (int th = 0; th < d->kh; ++th)
        for (int tw = 0; tw < d->kw; ++tw) {
            const int ow = tw * d->dw;
            p->toff[th * d->kw + tw] = d->sw == 1 ? th * d->dh * p->PWp + ow
                                                  : (ow & 1) * p->PH * p->PWp + th * d->dh * p->PWp + (ow >> 1);
        }
    p->tiles_r = rh_cdiv(d->h_out, TR);
    p->tiles_q = rh_cdiv(d->w_out, p->TQ);
    const long tiles = (long)d->batch * p->tiles_r * p->tiles_q;
    if (tiles >= 0x7fffffffl) return false;
    p->tiles_total = (int)tiles;
    pl->groups = rh_cdiv(d->c_in, 32);
    static const int z_env = [] { const char* e = getenv("RH_WGRAD2D_X6_SLICES"); return e ? atoi(e) : 0; }();
    int Z = z_env > 0 ? z_env : 256 / pl->groups;       // one workgroup per CU (up to 143 KB of LDS)
    if (Z < 1) Z = 1;
    if (Z > tiles) Z = (int)tiles;
    pl->Z = Z;
    p->g_bytes = (unsigned)g_b; p->x_bytes = (unsigned)x_b;
    return true;
}

// One instantiation -- and therefore one once_flag -- per kernel variant: every variant gets its own
// MaxDynamicSharedMemorySize attribute (plans ask for up to 145 KB of dynamic LDS).
template <int NQX, int SW>
void w2x_go(const W2X& p, const W2XPlan& pl, hipStream_t stream) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad2d_x6_kernel<NQX, SW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    });
    rh_launch_main(wgrad2d_x6_kernel<NQX, SW>, dim3((unsigned)pl.Z, (unsigned)pl.groups), dim3(kW2Threads), pl.lds, stream, p);
}

}  // namespace

// bytes of partial-tile scratch the bf16x6 weight gradient wants; -1 = geometry not eligible
int64_t rh_wgrad2d_x6_workspace(const rh_conv2d_desc* d) {
    W2X p;
    W2XPlan pl{};
    if (!plan_w2x(d, &p, &pl)) return -1;
    return (int64_t)pl.Z * d->c_out * d->c_in * d->kh * d->kw * (int64_t)sizeof(float);
}

// dw = sum over positions of dy (x) x-patches; dy already carries act'(y).  *used = false when the geometry (or the scratch
// offered) does not fit.
int rh_wgrad2d_x6_launch(const rh_conv2d_desc* d, const float* dy, const float* x, float* dw, void* ws, int64_t ws_bytes,
                         hipStream_t stream, bool* used, const unsigned* dy_range, const unsigned* x_range) {
    *used = false;
    W2X p;
    W2XPlan pl{};
    if (RH_X6_F16 && (!dy_range || !x_range)) return RH_OK;      // no range slots (rh_x6_set_ranges): f32-input MFMA kernels
    if (!plan_w2x(d, &p, &pl)) return RH_OK;
    p.g_range = dy_range; p.x_range = x_range;
    const long nw = (long)d->c_out * d->c_in * d->kh * d->kw;
    const int64_t need = (int64_t)pl.Z * nw * (int64_t)sizeof(float);
    if (((uintptr_t)dy & 3) || ((uintptr_t)x & 3)) return RH_OK;
    if (pl.Z > 1 && (!ws || ws_bytes < need)) return RH_OK;
    p.G = dy; p.X = x;
    p.out = pl.Z > 1 ? (float*)ws : dw;
    if (d->sw == 2) w2x_go<3, 2>(p, pl, stream);
    else if (pl.nqx == 3) w2x_go<3, 1>(p, pl, stream);
    else w2x_go<6, 1>(p, pl, stream);
    if (int e = rh_check_launch("wgrad2d_x6")) return e;
    *used = true;
    if (pl.Z > 1) return rh_reduce_partials_launch((const float*)ws, dw, nw, pl.Z, stream, "wgrad2d_x6_reduce");
    return RH_OK;
}
