// Fused Residual(DilatedUnit) forward (rave/blocks.py:31-45, 83-112):
//     h = conv_k_dil(act(x), W3)          C -> C, "same" padding, no bias
//     y = conv_1x1(act(h), W1) + x
// as ONE launch of the bf16x6 machinery (conv_x6_kernel.inc; same numerics, same accumulation order per output element
// as the two separate launches -> bit-identical results) for C = 32, 64, 96:
//   * GEMM 1 is conv_x6_kernel's main loop (stride-1 taps, one row tile: a wave owns ALL C rows of its columns);
//   * h never makes the round trip through HBM: the accumulator tile of a wave already holds every channel of its
//     columns, lane (j, g) the rows 4g + (r & 3) + 8 (r >> 2).  The B operand of GEMM 2 wants, per 16-channel K block and
//     lane, the eight channels 8g .. 8g+7: activation + exact 3-way bf16 split + pair packing in registers, then two
//     v_permlane32_swap per piece exchange the quads {4..7} <-> {8..11} between the two half-waves -- the standard
//     fragment order, so the 1x1 conv's ordinary packed weights serve as A operand.  No LDS traffic for h at all;
//   * W1 (C x C, 6 bytes per weight: 55 KB at C = 96) is loaded into LDS once after GEMM 1 and GEMM 2 runs without
//     barriers, one column tile at a time (accumulators: 16 TM registers instead of 32 TM);
//   * training: h is stored (f32, once) for the backward pass before GEMM 2 starts, so that the stores drain under its
//     MFMAs; a no-grad forward never writes it.  The epilogue adds the residual x and stores y.
// What it saves per unit against two launches: one launch + prologue + store burst (~14 us), the read of h (and its
// write in inference).  C >= 192 needs the rows of two waves per column (LDS exchange) or 6 row tiles per wave and stays
// on the two-launch path.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "conv_params.hpp"
#include "conv_x6_kernel.inc"

bool rh_conv_x6_plan_fixed(ConvP& p, int tm, int tn, int wm, size_t* lds, dim3* grid);   // conv_x6.hip

namespace {

struct UnitP {
    ConvP c;                   // geometry / operands of the dilated conv (in = x, wq = its fragments, out = y, add = x)
    const unsigned* wq1;       // fragments of the 1x1 conv: [chunk][g][piece][Mp]
    unsigned wq1_bytes;
    float slope2;              // LeakyReLU slope applied to h (1 = none)
    float* h_out;              // h for the backward pass, or null (no-grad forward)
    const unsigned* w1_range;  // range record of the 1x1 conv's weights (behind its fragments)
    unsigned* h_range;         // where max |h| goes (range slot of the backward pass's consumers), or null
};

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void unit_x6_kernel(const UnitP u) {
    const ConvP& p = u.c;
    constexpr int WN = 4;
    constexpr int BM = 32 * TM;
    constexpr int A_UNITS = 2 * kX6P * BM;
    constexpr int NAL = (A_UNITS + 255) / 256;
    constexpr int NQ = WN * TN == 8 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const a_st = reinterpret_cast<u32x4*>(smem_raw);
    u32x4* const b_st = a_st + 2 * A_UNITS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;

    const int tap0 = p.ph_tap0[0];
    const int minoff = p.ph_minoff[0];
    const int nu = p.ph_ntaps[0];
    const int bt = blockIdx.x / p.tiles_per_b;
    const int nt = blockIdx.x - bt * p.tiles_per_b;
    const int b0 = bt * p.nb;
    const int n0 = nt * p.bnl;
    const int pitch = p.pitch, P = p.x6_P;
    const float slope1 = p.in_act == RH_ACT_LEAKY ? p.in_slope : 1.f;        // max(v, 1 v) = v

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.wq), 0, p.wq_bytes, 0x00020000);
#if RH_X6_F16
    // scales (common.hpp).  x: its range slot; W3, W1: their records.  h is converted in registers before its maximum can be
    // known: it is scaled by the BOUND max |x| * max_m sum_k |W3[m][k]| (the record's second word) -- a few bits above the
    // true maximum, paid out of the 15 bits of headroom the two-piece representation has below the tensor maximum.
    int inv_x, inv_w3, inv_h, inv_w1;
    const unsigned xmax = rh_range_max(p.in_range);
    const float xsc = __uint_as_float(rh_x6_scale_bits(xmax, &inv_x));
    (void)rh_x6_scale_bits(p.w_range[0], &inv_w3);
    const float osc1 = __uint_as_float(rh_x6_unscale_bits(inv_w3, inv_x));                   // accumulators of GEMM 1 -> h
    const float hbound = __uint_as_float(xmax) * __uint_as_float(p.w_range[1]);
    const unsigned hsc_bits = rh_x6_scale_bits(__float_as_uint(hbound), &inv_h);
    (void)rh_x6_scale_bits(u.w1_range[0], &inv_w1);
    const float osc2 = __uint_as_float(rh_x6_unscale_bits(inv_w1, inv_h));                   // accumulators of GEMM 2 -> conv1(h)
    // accumulators of GEMM 1 -> scaled h in ONE multiplication: osc1 * hsc, exponents added and clamped
    const float h2sc = __uint_as_float(rh_x6_unscale_bits((int)(__float_as_uint(osc1) >> 23), (int)(hsc_bits >> 23)));
#else
    const float xsc = 1.f, osc1 = 1.f, osc2 = 1.f, h2sc = 1.f;
#endif

    int bpos[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = min(n0 + nl, p.ncols - 1);
        bpos[tn] = g * kX6P * P + bl * pitch + (n - n0);
    }
    const int arow = g * kX6P * BM + j;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- GEMM 1: the stride-1 main loop of conv_x6_kernel (see there for the staging pipeline)
    unsigned aoff[NAL];
#pragma unroll
    for (int r = 0; r < NAL; ++r) {
        const int uu = tid + 256 * r;
        const int gs = uu / BM, mrow = uu - gs * BM;
        aoff[r] = (uu < A_UNITS && mrow < p.Mp) ? (unsigned)((gs * p.Mp + mrow) * 16) : kOOB;
    }
    const unsigned step_bytes = (unsigned)(2 * kX6P * p.Mp * 16);
    unsigned xoff[NQ];
    int xdst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int o = e >= P ? 1 : 0;
        const int pp = e - o * P;
        const int bl = pp / pitch, qq = pp - bl * pitch;
        const bool task = e < 2 * P;
        const bool bok = task && b0 + bl < p.B;
        xdst[q] = task ? o * kX6P * P + pp : -1;
        const int f = n0 + minoff + qq;
        const bool ok = bok && f >= 0 && f < p.in_valid;
        xoff[q] = ok ? (unsigned)((((b0 + bl) * p.C + 8 * o) * p.in_row + f) * 4) : kOOB;
    }
    const unsigned chunk_bytes = (unsigned)(16 * p.in_row * 4);
    unsigned rowc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowc[i] = (unsigned)(i * p.in_row * 4);

    float xr[NQ][8];
    u32x4 ar[NAL];
    auto load_x = [&](int chunk) {
        const unsigned cbv = (unsigned)chunk * chunk_bytes;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xr[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, xoff[q], cbv + rowc[i], 0));
    };
    // activation (max(v, slope v)) + split of 8 values that are multiplied by `sc` first (a power of two; 1 in the bf16 build)
    auto split8 = [&](const float (&v)[8], float slope, float sc, u32x4 (&pk)[kX6P]) {
#if RH_X6_F16
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = v[2 * k] * sc, b = v[2 * k + 1] * sc;
            a = rh_max1(a, a * slope);
            b = rh_max1(b, b * slope);
            const rh_h2 h = rh_h2_split(a, b);
            pk[0][k] = h.hi; pk[1][k] = h.lo;
        }
#else
        unsigned h[3][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = rh_max1(v[i], v[i] * slope);
            rh_bf3_split(a, h[0][i], h[1][i], h[2][i]);
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[s3][k] = __builtin_amdgcn_perm(h[s3][2 * k + 1], h[s3][2 * k], 0x07060302u);
#endif
    };
    auto convert_x = [&](int stage) {
        u32x4* dst0 = b_st + stage * 2 * kX6P * P;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (xdst[q] < 0) continue;
            u32x4 pk[kX6P];
            split8(xr[q], slope1, xsc, pk);
#pragma unroll
            for (int s3 = 0; s3 < kX6P; ++s3) dst0[xdst[q] + s3 * P] = pk[s3];
        }
    };
    const int nchunks = p.C >> 4;
    const int S = nchunks * nu;
    const unsigned sbytes0 = (unsigned)(p.ph_q2ofs[0] * 16);
    auto load_a = [&](int step) {
        const unsigned sb = sbytes0 + (unsigned)step * step_bytes;
#pragma unroll
        for (int r = 0; r < NAL; ++r) {
            if (r * 256 >= A_UNITS) continue;
            ar[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, aoff[r], sb, 0));
        }
    };
    auto store_a = [&](int stage) {
#pragma unroll
        for (int r = 0; r < NAL; ++r) {
            const int uu = tid + 256 * r;
            if (uu < A_UNITS) a_st[stage * A_UNITS + uu] = ar[r];
        }
    };
    load_a(0);
    load_x(0);
    store_a(0);
    if (S > 1) load_a(1);
    convert_x(0);
    if (nchunks > 1) load_x(1);
    __syncthreads();
    constexpr int SA[RH_X6_NPROD] = RH_X6_SA, SB[RH_X6_NPROD] = RH_X6_SB;     // smallest terms first
    int st = 0;
    int toff_next = p.off[tap0] - minoff;
    for (int ci = 0; ci < nchunks; ++ci) {
        const u32x4* bl_ = b_st + (ci & 1) * 2 * kX6P * P;
        for (int t = 0; t < nu; ++t, ++st) {
            const int toff = toff_next;
            {
                const int t1 = t + 1 < nu ? t + 1 : 0;
                toff_next = p.off[tap0 + t1] - minoff;
            }
            const u32x4* al = a_st + (st & 1) * A_UNITS + arow;
            rh_x6_frag bfr[TN][kX6P], afr[TM][kX6P];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) bfr[tn][s3] = __builtin_bit_cast(rh_x6_frag, bl_[bpos[tn] + s3 * P + toff]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) afr[tm][s3] = __builtin_bit_cast(rh_x6_frag, al[s3 * BM + tm * 32]);
            if (st + 1 < S) {
                store_a((st + 1) & 1);
                if (st + 2 < S) load_a(st + 2);
            }
            if (t == nu - 1 && ci + 1 < nchunks) {
                convert_x((ci + 1) & 1);
                if (ci + 2 < nchunks) load_x(ci + 2);
            }
#pragma unroll
            for (int q = 0; q < RH_X6_NPROD; ++q)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = RH_X6_MFMA(afr[tm][SA[q]], bfr[tn][SB[q]], acc[tm][tn]);
            __syncthreads();
        }
    }

    // ---- output addressing of this lane's column tiles (x, h and y share the layout)
    unsigned cb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int bl = col >> p.bnl_shift;
        const int nl = col & (p.bnl - 1);
        const int n = n0 + nl, b = b0 + bl;
        const bool ok = n < p.ncols && b < p.B && n < p.out_valid;
        cb[tn] = ok ? (unsigned)(((long)b * p.M * p.out_row + n) * 4) : kOOB;
    }
    const unsigned obytes = (unsigned)(p.part_stride * 4);
    const auto y_r = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, obytes, 0x00020000);
    const auto x_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.add), 0, obytes, 0x00020000);
    const auto none_r = __builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, 0x00020000);
    const bool full = BM <= p.M;
    float hmax_all = 0.f;
    if (u.h_out) {      // training: h for the backward pass; the stores drain under GEMM 2
        const auto h_r = __builtin_amdgcn_make_buffer_rsrc(u.h_out, 0, obytes, 0x00020000);
        x6_store_tile<0, TM, TN>(acc, cb, 0, g, p.M, full, p.out_row, h_r, none_r, none_r, none_r, 1.f, 0.f, osc1, hmax_all);
        // (published together with y's at the end: the LDS is about to be reused for W1)
    }

    // ---- W1 fragments -> LDS, whole operand ([chunk][g][piece][Mp], Mp == BM): GEMM 2 runs without barriers
    constexpr int W1_UNITS = (BM / 16) * 2 * kX6P * BM;
    u32x4* const w1s = reinterpret_cast<u32x4*>(smem_raw);
    {
        const auto w1_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(u.wq1), 0, u.wq1_bytes, 0x00020000);
        constexpr int NW = (W1_UNITS + 255) / 256;
        u32x4 wv[NW];
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int uu = tid + 256 * r;
            wv[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w1_r, uu < W1_UNITS ? (unsigned)(uu * 16) : kOOB, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int uu = tid + 256 * r;
            if (uu < W1_UNITS) w1s[uu] = wv[r];
        }
    }
    __syncthreads();

    // ---- GEMM 2, one column tile at a time: B fragments straight from the accumulators of GEMM 1
    float ymax = 0.f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        f32x16 acc2[TM][1];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[tm][0][r] = 0.f;
#pragma unroll
        for (int b = 0; b < 2 * TM; ++b) {
            const int tmh = b >> 1, hf = b & 1;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = acc[tmh][tn][8 * hf + i];
            u32x4 pk[kX6P];
            split8(v, u.slope2, h2sc, pk);
            rh_x6_frag bfr2[kX6P], afr2[TM][kX6P];
#pragma unroll
            for (int s3 = 0; s3 < kX6P; ++s3) {
                // lane (j, 0) holds channels {0..3, 8..11} of the block, lane (j, 1) {4..7, 12..15}: exchanging the second
                // pair of dwords of the lower half-wave with the first pair of the upper one gives 8g .. 8g+7 per lane
                auto r0 = __builtin_amdgcn_permlane32_swap(pk[s3][0], pk[s3][2], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(pk[s3][1], pk[s3][3], false, false);
                u32x4 f;
                f[0] = r0[0]; f[2] = r0[1]; f[1] = r1[0]; f[3] = r1[1];
                bfr2[s3] = __builtin_bit_cast(rh_x6_frag, f);
            }
            const u32x4* al = w1s + (b * 2 * kX6P + g * kX6P) * BM + j;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) afr2[tm][s3] = __builtin_bit_cast(rh_x6_frag, al[s3 * BM + tm * 32]);
#pragma unroll
            for (int q = 0; q < RH_X6_NPROD; ++q)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc2[tm][0] = RH_X6_MFMA(afr2[tm][SA[q]], bfr2[SB[q]], acc2[tm][0]);
        }
        const unsigned cb1[1] = {cb[tn]};
        x6_store_tile<4, TM, 1>(acc2, cb1, 0, g, p.M, full, p.out_row, y_r, none_r, x_r, none_r, 1.f, 0.f, osc2, ymax);
    }
    if (RH_X6_F16 && (p.out_range || u.h_range)) {      // uniform; W1's LDS image is dead after a barrier
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
        if (p.out_range) rh_range_publish(p.out_range, ymax, blockIdx.x, red);
        if (u.h_range) rh_range_publish(u.h_range, hmax_all, blockIdx.x + 5u, red + 4);
    }
}

bool unit_enabled() {
    const char* e = getenv("RH_UNIT_FUSED");        // read per call (tests): 0 = two launches
    if (e && atoi(e) == 0) return false;
    const char* x6 = getenv("RH_CONV_X6");
    return !(x6 && atoi(x6) == 0);
}

bool unit_shape_ok(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1) {
    if (!d3 || !d1) return false;
    const int C = d3->c_in;
    auto act_ok = [](const rh_conv1d_desc* d) {
        return d->act == RH_ACT_NONE || (d->act == RH_ACT_LEAKY && d->act_slope >= 0.f && d->act_slope <= 1.f);
    };
    return (C == 32 || C == 64 || C == 96) && d3->c_out == C && d1->c_in == C && d1->c_out == C && !d3->transposed &&
           !d1->transposed && d3->stride == 1 && d1->stride == 1 && d3->inner == 1 && d1->inner == 1 && d3->groups == 1 &&
           d1->groups == 1 && d1->kernel == 1 && d1->dilation == 1 && d1->pad_left == 0 && d3->in_valid == 0 &&
           d1->in_valid == 0 && d3->out_act == RH_ACT_NONE && d1->out_act == RH_ACT_NONE && act_ok(d3) && act_ok(d1) &&
           d3->batch > 0 && d3->batch == d1->batch && d3->l_in > 0 && d3->l_out == d3->l_in && d1->l_in == d3->l_in &&
           d1->l_out == d3->l_in;
}

int fill_unit(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1, UnitP* u, size_t* lds, dim3* grid) {
    ConvP& p = u->c;
    if (int e = rh_conv_fill_fwd(d3, &p)) return e;
    ConvP p1{};
    if (int e = rh_conv_fill_fwd(d1, &p1)) return e;
    if (p.x6_mode != 1 || p1.x6_mode != 1 || p.nphase != 1) return RH_ERR_UNSUPPORTED;
    const int tm = p.M / 32;
    if (!rh_conv_x6_plan_fixed(p, tm, 2, 1, lds, grid)) return RH_ERR_UNSUPPORTED;
    const size_t w1_bytes = (size_t)(p.M / 16) * 2 * kX6P * p.M * 16;
    if (w1_bytes > *lds) *lds = w1_bytes;
    if (*lds > 160 * 1024 || (unsigned long long)p1.wq_bytes < w1_bytes) return RH_ERR_UNSUPPORTED;
    u->wq1_bytes = p1.wq_bytes;
    u->slope2 = d1->act == RH_ACT_LEAKY ? d1->act_slope : 1.f;
    // (the launcher fills the pointers; p1.x6_wofs is the float offset of the 1x1 conv's fragments in its packed operand)
    u->h_out = nullptr;
    u->wq1 = reinterpret_cast<const unsigned*>((uintptr_t)p1.x6_wofs);    // offset, resolved by the caller
    return RH_OK;
}

}  // namespace

// 1 = rh_residual_unit_fwd_f32 takes this pair of geometries (the bf16x6 kernels are enabled and the unit fits one launch)
extern "C" int rh_residual_unit_fused(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1) {
    if (!unit_enabled() || !unit_shape_ok(d3, d1)) return 0;
    UnitP u{};
    size_t lds = 0;
    dim3 grid;
    static const unsigned any_range[kRangeSlotWords] = {};
    u.c.in_range = any_range;            // planning only
    return fill_unit(d3, d1, &u, &lds, &grid) == RH_OK ? 1 : 0;
}

// y = conv_1x1(act1(h), w1) + x with h = conv_k(act3(x), w3) ("same" padding, no biases) in one launch.  wp3_fwd /
// wp1_fwd: the packed forward operands of the two convs (rh_conv1d_pack_*); h: C x L buffer that receives the
// intermediate for the backward pass, or NULL (inference).  Replaces Residual(DilatedUnit(...)) (rave/blocks.py:31-45,83-112).
extern "C" int rh_residual_unit_fwd_f32(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1, const float* x, const float* wp3_fwd,
                                        const float* wp1_fwd, float* h, float* y, rh_stream_t stream) {
    RH_REQUIRE(unit_enabled() && unit_shape_ok(d3, d1), RH_ERR_UNSUPPORTED, "residual_unit_fwd: geometry not fusable");
    RH_REQUIRE(x && wp3_fwd && wp1_fwd && y, RH_ERR_INVALID, "residual_unit_fwd: null pointer");
    UnitP u{};
    size_t lds = 0;
    dim3 grid;
    const unsigned* x_range = nullptr;
    unsigned *y_range = nullptr, *h_range = nullptr;
    rh_take_ranges(nullptr, &x_range, &y_range, &h_range);
    RH_REQUIRE(!RH_X6_F16 || x_range, RH_ERR_INVALID, "residual_unit_fwd: the f16 kernels need the input's range slot (rh_x6_set_ranges)");
    u.c.in_range = x_range;
    if (int e = fill_unit(d3, d1, &u, &lds, &grid)) {
        rh_set_error("residual_unit_fwd: geometry not fusable");
        return e;
    }
    ConvP& p = u.c;
    p.in_range = x_range; p.out_range = y_range; u.h_range = h ? h_range : nullptr;
    RH_REQUIRE((((uintptr_t)wp3_fwd | (uintptr_t)wp1_fwd) & 15) == 0 && ((uintptr_t)x & 3) == 0, RH_ERR_INVALID,
               "residual_unit_fwd: misaligned operand");
    p.in = x; p.wp = wp3_fwd; p.out = y; p.bias = nullptr; p.add = x; p.mul_src = nullptr;
    p.wq = reinterpret_cast<const unsigned*>(wp3_fwd + p.x6_wofs);
    p.in_alpha = nullptr; p.mul_alpha = nullptr;
    p.in_bytes = (unsigned)(4ull * p.B * p.C * (unsigned long long)p.in_row);
    u.wq1 = reinterpret_cast<const unsigned*>(wp1_fwd + (long)(uintptr_t)u.wq1);
    p.w_range = p.wq + p.wq_bytes / 4;
    u.w1_range = u.wq1 + u.wq1_bytes / 4;
    u.h_out = h;
    auto go = [&](auto kern) {
        static std::once_flag once;
        std::call_once(once, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        rh_launch_main(kern, grid, dim3(256), lds, (hipStream_t)stream, u);
    };
    const int tm = p.M / 32;
    if (tm == 1) go(unit_x6_kernel<1, 2>);
    else if (tm == 2) go(unit_x6_kernel<2, 2>);
    else go(unit_x6_kernel<3, 2>);
    return rh_check_launch("residual_unit_fwd");
}
