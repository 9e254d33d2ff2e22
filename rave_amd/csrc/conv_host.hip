// Host side of the convolution entry points: geometry -> tap plan -> kernel parameters, weight
// packing, and the extern "C" functions declared in include/rave_hip.h.
#include "conv_params.hpp"

int64_t rh_wgrad_workspace(const rh_conv1d_desc* d);

namespace {

struct TapPlan {
    int nphase = 0;
    int oph[kMaxPhases], ntaps[kMaxPhases], tap0[kMaxPhases], minoff[kMaxPhases], maxoff[kMaxPhases];
    int nslots = 0;
    int kk[kMaxTaps], off[kMaxTaps];
    int is = 1, os = 1;
    int rows = 0;   // output rows per phase
    int C = 0, M = 0;  // GEMM K-channels / GEMM rows
    bool src_m_major = true;  // source index = (m*C + c)*k + kk, else (c*M + m)*k + kk
};

int validate(const rh_conv1d_desc* d) {
    RH_REQUIRE(d, RH_ERR_INVALID, "conv1d: null descriptor");
    RH_REQUIRE(d->batch >= 0 && d->c_in > 0 && d->c_out > 0 && d->l_in >= 0 && d->l_out >= 0,
               RH_ERR_INVALID, "conv1d: bad sizes");
    RH_REQUIRE(d->kernel >= 1 && d->stride >= 1 && d->dilation >= 1 && d->inner >= 1,
               RH_ERR_INVALID, "conv1d: kernel/stride/dilation/inner must be >= 1");
    RH_REQUIRE(d->groups == 1, RH_ERR_UNSUPPORTED, "conv1d: groups = %d not implemented", d->groups);
    RH_REQUIRE(d->kernel <= kMaxTaps, RH_ERR_UNSUPPORTED, "conv1d: kernel %d > %d", d->kernel, kMaxTaps);
    RH_REQUIRE(d->act >= RH_ACT_NONE && d->act <= RH_ACT_SNAKE, RH_ERR_INVALID, "conv1d: bad act");
    RH_REQUIRE(d->out_act == RH_ACT_NONE || d->out_act == RH_ACT_LEAKY, RH_ERR_UNSUPPORTED,
               "conv1d: out_act must be none or leaky");
    RH_REQUIRE(d->in_valid >= 0 && d->in_valid <= (int64_t)d->l_in * d->inner, RH_ERR_INVALID,
               "conv1d: in_valid out of range");
    if (d->transposed) {
        RH_REQUIRE(d->dilation == 1 && d->inner == 1, RH_ERR_UNSUPPORTED,
                   "conv_transpose1d: dilation/inner must be 1");
        RH_REQUIRE(d->stride <= kMaxPhases, RH_ERR_UNSUPPORTED, "conv_transpose1d: stride > %d", kMaxPhases);
    } else if (d->stride > 1) {
        RH_REQUIRE(d->stride <= kMaxPhases, RH_ERR_UNSUPPORTED, "conv1d: stride > %d", kMaxPhases);
    }
    return RH_OK;
}

// which: 0 = forward operand, 1 = data-gradient operand
int build_plan(const rh_conv1d_desc* d, int which, TapPlan* tp) {
    TapPlan& t = *tp;
    const int k = d->kernel, s = d->stride, dil = d->dilation, P = d->pad_left;
    auto phases = [&](int pad) {
        t.nphase = s;
        t.is = 1;
        t.os = s;
        for (int ph = 0; ph < s; ++ph) {
            const int r0 = ((ph + pad) % s + s) % s;
            const int base = (ph + pad - r0) / s;
            t.oph[ph] = ph;
            t.tap0[ph] = t.nslots;
            int n = 0;
            for (int m = 0; r0 + m * s < k; ++m, ++n) {
                t.kk[t.nslots] = r0 + m * s;
                t.off[t.nslots] = base - m;
                ++t.nslots;
            }
            t.ntaps[ph] = n;
        }
    };
    auto single = [&]() {
        t.nphase = 1;
        t.oph[0] = 0;
        t.tap0[0] = 0;
        t.ntaps[0] = k;
        t.nslots = k;
    };
    if (which == 0) {
        t.C = d->c_in;
        t.M = d->c_out;
        if (!d->transposed) {
            single();
            for (int i = 0; i < k; ++i) { t.kk[i] = i; t.off[i] = i * dil - P; }
            t.is = s; t.os = 1; t.rows = d->l_out;
            t.src_m_major = true;   // w[co=m][ci=c][kk]
        } else {
            phases(P);
            t.rows = rh_cdiv(d->l_out, s);
            t.src_m_major = false;  // w[ci=c][co=m][kk]
        }
    } else {
        t.C = d->c_out;
        t.M = d->c_in;
        if (!d->transposed) {
            if (s == 1) {
                single();
                for (int i = 0; i < k; ++i) { t.kk[i] = i; t.off[i] = P - i * dil; }
                t.is = 1; t.os = 1; t.rows = d->l_in;
            } else {
                RH_REQUIRE(dil == 1, RH_ERR_UNSUPPORTED, "conv1d bwd_data: stride > 1 with dilation > 1");
                phases(P);
                t.rows = rh_cdiv(d->l_in, s);
            }
            t.src_m_major = false;  // w[co=c][ci=m][kk]
        } else {
            single();
            for (int i = 0; i < k; ++i) { t.kk[i] = i; t.off[i] = i - P; }
            t.is = s; t.os = 1; t.rows = d->l_in;
            t.src_m_major = true;   // w[ci=m][co=c][kk]
        }
    }
    for (int ph = 0; ph < t.nphase; ++ph) {
        int lo = 0, hi = 0;
        for (int i = 0; i < t.ntaps[ph]; ++i) {
            const int o = t.off[t.tap0[ph] + i];
            if (i == 0 || o < lo) lo = o;
            if (i == 0 || o > hi) hi = o;
        }
        t.minoff[ph] = lo;
        t.maxoff[ph] = hi;
    }
    return RH_OK;
}

inline int round32(int m) { return (m + 31) & ~31; }

void plan_to_params(const TapPlan& t, ConvP* p) {
    p->C = t.C;
    p->M = t.M;
    p->Mp = round32(t.M);
    p->is = t.is;
    p->os = t.os;
    p->nphase = t.nphase;
    long wofs = 0;
    for (int ph = 0; ph < t.nphase; ++ph) {
        p->ph_oph[ph] = t.oph[ph];
        p->ph_ntaps[ph] = t.ntaps[ph];
        p->ph_tap0[ph] = t.tap0[ph];
        p->ph_minoff[ph] = t.minoff[ph];
        p->ph_maxoff[ph] = t.maxoff[ph];
        p->ph_wofs[ph] = wofs;
        wofs += (long)t.ntaps[ph] * t.C * p->Mp;
    }
    for (int i = 0; i < t.nslots; ++i) p->off[i] = t.off[i];
}

// "Virtual rows": the s output phases of a transposed convolution / of the data gradient of a stride-s convolution,
// taken as s * M GEMM rows of ONE stride-1 gather over the same input.  Phase ph reads in[q + base_ph - m] for its
// taps m = 0 .. ntaps_ph - 1 and writes out[s q + ph]; base_ph is b0 for ph < ph*, b0 + 1 from ph* on.  With column
// n = q for the first group and n = q + 1 for the second, every phase reads in[n + b0 - u] (u < U = max ntaps) and the
// s outputs of a column are the CONTIGUOUS run out[s n + o0 + j], j = (ph - ph*) mod s, o0 = ph* ? ph* - s : 0 --
// one 16-byte store per lane and row group instead of s launches' worth of 4-byte stores s elements apart.
struct VPlan { int s, U, phstar, o0, b0, jof[kMaxPhases]; };
bool vplan_of(const TapPlan& t, int inner, VPlan* v) {
    static const bool on = [] { const char* e = getenv("RH_X6_VROWS"); return !(e && atoi(e) == 0); }();
    const int s = t.nphase;
    if (!on || (s != 2 && s != 4) || t.os != s || t.is != 1 || inner != 1 || (t.C & 15) || ((t.M * s) & 31)) return false;
    int U = 0;
    for (int ph = 0; ph < s; ++ph) {
        if (t.ntaps[ph] < 1) return false;
        for (int m = 0; m < t.ntaps[ph]; ++m)
            if (t.off[t.tap0[ph] + m] != t.off[t.tap0[ph]] - m) return false;
        U = U > t.ntaps[ph] ? U : t.ntaps[ph];
    }
    const int b0 = t.off[t.tap0[0]];
    int phstar = 0;
    for (int ph = 0; ph < s; ++ph) {
        const int b = t.off[t.tap0[ph]];
        if (b == b0 + 1) { if (!phstar) phstar = ph; }
        else if (b != b0 || phstar) return false;          // bases must be b0 ... b0, b0+1 ... b0+1
    }
    v->s = s; v->U = U; v->phstar = phstar; v->b0 = b0;
    v->o0 = phstar ? phstar - s : 0;
    for (int ph = 0; ph < s; ++ph) v->jof[ph] = (ph - phstar + s) % s;
    return true;
}

// bf16x6 section of a packed operand (conv_x6.hip): layout mode and size in 16-byte fragments
int x6_mode_of(const TapPlan& t, int inner) {
    return rh_x6_mode(t.C, t.nphase, t.is, inner, t.ntaps[0], t.off, t.kk);
}
// fragments of the bf16x6 section(s); *vofs = first fragment of the virtual-row section (-1: none)
long x6_units(const TapPlan& t, int mode, long* ph_ofs = nullptr, int inner = 1, long* vofs = nullptr) {
    const long Mp = round32(t.M);
    if (vofs) *vofs = -1;
    if (mode == 0) return 0;
    if (mode == 1) {
        long u = 0;
        for (int ph = 0; ph < t.nphase; ++ph) {
            if (ph_ofs) ph_ofs[ph] = u;
            u += (long)(t.C >> 4) * t.ntaps[ph] * 2 * kX6P * Mp;
        }
        VPlan v;
        if (vplan_of(t, inner, &v)) {
            if (vofs) *vofs = u;
            u += (long)(t.C >> 4) * v.U * 2 * kX6P * round32(t.M * v.s);
        }
        return u;
    }
    if (ph_ofs) ph_ofs[0] = 0;
    return (long)((t.C * mode) >> 4) * rh_cdiv(t.ntaps[0], mode) * 2 * kX6P * Mp;
}
void plan_to_x6(const TapPlan& t, int inner, ConvP* p) {
    p->x6_mode = x6_mode_of(t, inner);
    long vofs = -1;
    long units = x6_units(t, p->x6_mode, p->ph_q2ofs, inner, &vofs);
    if (units * 4 >= 0x7fffffffl) { p->x6_mode = 0; units = 0; vofs = -1; }      // same bound as the packers
    p->x6_nu = p->x6_mode > 1 ? rh_cdiv(t.ntaps[0], p->x6_mode) : 0;
    p->vs = 0;
    p->v_s = 0;
    if (vofs >= 0) {               // the launcher (conv_x6.hip) rewrites its copy of the parameters from these
        VPlan v;
        vplan_of(t, inner, &v);
        p->v_s = v.s; p->v_q2ofs = vofs;
        p->v_o0 = v.o0; p->v_U = v.U; p->v_b0 = v.b0;
    }
    p->x6_wofs = (long)t.nslots * t.C * round32(t.M);
    p->wq_bytes = units * 16 < 0xffffffffl ? (unsigned)(units * 16) : 0xffffffffu;
}

// One 32 (c) x 32 (m) tile of ALL slots of one packed copy per workgroup, through LDS (8 slots per pass): the source is
// read along its contiguous index ((m*C + c)*k + kk for the m-major copies, (c*M + m)*k + kk for the others), the
// packed f32 tensor is written along m, and -- for geometries the bf16x6 kernels take (conv_x6.hip) -- the same values
// are split exactly into three bf16 pieces and written as 16-byte fragments of 8 K values in the order that kernel
// walks them ([phase][step][g][piece][Mp], layouts in conv_params.hpp).
constexpr int kPackSlots = 4;      // slots per pass: 16.9 KB of LDS (8 slots: 385 us for the v2 model's repack, 4: 314)

__host__ __device__ __forceinline__ long pack_tiles(const PackP& q) {
    return q.total ? (long)(q.Mp / 32) * ((q.C + 31) / 32) : 0;
}

// The kX6P pieces of 8 K values as 16-byte fragments.  f16 build: the values are scaled by `sc` (the power of two that takes
// the tensor's max |w| into [2^14, 2^15): common.hpp) and split hi / lo; bf16 build: exact 3-way truncation split, sc unused.
__device__ __forceinline__ void emit_fragment_bf3(const float (&v)[8], unsigned* dst, long piece_stride_u32) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned h[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rh_bf3_split(v[i], h[0][i], h[1][i], h[2][i]);
        h[0][i] &= 0xffff0000u; h[1][i] &= 0xffff0000u; h[2][i] &= 0xffff0000u;
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = (h[s3][2 * k] >> 16) | h[s3][2 * k + 1];
        *reinterpret_cast<u32x4*>(dst + s3 * piece_stride_u32) = pk;
    }
}
__device__ __forceinline__ void emit_fragment(const float (&v)[8], unsigned* dst, long piece_stride_u32, float sc) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#if RH_X6_F16
    u32x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const rh_h2 h = rh_h2_split(v[2 * k] * sc, v[2 * k + 1] * sc);
        hi[k] = h.hi; lo[k] = h.lo;
    }
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + piece_stride_u32) = lo;
#else
    unsigned h[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rh_bf3_split(v[i], h[0][i], h[1][i], h[2][i]);
        h[0][i] &= 0xffff0000u; h[1][i] &= 0xffff0000u; h[2][i] &= 0xffff0000u;
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = (h[s3][2 * k] >> 16) | h[s3][2 * k + 1];
        *reinterpret_cast<u32x4*>(dst + s3 * piece_stride_u32) = pk;
    }
#endif
}

// (sg, sn): this workgroup takes the slot groups sg, sg + sn, ... of the tile -- a conv with ONE 32 x 32 tile and 27 taps
// (the 2-D discriminator layers) otherwise packs all its slots serially in one workgroup: 36 us per layer, 108 layers per
// v3 step
__device__ __forceinline__ void pack_tile(const PackP& q, long tile, float* lds /* [kPackSlots][32][33] */, int sg = 0, int sn = 1) {
    const int ct = (q.C + 31) / 32;
    const int c0 = (int)(tile % ct) * 32;
    const int m0 = (int)(tile / ct) * 32;
    const int tid = threadIdx.x;
    float fsc = 1.f;              // f16 build: scale of the pieces, from the range record the range kernel left behind the fragments
#if RH_X6_F16
    if (q.wq && !q.bf16x3) {
        int inv;
        fsc = __uint_as_float(rh_x6_scale_bits(q.range[0], &inv));
    }
#endif
    for (int s0 = sg * kPackSlots; s0 < q.nslots; s0 += sn * kPackSlots) {
        const int ns = min(kPackSlots, q.nslots - s0);
        // U elements per thread are LOADED before the first one is stored: one element per iteration (index -> tap table ->
        // weight -> scale, each a dependent memory access) made this phase latency-bound -- the batched repack of the v2
        // generator moved 0.88 GB in 310 us.  The tap indices of the pass are wave-uniform and fetched once.
        constexpr int U = 8;
        int kks[kPackSlots];
#pragma unroll
        for (int j = 0; j < kPackSlots; ++j) kks[j] = q.kk[min(s0 + j, q.nslots - 1)];
        const int total = ns * 1024;
        for (int e0 = tid; e0 < total; e0 += 256 * U) {
            float v[U], sc[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 256 * u;
                int sl, r;
                if (ns == 4) { sl = e & 3; r = e >> 2; }
                else if (ns == 3) { r = e / 3; sl = e - 3 * r; }
                else if (ns == 2) { sl = e & 1; r = e >> 1; }
                else { sl = 0; r = e; }
                int cl, ml;
                if (q.m_major) { cl = r & 31; ml = r >> 5; }
                else { ml = r & 31; cl = r >> 5; }
                const int m = m0 + ml, c = c0 + cl;
                const bool in = e < total;
                const bool ok = in && m < q.M && c < q.C;
                const int kk = sl == 0 ? kks[0] : (sl == 1 ? kks[1] : (sl == 2 ? kks[2] : kks[3]));
                const long idx = q.m_major ? ((long)m * q.C + c) * q.k + kk : ((long)c * q.M + m) * q.k + kk;
                v[u] = ok ? q.w[idx] : 0.f;
                sc[u] = (ok && q.scale) ? q.scale[q.m_major ? m : c] : 1.f;       // dim 0 of the PyTorch weight tensor
                dst[u] = in ? (sl * 32 + cl) * 33 + ml : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) lds[dst[u]] = v[u] * sc[u];
        }
        __syncthreads();
        for (int e = tid; e < ns * 1024; e += 256) {
            const int ml = e & 31, cl = (e >> 5) & 31, sl = e >> 10;
            const int c = c0 + cl;
            if (c < q.C) q.wp[((long)(s0 + sl) * q.C + c) * q.Mp + m0 + ml] = lds[(sl * 32 + cl) * 33 + ml];
        }
        if (q.wq && q.x6_mode == 1) {
            for (int e = tid; e < ns * 128; e += 256) {
                const int ml = e & 31, oct = (e >> 5) & 3, sl = e >> 7;
                const int cb = c0 + oct * 8;
                if (cb >= q.C) continue;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = lds[(sl * 32 + oct * 8 + i) * 33 + ml];
                const int slot = s0 + sl;
                if (q.bf16x3) {          // (the 2-D kernels' operand: q2a / q2n were filled for three pieces)
                    const long unit3 = (long)q.q2a[slot] + (long)(cb >> 4) * q.q2n[slot] + (long)(((cb >> 3) & 1) * 3) * q.Mp + m0 + ml;
                    emit_fragment_bf3(v, q.wq + unit3 * 4, (long)q.Mp * 4);
                    continue;
                }
                const long unit = (long)q.q2a[slot] + (long)(cb >> 4) * q.q2n[slot] + (long)(((cb >> 3) & 1) * kX6P) * q.Mp + m0 + ml;
                emit_fragment(v, q.wq + unit * 4, (long)q.Mp * 4, fsc);
            }
        }
        if (q.wq && q.x6_vs > 1) {         // second section: virtual rows
            const int vs = q.x6_vs;
            for (int e = tid; e < ns * 128; e += 256) {
                const int ml = e & 31, oct = (e >> 5) & 3, sl = e >> 7;
                const int cb = c0 + oct * 8;
                const int r = (m0 + ml) * vs + q.q2j[s0 + sl];
                if (cb >= q.C || r >= q.x6_Mvp) continue;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = lds[(sl * 32 + oct * 8 + i) * 33 + ml];
                const int slot = s0 + sl;
                const long unit = q.x6_vofs + q.vq2a[slot] + (long)(cb >> 4) * q.x6_U * 2 * kX6P * q.x6_Mvp + (long)(((cb >> 3) & 1) * kX6P) * q.x6_Mvp + r;
                emit_fragment(v, q.wq + unit * 4, (long)q.x6_Mvp * 4, fsc);
            }
            if (s0 == 0) {         // (row group j, tap u) pairs no slot covers: phases with fewer taps than the longest
                const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int e = tid; e < q.vz_n * 128; e += 256) {
                    const int ml = e & 31, oct = (e >> 5) & 3, zi = e >> 7;
                    const int cb = c0 + oct * 8;
                    const int r = (m0 + ml) * vs + q.vz_j[zi];
                    if (cb >= q.C || r >= q.x6_Mvp) continue;
                    const long unit = q.x6_vofs + (long)q.vz_u[zi] * 2 * kX6P * q.x6_Mvp + (long)(cb >> 4) * q.x6_U * 2 * kX6P * q.x6_Mvp +
                                      (long)(((cb >> 3) & 1) * kX6P) * q.x6_Mvp + r;
                    emit_fragment(zero, q.wq + unit * 4, (long)q.x6_Mvp * 4, fsc);
                }
            }
        }
        if (q.wq && q.x6_mode > 1) {
            const int IS = q.x6_mode, cpc = 16 / IS, nvc = 32 / cpc;
            const int nug = (ns + IS - 1) / IS;                    // s0 is a multiple of IS (kPackSlots % IS == 0)
            for (int e = tid; e < nug * nvc * 64; e += 256) {
                const int ml = e & 31, g = (e >> 5) & 1;
                const int r = e >> 6;
                const int cv = r % nvc, ul = r / nvc;
                const int cbase = c0 + cv * cpc;
                if (cbase >= q.C) continue;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kap = 8 * g + i;
                    const int cl = cv * cpc + kap / IS, sl = ul * IS + kap % IS;
                    v[i] = sl < ns ? lds[(sl * 32 + cl) * 33 + ml] : 0.f;
                }
                const long unit = ((long)(cbase / cpc) * q.x6_nu + (s0 / IS + ul)) * 2 * kX6P * q.Mp + (long)(g * kX6P) * q.Mp + m0 + ml;
                emit_fragment(v, q.wq + unit * 4, (long)q.Mp * 4, fsc);
            }
        }
        __syncthreads();
    }
}

// Both packed copies in one launch (forward operand, then data-gradient operand).
__global__ __launch_bounds__(256) void pack_kernel(const PackP a, const PackP b) {
    __shared__ float lds[kPackSlots * 32 * 33];
    const long t = blockIdx.x;
    const long na = pack_tiles(a);
    if (t < na) pack_tile(a, t, lds, blockIdx.y, gridDim.y);
    else if (t - na < pack_tiles(b)) pack_tile(b, t - na, lds, blockIdx.y, gridDim.y);
}

// slot groups of one pack launch spread over grid.y when the launch has few tiles (<= 8 groups)
inline unsigned pack_slot_groups(const PackP& a, const PackP& b, long tiles) {
    if (tiles >= 64) return 1;
    const int ns = a.nslots > b.nslots ? a.nslots : b.nslots;
    const int g = (ns + kPackSlots - 1) / kPackSlots;
    return (unsigned)(g < 1 ? 1 : (g > 8 ? 8 : g));
}

int fill_pack(const rh_conv1d_desc* d, int which, const float* w, const float* scale, float* wp, PackP* p) {
    *p = PackP{};
    if (!wp) return RH_OK;
    TapPlan t;
    if (int e = build_plan(d, which, &t)) return e;
    p->w = w; p->scale = scale; p->wp = wp; p->C = t.C; p->M = t.M; p->Mp = round32(t.M); p->k = d->kernel;
    // dim 0 of the weight tensor is c_out for Conv1d and c_in for ConvTranspose1d:
    //   which=0: M = c_out. Conv1d -> m_major (dim0 = m);  ConvT -> c-major (dim0 = c)   -> scale index as coded
    //   which=1: M = c_in.  Conv1d -> c-major (dim0 = c);  ConvT -> m_major (dim0 = m)
    p->m_major = t.src_m_major ? 1 : 0;
    p->total = (long)t.nslots * t.C * p->Mp;
    p->nslots = t.nslots;
    for (int i = 0; i < t.nslots; ++i) p->kk[i] = t.kk[i];
    const int mode = x6_mode_of(t, d->inner);
    long ph_ofs[kMaxPhases];
    long vofs = -1;
    const long units = x6_units(t, mode, ph_ofs, d->inner, &vofs);
    if (mode && units * 4 < 0x7fffffffl) {
        p->wq = reinterpret_cast<unsigned*>(wp + p->total);            // 16-byte aligned: Mp % 32 == 0
        p->range = p->wq + units * 4;                                  // {max |w|, max row sum |w|, 0, 0} behind the fragments
        p->x6_mode = mode;
        if (vofs >= 0) {
            VPlan v;
            vplan_of(t, d->inner, &v);
            p->x6_vs = v.s;
            p->x6_vofs = vofs;
            p->x6_U = v.U;
            p->x6_Mvp = round32(t.M * v.s);
            bool have[kMaxPhases][kMaxTaps] = {};
            for (int ph = 0; ph < t.nphase; ++ph)
                for (int tl = 0; tl < t.ntaps[ph]; ++tl) {
                    const int slot = t.tap0[ph] + tl;
                    p->vq2a[slot] = tl * 2 * kX6P * p->x6_Mvp;
                    p->q2j[slot] = v.jof[ph];
                    have[v.jof[ph]][tl] = true;
                }
            for (int j = 0; j < v.s; ++j)
                for (int u = 0; u < v.U; ++u)
                    if (!have[j][u]) {
                        RH_REQUIRE(p->vz_n < 2 * kMaxPhases, RH_ERR_UNSUPPORTED, "conv1d_pack: too many empty (phase, tap) pairs");
                        p->vz_j[p->vz_n] = j; p->vz_u[p->vz_n] = u; ++p->vz_n;
                    }
        }
        if (mode == 1) {
            for (int ph = 0; ph < t.nphase; ++ph)
                for (int tl = 0; tl < t.ntaps[ph]; ++tl) {
                    p->q2a[t.tap0[ph] + tl] = (int)(ph_ofs[ph] + (long)tl * 2 * kX6P * p->Mp);
                    p->q2n[t.tap0[ph] + tl] = t.ntaps[ph] * 2 * kX6P * p->Mp;
                }
        } else {
            p->x6_nu = rh_cdiv(t.ntaps[0], mode);
        }
    }
    return RH_OK;
}

// Range record of a weight tensor (rows = dim 0, cols = the rest; scale = the weight-norm factor per row or null): max |w| and
// the largest row sum of |w| (for Conv1d a row of dim 0 is a GEMM row of the forward operand: |conv(x, w)| <= max |x| times it --
// the bound unit_x6.hip scales its intermediate with), as float bit patterns atomicMax'ed into the (zeroed) records of both
// packed copies.  One workgroup per row.  max |scale * v| == |scale| * max |v| bit for bit (rounding is monotonic).
__global__ __launch_bounds__(256) void pack_range_kernel(const float* __restrict__ w, const float* __restrict__ scale, long cols,
                                                         unsigned* ra, unsigned* rb) {
    __shared__ float red[8];
    const long r = blockIdx.x;
    const float* wr = w + r * cols;
    float mx = 0.f, sm = 0.f;
    for (long e = threadIdx.x; e < cols; e += 256) { const float a = fabsf(wr[e]); mx = fmaxf(mx, a); sm += a; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); sm += __shfl_down(sm, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; red[4 + (threadIdx.x >> 6)] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sc = scale ? fabsf(scale[r]) : 1.f;
        const float m = sc * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float l1 = sc * ((red[4] + red[5]) + (red[6] + red[7]));
        if (ra) { atomicMax(ra, __float_as_uint(m)); atomicMax(ra + 1, __float_as_uint(l1)); }
        if (rb) { atomicMax(rb, __float_as_uint(m)); atomicMax(rb + 1, __float_as_uint(l1)); }
    }
}

int pack_both(const rh_conv1d_desc* d, const float* w, const float* scale, float* wp_fwd, float* wp_bwd,
              hipStream_t stream) {
    PackP a, b;
    if (int e = fill_pack(d, 0, w, scale, wp_fwd, &a)) return e;
    if (int e = fill_pack(d, 1, w, scale, wp_bwd, &b)) return e;
    const long tiles = pack_tiles(a) + pack_tiles(b);
    if (tiles == 0) return RH_OK;
    if (a.range || b.range) {
        if (a.range && hipMemsetAsync(a.range, 0, 16, stream) != hipSuccess) return rh_check_launch("conv1d_pack_range");
        if (b.range && hipMemsetAsync(b.range, 0, 16, stream) != hipSuccess) return rh_check_launch("conv1d_pack_range");
        const long rows = d->transposed ? d->c_in : d->c_out;
        const long cols = (long)(d->transposed ? d->c_out : d->c_in) * d->kernel;
        hipLaunchKernelGGL(pack_range_kernel, dim3((unsigned)rows), dim3(256), 0, stream, w, scale, cols, a.range, b.range);
        if (int e = rh_check_launch("conv1d_pack_range")) return e;
    }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)tiles, pack_slot_groups(a, b, tiles)), dim3(256), 0, stream, a, b);
    return rh_check_launch("conv1d_pack");
}

}  // namespace

// w / rows / cols given: the range records of both copies are computed first (pack_range_kernel: rows = dim 0 of w)
int rh_pack_launch(const PackP& a, const PackP& b, hipStream_t stream, const char* what, const float* w, long rows, long cols) {
    const long tiles = pack_tiles(a) + pack_tiles(b);
    if (tiles == 0) return RH_OK;
    if (w && (a.range || b.range)) {
        if (a.range && hipMemsetAsync(a.range, 0, 16, stream) != hipSuccess) return rh_check_launch(what);
        if (b.range && hipMemsetAsync(b.range, 0, 16, stream) != hipSuccess) return rh_check_launch(what);
        hipLaunchKernelGGL(pack_range_kernel, dim3((unsigned)rows), dim3(256), 0, stream, w, (const float*)nullptr, cols, a.range, b.range);
        if (int e = rh_check_launch(what)) return e;
    }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)tiles, pack_slot_groups(a, b, tiles)), dim3(256), 0, stream, a, b);
    return rh_check_launch(what);
}

int rh_conv_fill_fwd(const rh_conv1d_desc* d, ConvP* p) {
    if (int e = validate(d)) return e;
    TapPlan t;
    if (int e = build_plan(d, 0, &t)) return e;
    plan_to_params(t, p);
    p->B = d->batch;
    p->inner = d->inner;
    const int in_full = d->l_in * d->inner;
    p->in_valid = d->in_valid ? d->in_valid : in_full;
    p->in_row = p->in_valid;          // memory row stride of x
    p->out_row = d->l_out * d->inner;
    p->out_valid = p->out_row;
    p->ncols = t.rows * d->inner;
    p->in_act = d->act;
    p->in_slope = d->act_slope;
    p->epi_act = RH_ACT_NONE;
    p->epi_slope = 0.f;
    p->out_act = d->out_act;
    p->out_slope = d->out_slope;
    plan_to_x6(t, d->inner, p);
    return RH_OK;
}

int rh_conv_fill_dgrad(const rh_conv1d_desc* d, ConvP* p) {
    if (int e = validate(d)) return e;
    TapPlan t;
    if (int e = build_plan(d, 1, &t)) return e;
    plan_to_params(t, p);
    p->B = d->batch;
    p->inner = d->inner;
    p->in_row = d->l_out * d->inner;  // dy rows
    p->in_valid = p->in_row;
    const int x_full = d->l_in * d->inner;
    p->out_valid = d->in_valid ? d->in_valid : x_full;
    p->out_row = p->out_valid;        // memory row stride of x / dx
    p->ncols = t.rows * d->inner;
    p->in_act = RH_ACT_NONE;
    p->in_slope = 0.f;
    p->epi_act = d->act;
    p->epi_slope = d->act_slope;
    p->out_act = RH_ACT_NONE;      // the caller pre-multiplies dy by out_act'(y) (rh_act_bwd_f32)
    p->out_slope = 0.f;
    plan_to_x6(t, d->inner, p);
    return RH_OK;
}

extern "C" int64_t rh_conv1d_packed_floats(const rh_conv1d_desc* d, int which) {
    if (validate(d)) return -1;
    const int64_t M = which == 0 ? d->c_out : d->c_in;
    const int64_t C = which == 0 ? d->c_in : d->c_out;
    const int64_t n32 = (int64_t)d->kernel * C * round32((int)M);
    TapPlan t;
    if (build_plan(d, which, &t)) return -1;
    // + the bf16x6 section (3 x 2 bytes per K value, K padded to whole steps) for the geometries conv_x6.hip takes
    const long units = x6_units(t, x6_mode_of(t, d->inner), nullptr, d->inner);
    return (units > 0 && units * 4 < 0x7fffffffl) ? n32 + units * 4 + 4 : n32;      // (+ the range record)
}

extern "C" int rh_conv1d_pack_f32(const rh_conv1d_desc* d, const float* w, float* wp_fwd,
                                  float* wp_bwd, rh_stream_t stream) {
    if (int e = validate(d)) return e;
    RH_REQUIRE(w, RH_ERR_INVALID, "conv1d_pack: null weight");
    return pack_both(d, w, nullptr, wp_fwd, wp_bwd, (hipStream_t)stream);
}

int rh_weight_norm_scales(const float* v, const float* g, int64_t rows, int64_t cols, float* norms, float* scale,
                          hipStream_t stream);

// ---- batched weight preparation: every weight-normalised conv of a model in TWO launches ------------
// (per-layer launches of weight_norm_scales + pack cost ~1.2 ms of a 21 ms v2 step: ~110 tiny kernels)
struct PrepItem {
    const float* v;
    const float* g;
    float* norms;
    float* scale;
    PackP fwd, bwd;          // .w = v, .scale = scale, .wp = persistent packed buffers
    long rows, cols;
    long row_begin;          // prefix over items (rows)
    long blk_begin;          // prefix over items (32 x 32 pack tiles)
};

namespace {

// Which item does row / pack tile `key` belong to?  The prefix sums live a SECOND time in two compact tables behind the n items
// (rh_prep_link: long row_begin[n], long blk_begin[n]): the binary search then walks ~0.5 KB that stays in the scalar cache,
// instead of one ~3 KB PrepItem per probe -- six dependent cache misses at the head of every one of ~35 k tiny workgroups were
// most of the two kernels' time (round 6).
__device__ __forceinline__ int find_item(const PrepItem* items, int n, long key, bool rows) {
    const long* tab = reinterpret_cast<const long*>(items + n) + (rows ? 0 : n);
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid] <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Range record of every layer from the per-row maxima prep_scales_kernel left behind `scale` (scale[rows + r] = max |w| of row
// r, scale[2 rows + r] = its sum |w|): one workgroup per item, plain stores into both packed copies' records.  (Round 6 first
// had every row's workgroup atomicMax into the records: ~70 k same-line atomics made prep_scales_kernel 52 -> 120 us.)
__global__ __launch_bounds__(256) void prep_range_kernel(const PrepItem* __restrict__ items, int n) {
    __shared__ float red[8];
    const PrepItem& p = items[blockIdx.x];
    float mx = 0.f, l1 = 0.f;
    for (long r = threadIdx.x; r < p.rows; r += 256) { mx = fmaxf(mx, p.scale[p.rows + r]); l1 = fmaxf(l1, p.scale[2 * p.rows + r]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); l1 = fmaxf(l1, __shfl_down(l1, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; red[4 + (threadIdx.x >> 6)] = l1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        const unsigned l = __float_as_uint(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        for (unsigned* r : {p.fwd.range, p.bwd.range})
            if (r) { r[0] = m; r[1] = l; r[2] = 0u; r[3] = 0u; }
    }
}

__global__ __launch_bounds__(256) void prep_scales_kernel(const PrepItem* __restrict__ items, int n) {
    __shared__ float red[12];
    const int it = find_item(items, n, blockIdx.x, true);
    const PrepItem& p = items[it];
    const long r = (long)blockIdx.x - p.row_begin;
    const float* vr = p.v + r * p.cols;
    float s = 0.f, mx = 0.f, sm = 0.f;
    // (per-thread order of the sum of squares: element e, e + 256, ... as ever -- the norms keep their bits; the loads go four
    // rounds at a time so that a row of a few KB is not one dependent load per 1 KB)
    long e = threadIdx.x;
    for (; e + 768 < p.cols; e += 1024) {
        const float a0 = vr[e], a1 = vr[e + 256], a2 = vr[e + 512], a3 = vr[e + 768];
        s += a0 * a0; s += a1 * a1; s += a2 * a2; s += a3 * a3;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(a0), fabsf(a1))), fmaxf(fabsf(a2), fabsf(a3)));
        sm += fabsf(a0); sm += fabsf(a1); sm += fabsf(a2); sm += fabsf(a3);
    }
    for (; e < p.cols; e += 256) {
        const float a = vr[e];
        s += a * a;
        mx = fmaxf(mx, fabsf(a));
        sm += fabsf(a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
        sm += __shfl_down(sm, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = mx; red[8 + (threadIdx.x >> 6)] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        const float sc = p.g[r] / norm;
        p.norms[r] = norm;
        p.scale[r] = sc;
        // per-row range statistics (pack_range_kernel's arithmetic), reduced per layer by prep_range_kernel
        const float m = fabsf(sc) * fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        const float l1 = fabsf(sc) * ((red[8] + red[9]) + (red[10] + red[11]));
        p.scale[p.rows + r] = m;
        p.scale[2 * p.rows + r] = l1;
    }
}

__global__ __launch_bounds__(256) void prep_pack_kernel(const PrepItem* __restrict__ items, int n) {
    // one tile per workgroup: the kernel lives off many short workgroups in flight (four consecutive tiles per
    // workgroup, one descriptor search for all of them: 314 -> 428 us for the v2 model)
    __shared__ float lds[kPackSlots * 32 * 33];
    const int it = find_item(items, n, blockIdx.x, false);
    const PrepItem& p = items[it];
    const long t = (long)blockIdx.x - p.blk_begin;
    const long nf = pack_tiles(p.fwd);
    if (t < nf) pack_tile(p.fwd, t, lds);
    else if (t - nf < pack_tiles(p.bwd)) pack_tile(p.bwd, t - nf, lds);
}

}  // namespace

extern "C" int64_t rh_prep_item_bytes(void) { return (int64_t)sizeof(PrepItem); }
// bytes of the whole item array for n items: the items + the two compact prefix tables rh_prep_link writes behind them
extern "C" int64_t rh_prep_array_bytes(int32_t n) { return (int64_t)n * (int64_t)sizeof(PrepItem) + 2 * (int64_t)n * (int64_t)sizeof(long); }

extern "C" int rh_prep_fill_item(const rh_conv1d_desc* d, const float* v, const float* g, float* norms, float* scale,
                                 float* wp_fwd, float* wp_bwd, void* item) {
    if (int e = validate(d)) return e;
    RH_REQUIRE(v && g && norms && scale && wp_fwd && item, RH_ERR_INVALID, "prep_fill_item: null pointer");
    PrepItem* p = (PrepItem*)item;
    *p = PrepItem{};
    p->v = v; p->g = g; p->norms = norms; p->scale = scale;
    p->rows = d->transposed ? d->c_in : d->c_out;
    p->cols = (long)(d->transposed ? d->c_out : d->c_in) * d->kernel;
    if (int e = fill_pack(d, 0, v, scale, wp_fwd, &p->fwd)) return e;
    if (int e = fill_pack(d, 1, v, scale, wp_bwd, &p->bwd)) return e;
    return RH_OK;
}

extern "C" int rh_prep_link(void* items, int32_t n, int64_t* total_rows, int64_t* total_blocks) {
    RH_REQUIRE(items && total_rows && total_blocks && n >= 0, RH_ERR_INVALID, "prep_link: bad arguments");
    PrepItem* p = (PrepItem*)items;
    long rows = 0, blks = 0;
    for (int i = 0; i < n; ++i) {
        p[i].row_begin = rows;
        p[i].blk_begin = blks;
        rows += p[i].rows;
        for (const PackP* q : {&p[i].fwd, &p[i].bwd}) blks += pack_tiles(*q);
    }
    long* tab = reinterpret_cast<long*>(p + n);           // (the caller allocated rh_prep_array_bytes(n))
    for (int i = 0; i < n; ++i) { tab[i] = p[i].row_begin; tab[n + i] = p[i].blk_begin; }
    *total_rows = rows;
    *total_blocks = blks;
    return RH_OK;
}

extern "C" int rh_prep_run_f32(const void* items_dev, int32_t n, int64_t total_rows, int64_t total_blocks,
                               rh_stream_t stream) {
    RH_REQUIRE(items_dev && n > 0, RH_ERR_INVALID, "prep_run: bad arguments");
    if (total_rows > 0) {
        hipLaunchKernelGGL(prep_scales_kernel, dim3((unsigned)total_rows), dim3(256), 0, (hipStream_t)stream,
                           (const PrepItem*)items_dev, n);
        if (int e = rh_check_launch("prep_scales")) return e;
        hipLaunchKernelGGL(prep_range_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const PrepItem*)items_dev, n);
        if (int e = rh_check_launch("prep_range")) return e;
    }
    if (total_blocks > 0) {
        hipLaunchKernelGGL(prep_pack_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                           (const PrepItem*)items_dev, n);
        return rh_check_launch("prep_pack");
    }
    return RH_OK;
}

extern "C" int rh_conv1d_pack_wn_f32(const rh_conv1d_desc* d, const float* v, const float* g, float* norms,
                                     float* scale, float* wp_fwd, float* wp_bwd, rh_stream_t stream) {
    if (int e = validate(d)) return e;
    RH_REQUIRE(v && g && norms && scale, RH_ERR_INVALID, "conv1d_pack_wn: null pointer");
    const int64_t rows = d->transposed ? d->c_in : d->c_out;
    const int64_t cols = (int64_t)(d->transposed ? d->c_out : d->c_in) * d->kernel;
    if (int e = rh_weight_norm_scales(v, g, rows, cols, norms, scale, (hipStream_t)stream)) return e;
    return pack_both(d, v, scale, wp_fwd, wp_bwd, (hipStream_t)stream);
}

extern "C" int64_t rh_conv1d_fwd_workspace_bytes(const rh_conv1d_desc* d) {
    ConvP p{};
    if (rh_conv_fill_fwd(d, &p)) return -1;
    return rh_conv_splitk_workspace(p);
}

extern "C" int64_t rh_conv1d_bwd_data_workspace_bytes(const rh_conv1d_desc* d) {
    ConvP p{};
    if (rh_conv_fill_dgrad(d, &p)) return -1;
    p.epi_act = d->act;
    return rh_conv_splitk_workspace(p);
}

extern "C" int rh_conv1d_kernel_family(const rh_conv1d_desc* d, int which, int has_bias, int has_add) {
    ConvP p{};
    if (int e = which == 0 ? rh_conv_fill_fwd(d, &p) : rh_conv_fill_dgrad(d, &p)) return e;
    if (which == 0 ? rh_smallc_fwd_eligible(d, has_add != 0) : rh_smallc_dgrad_eligible(d, has_add != 0)) return 2;
    alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};     // only tested against NULL / alignment by the planner
    p.in = dummy; p.wp = dummy; p.wq = reinterpret_cast<const unsigned*>(dummy);
    p.bias = has_bias ? dummy : nullptr;
    p.add = has_add ? dummy : nullptr;
    p.mul_src = (which == 1 && d->act != RH_ACT_NONE) ? dummy : nullptr;
    if (p.B <= 0 || p.ncols <= 0 || p.M <= 0 || !rh_conv_dma_eligible(p)) return 0;
    return (rh_conv_x6_workspace(p) >= 0 || rh_conv_c2x_query(p)) ? 1 : 0;
}

// Diagnostics: out8 = {family (rh_conv1d_kernel_family), tm, tn, wm, K slices, fragment-layout input stride, virtual rows, workgroups}
// of the launch rh_conv1d_fwd_f32 (which = 0) / rh_conv1d_bwd_data_f32 (which = 1) would issue for this geometry.
extern "C" int rh_conv1d_plan_info(const rh_conv1d_desc* d, int which, int has_bias, int has_add, int32_t* out8) {
    RH_REQUIRE(out8, RH_ERR_INVALID, "conv1d_plan_info: null output");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    ConvP p{};
    if (int e = which == 0 ? rh_conv_fill_fwd(d, &p) : rh_conv_fill_dgrad(d, &p)) return e;
    if (which == 0 ? rh_smallc_fwd_eligible(d, has_add != 0) : rh_smallc_dgrad_eligible(d, has_add != 0)) { out8[0] = 2; return RH_OK; }
    alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
    p.in = dummy; p.wp = dummy; p.wq = reinterpret_cast<const unsigned*>(dummy);
    p.bias = has_bias ? dummy : nullptr;
    p.add = has_add ? dummy : nullptr;
    p.mul_src = (which == 1 && d->act != RH_ACT_NONE) ? dummy : nullptr;
    if (p.B <= 0 || p.ncols <= 0 || p.M <= 0 || !rh_conv_dma_eligible(p)) return RH_OK;
    int q[7];
    if (rh_conv_x6_plan_query(p, q)) {
        out8[0] = 1;
        for (int i = 0; i < 7; ++i) out8[i + 1] = q[i];
    }
    return RH_OK;
}

extern "C" int rh_conv1d_fwd_f32(const rh_conv1d_desc* d, const float* x, const float* wp_fwd,
                                 const float* bias, const float* snake_alpha, const float* residual,
                                 float* y, void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    ConvP p{};
    rh_take_ranges(nullptr, &p.in_range, &p.out_range, nullptr);       // consumed by this call whatever happens below
    const unsigned* const in_range = p.in_range;
    unsigned* const out_range = p.out_range;
    if (int e = rh_conv_fill_fwd(d, &p)) return e;
    p.in_range = in_range; p.out_range = out_range;
    if (d->batch == 0 || d->l_out == 0) return RH_OK;
    RH_REQUIRE(x && wp_fwd && y, RH_ERR_INVALID, "conv1d_fwd: null pointer");
    RH_REQUIRE(d->act != RH_ACT_SNAKE || snake_alpha, RH_ERR_INVALID, "conv1d_fwd: snake needs alpha");
    if (rh_smallc_fwd_eligible(d, residual != nullptr)) {
        return rh_smallc_fwd(d, x, wp_fwd, bias, y, (hipStream_t)stream, p.out_range);     // (publishes max |y| itself)
    }
    p.in = x; p.wp = wp_fwd; p.out = y; p.bias = bias; p.add = residual; p.mul_src = nullptr;
    p.wq = reinterpret_cast<const unsigned*>(wp_fwd + p.x6_wofs);
    p.in_alpha = snake_alpha; p.mul_alpha = nullptr;
    return rh_conv_launch(p, (hipStream_t)stream, d->transposed ? "conv_transpose1d_fwd" : "conv1d_fwd", workspace,
                          workspace_bytes);
}

extern "C" int rh_conv1d_bwd_data_f32(const rh_conv1d_desc* d, const float* dy, const float* wp_bwd,
                                      const float* x, const float* snake_alpha, const float* add,
                                      float* dx, void* workspace, int64_t workspace_bytes,
                                      rh_stream_t stream) {
    ConvP p{};
    rh_take_ranges(nullptr, &p.in_range, &p.out_range, nullptr);
    const unsigned* const in_range = p.in_range;
    unsigned* const out_range = p.out_range;
    if (int e = rh_conv_fill_dgrad(d, &p)) return e;
    p.in_range = in_range; p.out_range = out_range;
    if (d->batch == 0 || d->l_in == 0) return RH_OK;
    RH_REQUIRE(dy && wp_bwd && dx, RH_ERR_INVALID, "conv1d_bwd_data: null pointer");
    RH_REQUIRE(d->act == RH_ACT_NONE || x, RH_ERR_INVALID, "conv1d_bwd_data: act needs the forward input");
    RH_REQUIRE(d->act != RH_ACT_SNAKE || snake_alpha, RH_ERR_INVALID, "conv1d_bwd_data: snake needs alpha");
    if (rh_smallc_dgrad_eligible(d, add != nullptr)) {
        TapPlan t;
        if (int e = build_plan(d, 1, &t)) return e;
        int slot_of_tap[kMaxTaps];
        for (int i = 0; i < t.nslots; ++i) slot_of_tap[t.kk[i]] = i;
        if (int e = rh_smallc_dgrad(d, dy, wp_bwd, slot_of_tap, dx, (hipStream_t)stream)) return e;
        p.out = dx;
        return rh_range_after(p, (hipStream_t)stream);
    }
    p.in = dy; p.wp = wp_bwd; p.out = dx; p.bias = nullptr; p.add = add;
    p.wq = reinterpret_cast<const unsigned*>(wp_bwd + p.x6_wofs);
    p.mul_src = d->act == RH_ACT_NONE ? nullptr : x;
    p.in_alpha = nullptr; p.mul_alpha = snake_alpha;
    return rh_conv_launch(p, (hipStream_t)stream, d->transposed ? "conv_transpose1d_bwd_data" : "conv1d_bwd_data",
                          workspace, workspace_bytes);
}

extern "C" int64_t rh_conv1d_workspace_bytes(const rh_conv1d_desc* d) {
    if (validate(d)) return -1;
    return rh_wgrad_workspace(d);
}

extern "C" int rh_conv1d_bwd_weight_f32(const rh_conv1d_desc* d, const float* dy, const float* x,
                                        const float* snake_alpha, float* dw, float* dbias,
                                        void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    if (int e = validate(d)) return e;
    RH_REQUIRE(dw && (d->batch == 0 || (dy && x)), RH_ERR_INVALID, "conv1d_bwd_weight: null pointer");
    RH_REQUIRE(d->act != RH_ACT_SNAKE || snake_alpha, RH_ERR_INVALID, "conv1d_bwd_weight: snake needs alpha");
    return rh_wgrad_run(d, dy, x, snake_alpha, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int rh_conv1d_bwd_weight_wn_f32(const rh_conv1d_desc* d, const float* dy, const float* x, const float* snake_alpha,
                                           const float* v, const float* g, const float* norms, float* dw_scratch, float* dv,
                                           float* dg, float* dbias, void* workspace, int64_t workspace_bytes, rh_stream_t stream) {
    if (int e = validate(d)) return e;
    RH_REQUIRE(dw_scratch && v && g && norms && dv && dg && (d->batch == 0 || (dy && x)), RH_ERR_INVALID,
               "conv1d_bwd_weight_wn: null pointer");
    RH_REQUIRE(d->act != RH_ACT_SNAKE || snake_alpha, RH_ERR_INVALID, "conv1d_bwd_weight_wn: snake needs alpha");
    const RhWnTail tail{v, g, norms, dv, dg};
    return rh_wgrad_run(d, dy, x, snake_alpha, dw_scratch, dbias, workspace, workspace_bytes, (hipStream_t)stream, &tail);
}
