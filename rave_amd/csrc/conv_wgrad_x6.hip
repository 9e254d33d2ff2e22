// Weight gradient of the 1-D convolutions, exact f32 on the bf16 matrix cores ("x6", see conv_x6_kernel.inc for the
// numerics: 3-way exact bf16 split of both operands, six v_mfma_f32_32x32x16_bf16 per product block, smallest terms
// first):
//   out[z][m][c*T + t] = sum_{(b,n) in K slice z} actR(R[b][m][n]) * actS(S[b][c][n*is + off[t]])
// (conv_params.hpp: WgradP; R = dy, S = x for Conv1d, the roles swap for ConvTranspose1d).
//
// GEMM view: rows m, columns (c, t), reduction over POSITIONS -- the contiguous axis of both operands.  A 16-deep MFMA
// k block is 16 consecutive positions of one batch item, and lane (j, g)'s fragment is 8 consecutive samples of one
// row: no transposition anywhere, every thread converts whole fragments.  Per step (16 positions):
//   * every thread owns (row, octet) tasks of the R tile and (column, octet) tasks of the S tile: it loads the 8 samples
//     straight from HBM/L2 one step ahead (buffer loads: padding and ragged tails read 0.0; consecutive lanes take
//     consecutive octets of one row, i.e. whole 128-byte lines), applies LeakyReLU, splits exactly and writes three
//     16-byte fragments [k block][g][piece][row] to LDS;
//   * a step is two k blocks (32 positions): wave tile 32*TM x 64, fragments by ds_read_b128, 24*TM MFMAs per step and
//     wave; one LDS stage (the next step's samples wait in registers), two barriers per step.
// Both operands need the conversion (the forward kernel gets its weights pre-split), ~4 VALU instructions per MFMA, so
// this kernel lives off the overlap of one workgroup's conversion with the other's MFMAs (2 workgroups per CU).
// K is split over (batch, position) ranges; the partial sums are combined in slice order by reduce_partials_kernel.
//
// Measured (C = 192 k = 3 layer, 83 us; ablation builds): MFMA + barriers alone 36 us of loop, loads + conversion alone
// 33 us, together 58 us -- the phases do not overlap, and no schedule made them: wave priorities (s_setprio by wave slot
// / block parity) changed nothing; a barrier ping-pong of two 4-wave teams in one workgroup (one multiplies while the
// other converts) was 1.2 ... 2x slower; converting the next step INSIDE the MFMA sequence of the same wave (two LDS
// stages, 16-position steps, sched_group_barrier interleave: 1 MFMA + 6 VALU) gained 3 % on long rows and lost up to
// 40 % on short ones.  The probe tools/probe/mfma_valu_overlap.hip shows why: on this SIMD the time of a VALU
// instruction ADDS to the MFMA time whichever wave issues it (MFMA alone 34-40 cycles, + 6 VALU = 46, two such waves
// 90 per pair).  What pays is fewer VALU instructions per sample.
#include <cstdlib>
#include <mutex>
#include "conv_params.hpp"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;

struct Wx6P {
    const float* R;
    const float* S;
    float* out;                 // [Z][M][N] partials (or dw itself when Z == 1)
    float* rsum;                // [Z][M] row sums of R over the K slice = partial bias gradients (R = dy), or null
    int B, M, C, T, N;          // N = C * T
    int r_row, s_row, s_valid, is;
    float r_slope, s_slope;     // LeakyReLU slope of the operand's activation; 1 = none
    int steps_per_b;            // ceil(r_row / (16 * kKS))
    int total_steps, steps_per_z;
    unsigned r_bytes, s_bytes;
    int minoff, maxoff;
    int p8, cpb, chmax;         // plane mode (PL): octets of a channel image, channel pitch in bytes, channel images reserved
    const unsigned* r_range;    // range slots of R and S (f16 build: common.hpp)
    const unsigned* s_range;
    int off[kMaxTaps];
};

typedef u32x4 u32x4_u __attribute__((aligned(2)));      // 16 bytes at any 2-byte alignment: ds_read_b128 serves it at 1.4x the aligned cost

// LeakyReLU (slope in [0, 1]; 1 = none) + split of 8 samples -> kX6P 16-byte fragments.  f16 build: the samples are scaled
// by `sc` (power of two, from the operand's range slot) and split hi / lo with packed converts, ~5 VALU per sample; bf16 build:
// exact 3-way truncation split, ~7.5.  VALU work is NOT free next to the matrix cores on this chip
// (tools/probe/mfma_valu_overlap.hip) -- every instruction saved here is matrix time.
__device__ __forceinline__ void emit(const float (&v)[8], float slope, float sc, u32x4* dst, int piece_stride) {
#if RH_X6_F16
    u32x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = v[2 * k] * sc, b = v[2 * k + 1] * sc;
        a = rh_max1(a, a * slope);
        b = rh_max1(b, b * slope);
        const rh_h2 h = rh_h2_split(a, b);
        hi[k] = h.hi; lo[k] = h.lo;
    }
    dst[0] = hi;
    dst[piece_stride] = lo;
#else
    unsigned h[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = rh_max1(v[i], v[i] * slope);
        rh_bf3_split(x, h[0][i], h[1][i], h[2][i]);
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = __builtin_amdgcn_perm(h[s3][2 * k + 1], h[s3][2 * k], 0x07060302u);   // high halves
        dst[s3 * piece_stride] = pk;
    }
#endif
}

constexpr int kKS = 2;     // MFMA k blocks (16 positions each) per step

// AV: rows of R are 16-byte aligned -> the 8 samples of an R task are two 16-byte loads instead of eight 4-byte ones
// PART: the last column tile holds whole 32-column MFMA tiles past the end of the weight tensor -- their columns are
// neither converted nor multiplied (a separate instantiation: the two uniform branches cost 14 VGPRs, which would take the
// 64-row variants from three workgroups per CU to two)
// PL ("planes", round 5): the S operand of a stride-1 layer with several taps is staged ONCE PER POSITION -- per step the
// 32 + (maxoff - minoff) positions of every channel the column tile touches, converted into bf16 ELEMENT planes
// [piece][channel][position] -- and the fragment of column (c, t) is the 16-byte LDS read at channel c's image + the tap's
// element offset: any 2-byte alignment, which ds_read_b128 serves at 1.4x the aligned cost (tools/probe/lds_unaligned.hip; the
// technique of wgrad2d_x6.hip).  A tap is an address offset; nothing is converted per tap: C = 96, k = 3 converts
// (96 + 91) x 32 samples per step instead of (96 + 256) x 32, dilation 9 (96 + 134) x 32; C >= 192 (192 + 46) instead of
// (192 + 128).  Same products in the same order: bit-identical weight gradients (tests/test_gpu_dispatch.py).
// MEASURED SLOWER and therefore opt-in (profiles/round5_negative_wgrad_planes.txt): C = 96 k = 3 105.8 us against 104.2 (dilation 9:
// 115.4 / 103.2), C = 192 82.4 / 75.2, C = 384 89.8 / 67.9, C = 768 95.7 / 70.8, the step 10.13 ms against 9.93 on the same box.  The
// loop was not conversion-bound: the unaligned fragment reads are inline asm, so each column tile waits for its three
// fragments with nothing else of the wave in flight (the aligned path's fragment loads are scheduled by the compiler across the
// MFMA sequence), and the 64-row variants need 170-190 registers: two workgroups per CU instead of three (forced under 168 they
// spill 3-10).  What it would take: the reads of
// column tile 1 / of the next k block issued under the current MFMAs (s_waitcnt lgkmcnt(3)), i.e. six more live fragments.
template <int TM, int WM, int WN, bool AV, bool PART, bool PL>
__global__ __launch_bounds__(256, 2) void wgrad_x6_kernel(const Wx6P p) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * TM * WM, BN = 64 * WN;
    // fragments of ONE k block: [g][piece][rows], with 4 fragments of padding per g block.  The OCT = 4 lanes that convert
    // the four octets of one row write to (k block, g) = (0,0), (0,1), (1,0), (1,1): without the padding those blocks are
    // 3 * BM fragments = a multiple of 256 bytes apart and every ds_write_b128 was a 4-way bank conflict (PMC:
    // SQ_LDS_BANK_CONFLICT 60 % of SQ_LDS_IDX_ACTIVE); with it the four lanes land 64 bytes apart.
    constexpr int A_GS = kX6P * BM + 4, B_GS = kX6P * BN + 4;             // g stride
    constexpr int A_UNITS = 2 * A_GS, B_UNITS = 2 * B_GS;                 // k-block stride
    constexpr int OCT = 2 * kKS;                                          // 8-sample octets per row and step
    constexpr int NA = (OCT * BM + 255) / 256;                                // tasks per thread and step
    // PL: (channel, octet) tasks -- at most BN / 2 + 2 channel images of <= 7 octets
    constexpr int NB = PL ? ((BN / 2 + 2) * 7 + 255) / 256 : (OCT * BN + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const a_st = reinterpret_cast<u32x4*>(smem_raw);              // [kKS][g][piece][BM]
    u32x4* const b_st = a_st + kKS * A_UNITS;                             // [kKS][g][piece][BN]   (PL: [piece][channel][position] bf16)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;

    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);
    const auto s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.S), 0, p.s_bytes, 0x00020000);
#if RH_X6_F16
    int inv_r, inv_s;
    const float rsc = __uint_as_float(rh_x6_scale_bits(rh_range_max(p.r_range), &inv_r));
    const float ssc = __uint_as_float(rh_x6_scale_bits(rh_range_max(p.s_range), &inv_s));
    const float osc = __uint_as_float(rh_x6_unscale_bits(inv_r, inv_s));
#else
    const float rsc = 1.f, ssc = 1.f, osc = 1.f;
#endif

    // ---- tasks (step-invariant).  A task = 8 consecutive samples of one row; consecutive lanes take consecutive
    // octets of the SAME row (OCT lanes x 32 bytes = one 128-byte line per row), so that a load instruction touches
    // 64 / OCT rows instead of 64.
    unsigned aoff[NA], boff[NB];
    int adst[NA], bdst[NB], apos[NA], bp0[NB];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int u = tid + 256 * q;
        const int o = u % OCT, m = u / OCT;
        const bool ok = m < BM && m0 + m < p.M;
        adst[q] = m < BM ? (o >> 1) * A_UNITS + (o & 1) * A_GS + m : -1;
        apos[q] = 8 * o;
        aoff[q] = ok ? (unsigned)(((m0 + m) * p.r_row + 8 * o) * 4) : kOOB;
    }
    const int c_lo = PL ? n0 / p.T : 0;                                                  // first channel of this column tile
    const int n_ch = PL ? min(p.N - 1, n0 + BN - 1) / p.T - c_lo + 1 : 0;                // channel images it needs
    const unsigned b_piece = PL ? (unsigned)(p.chmax * p.cpb) : 0u;                      // bytes between the pieces of the image
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int u = tid + 256 * q;
        if constexpr (PL) {
            const int ch = u / p.p8, o = u - ch * p.p8;
            const bool ok = ch < n_ch;
            bdst[q] = ok ? ch * p.cpb + o * 16 : -1;                                     // byte offset inside a piece
            bp0[q] = p.minoff + 8 * o;                                                   // position of sample 0 relative to n
            boff[q] = ok ? (unsigned)(((c_lo + ch) * p.s_row + bp0[q]) * 4) : kOOB;
        } else {
            const int o = u % OCT, col = u / OCT;
            const int cc = (n0 + col) / p.T, t = (n0 + col) - cc * p.T;
            const bool ok = col < BN && n0 + col < p.N;
            // columns past the end of the weight tensor convert nothing (their LDS slots keep whatever they hold: a column of
            // the B operand only reaches its own output column, which is never stored)
            bdst[q] = (PART ? ok : col < BN) ? (o >> 1) * B_UNITS + (o & 1) * B_GS + col : -1;
            bp0[q] = 8 * o * p.is + (ok ? p.off[t] : 0);                  // position of sample 0 relative to n * is
            boff[q] = ok ? (unsigned)((cc * p.s_row + bp0[q]) * 4) : kOOB;
        }
    }
    // PL: byte offset of this lane's two columns inside a piece of the image: channel image + element offset of the tap
    unsigned bcolb[2] = {0u, 0u};
    if constexpr (PL) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + wn * 64 + tn * 32 + j;
            if (n < p.N) {
                const int cc = n / p.T, t = n - cc * p.T;
                bcolb[tn] = (unsigned)((cc - c_lo) * p.cpb + (p.off[t] - p.minoff) * 2);
            }
        }
    }

    f32x16 acc[TM][2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    constexpr int SPAN = 16 * kKS;                                        // positions per step
    float ra[NA][8], rb[NB][8];
    auto load = [&](int st) {
        const int b = st / p.steps_per_b;
        const int n = (st - b * p.steps_per_b) * SPAN;
        const unsigned rs = (unsigned)((b * p.M * p.r_row + n) * 4);
        const unsigned ss = (unsigned)((b * p.C * p.s_row + n * p.is) * 4);
        const bool r_tail = n + SPAN > p.r_row;                                     // uniform
        const bool s_edge = PL ? (n + p.minoff < 0 || n + p.minoff + 8 * p.p8 > p.s_valid || r_tail)
                               : (n * p.is + p.minoff < 0 || (n + SPAN - 1) * p.is + p.maxoff >= p.s_valid || r_tail);
        // the whole element offset goes into the per-lane operand (the bounds check must see it: a tap offset alone
        // can be negative for a valid sample), nothing into the scalar offset
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const unsigned base = aoff[q] == kOOB ? kOOB : aoff[q] + rs;
            if constexpr (AV) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    unsigned off = base == kOOB ? kOOB : base + 16u * h;
                    if (r_tail) off = n + apos[q] + 4 * h < p.r_row ? off : kOOB;      // r_row % 4 == 0: all or nothing
                    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0));
#pragma unroll
                    for (int i = 0; i < 4; ++i) ra[q][4 * h + i] = __uint_as_float(v[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned off = base == kOOB ? kOOB : base + 4u * i;
                    if (r_tail) off = n + apos[q] + i < p.r_row ? off : kOOB;
                    ra[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, off, 0, 0));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const unsigned base = boff[q] == kOOB ? kOOB : boff[q] + ss;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned off = base == kOOB ? kOOB : base + 4u * (unsigned)(i * p.is);
                if (s_edge) {
                    const int pos = n * p.is + bp0[q] + i * p.is;
                    off = (pos >= 0 && pos < p.s_valid) ? off : kOOB;
                }
                rb[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rsrc, off, 0, 0));
            }
        }
    };
    // bias gradient = row sums of dy: the first column tile of every row tile adds up the samples it converts anyway
    // (one pass over dy for weight AND bias gradient; the separate bias kernel re-read every dy tensor)
    const bool want_rsum = p.rsum != nullptr && blockIdx.x == 0;      // uniform
    float rs_acc[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) rs_acc[q] = 0.f;
    auto convert = [&]() {
        if (want_rsum) {
#pragma unroll
            for (int q = 0; q < NA; ++q)
                rs_acc[q] += ((ra[q][0] + ra[q][1]) + (ra[q][2] + ra[q][3])) + ((ra[q][4] + ra[q][5]) + (ra[q][6] + ra[q][7]));
        }
#pragma unroll
        for (int q = 0; q < NA; ++q)
            if (adst[q] >= 0) emit(ra[q], p.r_slope, rsc, a_st + adst[q], BM);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (bdst[q] < 0) continue;
            if constexpr (PL) emit(rb[q], p.s_slope, ssc, reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(b_st) + bdst[q]), (int)(b_piece >> 4));
            else emit(rb[q], p.s_slope, ssc, b_st + bdst[q], BN);
        }
    };

    const int st0 = z * p.steps_per_z;
    const int nst = min(p.steps_per_z, p.total_steps - st0);
    if (nst > 0) {
        load(st0);
        convert();
    }
    __syncthreads();
    const int arow = g * A_GS + wm * TM * 32 + j;
    const int bcol = g * B_GS + wn * 64 + j;
    // one LDS stage: the next step's samples wait in registers while the matrix cores work on this step's fragments
    // (the other workgroup of the CU runs its MFMAs while this one converts)
    // How many of this wave's two 32-column tiles hold columns of the weight tensor at all (wave-uniform).  The 96-row layers
    // run 96 x 256 workgroup tiles over N = 288 (k = 3) or 96 (k = 1) columns: the second tile of the former holds 32 live
    // columns, the only tile of the latter 96 -- multiplying (and converting) the dead ones cost those layers 30-40 %.
    const int live_tn = !PART ? 2 : (n0 + wn * 64 >= p.N ? 0 : (n0 + wn * 64 + 32 >= p.N ? 1 : 2));
    const bool live1 = live_tn >= 1, live2 = live_tn == 2;            // scalar (wn, n0, N are): uniform branches below
    for (int s = 0; s < nst; ++s) {
        const bool more = s + 1 < nst;
        if (more) load(st0 + s + 1);
        if (live1) {
#pragma unroll
            for (int kb = 0; kb < kKS; ++kb) {
                const u32x4* al = a_st + kb * A_UNITS + arow;
                const u32x4* bl = b_st + kb * B_UNITS + bcol;
                constexpr int SA[RH_X6_NPROD] = RH_X6_SA, SB[RH_X6_NPROD] = RH_X6_SB;     // smallest terms first
                rh_x6_frag afr[TM][kX6P];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int s3 = 0; s3 < kX6P; ++s3) afr[tm][s3] = __builtin_bit_cast(rh_x6_frag, al[s3 * BM + tm * 32]);
                if constexpr (PL) {
                    // the 8 positions 16 kb + 8 g .. + 7 of the column's channel image, shifted by its tap: a 16-byte LDS read at
                    // 2-byte alignment.  Round 6: a plain C++ load through a 2-byte-aligned vector type -- the compiler emits
                    // ds_read_b128 for it on gfx950, counts it (lgkmcnt) and schedules it like the aligned path's reads (round 5
                    // issued it as inline asm and waited for every column tile with nothing else in flight).
                    const unsigned char* bimg = reinterpret_cast<const unsigned char*>(b_st) + (kb * 16 + g * 8) * 2;
                    rh_x6_frag bfr[2][kX6P];
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int s3 = 0; s3 < kX6P; ++s3)
                            bfr[tn][s3] = __builtin_bit_cast(rh_x6_frag, *reinterpret_cast<const u32x4_u*>(bimg + bcolb[tn] + (unsigned)s3 * b_piece));
#pragma unroll
                    for (int q = 0; q < RH_X6_NPROD; ++q)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            acc[tm][0] = RH_X6_MFMA(afr[tm][SA[q]], bfr[0][SB[q]], acc[tm][0]);
                            if (live2) acc[tm][1] = RH_X6_MFMA(afr[tm][SA[q]], bfr[1][SB[q]], acc[tm][1]);
                        }
                } else {
                    rh_x6_frag bfr[2][kX6P];
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)           // (a dead tile's slots hold stale data: read, never multiplied)
#pragma unroll
                        for (int s3 = 0; s3 < kX6P; ++s3) bfr[tn][s3] = __builtin_bit_cast(rh_x6_frag, bl[s3 * BN + tn * 32]);
#pragma unroll
                    for (int q = 0; q < RH_X6_NPROD; ++q)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            acc[tm][0] = RH_X6_MFMA(afr[tm][SA[q]], bfr[0][SB[q]], acc[tm][0]);
                            if (live2) acc[tm][1] = RH_X6_MFMA(afr[tm][SA[q]], bfr[1][SB[q]], acc[tm][1]);
                        }
                }
            }
        }
        __syncthreads();                 // every wave is done reading this step's fragments
        if (more) convert();
        __syncthreads();
    }

    if (want_rsum) {            // the OCT lanes of a row are neighbours: fixed-order butterfly, lane of octet 0 writes
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            float v = rs_acc[q];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            const int u = tid + 256 * q;
            const int m = u / OCT;
            if ((u % OCT) == 0 && m < BM && m0 + m < p.M) p.rsum[(long)z * p.M + m0 + m] = v;
        }
    }
    // ---- partial sums of this K slice
    float* __restrict__ outz = p.out + (long)z * p.M * p.N;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + j;
        if (col >= p.N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mb = m0 + (wm * TM + tm) * 32 + 4 * g;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) outz[(long)m * p.N + col] = RH_X6_F16 ? acc[tm][tn][r] * osc : acc[tm][tn][r];
            }
        }
    }
}

struct Wx6Plan {
    int tm, wm, rt, ct, Z, steps_per_z;
    int planes;              // the S operand staged once per position as bf16 element planes (PL)
    size_t lds;
};

bool plan_wx6(const WgradP& w, Wx6P* p, Wx6Plan* pl, const unsigned* r_range, const unsigned* s_range) {
    const char* e = getenv("RH_WGRAD_X6");        // read per call: the parity tests flip it at run time
    if (e && atoi(e) == 0) return false;
    if (RH_X6_F16 && (!r_range || !s_range)) return false;      // no range slots (rh_x6_set_ranges): f32-input MFMA kernels
    if (w.inner != 1 || w.T > kMaxTaps || w.B <= 0 || w.r_row <= 0) return false;
    if ((w.r_act != RH_ACT_NONE && w.r_act != RH_ACT_LEAKY) || (w.s_act != RH_ACT_NONE && w.s_act != RH_ACT_LEAKY)) return false;
    // the conversion applies LeakyReLU as max(x, slope x)
    if ((w.r_act == RH_ACT_LEAKY && !(w.r_slope >= 0.f && w.r_slope <= 1.f)) || (w.s_act == RH_ACT_LEAKY && !(w.s_slope >= 0.f && w.s_slope <= 1.f))) return false;
    if (w.M < 32 || (long)w.C * w.T < 64) return false;            // tiny GEMMs (first discriminator layers): f32 kernels
    // Measured per layer (profiles/round2_layer_table_b32.txt): 1.2x ... 1.6x over the f32-MFMA kernel (wgrad_dma_kernel)
    // wherever the weight tensor offers a few output tiles -- everything but the <= 96-row layers on long sequences
    // (C = 96 at 4096 positions: one to three tiles, so K is cut into hundreds of slices whose partial tiles cost more
    // than the matrix work; the f32 kernel's 96-column tiles and LDS-DMA row segments win there).
    static const int force = [] { const char* e2 = getenv("RH_WGRAD_X6_ALL"); return e2 ? atoi(e2) : 0; }();
    // (round 3: with conflict-free fragment writes and the cheaper conversion the 96-row layers moved over too -- C = 96 k = 3
    // 115 -> 100 us, k = 1 59 -> 56; the 32-row output layer stays: 94 us on the f32 kernel against 120 here)
    if (!force && w.M <= 64 && w.r_row > 1024) return false;
    const unsigned long long rb = 4ull * w.B * w.M * (unsigned long long)w.r_row;
    const unsigned long long sb = 4ull * w.B * w.C * (unsigned long long)w.s_row;
    if (rb >= 0x7fffffffull || sb >= 0x7fffffffull) return false;
    *p = Wx6P{};
    p->R = w.R; p->S = w.S;
    p->B = w.B; p->M = w.M; p->C = w.C; p->T = w.T; p->N = w.C * w.T;
    p->r_row = w.r_row; p->s_row = w.s_row; p->s_valid = w.s_valid; p->is = w.is;
    p->r_slope = w.r_act == RH_ACT_LEAKY ? w.r_slope : 1.f;
    p->s_slope = w.s_act == RH_ACT_LEAKY ? w.s_slope : 1.f;
    p->steps_per_b = rh_cdiv(w.r_row, 16 * kKS);
    p->total_steps = w.B * p->steps_per_b;
    p->r_bytes = (unsigned)rb; p->s_bytes = (unsigned)sb;
    p->minoff = w.minoff; p->maxoff = w.maxoff;
    p->r_range = r_range; p->s_range = s_range;
    for (int t = 0; t < w.T; ++t) p->off[t] = w.off[t];
    const int Mp = (w.M + 31) & ~31;
    // 96-row wave tiles (TM = 3: 216 VGPRs, two workgroups per CU) convert the least per MFMA, but 64-row ones (TM = 2:
    // 158 VGPRs, 50 KB of LDS) fit THREE workgroups per CU, and a third set of phases to interleave is worth more wherever
    // 128-row workgroup tiles divide the rows (measured per layer: C = 384 k = 3 76 -> 68 us, 384 -> 768 k = 8 111 -> 96,
    // C = 768 91 -> 85; M = 192 loses a quarter of the tile and stays on TM = 3)
    // -- and the reduction is long enough: pointwise layers (C = 384 / 768, k = 1) lose 5 ... 19 % with it
    pl->tm = (Mp % 128 == 0 && Mp >= 256 && w.T > 1) ? 2 : (Mp % 96 == 0 ? 3 : (Mp % 64 == 0 ? 2 : 1));
    static const int tm_env = [] { const char* e2 = getenv("RH_WGRAD_X6_TM"); return e2 ? atoi(e2) : 0; }();
    if (tm_env >= 1 && tm_env <= 3) pl->tm = tm_env;
    pl->wm = Mp >= 64 * pl->tm ? 2 : 1;
    const int BM = 32 * pl->tm * pl->wm, BN = 64 * (4 / pl->wm);
    pl->rt = rh_cdiv(w.M, BM);
    pl->ct = rh_cdiv(p->N, BN);
    // plane mode (round 5, RH_WGRAD_X6_PLANES=1): stride-1 layers with several taps whose reach fits the image (<= 7 octets
    // per channel: the k = 3 units up to dilation 9 -- reach 18 --, the k = 7 stem).  Default: the per-tap conversion.
    {
        // Round 6 (compiler-visible unaligned reads, two f16 pieces): faster where the rows are few and the conversion dominates
        // -- M <= 96: C = 96 k = 3 104 -> 89 us, stem / output layer +3 % -- slower from C = 192 on (C = 384 k = 3 59 -> 79 us):
        // on by default for M <= 96, RH_WGRAD_X6_PLANES = 1 / 0 forces it everywhere / nowhere (read per call: tests).
        const char* pe = getenv("RH_WGRAD_X6_PLANES");
        const bool on = pe ? pe[0] == '1' : Mp <= 96;
        const int reach = w.maxoff - w.minoff;
        const int p8 = (32 + reach + 7) / 8;
        pl->planes = on && pl->tm >= 2 && w.is == 1 && w.T >= 2 && reach >= 0 && p8 <= 7 && 4l * (w.s_row + 64) * w.C * w.B < 0x7fffffffl;
        p->p8 = p8;
        p->cpb = 16 * ((p8 & 1) ? p8 : p8 + 1);                 // odd number of 16-byte slots: channel images start on different banks
        p->chmax = BN / w.T + 2;
    }
    const size_t a_bytes = (size_t)kKS * 2 * (kX6P * BM + 4) * 16;
    pl->lds = a_bytes + (pl->planes ? (size_t)kX6P * p->chmax * p->cpb + 32 : (size_t)kKS * 2 * (kX6P * BN + 4) * 16);
    // K slices: one round of workgroups (512) -- every extra slice is another copy of the whole weight tensor written and
    // re-read.  (Rounds 2-4 ran two rounds, 1024, when the weight tensor has >= 32 tiles: measured per layer then, slower
    // in the step now.)
    const char* te = getenv("RH_WGRAD_X6_BLOCKS");      // (read per call: the fused-tail test asks for the finer split)
    const int target_env = te ? atoi(te) : 0;
    // (round 5: -1 = one round of the RESIDENT workgroups -- 64-row wave tiles fit three per CU, the others two)
    // round 5, A/B on two boxes (tools/debug/exp_r5_*.sh): 512 everywhere 10.02-10.04 ms per step against 10.08-10.09 with two
    // rounds (1024) for the many-tile layers, 10.11 with "resident slots" (768 for TM = 2), 10.10 at 384, 10.35 at 640
    const int target = target_env > 0 ? target_env : (target_env < 0 ? (pl->tm == 2 ? 768 : 512) : 512);
    int Z = rh_cdiv(target, pl->rt * pl->ct);
    const int zmax = p->total_steps / 4 > 0 ? p->total_steps / 4 : 1;     // at least 4 steps (128 positions) per slice
    if (Z > zmax) Z = zmax;
    if (Z < 1) Z = 1;
    pl->steps_per_z = rh_cdiv(p->total_steps, Z);
    pl->Z = rh_cdiv(p->total_steps, pl->steps_per_z);
    p->steps_per_z = pl->steps_per_z;
    return true;
}

template <int TM, int WM, bool AV, bool PART, bool PL>
void go4(const Wx6P& p, const Wx6Plan& pl, hipStream_t stream) {
    auto kern = wgrad_x6_kernel<TM, WM, 4 / WM, AV, PART, PL>;
    static std::once_flag once;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    rh_launch_main(kern, dim3(pl.ct, pl.rt, pl.Z), dim3(256), pl.lds, stream, p);
}

template <int TM, int WM, bool AV, bool PART>
void go3(const Wx6P& p, const Wx6Plan& pl, hipStream_t stream) {
    // (no plane-mode instance for 32-row wave tiles: the opt-in mode is slower anyway and that instance needed scratch)
    if constexpr (TM >= 2) {
        if (pl.planes) return go4<TM, WM, AV, PART, true>(p, pl, stream);
    }
    go4<TM, WM, AV, PART, false>(p, pl, stream);
}

template <int TM, int WM, bool AV>
void go2(const Wx6P& p, const Wx6Plan& pl, hipStream_t stream) {
    constexpr int TPW = 2 * (4 / WM);                              // 32-column MFMA tiles per workgroup tile
    if (rh_cdiv(p.N, 32) % TPW != 0) go3<TM, WM, AV, true>(p, pl, stream);
    else go3<TM, WM, AV, false>(p, pl, stream);
}

template <int TM, int WM>
void go(const Wx6P& p, const Wx6Plan& pl, hipStream_t stream) {
    if (p.r_row % 4 == 0 && ((uintptr_t)p.R & 15) == 0) go2<TM, WM, true>(p, pl, stream);
    else go2<TM, WM, false>(p, pl, stream);
}

}  // namespace

// Scratch (bytes) the bf16x6 weight-gradient path wants for its K-slice partials; -1 = geometry not eligible.
int64_t rh_wgrad_x6_workspace(const WgradP& w) {
    Wx6P p;
    Wx6Plan pl{};
    static const unsigned any_range[kRangeSlotWords] = {};      // planning only: the answer does not depend on the slots
    if (!plan_wx6(w, &p, &pl, any_range, any_range)) return -1;
    // partial weight tiles (Z > 1) + partial row sums for the fused bias gradient (always reserved)
    return (pl.Z > 1 ? (int64_t)pl.Z * w.M * w.C * w.T : 0) * (int64_t)sizeof(float) + (int64_t)pl.Z * w.M * (int64_t)sizeof(float);
}

// Returns RH_OK with *used = false when the geometry does not fit this path.  ws must hold rh_wgrad_x6_workspace bytes.
// rsum_out != null: also the row sums of R over (batch, position) -- the bias gradient when R = dy -- from the same pass
// left_z != null and the K range was split: the weight partials stay UNREDUCED in ws ([Z][M][C*T], *left_z = Z) for a caller
// that folds the reduction into its next pass (conv_wgrad.hip: reduce_wn_bwd_kernel); the bias partials are reduced here.
int rh_wgrad_x6_launch(const WgradP& w, float* dw, float* rsum_out, void* ws, hipStream_t stream, bool* used, int* left_z,
                       const unsigned* r_range, const unsigned* s_range) {
    *used = false;
    if (left_z) *left_z = 0;
    Wx6P p;
    Wx6Plan pl{};
    if (!plan_wx6(w, &p, &pl, r_range, s_range)) return RH_OK;
    p.out = pl.Z > 1 ? (float*)ws : dw;
    float* const rs_part = (float*)ws + (pl.Z > 1 ? (long)pl.Z * w.M * w.C * w.T : 0);
    p.rsum = rsum_out ? (pl.Z > 1 ? rs_part : rsum_out) : nullptr;
    if (pl.wm == 1) {
        if (pl.tm == 1) go<1, 1>(p, pl, stream);
        else if (pl.tm == 2) go<2, 1>(p, pl, stream);
        else go<3, 1>(p, pl, stream);
    } else {
        if (pl.tm == 1) go<1, 2>(p, pl, stream);
        else if (pl.tm == 2) go<2, 2>(p, pl, stream);
        else go<3, 2>(p, pl, stream);
    }
    if (int e = rh_check_launch("conv1d_bwd_weight_x6")) return e;
    *used = true;
    if (pl.Z > 1) {
        if (left_z) *left_z = pl.Z;
        else if (int e = rh_reduce_partials_launch((const float*)ws, dw, (long)w.M * w.C * w.T, pl.Z, stream, "conv1d_bwd_weight_reduce")) return e;
        if (rsum_out) return rh_reduce_partials_launch(rs_part, rsum_out, w.M, pl.Z, stream, "conv1d_bwd_bias_reduce");
    }
    return RH_OK;
}
