// Host+device parameter blocks of the implicit-GEMM convolution kernels.
#pragma once
#include "common.hpp"

constexpr int kMaxPhases = 8;
constexpr int kMaxTaps = 64;

// Generalised 1-D "gather" convolution (forward of Conv1d / ConvTranspose1d and the data
// gradient of both):
//   out[b][m][oidx(n)] = epi( sum_{t<ntaps} sum_{c<C} wp[phase][t][c][m] * act(in[b][c][ibase(n) + off[t]*inner]) )
//   ibase(n) = (n / inner) * is * inner + n % inner          (n = column index, 0 <= n < ncols)
//   oidx(n)  = ((n / inner) * os + oph) * inner + n % inner
//   epi(v)   = (v + bias[m]) * act'(mul_src[...]) + add[...]
struct ConvP {
    const float* in;
    const float* wp;
    float* out;
    const float* bias;       // [M] or null
    const float* mul_src;    // [B][M][out_row] or null
    const float* add;        // [B][M][out_row] or null
    const float* in_alpha;   // snake alpha of the input activation [C] or null
    const float* mul_alpha;  // snake alpha of the epilogue derivative [M] or null
    int B, C, M, Mp;
    int in_row, in_valid, out_row, out_valid;  // row strides (elements) and valid prefixes
    int ncols, inner, is, os;
    int bnl, bnl_shift, nb, tiles_per_b;       // column tile = nb batch items x bnl columns
    int pitch;                                 // LDS pitch of one staged input row
    int ck;                                    // channels per K chunk (even)
    int wlds_floats;                           // floats reserved for the weight tile in LDS
    int in_act, epi_act, out_act;              // out_act: LeakyReLU on the OUTPUT (after bias / add)
    float in_slope, epi_slope, out_slope;
    // --- LDS-DMA (async global->LDS) pipeline only ---
    int segs;                                  // 64-float segments per staged input row (pitch = 64*segs)
    unsigned magic_segs, magic_ck, magic_lpr;  // ceil(2^20 / d): exact n/d for n*d < 2^20
    int stage_floats;                          // floats per pipeline stage (weights + input tile)
    int ksplit;                                // K (channel-chunk) slices; > 1 -> raw partial sums go to `part`
    int chunks_per_split;
    float* part;                               // [ksplit][B][M][out_row] scratch (caller workspace); x6 path: [ksplit][tile]
                                               // register dumps of whole tiles (conv_x6_kernel.inc: x6_combine)
    unsigned* tickets;                         // x6 path, ksplit > 1: one arrival counter per output tile, zero between launches
    long part_stride;                          // B*M*out_row
    unsigned in_bytes, w_bytes;                // buffer sizes for the bounds-checked DMA descriptors
    // --- x6 path (conv_x6.hip) only ---
    const unsigned* wq;                        // pre-split weights: 16-byte fragments [phase][step][g][piece][Mp] (8 halves each)
    unsigned wq_bytes;
    // range slots (common.hpp: RH_X6_F16): max |x| of the input tensor (kRangeWords words; required by the f16 kernels),
    // where the launch leaves max |out| (kRangeWords words, zeroed by the caller; may be null), and the weights' own record
    // behind their fragments: {max |w|, max over dim-0 rows of sum |w|, 0, 0} as float bit patterns
    const unsigned* in_range;
    unsigned* out_range;
    const unsigned* w_range;
    int x6_mode;                               // 0 = the packed operand has no bf16x6 section; else the input stride IS the
                                               // section was packed for (1 = plain taps, 2 / 4 = phase-interleaved octets)
    int x6_P;                                  // positions (16-byte fragments) per (octet, piece) plane of the B tile in LDS
    int x6_a_units, x6_b_units;                // 16-byte units of one A stage / one B stage in LDS
    int x6_nu;                                 // steps (taps or tap groups) per K chunk
    // "virtual rows" (a second bf16x6 section behind the per-phase one, v_s = s > 1 when the packer wrote it: the s
    // output phases of a transposed convolution / strided data gradient as s * M GEMM rows of ONE stride-1 gather;
    // row r = m * s + j, a column's s outputs are contiguous in memory):
    int v_s;                                   // 0 = the operand has no such section
    long v_q2ofs;                              // its first fragment (16-byte units) in wq
    int vs;                                    // 0 / 1 = off; s in the kernel's copy of the parameters
    int Mr;                                    // real output channels (M = Mr * vs there)
    int v_o0;                                  // output index of row j of column n = n * vs + v_o0 + j
    int v_U, v_b0;                             // taps of the virtual gather, offsets b0 - u
    long ph_q2ofs[kMaxPhases];                 // first fragment (16-byte units) of each phase in wq
    long x6_wofs;                              // floats between wp and the bf16x6 section of the packed operand
    int nphase;
    int ph_oph[kMaxPhases], ph_ntaps[kMaxPhases], ph_tap0[kMaxPhases], ph_minoff[kMaxPhases],
        ph_maxoff[kMaxPhases];
    long ph_wofs[kMaxPhases];
    int off[kMaxTaps];
};

// Weight gradient:  out[z][m][c*T + t] = sum_{(b,n) in slice z} actR(R[b][m][n]) * actS(S[b][c][ibase(n) + off[t]*inner])
struct WgradP {
    const float* R;
    const float* S;
    float* out;              // [Z][M][C*T] partials (or dw itself when Z == 1)
    const float* r_alpha;    // snake alpha for R rows [M] or null
    const float* s_alpha;    // snake alpha for S rows [C] or null
    int B, M, C, T;
    int r_row;               // row stride of R (elements) == number of n per batch item
    int s_row, s_valid;      // row stride / valid prefix of S
    int inner, is;
    int r_act, s_act;
    float r_slope, s_slope;
    int rk;                  // K rows per chunk (chunk covers rk*inner consecutive n)
    int chunks_per_b;        // ceil((r_row/inner) / rk)
    int total_chunks;        // B * chunks_per_b
    int chunks_per_z;        // chunks handled by one K slice
    int pr;                  // LDS pitch of an R row
    int ps;                  // LDS pitch of an S row
    int nc_max;              // S rows staged per chunk
    int minoff, maxoff;
    // --- LDS-DMA pipeline only ---
    unsigned magic_ps, magic_cpb, magic_pr, magic_lpr_r, magic_lpr_s;   // ceil(2^32/d) reciprocals (0 encodes d == 1)
    int vec;                        // 16-byte DMA fast path enabled
    int r_floats, s_floats;         // flat tile sizes (multiples of 64 floats)
    int stage_floats;
    unsigned r_bytes, s_bytes;
    int off[kMaxTaps];
};

// Weight repack: wp[slot][c][Mp] = scale * w[src(m, c, kk[slot])]   (one slot per (phase, tap))
struct PackP {
    const float* w;
    const float* scale;   // per dim-0 slice (weight-norm g/||v||) or null
    float* wp;
    unsigned* wq;         // x6 section behind the f32 section (conv_x6.hip) or null: 16-byte fragments of 8 halves,
                          //   [phase][step][g = octet of the 16-deep MFMA k block][piece 0..kX6P-1][Mp]
    unsigned* range;      // one 16-byte record behind the fragments: {max |w|, max row sum |w|, 0, 0} (float bits); written by
                          //   the range kernels BEFORE the pack kernel, which scales the f16 pieces by it
    long total;           // nslots * C * Mp   (0 = nothing to do)
    int C, M, Mp, k;      // k = taps per (m, c) pair in the source tensor
    int nslots;
    int m_major;          // source index = (m*C + c)*k + kk, else (c*M + m)*k + kk
    int x6_mode;          // 0 none; 1: step = (16-channel chunk, tap); IS > 1: step = (16/IS-channel chunk, tap group u),
                          //   k slot kappa of the block <-> channel kappa / IS, source tap u*IS + kappa % IS
    int x6_nu;            // IS > 1: tap groups per chunk = ceil(k / IS)
    int bf16x3;           // 1: the mode-1 section holds THREE bf16 truncation pieces per value whatever the build (the 2-D
                          //   kernels' operand: conv2d.hip, conv2d_x6.hip), no range record
    // x6_vs = s > 1: a second section ("virtual rows") at fragment x6_vofs: step = (16-channel chunk, tap u < x6_U),
    // row r = m * s + q2j[slot], rows padded to x6_Mvp, fragment of slot = x6_vofs + vq2a[slot] + chunk * x6_U*6*x6_Mvp;
    // (j, u) pairs no slot covers are written as zeros
    int x6_vs, x6_U, x6_Mvp, vz_n;
    long x6_vofs;
    int q2j[kMaxTaps], vq2a[kMaxTaps];
    int vz_j[2 * kMaxPhases], vz_u[2 * kMaxPhases];
    int kk[kMaxTaps];
    int q2a[kMaxTaps];    // mode 1: fragment offset of (phase of the slot, chunk 0, tap-in-phase of the slot)
    int q2n[kMaxTaps];    // mode 1: fragments per chunk in the phase of the slot (ntaps * 6 * Mp)
};
int rh_pack_launch(const PackP& a, const PackP& b, hipStream_t stream, const char* what, const float* w = nullptr, long rows = 0,
                   long cols = 0);
// dbias[m] = sum_{b,e} dy[b][m][e] * act'(y[b][m][e]) (y may be null): grid (M, 64) partials in `part`
// (rh_bias_grad_workspace(M) bytes) + ordered finalize -> deterministic
int64_t rh_bias_grad_workspace(int M);
int rh_bias_grad_launch(const float* dy, const float* y, float* part, float* db, int B, int M, long plane, int act,
                        float slope, hipStream_t stream, float* g_out = nullptr, unsigned* g_range = nullptr);
int rh_reduce_partials_launch(const float* part, float* out, long n, int Z, hipStream_t stream, const char* what);
// weight norm behind a weight gradient: the caller wants dv, dg of w = g v/||v|| (dim 0) instead of dw
struct RhWnTail {
    const float* v;
    const float* g;
    const float* norms;
    float* dv;
    float* dg;
};
int rh_wgrad_run(const rh_conv1d_desc* d, const float* dy, const float* x, const float* alpha, float* dw, float* dbias, void* ws,
                 int64_t ws_bytes, hipStream_t stream, const RhWnTail* tail = nullptr);

int rh_conv_fill_fwd(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_fill_dgrad(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_launch(ConvP& p, hipStream_t stream, const char* what, void* ws = nullptr, int64_t ws_bytes = 0);
int rh_conv_launch_sync(ConvP& p, hipStream_t stream, const char* what);   // register-staged (all activations)
int rh_range_after(const ConvP& p, hipStream_t stream);                    // fills p.out_range from the output (kernels that do not publish)
int rh_conv_launch_dma(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes);
bool rh_conv_dma_eligible(const ConvP& p);
int64_t rh_conv_splitk_workspace(ConvP p);     // bytes of scratch the launch would like (0 = none)
int64_t rh_conv_x6_workspace(ConvP p);         // bf16x6 path: scratch it needs, -1 = geometry not eligible
// Which bf16x6 operand layout a geometry gets (decided from the tap plan alone, so that packers and launchers agree):
// 0 = none, 1 = plain taps (input stride 1, any number of output phases), 2 / 4 = phase-interleaved octets of a
// strided gather (one phase, contiguous taps, dilation 1).
inline int rh_x6_mode(int C, int nphase, int is, int inner, int ntaps0, const int* off, const int* kk) {
    if (is == 1) return (C & 15) == 0 ? 1 : 0;
    // stride 3 (descript MPD's (5,1) stride-(3,1) convolutions, rave/descript_discriminator.py:36-42): plain-tap fragments
    // as for stride 1; conv_x6_kernel has no stride-3 instance (3 does not divide the 8-sample octet) -- the launcher hands
    // these geometries to the 2-D kernel (conv2d_x6.hip, W = 1), where a stride is a multiplier on the lane's patch position
    if (is == 3) return (nphase == 1 && inner == 1 && (C & 15) == 0) ? 1 : 0;
    if (nphase != 1 || inner != 1 || (is != 2 && is != 4) || (C * is) % 16 != 0) return 0;
    for (int t = 0; t < ntaps0; ++t)
        if (off[t] != off[0] + t || kk[t] != t) return 0;
    return is;
}
// conv_x6_kernel instances (conv_x6_kernel.inc: x6_launch): epi = operand mode (bit 0 bias, 1 derivative, 2 add, 3 output
// activation) + 16 * (virtual rows: 1 = runs of 2, 2 = runs of 4); is = input stride of the fragment layout, tm = row tiles per wave
inline bool rh_x6_epi_instantiated(int is, int tm, bool leaky, int epi) {
    const int m = epi & 15, v = epi >> 4;
    if (v) {
        if (is != 1 || tm < 2 || v > 2) return false;
        return leaky ? (m == 0 || m == 1 || m == 5) : (m == 0 || m == 1 || m == 2 || m == 4 || m == 5 || m == 6);
    }
    return leaky ? (m == 0 || m == 1 || m == 4 || m == 5) : (m == 0 || m == 1 || m == 2 || m == 4 || m == 5 || m == 6 || m == 9 || m == 13);
}
int rh_conv_launch_x6(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes, bool* used);
// stride-3 gathers with plain-tap fragments on the 2-D bf16x6 kernel (conv2d_x6.hip); query = would this launch take it?
int rh_conv_launch_c2x(ConvP& p, hipStream_t stream, const char* what, bool* used);
bool rh_conv_c2x_query(ConvP p);
bool rh_conv_x6_plan_query(ConvP p, int* out7);

// Vector-ALU kernels for the 1- / 2-channel first layers of the discriminators (conv_smallc.hip)
bool rh_smallc_fwd_eligible(const rh_conv1d_desc* d, bool has_residual);
int rh_smallc_fwd(const rh_conv1d_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, hipStream_t stream,
                  unsigned* out_range = nullptr);
bool rh_smallc_dgrad_eligible(const rh_conv1d_desc* d, bool has_add);
int rh_smallc_dgrad(const rh_conv1d_desc* d, const float* dy, const float* wp_bwd, const int* slot_of_tap, float* dx,
                    hipStream_t stream);
bool rh_smallc_wgrad_eligible(const rh_conv1d_desc* d);
int64_t rh_smallc_wgrad_workspace(const rh_conv1d_desc* d);
int rh_smallc_wgrad(const rh_conv1d_desc* d, const float* dy, const float* x, float* dw, float* dbias, void* ws,
                    hipStream_t stream);
