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
    float* part;                               // [ksplit][B][M][out_row] scratch (caller workspace)
    long part_stride;                          // B*M*out_row
    unsigned in_bytes, w_bytes;                // buffer sizes for the bounds-checked DMA descriptors
    // --- bf16x6 path (conv_x6.hip) only ---
    const unsigned short* wq;                  // pre-split weights [slot][C/8][3][Mp][8] bf16 (caller scratch)
    unsigned wq_bytes;
    int x6_xf_floats, x6_xb_bytes, x6_w_bytes; // LDS regions: f32 input stage, bf16 tile, one weight stage
    int x6_packed;                             // the packed operand carries the bf16x6 section (rh_conv1d_* packers)
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
    unsigned short* wq;   // bf16x6 section [slot][C/8][3][Mp][8] behind the f32 section (conv_x6.hip) or null
    long total;           // nslots * C * Mp   (0 = nothing to do)
    int C, M, Mp, k;      // k = taps per (m, c) pair in the source tensor
    int m_major;          // source index = (m*C + c)*k + kk, else (c*M + m)*k + kk
    int kk[kMaxTaps];
};
int rh_pack_launch(const PackP& a, const PackP& b, hipStream_t stream, const char* what);
// dbias[m] = sum_{b,e} dy[b][m][e] * act'(y[b][m][e]) (y may be null): grid (M, 64) partials in `part`
// (rh_bias_grad_workspace(M) bytes) + ordered finalize -> deterministic
int64_t rh_bias_grad_workspace(int M);
int rh_bias_grad_launch(const float* dy, const float* y, float* part, float* db, int B, int M, long plane, int act,
                        float slope, hipStream_t stream);
int rh_reduce_partials_launch(const float* part, float* out, long n, int Z, hipStream_t stream, const char* what);

int rh_conv_fill_fwd(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_fill_dgrad(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_launch(ConvP& p, hipStream_t stream, const char* what, void* ws = nullptr, int64_t ws_bytes = 0);
int rh_conv_launch_sync(ConvP& p, hipStream_t stream, const char* what);   // register-staged (all activations)
int rh_conv_launch_dma(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes);
bool rh_conv_dma_eligible(const ConvP& p);
int64_t rh_conv_splitk_workspace(ConvP p);     // bytes of scratch the launch would like (0 = none)
int64_t rh_conv_x6_workspace(ConvP p);         // bf16x6 path: scratch it needs, -1 = geometry not eligible
// weights eligible for the bf16x6 kernels (decided from the geometry alone, so that packers and launchers agree)
inline bool rh_x6_weights(int M, int C, int ntaps, int nphase, int is, int os, int inner) {
    return nphase == 1 && is == 1 && os == 1 && inner == 1 && ntaps >= 1 && ntaps <= 3 && (C & 15) == 0 && M % 96 == 0;
}
int rh_conv_launch_x6(ConvP& p, hipStream_t stream, const char* what, void* ws, int64_t ws_bytes, bool* used);
