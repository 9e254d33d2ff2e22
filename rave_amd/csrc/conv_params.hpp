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
    int in_act, epi_act;
    float in_slope, epi_slope;
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
    int off[kMaxTaps];
};

int rh_conv_fill_fwd(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_fill_dgrad(const rh_conv1d_desc* d, ConvP* p);
int rh_conv_launch(ConvP& p, hipStream_t stream, const char* what);
