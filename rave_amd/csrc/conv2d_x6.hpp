// Parameter block + host entry points of the bf16x6 general Conv2d kernels (conv2d_x6.hip); filled by conv2d.hip.
#pragma once
#include "conv_params.hpp"

constexpr int kPh2x = 8;   // max output phases of a data gradient (sh * sw)

// Generalised 2-D gather convolution, one launch per direction (forward, or all output phases of a data gradient):
//   out[b][m][r*os_h + oph_h][q*os_w + oph_w] = epi( sum_t sum_c W[phase][t][c][m] * in[b][c][r*is_h + offh[t]][q*is_w + offw[t]] )
// with zero outside the input plane (all four paddings), (r, q) over the per-phase output grid rows x qcols.
struct C2X {
    const float* in;
    const unsigned* wq;        // pre-split weights: 16-byte fragments [phase][chunk][tap][g][piece][Mp]
    float* out;
    const float* bias;         // [M] or null
    int B, C, M, Mp;
    int in_h, in_w, out_h, out_w;
    int rows, qcols;
    int is_h, is_w, os_h, os_w;
    int out_act;               // RH_ACT_NONE / RH_ACT_LEAKY on the output (after the bias)
    float out_slope;
    int nphase;
    int ph_oph_h[kPh2x], ph_oph_w[kPh2x], ph_ntaps[kPh2x], ph_tap0[kPh2x];
    int ph_minh[kPh2x], ph_minw[kPh2x], ph_maxh[kPh2x], ph_maxw[kPh2x];
    long ph_q2ofs[kPh2x];      // first fragment (16-byte units) of each phase in wq
    int offh[kMaxTaps], offw[kMaxTaps];
    unsigned in_bytes, wq_bytes;
    // range slots (f16 build, common.hpp): max |in| (required), where max |out| goes (or null), the weights' record behind wq
    const unsigned* in_range;
    unsigned* out_range;
    const unsigned* w_range;
    // ---- tile plan (rh_conv2d_x6_plan)
    int TQ, TR, tq_shift, tr_shift, nb;      // column tile = nb batch items x TR rows x TQ columns (powers of two)
    int tiles_q, tiles_r;
    int PH, PW, P;                           // staged patch per batch item PH x PW positions; P = nb * PH * PW
    int toff[kMaxTaps];                      // fragment offset of a tap inside the patch of its phase
};

// bf16x6 section of a packed 2-D operand: fragments (16-byte units) and the first fragment of every phase
long rh_conv2d_x6_units(int C, int M, int nphase, const int* ntaps, long* ph_ofs);
// Launches the gather convolution on the bf16 matrix cores; *used = false (and RH_OK) when the geometry does not take
// this path (the caller then runs the f32-MFMA kernels).
int rh_conv2d_x6_launch(C2X& p, hipStream_t stream, const char* what, bool* used);
// Diagnostics: {tm, tn, nq, TR, TQ, nb, lds bytes, workgroups}; false = not eligible
bool rh_conv2d_x6_plan_query(C2X p, long* out8);
