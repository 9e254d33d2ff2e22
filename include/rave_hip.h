/*
 * rave_hip.h -- C ABI of librave_hip.so: the MI355X (gfx950 / CDNA4) hot path of RAVE.
 *
 * The reference (acids-ircam/RAVE) has NO native/FFI interface: its "operator API" for this
 * path is the set of Python leaf-module signatures that gin hands to rave.model.RAVE
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference *Python* site it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer would add on the reference
 * side.  All citations are relative to the reference tree.
 *
 * Conventions
 *   - tensors are contiguous fp32, layout (B, C, L) with L innermost (PyTorch default);
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch caching allocator in
 *     the Python binding); the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - return value: RH_OK (0), <0 = invalid / unsupported argument (nothing enqueued),
 *     >0 = hipError_t of the failed launch; rh_last_error() gives a thread-local message;
 *   - re-entrant: no mutable global state (forward runs on the Python thread, backward on
 *     PyTorch's autograd thread with the GIL released by ctypes).
 */
#ifndef RAVE_HIP_H
#define RAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_VERSION 100 /* 0.1.0 */

#define RH_OK 0
#define RH_ERR_INVALID (-1)     /* inconsistent descriptor / null pointer */
#define RH_ERR_UNSUPPORTED (-2) /* valid but not implemented (e.g. groups != 1) */
#define RH_ERR_WORKSPACE (-3)   /* workspace too small; see rh_conv1d_workspace_bytes */

typedef void* rh_stream_t; /* hipStream_t */

/* Activation fused on the conv INPUT (the reference applies it as a separate nn.Module right
 * before the conv: rave/blocks.py:93-105 DilatedUnit, :559, :578, :648, :673). */
enum rh_act {
    RH_ACT_NONE = 0,
    RH_ACT_LEAKY = 1, /* nn.LeakyReLU(slope)          rave/blocks.py:90,527,612; discriminator.py:101 */
    RH_ACT_SNAKE = 2  /* x + sin^2(a x)/(a+1e-9)      rave/blocks.py:852-860 (per-channel alpha)      */
};

/*
 * One 1-D convolution geometry.  Replaces cached_conv.Conv1d / cached_conv.ConvTranspose1d as
 * used by the reference (call sites: rave/pqmf.py:256-273, rave/blocks.py:96-108,536-592,
 * 637-692) and torch.nn.Conv1d / torch.nn.Conv2d with (k,1) kernels as used by the
 * discriminators (rave/discriminator.py:77-119,174-195).
 *
 * transposed == 0:  y[b,co,l] = bias[co] + sum_{ci,t} w[co,ci,t] * act(x)[b,ci, l*stride + t*dilation - pad_left]
 *                   (zero outside [0,l_in)); pad_right is implied by l_out.            w: (c_out, c_in, kernel)
 * transposed == 1:  torch conv_transpose1d(act(x), w, stride, padding=pad_left)         w: (c_in, c_out, kernel)
 *                   l_out = (l_in-1)*stride - 2*pad_left + kernel ; dilation must be 1.
 *
 * inner > 1 describes a Conv2d with kernel (k,1), stride (s,1), padding (p,0) on a (B,C,H,W)
 * tensor with W == inner (MultiPeriodDiscriminator, rave/discriminator.py:186-195): l_in/l_out
 * are H_in/H_out and every position is `inner` contiguous elements.  in_valid (0 = all) is the
 * number of leading elements of each (b,c) input row that are real data; the remainder reads as
 * zero (this is MultiPeriodDiscriminator.fold's zero pad, done in-kernel).
 */
typedef struct rh_conv1d_desc {
    int32_t batch;
    int32_t c_in;
    int32_t c_out;
    int32_t l_in;
    int32_t l_out;
    int32_t kernel;
    int32_t stride;
    int32_t dilation;
    int32_t pad_left;
    int32_t transposed;
    int32_t groups;   /* must be 1 */
    int32_t inner;    /* 1 for Conv1d */
    int32_t in_valid; /* 0 = l_in*inner */
    int32_t act;      /* enum rh_act, fused on the conv input */
    float act_slope;  /* LeakyReLU slope */
    int32_t out_act;  /* RH_ACT_NONE / RH_ACT_LEAKY applied to the OUTPUT (forward only): y = out_act(conv + bias
                       * + residual), as WNConv2d(...) + LeakyReLU(0.1) of rave/descript_discriminator.py:22-27.
                       * The backward entry points ignore it: pass dy * out_act'(y) (rh_act_bwd_f32). */
    float out_slope;
} rh_conv1d_desc;

int rh_version(void);
const char* rh_last_error(void);


/* ---- weight preparation --------------------------------------------------------------- */

/* torch.nn.utils.weight_norm(dim=0) as applied by rave/blocks.py:15-22 `normalization`:
 *   w[r,:] = g[r] * v[r,:] / ||v[r,:]||_2 , rows = size of dim 0, cols = product of the others
 *   (dim 0 = out-channels for Conv1d, IN-channels for ConvTranspose1d).  norms[r] = ||v[r,:]||. */
int rh_weight_norm_fwd_f32(const float* v, const float* g, int64_t rows, int64_t cols, float* w,
                           float* norms, rh_stream_t stream);
/* backward of the above: dv, dg from dw (torch _weight_norm_interface_backward). */
int rh_weight_norm_bwd_f32(const float* dw, const float* v, const float* g, const float* norms,
                           int64_t rows, int64_t cols, float* dv, float* dg, rh_stream_t stream);
/* The same for MANY weight tensors in ONE launch (one workgroup per row over the concatenated rows; table by value in the
 * kernel arguments, <= 64 tensors per launch): the weight-norm backward of every `normalization(conv)` of a backward pass
 * (rave/blocks.py:15-22) collected and run once -- same arithmetic and bits as rh_weight_norm_bwd_f32 per tensor. */
typedef struct rh_wn_bwd_item {
    const float* dw;
    const float* v;
    const float* g;
    const float* norms;
    float* dv;
    float* dg;
    int64_t rows;
    int64_t cols;
} rh_wn_bwd_item;
int rh_weight_norm_bwd_batched_f32(const rh_wn_bwd_item* items, int32_t n_items, rh_stream_t stream);
/* Batched reduction of the weight gradients' K-slice partials (round 6; the per-layer ordered reduction that follows every
 * split weight-gradient kernel -- 56 launches per v2 step -- collected over several layers into one launch, the same sums in the
 * same order).  rh_defer_reduce(item) arms the NEXT rh_conv1d_bwd_weight_f32 call of this thread (consumed by it, like
 * rh_x6_set_ranges): where that call would launch the reduction of its partials it leaves them in its workspace and fills *item
 * (part = the partials [Z][n] behind the bias scratch of the workspace, out = dw, n, Z > 1); item->Z = 0 when nothing is
 * pending (dw is complete).  The caller keeps workspace and dw alive until rh_reduce_partials_batched_f32 has run on the same
 * stream.  Reference site: the weight gradients autograd produces for every conv of rave/blocks.py:514-714 in
 * rave/model.py:288-344. */
typedef struct rh_reduce_item {
    const float* part;
    float* out;
    int64_t n;
    int32_t Z;
    int32_t reserved;
} rh_reduce_item;
int rh_defer_reduce(rh_reduce_item* item);
int rh_reduce_partials_batched_f32(const rh_reduce_item* items, int32_t n_items, rh_stream_t stream);

/* Number of floats of the two packed (MFMA-friendly, K-major) copies of a weight tensor:
 * which = 0 -> operand of rh_conv1d_fwd_f32, which = 1 -> operand of rh_conv1d_bwd_data_f32. */
int64_t rh_conv1d_packed_floats(const rh_conv1d_desc* d, int which);
/* Repack w (PyTorch layout, see rh_conv1d_desc) into wp_fwd and/or wp_bwd (either may be NULL). */
int rh_conv1d_pack_f32(const rh_conv1d_desc* d, const float* w, float* wp_fwd, float* wp_bwd,
                       rh_stream_t stream);

/* Weight norm folded into the repack: v (PyTorch weight layout), g (dim-0 gains) -> norms[r] = ||v[r]||,
 * scale[r] = g[r]/norms[r] (both `rows` floats, rows = dim 0 of v) and the packed copies of
 * w = g v/||v||, without materialising w.  Replaces `normalization(cc.Conv1d(...))`
 * (rave/blocks.py:15-22) on the forward path; the gradient goes through rh_weight_norm_bwd_f32. */
int rh_conv1d_pack_wn_f32(const rh_conv1d_desc* d, const float* v, const float* g, float* norms,
                          float* scale, float* wp_fwd, float* wp_bwd, rh_stream_t stream);

/* Batched form of rh_conv1d_pack_wn_f32 for a whole model: the caller fills one opaque item per
 * weight-normalised conv (host memory, rh_prep_item_bytes() each, pointers are DEVICE pointers to the
 * parameters and to persistent norms / scale / packed buffers; the array is rh_prep_array_bytes(n) long: the items and,
 * behind them, two compact lookup tables rh_prep_link writes), links them, copies the array to the
 * device once, and then refreshes every layer's packed weights with three launches (range clear, scales, pack) per training step. */
int64_t rh_prep_item_bytes(void);
int64_t rh_prep_array_bytes(int32_t n);
/* (rh_prep_fill_item: `scale` must hold 3 * rows floats -- the scales, then the per-row max |w| and sum |w| the range
 * records of the f16 kernels are reduced from.) */
int rh_prep_fill_item(const rh_conv1d_desc* d, const float* v, const float* g, float* norms, float* scale,
                      float* wp_fwd, float* wp_bwd, void* item_host);
int rh_prep_link(void* items_host, int32_t n, int64_t* total_rows, int64_t* total_blocks);
int rh_prep_run_f32(const void* items_dev, int32_t n, int64_t total_rows, int64_t total_blocks, rh_stream_t stream);

/* ---- convolution ---------------------------------------------------------------------- */

/* y = conv(act(x)) + bias + residual.   bias (c_out), residual (B,c_out,l_out*inner) and
 * snake_alpha (c_in) may be NULL.  Replaces `act -> cc.Conv1d -> (+ x)` of DilatedUnit /
 * Residual (rave/blocks.py:31-45,83-112) and every other `activation; conv` pair of
 * EncoderV2 / GeneratorV2 / discriminator.ConvNet. */
int rh_conv1d_fwd_f32(const rh_conv1d_desc* d, const float* x, const float* wp_fwd,
                      const float* bias, const float* snake_alpha, const float* residual,
                      float* y, void* workspace, int64_t workspace_bytes, rh_stream_t stream);
/* Optional scratch for rh_conv1d_fwd_f32 / rh_conv1d_bwd_data_f32 (0 = none needed).  Geometries
 * whose output is too small to fill 256 CUs (short sequences x many channels) split the
 * reduction over (tap, input-channel) chunks across workgroups; the partial sums live in this
 * scratch and are combined in a fixed order (deterministic).  Passing NULL / too few bytes is
 * legal: the launch then runs unsplit (slower, same accumulation class). */
int64_t rh_conv1d_fwd_workspace_bytes(const rh_conv1d_desc* d);
int64_t rh_conv1d_bwd_data_workspace_bytes(const rh_conv1d_desc* d);
/* Which kernel family rh_conv1d_fwd_f32 (which = 0) / rh_conv1d_bwd_data_f32 (which = 1) would launch for this
 * geometry and operand set: 1 = exact f32 on the bf16 matrix cores (3-way split, conv_x6_kernel), 2 = the HBM-bound
 * vector-ALU kernels of the 1- / 2-channel first discriminator layers (conv_smallc.hip), 0 = f32-input MFMA kernels,
 * < 0 = invalid descriptor.  Measurement only (bench.py prices every launch against the peak of the
 * instruction it issues); has_bias / has_add = the optional operands are non-NULL. */
int rh_conv1d_kernel_family(const rh_conv1d_desc* d, int which, int has_bias, int has_add);
/* Fused Residual(DilatedUnit) forward -- y = conv_1x1(act(h), w1) + x, h = conv_k_dilated(act(x), w3) -- in ONE launch
 * (rave/blocks.py:31-45 `Residual`, :83-112 `DilatedUnit`; C = 32 / 64 / 96, stride 1, "same" padding, no biases).
 * d3 / d1 describe the two convolutions (act / act_slope = the activation in front of each), wp3_fwd / wp1_fwd are
 * their packed forward operands.  h: receives the intermediate for the backward pass, or NULL (inference: never
 * written).  Results are bit-identical to rh_conv1d_fwd_f32(d3) followed by rh_conv1d_fwd_f32(d1, residual = x). */
int rh_residual_unit_fused(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1);   /* 1 = the pair fits the fused launch */
int rh_residual_unit_fwd_f32(const rh_conv1d_desc* d3, const rh_conv1d_desc* d1, const float* x, const float* wp3_fwd,
                             const float* wp1_fwd, float* h, float* y, rh_stream_t stream);

/* ---- range slots of the f16 matrix-core kernels (round 6) ---------------------------------------------------------------
 * The "x6" kernels (conv_x6_kernel, unit_x6_kernel, wgrad_x6_kernel) compute exact-f32-class products on the f16 matrix cores:
 * every operand is scaled by a power of two that takes its TENSOR's largest magnitude into [2^14, 2^15) and split into two
 * f16 pieces (csrc/common.hpp).  The scale needs max |x| of each activation operand; it travels in a RANGE SLOT: an array of
 * rh_x6_range_words() uint32 in device memory (32 words in use, one per 128-byte line so that the producers' atomics do not
 * serialise; the rest stays zero), each the bit pattern of a non-negative float -- max |x| is the largest word (any
 * upper bound within a factor of ~2^10 of it keeps f32 accuracy; a too SMALL value overflows f16: the slot must cover the
 * tensor).  rh_x6_set_ranges arms the slots of the NEXT rh_conv1d_fwd_f32 / rh_conv1d_bwd_data_f32 / rh_residual_unit_fwd_f32
 * / rh_conv1d_bwd_weight[_wn]_f32 / rh_conv2d_fwd_f32 / rh_conv2d_bwd_data_f32 / rh_conv2d_bwd_weight_f32 / rh_act_bwd_bias_f32
 * (out = the slot of g) / rh_pqmf_fold_k1_f32 / rh_reparam_fwd_f32 (out = the slot of their output) call of this thread (consumed by that call, like rh_set_kernel_events):
 *     in_a   weight gradient only: the slot of dy          in_b   the slot of the input activation (x; dy for bwd_data)
 *     out    where the call leaves max |output| (forward: y, bwd_data: dx; atomicMax, the caller zeroes it first) or NULL
 *     out2   rh_residual_unit_fwd_f32: the slot of the intermediate h, or NULL
 * A call whose input slot is NULL takes the f32-input MFMA kernels instead (same results to f32 rounding, slower).
 * rh_amax_f32 fills a (zeroed) slot from a tensor no kernel of this library produced.  rh_x6_uses_ranges() = 1 for this
 * build (0 = the bf16 x 6 comparison build, which ignores slots).  The weights' own range is recorded by the pack kernels
 * inside the packed operand.  Reference sites: every conv of rave/blocks.py:83-112,514-714 and rave/discriminator.py:77-119. */
int rh_x6_uses_ranges(void);
int rh_x6_range_words(void);
int rh_x6_set_ranges(const uint32_t* in_a, const uint32_t* in_b, uint32_t* out, uint32_t* out2);
int rh_amax_f32(const float* x, int64_t n, uint32_t* slot, rh_stream_t stream);

/* Diagnostics: out8 = {family, row tiles per wave, column tiles per wave, waves along the rows, K slices, input stride of the
 * fragment layout, virtual rows, workgroups} of the launch rh_conv1d_fwd_f32 (which = 0) / rh_conv1d_bwd_data_f32
 * (which = 1) would issue; zeros behind `family` for the non-bf16x6 families.  Lets the parity tests assert that the
 * benchmarked batch size runs other template instances than the small-batch tests. */
int rh_conv1d_plan_info(const rh_conv1d_desc* d, int which, int has_bias, int has_add, int32_t* out8);
/* Same question for rh_conv1d_bwd_weight_f32: 1 = wgrad_x6_kernel (bf16 matrix cores; for Conv1d it also produces the
 * bias gradient), 2 = first-layer vector-ALU kernel (weight + bias gradient in one pass), 0 = f32-input MFMA kernels. */
int rh_conv1d_bwd_weight_kernel_family(const rh_conv1d_desc* d);

/* dx = act'(x) * conv_bwd_data(dy) + add.   `x` is the forward input (needed when act != NONE),
 * `add` (B,c_in,l_in*inner) may be NULL (residual-branch gradient). */
int rh_conv1d_bwd_data_f32(const rh_conv1d_desc* d, const float* dy, const float* wp_bwd,
                           const float* x, const float* snake_alpha, const float* add, float* dx,
                           void* workspace, int64_t workspace_bytes, rh_stream_t stream);

/* Bytes of scratch rh_conv1d_bwd_weight_f32 needs for this geometry. */
int64_t rh_conv1d_workspace_bytes(const rh_conv1d_desc* d);
/* dw (PyTorch weight layout) = sum_{b,l} dy (x) act(x);  dbias (c_out, may be NULL) = sum dy.
 * Deterministic: split-K partials in `workspace`, then an ordered reduction. */
int rh_conv1d_bwd_weight_f32(const rh_conv1d_desc* d, const float* dy, const float* x,
                             const float* snake_alpha, float* dw, float* dbias, void* workspace,
                             int64_t workspace_bytes, rh_stream_t stream);
/* The same weight gradient pushed through torch._weight_norm's backward (rave/blocks.py:15-22: `normalization` =
 * weight_norm around every conv of the v2 / v3 generator; dim 0 = c_out for Conv1d, c_in for ConvTranspose1d):
 *   dg[r] = <dw[r,:], v[r,:]> / ||v[r,:]||,   dv = (g/||v||) (dw - v <dw,v>/||v||^2)
 * with `norms` = ||v[r,:]|| as the repack left them.  Where the K range was split, dv / dg come straight from the slice
 * partials in ONE launch when RH_WN_FUSED=1 (the summed weight gradient is never written; bitwise the same result as
 * rh_conv1d_bwd_weight_f32 + rh_weight_norm_bwd_f32 -- measured 3 % SLOWER on the v2 step, hence opt-in); otherwise dw
 * goes through `dw_scratch` (weight-sized, contents undefined on return) and the plain weight-norm backward kernel. */
int rh_conv1d_bwd_weight_wn_f32(const rh_conv1d_desc* d, const float* dy, const float* x, const float* snake_alpha,
                                const float* v, const float* g, const float* norms, float* dw_scratch, float* dv,
                                float* dg, float* dbias, void* workspace, int64_t workspace_bytes, rh_stream_t stream);
/* Diagnostics: how many times this process took the one-launch form above (tests assert that it really runs). */
int64_t rh_conv1d_bwd_weight_wn_fused_launches(void);

/* ---- PQMF (rave/pqmf.py CachedPQMF, 16 bands) ------------------------------------------- */

/* CachedPQMF.forward (rave/pqmf.py:279-283): strided Conv1d(1,M,K,stride=M,pad=(pad_left,*))
 * followed by reverse_half (:13-17).  x (rows, T) -> y (rows, M, n_frames) with
 * n_frames = (T + pad_left + pad_right - K)/M + 1.  w: forward_conv.weight (M,1,K). M must be 16. */
int rh_pqmf_analysis_fwd_f32(const float* x, const float* w, int32_t rows, int32_t t_len,
                             int32_t n_band, int32_t kernel, int32_t pad_left, int32_t n_frames,
                             float* y, rh_stream_t stream);
/* gradient of the above w.r.t. x. */
int rh_pqmf_analysis_bwd_f32(const float* dy, const float* w, int32_t rows, int32_t t_len,
                             int32_t n_band, int32_t kernel, int32_t pad_left, int32_t n_frames,
                             float* dx, rh_stream_t stream);
/* CachedPQMF.inverse (rave/pqmf.py:285-294): reverse_half, Conv1d(M,M,K2,pad=(pad_left,*)) * M,
 * band flip and interleave.  y (rows, M, n_frames) -> x (rows, n_out*M); n_out = n_frames +
 * pad_left + pad_right - K2 + 1.  w: inverse_conv.weight (M,M,K2). */
int rh_pqmf_synthesis_fwd_f32(const float* y, const float* w, int32_t rows, int32_t n_frames,
                              int32_t n_band, int32_t kernel, int32_t pad_left, int32_t n_out,
                              float* x, rh_stream_t stream);
/* gradient of the above w.r.t. y. */
int rh_pqmf_synthesis_bwd_f32(const float* dx, const float* w, int32_t rows, int32_t n_frames,
                              int32_t n_band, int32_t kernel, int32_t pad_left, int32_t n_out,
                              float* dy, rh_stream_t stream);

/* The same four transforms in the FOLDED fast form of the cosine-modulated bank (55.6 MAC/sample instead of 513;
 * rave/pqmf.py:32-52 get_qmf_bank + :245-294).  `tab` = 384 floats hs[tau] = (-1)^(tau/32) h[tau] (prototype
 * `pqmf.h`, zero padded) followed by the 16 x 32 matrix Cm[k][m] = 2 cos((2k+1) pi/32 (m - N/2) + (-1)^k pi/4).
 *   k1 (fold -> matrix):        out[row][k][n] = s(k,n) * scale * sum_m Cm[k][m] sum_j hs[m+32j] in[row][16n + m + 32j + o0]
 *   k2 (matrix -> overlap-add): out[row][q]    = scale * sum_n' hs[tau] * sum_c Cm[c][tau%32] s(c,n') in[row][c][n'],
 *                                                tau = q - 16 n' + dp in [0, 384)
 * with s = reverse_half's sign (-1 iff k odd and n even) and zero extension of `in`.  With lpad = (512 - len(h)) / 2:
 *   analysis forward   = k1(x,  o0 = lpad - pad_left, scale 1)        analysis backward  = k2(dy, dp = pad_left - lpad, 1)
 *   synthesis forward  = k2(y,  dp = 496 - 16 pad_left - lpad, 16)    synthesis backward = k1(dx, o0 = lpad - (496 - 16 pad_left), 16)
 * Valid only while forward_conv / inverse_conv hold the designed bank (the caller checks; they are never trained). */
int rh_pqmf_fold_k1_f32(const float* in, const float* tab, int32_t rows, int32_t t_len, int32_t n_frames, int32_t o0,
                        float scale, float* out, rh_stream_t stream);
int rh_pqmf_fold_k2_f32(const float* in, const float* tab, int32_t rows, int32_t n_frames, int32_t n_out, int32_t dp,
                        float scale, float* out, rh_stream_t stream);

/* ---- small fused elementwise ops ------------------------------------------------------- */

/* GeneratorV2 output head (rave/blocks.py:705-711): y = tanh(a * sigmoid(m)), with
 * x = [a | m] split on the channel axis: x (B, 2C, L) -> y (B, C, L). */
int rh_amp_tanh_fwd_f32(const float* x, int32_t batch, int32_t c, int32_t l, float* y,
                        rh_stream_t stream);
int rh_amp_tanh_bwd_f32(const float* dy, const float* x, int32_t batch, int32_t c, int32_t l,
                        float* dx, rh_stream_t stream);
/* Standalone activation (used where no conv follows). n = total elements, c/l for snake. */
int rh_act_fwd_f32(const float* x, const float* snake_alpha, int32_t act, float slope,
                   int32_t batch, int32_t c, int32_t l, float* y, rh_stream_t stream);
/* Snake backward (rave/blocks.py:852-860): dx and dalpha (c floats) from dy, x (B,c,l), alpha (c).
 * workspace: rh_snake_bwd_workspace_bytes(batch, c) bytes (ordered partial sums -> deterministic). */
int64_t rh_snake_bwd_workspace_bytes(int32_t batch, int32_t c);
int rh_snake_bwd_f32(const float* dy, const float* x, const float* alpha, int32_t batch, int32_t c,
                     int32_t l, float* dx, float* dalpha, void* workspace, int64_t workspace_bytes,
                     rh_stream_t stream);
/* g = dy * act'(y) for an OUTPUT LeakyReLU (sign(y) == sign(pre-activation)): the cotangent the backward
 * entry points of a conv with out_act expect. */
int rh_act_bwd_f32(const float* dy, const float* y, int32_t act, float slope, int64_t n, float* g, rh_stream_t stream);
/* One pass for the backward prologue of a conv with a LeakyReLU on its OUTPUT (rave/discriminator.py:50,
 * rave/descript_discriminator.py:27): g = dy * act'(y) and dbias[m] = sum over (batch, plane) of g, on (B, M, plane)
 * tensors; ordered partial sums (deterministic).  workspace: rh_act_bwd_bias_workspace_bytes(M) bytes.  With an output range
 * slot armed (rh_x6_set_ranges) the pass also leaves max |g| there. */
int64_t rh_act_bwd_bias_workspace_bytes(int32_t M);
int rh_act_bwd_bias_f32(const float* dy, const float* y, int32_t act, float slope, int32_t B, int32_t M, int64_t plane, float* g,
                        float* dbias, void* workspace, int64_t workspace_bytes, rh_stream_t stream);
/* AdaptiveInstanceNormalization in EVAL mode (rave/blocks.py:863-926; identity in training mode, :901-902).
 * stats_update = `mean = x.mean(-1); std = x.std(-1); update(mean_buf, mean, n); update(std_buf, std, n)` (:876-879, :904-909,
 * :915-920) on x (rows = batch * channels, l): per-row mean and unbiased standard deviation, then
 * buf[:rows] += (stat - buf[:rows]) / (num_updates[0] + 1); num_updates is the module's device buffer (the caller
 * increments it afterwards, as the reference does).  transfer = :887-895,
 * y = (x - mean_x) / (std_x + 1e-5) * std_y + mean_y with per-row statistics. */
int rh_adain_stats_update_f32(const float* x, int64_t rows, int32_t l, const float* num_updates, float* mean_buf,
                              float* std_buf, rh_stream_t stream);
int rh_adain_transfer_f32(const float* x, int64_t rows, int32_t l, const float* mean_x, const float* std_x,
                          const float* mean_y, const float* std_y, float* y, rh_stream_t stream);
/* nn.functional.avg_pool1d(x, 2) of MultiScaleDiscriminator (rave/discriminator.py:135). */
int rh_avgpool2_fwd_f32(const float* x, int64_t rows, int32_t l_in, float* y, rh_stream_t stream);
int rh_avgpool2_bwd_f32(const float* dy, int64_t rows, int32_t l_in, float* dx, rh_stream_t stream);

/* ---- general 2-D convolution (spectral / descript discriminators) ------------------------------ */

/*
 * torch.nn.Conv2d with zero padding on (B, C, H, W) tensors, W innermost, groups == 1, followed by the
 * LeakyReLU the reference puts right after it.  Replaces the Conv2d stacks of
 *   rave/discriminator.py:23-74  (rectified_2d_conv_block / EncodecConvNet: (9,3) kernels, stride (2,1),
 *                                 dilation (1,1|2|4); (3,3))
 *   rave/descript_discriminator.py:22-27,118-184 (WNConv2d; MRD: (3,9) stride (1,2) pad (1,4); (3,3))
 *   rave/descript_discriminator.py:30-66 (MPD: (5,1) stride (3,1) pad (2,0))
 *
 *   y[b,co,ho,wo] = act( bias[co] + sum_{ci,th,tw} w[co,ci,th,tw] * x[b,ci, ho*sh + th*dh - ph, wo*sw + tw*dw - pw] )
 *
 * `act` is applied to the OUTPUT (those networks keep the activated tensor as a feature map); the
 * backward entry points take y and apply act'(y) to dy on the fly (sign(y) == sign(pre-activation)
 * for a LeakyReLU with positive slope).  Only RH_ACT_NONE / RH_ACT_LEAKY.
 */
typedef struct rh_conv2d_desc {
    int32_t batch;
    int32_t c_in;
    int32_t c_out;
    int32_t h_in, w_in;
    int32_t h_out, w_out; /* must equal floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1 */
    int32_t kh, kw;
    int32_t sh, sw;
    int32_t dh, dw;
    int32_t ph, pw;
    int32_t act;     /* enum rh_act on the output (NONE or LEAKY) */
    float act_slope;
} rh_conv2d_desc;

/* Packed weight sizes (floats): which = 0 forward operand, 1 data-gradient operand. */
int64_t rh_conv2d_packed_floats(const rh_conv2d_desc* d, int which);
/* w: (c_out, c_in, kh, kw) as torch.nn.Conv2d.weight; wp_bwd may be null. */
int rh_conv2d_pack_f32(const rh_conv2d_desc* d, const float* w, float* wp_fwd, float* wp_bwd, rh_stream_t stream);
int rh_conv2d_fwd_f32(const rh_conv2d_desc* d, const float* x, const float* wp_fwd, const float* bias /* may be null */,
                      float* y, rh_stream_t stream);
/* dx = conv2d^T(dy * act'(y)); y may be null when act == RH_ACT_NONE. */
int rh_conv2d_bwd_data_f32(const rh_conv2d_desc* d, const float* dy, const float* y, const float* wp_bwd, float* dx,
                           rh_stream_t stream);
/* Diagnostics (tests): kernel family and tile plan of a forward (which = 0) / data-gradient (1) launch.
 * out[16] = {family: 0 f32-input MFMA, 1 bf16x6 (conv2d_x6.hip), 2 vector ALU (conv2d_smallm.hip: <= 4 output rows); tm; tn; conversion tasks per thread; TR; TQ; nb; LDS bytes;
 *            workgroups; PH; PW; P (patch positions); largest tap offset in the patch; phases; row tiles; column tiles} */
int rh_conv2d_plan_info(const rh_conv2d_desc* d, int32_t which, int64_t* out);
/* 1 = rh_conv2d_bwd_weight_f32 runs this geometry (act == NONE) on the bf16 matrix cores (wgrad2d_x6.hip), else 0. */
int rh_conv2d_bwd_weight_kernel_family(const rh_conv2d_desc* d);
int64_t rh_conv2d_workspace_bytes(const rh_conv2d_desc* d);
/* dw (c_out, c_in, kh, kw), dbias (c_out) or null; deterministic (ordered split-K partials in `workspace`). */
int rh_conv2d_bwd_weight_f32(const rh_conv2d_desc* d, const float* dy, const float* y, const float* x, float* dw,
                             float* dbias, void* workspace, int64_t workspace_bytes, rh_stream_t stream);

/* ---- residual vector quantisation ------------------------------- */

/* One EuclideanCodebook step (rave/quantization.py:131-181) on n_vectors x dim row-major vectors:
 * indices[n] = argmin_k (|x_n|^2 - 2 x_n.e_k) + |e_k|^2 (first index on ties, as torch.max of the negated
 * distance); residual[n] = x_n - e_ind (input of the next quantiser, ResidualVectorQuantization.forward
 * :283-300; may alias x; may be null); quantized_sum[n] += e_ind (may be null);
 * loss_partials[rh_vq_loss_partials(n_vectors)] = per-block sums of |e_ind - x_n|^2 (commitment loss
 * numerator, :263-266; may be null). */
int64_t rh_vq_loss_partials(int64_t n_vectors);
int rh_vq_assign_f32(const float* x, const float* embed, int64_t n_vectors, int32_t dim, int32_t codebook_size,
                     int64_t* indices, float* residual, float* quantized_sum, float* loss_partials, rh_stream_t stream);
/* The same step with caller-provided scratch: with rh_vq_assign_workspace_bytes(...) > 0 bytes of it the distance search
 * (rave/quantization.py:131-136) runs on the f32 matrix cores -- the same fmaf chains, hence the same distances, indices
 * and loss partials as the one-launch kernel -- in two launches; 0 bytes / NULL = rh_vq_assign_f32 (residual may alias x
 * on either path). */
int64_t rh_vq_assign_workspace_bytes(int64_t n_vectors, int32_t dim, int32_t codebook_size);
int rh_vq_assign_ws_f32(const float* x, const float* embed, int64_t n_vectors, int32_t dim, int32_t codebook_size,
                        int64_t* indices, float* residual, float* quantized_sum, float* loss_partials, void* workspace,
                        int64_t workspace_bytes, rh_stream_t stream);
/* Training-time codebook update (:165-179): cluster_size / embed_avg EMAs from per-code counts and ordered
 * vector sums (deterministic), then embed = embed_avg / (laplace_smoothing(cluster_size) * sum). */
int rh_vq_ema_update_f32(const float* x, const int64_t* indices, int64_t n_vectors, int32_t dim, int32_t codebook_size,
                         float decay, float epsilon, float* cluster_size, float* embed_avg, float* embed,
                         rh_stream_t stream);

/* ---- training data feed (SURVEY.md section 8f "next" #4) --------------------------------------------- */

/* One minibatch of the reference's per-item CPU chain (rave/dataset.py:75-78,218-229,246,283-299;
 * rave/transforms.py:96-115) in one launch: for every row r (one channel of one clip)
 *   x = float32(pcm[src_offset[r] + n]) / 32767                                   n < n_signal   (crop applied by the offset)
 *   y = lfilter([b0,b1,b2], [1,a1,a2], x) in float64 (direct form II transposed)  if coef[r][0] is not NaN
 *   out[r][n] = float32(y + noise[r][n] / 2^bit_depth)
 * coef: rows x 5 doubles (b0, b1, b2, a1, a2), b0 = NaN skips the filter (RandomApply miss); noise: U[0,1) floats.
 * The random draws (item, crop point, pole angle, noise) belong to the caller. */
int rh_feed_batch_i16_f32(const int16_t* pcm, const int64_t* src_offset, const double* coef, const float* noise,
                          int32_t rows, int32_t n_signal, int32_t bit_depth, float* out, rh_stream_t stream);

/* ---- spectral distance ("next" item #1 of SURVEY.md section 8f, beside the hot path) ------------- */

/* STFT framing of torchaudio.transforms.Spectrogram(center=True, pad_mode="reflect") as used by
 * MultiScaleSTFT (rave/core.py:269-319): frames (rows, n_frames, n_fft) = window * reflect-padded x,
 * n_frames = t_len / hop + 1; and its adjoint (gradient w.r.t. x).  The FFT stays on rocFFT. */
int rh_stft_frame_fwd_f32(const float* x, const float* window, int64_t rows, int32_t t_len, int32_t n_fft,
                          int32_t hop, int32_t n_frames, float* frames, rh_stream_t stream);
int rh_stft_frame_bwd_f32(const float* dframes, const float* window, int64_t rows, int32_t t_len, int32_t n_fft,
                          int32_t hop, int32_t n_frames, float* dx, rh_stream_t stream);
/* the same with accumulate != 0: dx += ... (the scales of MultiScaleSTFT, rave/core.py:269-319, add up in place) */
int rh_stft_frame_bwd_acc_f32(const float* dframes, const float* window, int64_t rows, int32_t t_len, int32_t n_fft,
                              int32_t hop, int32_t n_frames, float* dx, int32_t accumulate, rh_stream_t stream);

/* AudioDistanceV1 on one STFT scale (rave/core.py:330-344) from two complex spectrograms (interleaved
 * re,im; n_complex elements each):  sums[0] = sum (|Sx|-|Sy|)^2, sums[1] = sum |Sx|^2,
 * sums[2] = sum |log(|Sx|+eps) - log(|Sy|+eps)|   =>   distance = sums[0]/sums[1] + sums[2]/n_complex.
 * The STFT itself stays on rocFFT.  workspace: rh_spectral_distance_workspace_bytes(). */
int64_t rh_spectral_distance_workspace_bytes(void);
int rh_spectral_distance_fwd_f32(const float* sx, const float* sy, int64_t n_complex, float eps, float* sums,
                                 void* workspace, int64_t workspace_bytes, rh_stream_t stream);
/* d distance / d Sx and / d Sy (either output may be NULL), scaled by the device scalar grad_out[0].
 * half_bins = 0: the plain gradients.  half_bins = n_fft/2+1 (spectra laid out (..., half_bins)): the interior
 * bins are halved, which makes the outputs the operand of the UNNORMALISED C2R transform that is the adjoint of
 * rfft -- d distance / d frames = irfft(dsx, n_fft, norm="forward") -- instead of autograd's zero-fill + copy +
 * C2C + real-part chain. */
int rh_spectral_distance_bwd_f32(const float* sx, const float* sy, const float* sums, const float* grad_out,
                                 int64_t n_complex, float eps, float* dsx, float* dsy, int32_t half_bins,
                                 rh_stream_t stream);

/* AudioDistanceV1.forward's sum over the scales (rave/core.py:336-344): out[0] = sum_s sums[3s]/sums[3s+1] + sums[3s+2]*inv_n[s]
 * (sums: n_scales x 3 as written by rh_spectral_distance_fwd_f32, inv_n[s] = 1 / n_complex of scale s). */
int rh_spectral_total_f32(const float* sums, const float* inv_n, int32_t n_scales, float* out, rh_stream_t stream);

/* AudioDistanceV1 on one STFT scale (rave/core.py:269-344: Spectrogram(n_fft, hop = n_fft / 4, center, reflect) -> |.| ->
 * relative L2 + L1 of logs) with the transform INSIDE the kernel: x, y (rows, t_len) f32 waveforms, window (n_fft, the
 * normalised window the reference multiplies the frames by), twiddle (n_fft complex: e^{-2 pi i k / n_fft}, interleaved).
 * n_fft in {128, 256, 512, 1024, 2048}, hop = n_fft / 4, t_len > n_fft / 2 (rh_stft_loss_supported).  Same `sums` as
 * rh_spectral_distance_fwd_f32 (n_complex = rows * (t_len / hop + 1) * (n_fft / 2 + 1)); the backward recomputes the
 * spectra, and writes (accumulate = 0) or adds (!= 0) d distance / d x and / d y (either may be NULL) times grad_out[0]. */
int rh_stft_loss_supported(int32_t n_fft, int32_t hop, int32_t t_len, int64_t rows);
/* diagnostics: out6 = {frames per forward workgroup, forward workgroups per row, hop blocks per backward workgroup, backward
 * workgroups per row, hop blocks of the last backward workgroup, dynamic LDS bytes of the backward launch} */
int rh_stft_loss_plan_info(int32_t n_fft, int32_t t_len, int64_t rows, int64_t* out6);
int64_t rh_stft_loss_workspace_bytes(int32_t n_fft, int32_t t_len, int64_t rows);
int rh_stft_loss_fwd_f32(const float* x, const float* y, const float* window, const float* twiddle, int64_t rows,
                         int32_t t_len, int32_t n_fft, float eps, float* sums, void* workspace, int64_t workspace_bytes,
                         rh_stream_t stream);
/* sums == NULL above: the per-workgroup partials stay in `workspace` (rh_stft_loss_workspace_bytes of them) and ONE call
 * finalizes all scales of a distance: sums (n_scales x 3, as above) and total = sum_s sums[s][0] / sums[s][1] + sums[s][2] *
 * inv_n[s] (AudioDistanceV1 over MultiScaleSTFT, rave/core.py:269-344) -- one launch instead of n_scales + 1 between the
 * forward and the backward pass; same reduction order, same bits.  partials[s] / partial_bytes[s]: HOST arrays. */
int rh_stft_loss_finalize_all_f32(const float* const* partials, const int64_t* partial_bytes, int32_t n_scales,
                                  const float* inv_n, float* sums, float* total, rh_stream_t stream);
int rh_stft_loss_bwd_f32(const float* x, const float* y, const float* window, const float* twiddle, int64_t rows,
                         int32_t t_len, int32_t n_fft, float eps, const float* sums, const float* grad_out, float* dx,
                         float* dy, int32_t accumulate, rh_stream_t stream);

/* Measurement hook (bench.py's roofline leg; not part of the reference interface): arm a pair of HIP events (hipEvent_t,
 * created by the caller with timing enabled) for the calling thread's NEXT main kernel launch -- the convolution /
 * weight-gradient kernel of rh_conv1d_fwd_f32 / _bwd_data_f32 / _bwd_weight_f32 / rh_residual_unit_fwd_f32, not its split-K
 * finalize or reduction launches.  The kernel is dispatched with the events as its own start / stop events, so
 * hipEventElapsedTime(start, stop) is the duration rocprofv3 reports for that dispatch.  rh_kernel_events_used() tells whether
 * a launch consumed the pair (and disarms it). */
int rh_set_kernel_events(void* start_event, void* stop_event);
int rh_kernel_events_used(void);
int rh_event_create(void** event);                                   /* hipEventCreate (timing enabled) */
int rh_event_destroy(void* event);
int rh_event_elapsed_ms(void* start_event, void* stop_event, float* ms);   /* after the stream has been synchronised */

/* ---- the generator loss of a training step (rave/model.py:336-344, 392-412) ---------------------------------------
 * scaled[i] = w1_i * value_i  (the weighted distance the step logs: `self.weights[...] * v`, `reg * beta_factor`)
 * total     = sum_i scaled[i] * w2_i, in term order from 0  (`loss_gen_value += v * self.weights.get(k, 1.)`)
 * as one launch, and its gradient grads[i] = (grad_total * w2_i) * w1_i as another -- instead of ~2 scalar ATen launches per
 * term in each direction between the forward and the backward pass.  Each product is rounded separately and the sum runs
 * in order: the same bits as the ATen chain.  value / w1_dev: device scalars (w1_dev NULL -> the host value w1);
 * <= 16 terms; the table travels by value (hipGraph-capturable). */
typedef struct rh_loss_item {
    const float* value;
    const float* w1_dev;
    float w1;
    float w2;
} rh_loss_item;
int rh_loss_combine_fwd_f32(const rh_loss_item* items, int32_t n_items, float* scaled, float* total, rh_stream_t stream);
int rh_loss_combine_bwd_f32(const rh_loss_item* items, int32_t n_items, const float* grad_total, float* grads,
                            rh_stream_t stream);

/* Adam step over many tensors (torch.optim.Adam, weight_decay = 0, amsgrad = False; rave/model.py:226-233): for every
 * item  m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * with t = the item's OWN step count (torch keeps one per parameter: a parameter whose first gradient comes later -- the
 * v1 generator's noise branch after the warm-up, rave/blocks.py:418 -- starts at t = 1).
 * `items` is a HOST array (the pointers inside are device pointers; it is consumed during the call: the tables travel in
 * the kernel arguments, so the call can be recorded into a hipGraph); `lr` and every `step` are device scalars -- the call
 * first advances each DISTINCT step counter among the items by one (items may share a counter) -- and `aux` is
 * 2 * n_items floats of device scratch (the bias corrections per counter). */
typedef struct rh_adam_item {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float* step;
} rh_adam_item;
int rh_adam_step_f32(const rh_adam_item* items, int32_t n_items, const float* lr, float beta1, float beta2, float eps,
                     float* aux, rh_stream_t stream);

/* Feature-matching distance of the GAN phase (rave/model.py:359-372 over rave/core.py:236-252, norm "L1") on UNSPLIT
 * discriminator feature maps: item i is a dense f32 tensor of 2 * half elements, the real half of the batch first;
 * distance = sum_i w_i * (relative ? sum|r-f| / sum|r| : sum|r-f|)  (the host folds the 1 / count factors -- and for the
 * non-relative form 1 / half -- into w_i).  `items` is a HOST array consumed during the call (<= 80 items).  fwd writes
 * sums[2i] = sum|r-f|, sums[2i+1] = sum|r| and out[0]; bwd writes d distance / d feature * grad_out[0] into items[i].df
 * (same layout as f).  workspace: rh_feature_matching_workspace_bytes(). */
typedef struct rh_fm_item {
    const float* f;
    float* df;
    int64_t half;
    float w;
} rh_fm_item;
int64_t rh_feature_matching_workspace_bytes(const rh_fm_item* items, int32_t n_items);
int rh_feature_matching_fwd_f32(const rh_fm_item* items, int32_t n_items, int32_t relative, void* workspace,
                                int64_t workspace_bytes, float* sums, float* out, rh_stream_t stream);
int rh_feature_matching_bwd_f32(const rh_fm_item* items, int32_t n_items, int32_t relative, const float* sums,
                                const float* grad_out, rh_stream_t stream);

/* VariationalEncoder.reparametrize (rave/blocks.py:727-745): z (batch, 2c, l) = [mean | scale] along dim 1, eps (batch, c, l)
 * the noise draw;  std = softplus(scale) + 1e-4,  zs = eps * std + mean,
 * kl[0] = (mean^2 + std^2 - log(std^2) - 1).sum(1).mean().  bwd: dz from dzs (may be NULL) and the scalar dkl (may be NULL). */
int64_t rh_reparam_workspace_bytes(void);
int rh_reparam_fwd_f32(const float* z, const float* eps, int32_t batch, int32_t c, int32_t l, float* zs, float* kl,
                       void* workspace, int64_t workspace_bytes, rh_stream_t stream);
int rh_reparam_bwd_f32(const float* z, const float* eps, const float* dzs, const float* dkl, int32_t batch, int32_t c,
                       int32_t l, float* dz, rh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RAVE_HIP_H */
