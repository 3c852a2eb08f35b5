/*
 * mpa.h -- C ABI of libmpa_hip.so, the MI355X (gfx950) kernel library behind
 * multipitch_architectures_amd.nn_models.
 *
 * The reference (christofw/multipitch_architectures) has no FFI of its own: its
 * hot path is the Python class API of libdl/nn_models,
 * and every heavy op is a torch.nn call that lowers to cuDNN/cuBLAS/ATen.  Each
 * entry point below therefore names the torch.nn call site it replaces
 * (file:line in /root/reference) instead of a reference FFI symbol.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, device pointers; activations NCHW with
 *     dim2 = time frames, dim3 = frequency bins (hcqt_datasets.py:74-75)
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synced
 *   - no allocation inside: workspaces are sized by the *_workspace() helpers and
 *     passed in; no global state
 *   - return 0 on success, negative MPA_ERR_* otherwise (mpa_strerror())
 */
#ifndef MPA_H
#define MPA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MPA_OK 0
#define MPA_ERR_ARG (-1)
#define MPA_ERR_LAUNCH (-2)
#define MPA_ERR_UNSUPPORTED (-3)
#define MPA_ERR_WORKSPACE (-4)

#define MPA_ACT_NONE 0
#define MPA_ACT_RELU 1
#define MPA_ACT_LRELU 2
#define MPA_ACT_SIGMOID 3
#define MPA_ACT_SELU 5      /* nn.SELU: 1.0507 * (x > 0 ? x : 1.6733 * (exp(x) - 1)) -- the frequency U-Nets (unet_cnns.py:1711-1765); only
                               mpa_act_fwd / mpa_act_bwd take it (not a fused convolution epilogue) */
#define MPA_ACT_ELU 4       /* nn.ELU(alpha=1): x > 0 ? x : exp(x) - 1  (double_conv alt_order, unet_cnns.py:60-70) */

const char* mpa_strerror(int code);
int mpa_version(void);
/* Diagnostic switches (csrc/mpa_diag.h: MPA_FWD_FORCE, MPA_WG_VARIANT, MPA_HEAD_OFF ... -- tile / variant overrides for
 * tests and timing scripts, none needed in production) are read from the environment once per process; a process that
 * changes one of them afterwards calls this to read them again.  Returns 1 in a -DMPA_DIAG build (kernel-side debug
 * switches compiled in), 0 in the release library. */
int mpa_diag_reload(void);

/* ------------------------------------------------------------------ convolution
 * Replaces nn.Conv2d in double_conv (unet_cnns.py:49-59), conv1/prefilt_list
 * (basic_cnns.py:371-387), conv2/conv3/conv4 heads (unet_cnns.py:538-557) and
 * convP (unet_cnns.py:2311-2318).  Cross-correlation, zero padding.
 * Implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32).                         */
typedef struct mpa_conv_desc {
  int32_t B, Cin, H, W;     /* input  (B,Cin,H,W)                 */
  int32_t Cout, kh, kw;     /* weight (Cout,Cin,kh,kw)            */
  int32_t sh, sw, ph, pw;   /* stride, zero padding               */
} mpa_conv_desc;

/* number of floats of the packed filter bank used by fwd (mode 0) / bwd-data (mode 1) / fwd with the cout remainder fold (mode 2) */
int64_t mpa_conv2d_packed_floats(const mpa_conv_desc* d, int mode);
/* repack (Cout,Cin,kh,kw) filters for fwd (mode 0) or, flipped+transposed, for bwd-data (mode 1) */
int mpa_conv2d_pack(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, void* stream);
/* All filter banks of a model in one launch (the training step re-packs every bank after each optimizer step).
 * mpa_conv2d_pack_entry fills one table entry (mpa_conv2d_pack_entry_bytes() bytes of host memory) for what
 * mpa_conv2d_pack(d, mode, w, w_packed) would do; the caller copies the entries, back to back, into device memory once
 * and mpa_conv2d_pack_many(device_table, n) re-packs all of them (w and w_packed are read from the entries: they must
 * stay where they were). */
int mpa_conv2d_pack_entry_bytes(void);
int mpa_conv2d_pack_entry(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, void* host_entry);
int mpa_conv2d_pack_many(const void* device_table, int n, void* stream);
/* Forward pass of a 15x15 stride-1 layer whose output channels are not a multiple of 16 (70 = 64 + 6, basic_cnns.py:371-387)
 * as two launches into y: channels [0, C0) as an ordinary convolution, the R <= 8 remaining channels as V * R rows of one
 * 16-row MFMA tile (V = 2 or 4 vertically adjacent output rows per channel).  w_packed from mpa_conv2d_pack(d, 2, ...)
 * (mpa_conv2d_packed_floats(d, 2) floats).  mpa_conv2d_fold_supported: 1 when the layer qualifies.  Backward-data applies
 * the same split by itself (mode 1 banks). */
int mpa_conv2d_fold_supported(const mpa_conv_desc* d);
int mpa_conv2d_fwd_folded(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y, int act,
                          float slope, void* stream);
/* y = act(conv(x, w) + bias) ; bias may be NULL */
int mpa_conv2d_fwd(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                   float* y, int act, float slope, void* stream);
/* Forward convolution in front of a BatchNorm2d (double_conv, unet_cnns.py:49-59): as mpa_conv2d_fwd without activation,
 * and the store epilogue leaves per-(pixel tile, channel) partial sums of y and y^2 in `partials`
 * ([mpa_conv2d_fwd_stats_rows(d)][Cout][2] floats), so that the batch statistics need no extra pass over y
 * (mpa_bn_relu_train_fwd_partials reduces them in a fixed order, in float64).                                  */
int64_t mpa_conv2d_fwd_stats_rows(const mpa_conv_desc* d);
int mpa_conv2d_fwd_stats(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                         float* partials, void* stream);
/* dx = conv_transpose(dy, w)  (w_packed from mode 1) */
int mpa_conv2d_bwd_data(const mpa_conv_desc* d, const float* dy, const float* w_packed, float* dx, void* stream);
/* human-readable tiling chosen for fwd (mode 0), bwd-data (1), bwd-weight (2), fwd with the cout remainder fold (3) --
 * diagnostics / DESIGN.md tables */
int mpa_conv2d_describe_plan(const mpa_conv_desc* d, int mode, char* buf, int buflen);
/* dw = sum_b,y,x dy * x ; db = sum dy (db may be NULL).  workspace bytes from the helper. */
int64_t mpa_conv2d_bwd_weight_workspace(const mpa_conv_desc* d);
int mpa_conv2d_bwd_weight(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ split-bf16 ("bf16x3") convolution, opt-in
 * Same call sites as above, for the 15-row filters (double_conv of the 75x216 / 37x108 levels, unet_cnns.py:49-59;
 * conv1 / prefilt_list, basic_cnns.py:371-387), stride 1.  Every fp32 operand is carried as hi + lo bf16 halves and a
 * product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (error ~2e-5 of the result's rms:
 * inside the 1e-4 forward bound; NOT bit-identical to the exact-fp32 path, which stays the default).
 * Split activation layout: [B][ceil(C/8)][hi|lo][H][W][8 channels] bf16 (mpa_bf16x3_split_bytes() bytes).        */
int64_t mpa_bf16x3_split_bytes(int B, int C, int H, int W);
int mpa_bf16x3_split(const float* x, void* out, int B, int C, int H, int W, void* stream);
/* 1 when the kernels are built for this problem (mode 0: forward, 1: backward-data, 2: backward-weight), else 0 */
int mpa_conv2d_bf16x3_supported(const mpa_conv_desc* d, int mode);
int64_t mpa_conv2d_bf16x3_packed_bytes(const mpa_conv_desc* d, int mode);
int mpa_conv2d_bf16x3_pack(const mpa_conv_desc* d, int mode, const float* w, void* w_packed, void* stream);
/* rows of `partials` ([rows][Cout][2], as mpa_conv2d_fwd_stats) the forward writes when partials != NULL */
int64_t mpa_conv2d_bf16x3_stats_rows(const mpa_conv_desc* d);
/* y = act(conv(x, w) + bias); xs from mpa_bf16x3_split(x), w_packed from mode 0; partials nullable (then act must be NONE) */
int mpa_conv2d_bf16x3_fwd(const mpa_conv_desc* d, const void* xs, const void* w_packed, const float* bias, float* y, int act,
                          float slope, float* partials, void* stream);
/* dx = conv_transpose(dy, w); dys from mpa_bf16x3_split(dy) (Cout channels), w_packed from mode 1 */
int mpa_conv2d_bf16x3_bwd_data(const mpa_conv_desc* d, const void* dys, const void* w_packed, float* dx, void* stream);
/* dw = sum dy * x, db = sum dy (db nullable) from the split input and the split output gradient; fixed-order reduction */
int64_t mpa_conv2d_bf16x3_bwd_weight_workspace(const mpa_conv_desc* d);
int mpa_conv2d_bf16x3_bwd_weight(const mpa_conv_desc* d, const void* xs, const void* dys, float* dw, float* db,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ normalisation
 * Input LayerNorm([C,F]) applied on x.transpose(1,2) (unet_cnns.py:505,560;
 * basic_cnns.py:160,190): every (b,t) slice of C*F values is normalised jointly. */
int mpa_layernorm_cf_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                         int B, int C, int T, int F, float eps, void* stream);
/* bytes of the dw/db partial-sum workspace of both LayerNorm backward kernels (n = number of affine elements) */
int64_t mpa_layernorm_bwd_workspace(int n);
int mpa_layernorm_cf_bwd_ws(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                            float* dx /*nullable*/, float* dw, float* db, void* ws, int B, int C, int T, int F, void* stream);
/* LayerNorm over the last dim of (rows,E) with fused residual: y = LN(a + r) (unet_cnns.py:156,158) */
int mpa_layernorm_rows_fwd(const float* a, const float* r /*nullable*/, const float* w, const float* b, float* sum_out,
                           float* y, float* mean, float* rstd, int64_t rows, int E, float eps, void* stream);
int mpa_layernorm_rows_bwd_ws(const float* dy, const float* xs, const float* w, const float* mean, const float* rstd,
                              float* dx, float* dw, float* db, void* ws, int64_t rows, int E, void* stream);

/* nn.BatchNorm2d + nn.ReLU of double_conv (unet_cnns.py:51-52,55-56).
 * train: batch statistics (biased var), running stats updated with momentum (unbiased var);
 * stats workspace: 2*C doubles (zeroed inside). relu!=0 fuses the ReLU.                        */
int mpa_bn_relu_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked /*nullable, += 1*/, float* y, float* save_mean,
                          float* save_invstd, double* stats_ws, int B, int C, int HW, float momentum, float eps, int relu,
                          void* stream);
/* as mpa_bn_relu_train_fwd with the batch statistics taken from the producing convolution's partial sums;
 * stage_ws (nullable: single-stage reduction): 64 * C * 2 doubles of scratch for the two-stage reduction of many rows */
int mpa_bn_relu_train_fwd_partials(const float* x, const float* partials, int rows, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                   float* save_mean, float* save_invstd, double* stage_ws, int B, int C, int HW,
                                   float momentum, float eps, int relu, void* stream);
int mpa_bn_relu_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float* y, float* save_mean /*nullable*/, float* save_invstd /*nullable*/,
                         int B, int C, int HW, float eps, int relu, void* stream);
/* dx, dgamma, dbeta from dy (grad w.r.t. the post-ReLU output y). train!=0: batch-stat backward.
 * stats_ws: 128 * C doubles of scratch (up to 64 partial pairs of sums per channel, added in a fixed order: no zero-fill,
 * no atomics -- run-to-run reproducible).
 * ReLU mask: with beta != NULL it is recomputed from x with the forward's own arithmetic and y is never read
 * (y may then be NULL); with beta == NULL it is y > 0.                                                          */
int mpa_bn_relu_bwd(const float* dy, const float* x, const float* y /*nullable*/, const float* gamma,
                    const float* beta /*nullable*/, const float* save_mean, const float* save_invstd, float* dx,
                    float* dgamma, float* dbeta, double* stats_ws, int B, int C, int HW, int relu, int train,
                    void* stream);

/* The training passes above in two halves with the per-channel sums handed to the caller in between (optional
 * exactness mode "SyncBN"): a data-parallel rank all-reduces `sums` (2*C doubles: sum x, sum x^2 -- or sum g, sum g*xhat
 * in backward) over the ranks and passes the *global* element count to the second half, so that an N x B/N run normalises
 * with the statistics of the whole batch.  dgamma / dbeta of mpa_bn_relu_bwd_sums are this rank's share (the gradient
 * averager sums them).  The ReLU mask is recomputed from x (beta must be given).                                 */
int mpa_bn_batch_sums(const float* x, double* sums, int B, int C, int HW, void* stream);
int mpa_bn_relu_train_fwd_sums(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                               float* save_mean, float* save_invstd, int B, int C, int HW, float momentum, float eps, int relu,
                               void* stream);
int mpa_bn_relu_bwd_sums(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, double* stats, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                         void* stream);
int mpa_bn_relu_bwd_apply(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, const double* stats, double count, float* dx, int B, int C, int HW,
                          int relu, void* stream);

/* ------------------------------------------------------------------ pooling / upsampling
 * nn.MaxPool2d (unet_cnns.py:511-526; basic_cnns.py:376,393; unet_cnns.py:2314): -inf padding, floor mode.
 * idx holds the flat input offset (within the H*W plane) of each window's first maximum.              */
int mpa_maxpool2d_fwd(const float* x, float* y, int32_t* idx, int B, int C, int H, int W, int kh, int kw,
                      int sh, int sw, int ph, int pw, void* stream);
int mpa_maxpool2d_bwd(const float* dy, const int32_t* idx, float* dx, int B, int C, int H, int W, int kh, int kw,
                      int sh, int sw, int ph, int pw, void* stream);
/* dx = maxpool backward of dy + add: the tensor that was pooled also fed a second consumer (the U-Net's skip connection,
 * unet_cnns.py:562-571: x1..x4 go into the next level's MaxPool2d and into unet_up_concat_padding) and `add` is that
 * consumer's gradient -- C planes of H*W floats per sample, samples add_batch_stride floats apart (the first Cs channels of
 * the concatenated gradient, read in place).  One pass instead of the pool backward, a slice copy and an add.            */
int mpa_maxpool2d_bwd_add(const float* dy, const int32_t* idx, const float* add /*nullable*/, int64_t add_batch_stride,
                          float* dx, int B, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, void* stream);
/* nn.MaxUnpool2d(kernel) with the indices of a non-overlapping MaxPool2d(kernel, return_indices=True) (stride == kernel, no
 * padding; unet_cnns.py:1751-1765): y (B,C,OH*kh,OW*kw), y[idx[o]] = x[o], zeros elsewhere -- in gather form (every output
 * position belongs to exactly one window), so no zero fill and no scatter.  Backward: dx[o] = dy[idx[o]].               */
int mpa_maxunpool2d_fwd(const float* x, const int32_t* idx, float* y, int B, int C, int OH, int OW, int kh, int kw, void* stream);
int mpa_maxunpool2d_bwd(const float* dy, const int32_t* idx, float* dx, int B, int C, int OH, int OW, int kh, int kw, void* stream);
/* unet_up_concat_padding (unet_cnns.py:93-104): out = cat([skip, pad(bilinear_x2_align_corners(x1))], dim=1) */
int mpa_upcat_fwd(const float* x1, const float* skip, float* out, int B, int C1, int H1, int W1, int Cs, int Hs,
                  int Ws, void* stream);
/* the same with upsampling factors (fh, fw) <= 4 instead of (2, 2): unet_up_concat_padding((2,3)) of the temporal U-Nets
 * (unet_cnns.py:1185, 1326); (2, 2) takes the kernels above */
int mpa_upcat_scaled_fwd(const float* x1, const float* skip, float* out, int B, int C1, int H1, int W1, int Cs, int Hs,
                         int Ws, int fh, int fw, void* stream);
int mpa_upcat_scaled_bwd(const float* dout, float* dx1, float* dskip /*nullable*/, int B, int C1, int H1, int W1, int Cs,
                         int Hs, int Ws, int fh, int fw, void* stream);
/* dskip == NULL: only dx1 is formed (the caller reads the skip half dout[:, :Cs] in place, mpa_maxpool2d_bwd_add) */
int mpa_upcat_bwd(const float* dout, float* dx1, float* dskip /*nullable*/, int B, int C1, int H1, int W1, int Cs, int Hs,
                  int Ws, void* stream);

/* ------------------------------------------------------------------ pointwise
 * nn.LeakyReLU / nn.ReLU / nn.Sigmoid / nn.Dropout / residual adds                                         */
int mpa_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream);
int mpa_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, float slope, void* stream);
/* nn.LogSoftmax(dim=1) over the channels of cat([a, b], dim=3): a (B,C,R,Wa), b (B,C,R,Wb) or NULL with Wb = 0, y (B,C,R,Wa+Wb)
 * -- the output stage of basic_cnn_segm_logsoftmax / basic_cnn_segm_blank_logsoftmax (basic_cnns.py:254-255, 331-338).
 * Backward from y: da / db = dy - exp(y) * sum_c dy. */
int mpa_logsoftmax_cat_fwd(const float* a, const float* b, float* y, int B, int C, int R, int Wa, int Wb, void* stream);
int mpa_logsoftmax_cat_bwd(const float* dy, const float* y, float* da, float* db, int B, int C, int R, int Wa, int Wb,
                           void* stream);
/* rng_state (device memory): [0] seed, [1] base offset of the current training step; `offset` = position of this call's
 * elements inside the step.  Element i draws from the counter-based stream at rng_state[1] + offset + i.        */
int mpa_dropout(const float* x, float* y, int64_t n, float p, const uint64_t* rng_state, uint64_t offset, void* stream);
/* nn.MaxPool2d((kh,1), stride 1, padding (kh/2,0)) -> nn.Dropout(p) [-> + residual] in one pass, kh = 3 or 13: the tail
 * of the CNN families' prefilter stages (kh = 3, basic_cnns.py:374-377; the residual of deep_cnn_segm_sigmoid.forward,
 * basic_cnns.py:414-418) and of every model's head stage conv2 (kh = 13, basic_cnns.py:380-385, unet_cnns.py:538-543).
 * h, out, residual: [planes][H][W] fp32; residual may be NULL; p == 0 (evaluation) skips the mask and needs no rng_state.
 * The mask is mpa_dropout's for the same (rng_state, offset).  which ([planes][H][W] int8, may be NULL when no gradient
 * is needed) records the window row 0..kh-1 of each maximum (bits 0-3) and whether it is positive (bit 6 set: not) for mpa_poolrows_dropout_bwd, which returns d/dh
 * (d/dresidual is dout).  Other kh: MPA_ERR_UNSUPPORTED. */
int mpa_poolrows_dropout_add_fwd(const float* h, const float* residual, float* out, int8_t* which, int64_t planes, int H,
                                 int W, int kh, float p, const uint64_t* rng_state, uint64_t offset, void* stream);
int mpa_poolrows_dropout_bwd(const float* dout, const int8_t* which, float* dh, int64_t planes, int H, int W, int kh, float p,
                             const uint64_t* rng_state, uint64_t offset, void* stream);
/* the same with the backward pass of the ReLU / LeakyReLU folded in that the convolution producing h applied in its epilogue
 * (nn.Sequential(Conv2d, LeakyReLU, MaxPool2d, Dropout): basic_cnns.py:371-385, unet_cnns.py:538-543): d/d(pre-activation) =
 * (d/dh) * (h > 0 ? 1 : neg_slope), the sign taken from the flag mpa_poolrows_dropout_add_fwd leaves in `which`. */
int mpa_poolrows_dropout_act_bwd(const float* dout, const int8_t* which, float* dh, int64_t planes, int H, int W, int kh, float p,
                                 const uint64_t* rng_state, uint64_t offset, float neg_slope, void* stream);
/* table[i] = host_ptrs[i], i < n: device pointer table written by kernels whose arguments carry the pointers (no
 * memcpy, nothing for the host to keep alive; capturable in a HIP graph)                                       */
int mpa_store_ptrs(const void** table, const void* const* host_ptrs, int n, void* stream);
/* dst[dst_offsets[i] .. + sizes[i]) = srcs[i][0 .. sizes[i]) for i < n: many tensors into one flat buffer with one
 * launch per 32 tensors (the gradient buckets of the data-parallel averager; host arrays, read at launch)          */
int mpa_gather_copy(float* dst, const float* const* srcs, const int64_t* dst_offsets, const int64_t* sizes, int n,
                    void* stream);
/* counter[0] += delta on the device (end-of-step advance of the dropout stream; capturable in a HIP graph)      */
int mpa_u64_add(uint64_t* counter, uint64_t delta, void* stream);
int mpa_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int mpa_axpy(float alpha, const float* x, float* y, int64_t n, void* stream);           /* y += alpha*x */
int mpa_scale(float alpha, float* x, int64_t n, void* stream);
int mpa_scale_by(const float* x, const float* g /*device scalar*/, float* y, int64_t n, void* stream);
/* x (B,E,S) <-> (B,S,E) transposes of transformer_enc_layer.forward (unet_cnns.py:150,159); pe nullable (S,E) */
/* x (B,R,Cc) -> y (B,Cc,R); pe_mode 0: none, 1: add pe (Cc,R) to the output, 2: add pe (R,Cc) to the input */
int mpa_transpose_add(const float* x, const float* pe, float* y, int B, int R, int Cc, int pe_mode, void* stream);
/* per-channel sum over (B,HW): conv bias gradient */
int mpa_channel_sum(const float* x, float* out, int B, int C, int HW, void* stream);
/* broadcast positional encoding add: y[b,s,:] = x[b,s,:] + pe[s,:] */
int mpa_add_rows_bcast(const float* x, const float* pe, float* y, int B, int64_t SE, void* stream);

/* ------------------------------------------------------------------ GEMM (nn.Linear, LSTM projections)
 * C[M,N] (+)= A[M,K] * op(B) (+ bias[N]) with act; A(m,k)=A[m*lda_m+k*lda_k], B(k,n)=Bm[k*ldb_k+n*ldb_n]. */
int mpa_gemm(const float* A, int64_t lda_m, int64_t lda_k, const float* Bm, int64_t ldb_k, int64_t ldb_n,
             const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate, int act, void* stream);
/* C[M,N] = (A B) where mask > 0, else 0: the input gradient of a Linear layer whose input is a ReLU output (mask, laid out like the
 * contiguous C) with that ReLU's backward pass folded in -- dh = (dy W2) * (h > 0) of the transformer MLP, libdl/nn_models/
 * unet_cnns.py:137-141,176.  Strides as mpa_gemm.  Fused in the short-K panel kernel for K = 128 with B n-contiguous; any other
 * shape runs mpa_gemm followed by mpa_act_bwd in place (same values). */
int mpa_gemm_masked(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                    const float* mask, float* C, int M, int N, int K, void* stream);

/* nbatch <= 4 products of identical shape and strides in one launch (the q/k/v projections of transformer_enc_layer and
 * the three in-projections of nn.MultiheadAttention, unet_cnns.py:131-135,153): problem b uses A[b], B[b], bias[b], C[b]
 * (host arrays of device pointers, read at launch).  shared_c != 0: every C[b] is the same matrix and receives the SUM of
 * the products (input gradient of three projections of one tensor); it is zeroed first and added to atomically.   */
int mpa_gemm_batched(int nbatch, const float* const* A, int64_t lda_m, int64_t lda_k, const float* const* B, int64_t ldb_k,
                     int64_t ldb_n, const float* const* bias /*nullable*/, float* const* C, int64_t ldc, int M, int N, int K,
                     int shared_c, int act, void* stream);
/* opt-in split-bf16 variant of mpa_gemm (hi*hi + hi*lo + lo*hi on bf16 MFMA, fp32 accumulation: conv_bf16x3.hip says what
 * the mode trades): 16-byte aligned operands with unit stride along k or along the other dimension, K % 32 == 0; otherwise
 * mpa_gemm_bf16x3_supported returns 0 / mpa_gemm_bf16x3 MPA_ERR_UNSUPPORTED and the caller stays on mpa_gemm               */
int mpa_gemm_bf16x3_supported(const float* A, int64_t lda_m, int64_t lda_k, const float* Bm, int64_t ldb_k, int64_t ldb_n, int M,
                              int N, int K);
int mpa_gemm_bf16x3(const float* A, int64_t lda_m, int64_t lda_k, const float* Bm, int64_t ldb_k, int64_t ldb_n,
                    const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate, int act, void* stream);
/* column sums of a (rows, N) matrix (Linear bias gradients) */
int mpa_colsum(const float* x, float* out, int64_t rows, int N, int accumulate, void* stream);

/* ------------------------------------------------------------------ attention over the batch axis
 * nn.MultiheadAttention fed (B,S,E) with batch_first=False (unet_cnns.py:134,153):
 * for every position s and head, softmax((q*d^-1/2) k^T) over the B samples.  q,k,v,o: (B,S,E).           */
int mpa_attn_batchaxis_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int S,
                           int E, int heads, void* stream);
int mpa_attn_batchaxis_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse,
                           const float* do_, float* dq, float* dk, float* dv, int B, int S, int E, int heads,
                           void* stream);

/* Bq queries against Bk keys / values (k, v, dk, dv: (Bk,S,E); q, o, do, dq: (Bq,S,E); lse: (S,heads,Bq)): a data-parallel
 * rank that has gathered the keys and values of all ranks attends over the whole batch (optional exactness mode) */
int mpa_attn_batchaxis_fwd_kv(const float* q, const float* k, const float* v, float* o, float* lse, int Bq, int Bk, int S,
                              int E, int heads, void* stream);
int mpa_attn_batchaxis_bwd_kv(const float* q, const float* k, const float* v, const float* o, const float* lse,
                              const float* do_, float* dq, float* dk, float* dv, int Bq, int Bk, int S, int E, int heads,
                              void* stream);

/* ------------------------------------------------------------------ LSTM cell (nn.LSTM, unet_cnns.py:232)
 * gates (B,4H) pre-activations in order i,f,g,o; c_prev nullable (zeros).                                  */
int mpa_lstm_cell_fwd(const float* gates, int64_t g_stride, const float* c_prev, float* c, float* h, int64_t h_stride,
                      float* acts, int B, int H, void* stream);
int mpa_lstm_cell_bwd(const float* dh, int64_t dh_stride, const float* dh_rec /*nullable*/, const float* dc_next /*nullable*/,
                      const float* acts, const float* c_prev, const float* c, float* dgates, int64_t dg_stride,
                      float* dc_prev, int B, int H, void* stream);

/* ------------------------------------------------------------------ losses (caller side, exp126a...py:87; exp195f...py:331-334)
 * BCELoss(mean) on probabilities with the -100 clamp; loss_out[0] = sum / n (zeroed by the call).                 */
int mpa_bce_fwd(const float* p, const float* y, float* loss_out, int64_t n, void* stream);
/* dp = g[0] * d(mean BCE)/dp ; g is a device scalar (the upstream gradient), nullable = 1 */
int mpa_bce_bwd(const float* p, const float* y, float* dp, int64_t n, const float* g, void* stream);
/* CrossEntropyLoss(mean) over rows of (B,K) logits with int64 targets; loss_out[0] = scale*mean (zeroed by the call). */
int mpa_ce_fwd_bwd(const float* logits, const int64_t* target, float* loss_out, float* dlogits, int B, int K,
                   float scale, void* stream);

/* ------------------------------------------------------------------ AdamW (exp126a...py:103-108,293)
 * multi-tensor step: tensor lists as device-resident pointer tables.  `hyper` is device memory, double[4]:
 * [0] learning rate (written by the host / ReduceLROnPlateau), [1] steps taken so far (the call increments it),
 * [2],[3] scratch for the bias corrections -- nothing that changes from step to step is a kernel argument, so the
 * call can be captured once in a HIP graph and replayed.                                                      */
int mpa_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* sizes, int ntensors, int64_t max_size, double* hyper, double beta1, double beta2,
                   double eps, double weight_decay, void* stream);

/* ------------------------------------------------------------------ patch extraction + augmentation
 * replaces libdl/data_loaders/hcqt_datasets.py:67-141 (dataset_context.__getitem__) and :199-289
 * (dataset_context_segm.__getitem__ without time scaling), for B patches per launch.  Recordings are resident
 * (n_harm, T_file, n_bins) fp32 tensors, targets (T_file, n_out) fp32.                                          */
enum { MPA_CTX_EQ = 1, MPA_CTX_NOISE = 2, MPA_CTX_LOG = 4, MPA_CTX_TUNE = 8, MPA_CTX_TRANSP = 16,
       MPA_CTX_SEGM_TARGETS = 32 /* dataset_context_segm's 4-D target: transposition clears frames, not bins (:273-277) */ };
typedef struct mpa_context_desc {
  int32_t n_harm;     /* 6 */
  int32_t n_bins;     /* 216 */
  int32_t frames;     /* frames per patch: 2*(context/2) + seglength */
  int32_t n_out;      /* target bins: 72 (pitch) or 12 (pitch class: circular roll) */
  int32_t seglength;  /* target rows per patch (1 for dataset_context) */
  int32_t flags;      /* MPA_CTX_* stages to apply, in the reference's order */
  float compression;  /* gamma of log(1 + gamma*x) */
  float noisestd;     /* 'aug:noisestd' */
} mpa_context_desc;
/* src[b]: device address of (harmonic 0, first frame of window b, bin 0); chan_stride[b]: elements between harmonics;
 * tgt[b]: device address of the first target row of window b.  aug: (B,4) int32 = alpha, beta ('aug:randomeq'
 * parabola, already accepted by the reference's non-negativity test), tuning shift in half bins (-2..2), transposition
 * in semitones.  n1 (B,n_harm,frames,n_bins), n2 (B,n_harm,frames), n3 (B,n_harm,frames,15): explicit Gaussian draws
 * (already scaled by their std) for parity tests; NULL = counter-based generator seeded by `seed`.
 * X: (B,n_harm,frames,n_bins), y: (B,1,seglength,n_out).                                                        */
int mpa_context_batch(const mpa_context_desc* d, int B, const uint64_t* src, const int64_t* chan_stride,
                      const uint64_t* tgt, const int32_t* aug, const float* n1, const float* n2, const float* n3,
                      uint64_t seed, float* X, float* y, void* stream);

/* 'aug:scalingfactor' of dataset_context_segm (libdl/data_loaders/hcqt_datasets.py:211-225): src = (harmonic 0, first frame
 * of the window incl. context, bin 0) of an (n_harm, >= half_context + seglength + half_context, n_bins) window with
 * chan_stride elements between harmonics; the seglength frames in the middle are resampled to new_len frames (linear,
 * at linspace(0, seglength - 1, new_len)), the context halves copied.  out: (n_harm, new_len + 2 half_context, n_bins). */
int mpa_time_scale(const float* src, int64_t chan_stride, int n_harm, int n_bins, int half_context, int seglength,
                   int new_len, float* out, void* stream);

/* ------------------------------------------------------------------ evaluation measures
 * replaces libdl/metrics/eval_metrics.py:8-116 (calculate_single_measure; the 11 measures of exp180d...py:150-151)
 * incl. libfmp/c5/c5s2_chord_rec_template.py:238-261 and libfmp/c3/c3s1_post_processing.py:60-68.
 * targ, pred: (n_frames, n_bins) fp32 device arrays of one recording; arithmetic in float64 like the reference.
 * out (device, 16 doubles): [0..10] precision, recall, f_measure, cosine_sim, binary_crossentropy,
 * euclidean_distance, binary_accuracy, soft_accuracy, accum_energy, roc_auc_measure, average_precision_score;
 * [11..13] TP, FP, FN; [14..15] number of positive / negative targets (roc_auc is not finite if either is 0).   */
int64_t mpa_eval_measures_workspace(int64_t n_frames, int n_bins);
int mpa_eval_measures(const float* targ, const float* pred, int64_t n_frames, int n_bins, double threshold,
                      double* out, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ note list -> piano roll
 * replaces compute_annotation_array_nooverlap (libdl/data_preprocessing/hcqt.py:205-272; called by
 * 01_precompute_features.ipynb cell 7).  note_events: device array [n_events][ev_stride >= 3] of float64 rows
 * (start_sec, end_sec, pitch, ...); kind 0 'pitch_class' (12 rows), 1 'pitch' (128 rows), 2 'instruments' (1 row);
 * out: device float64 [rows][n_frames] (zeroed inside).  Integer / index work: bit-exact with the reference.
 * workspace: mpa_annotation_workspace(n_events) bytes; its first int32 is the status the caller must read after the
 * stream has finished: 0 ok, 1 the reference's assertion fired ("still events of length<1 after correction!"),
 * 2 a row index out of bounds (numpy's IndexError), 3 more than 8192 vanishing events (not built).              */
int64_t mpa_annotation_workspace(int n_events);
int mpa_annotation_array_nooverlap(const double* note_events, int ev_stride, int n_events, double fs_hcqt, double shorten,
                                   int kind, int n_frames, double* out, void* workspace, int64_t workspace_bytes,
                                   void* stream);

/* ------------------------------------------------------------------ HCQT front-end (second half) -- PARITY UNPINNED
 * The arithmetic of compute_efficient_hcqt (libdl/data_preprocessing/hcqt.py:89-164).  The reference delegates to librosa 0.8
 * (librosa.cqt, librosa.estimate_tuning -- third-party, absent from the image): these entry points implement the published
 * algorithm as restated in oracle/restate_hcqt.py (librosa's constant-Q filter bank evaluated directly at the original rate
 * instead of by its octave-wise resampling recursion) and are checked against that restatement only.  The products
 * "signal frames x basis" run through mpa_gemm with a strided A operand (A(m,k) = y[m*hop + k]).                          */
/* out[i] = y[reflect(i - pad_l)], i < n + pad_l + pad_r (numpy mode="reflect") */
int mpa_reflect_pad(const float* y, int64_t n, int64_t pad_l, int64_t pad_r, float* out, void* stream);
/* B [n_fft][2*(n_fft/2+1)]: periodic-Hann-windowed DFT basis, columns (re, -im) per frequency bin */
int mpa_stft_basis(float* B, int n_fft, void* stream);
/* S[i] = |(C[2i], C[2i+1])| */
int mpa_complex_mag(const float* C, float* S, int64_t n, void* stream);
/* librosa piptrack on S [frames][n_fft/2+1]: candidates (pitch in Hz, interpolated magnitude) appended to pitch / mag
 * (capacity frames * ((nb+1)/2)), their number in *count (device int)                                                    */
int mpa_piptrack(const float* S, int64_t frames, int nb, double sr, int n_fft, double fmin, double fmax, double threshold,
                 double* pitch, float* mag, int* count, void* stream);
/* librosa estimate_tuning's tail: candidates at or above the median magnitude -> pitch_tuning histogram -> *tuning_out
 * (device double, fraction of a bin in [-0.5, 0.5)); n = the count read back from mpa_piptrack                           */
int64_t mpa_pitch_tuning_workspace(int64_t n);
int mpa_pitch_tuning(const double* pitch, const float* mag, int64_t n, int bins_per_octave, double resolution,
                     double* tuning_out, void* ws, int64_t ws_bytes, void* stream);
/* constant-Q basis of nb consecutive bins starting at frequency f0: B [K][ncols], columns (re, -im) per bin, row k = sample
 * offset k - K0 from the frame centre; filters of librosa.filters.constant_q (Hann, L1-normalised) times sqrt(length)      */
int mpa_cqt_basis(float* B, int64_t K, int64_t K0, int ncols, double f0, int nb, int bins_per_octave, double sr, void* stream);
/* magnitudes of one bin group C [frames][ncols] into the HCQT tensor out [n_bins_out][frames][n_harm]: bin bin0+j becomes
 * row bin0+j-fac_bins[m] of harmonic hidx[m] for each of the nmem <= 8 members whose slice contains it (host arrays)     */
int mpa_cqt_mag_scatter(const float* C, int64_t frames, int ncols, int nb, int bin0, float* out, int n_bins_out, int n_harm,
                        const int* fac_bins, const int* hidx, int nmem, void* stream);

#ifdef __cplusplus
}
#endif
#endif
