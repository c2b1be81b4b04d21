/*
 * misamd — C-ABI of the MI355X (gfx950) U-Net training hot path.
 *
 * The reference project (Runner-xc/Medical_image_segmentation) is pure Python on
 * torch ATen and has no FFI of its own; each entry point below replaces the ATen
 * call sequence at the cited reference lines.  Paths are relative to the
 * reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the library never
 *     allocates, frees or synchronises: the caller owns all memory (workspaces
 *     included, sized by the *_workspace_bytes queries) and passes the HIP stream
 *     the work is enqueued on (hipStream_t as void*; NULL = default stream);
 *   - activations are channels-last:  x[b][y][x][c]  ("NHWC"), element type given
 *     by a mis_dtype (float32 or bfloat16); all accumulation is float32;
 *   - parameters and their gradients are float32 in the reference's layouts
 *     (conv weight [Cout][Cin][3][3], as nn.Conv2d stores it);
 *   - targets are int64 class indices [B][H][W], counts are int64;
 *   - "lazy activations": where an argument `X_affine` accompanies an activation
 *     pointer X, a non-NULL value is a float32 [2][C] table (scale[C] then shift[C])
 *     and the kernel reads relu(X*scale + shift) instead of X — the BatchNorm + ReLU
 *     of the layer that produced X, applied on load so that the normalised
 *     activation is never written to memory.  NULL = X is used as stored;
 *   - every function returns 0 on success or a mis_status; mis_last_error() gives
 *     a thread-local message.  The Python host raises RuntimeError on non-zero,
 *     matching the reference's exception-based error convention.
 */
#ifndef MISAMD_H
#define MISAMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mis_stream_t;

enum mis_status {
  MIS_OK = 0,
  MIS_ERR_ARG = 1,          /* bad shape / null pointer / unsupported size */
  MIS_ERR_LAUNCH = 2,       /* hipLaunch / runtime error */
  MIS_ERR_UNSUPPORTED = 3   /* valid request the library has no kernel for */
};

enum mis_dtype { MIS_F32 = 0, MIS_BF16 = 1 };
/* region loss formed from the pooled sums: diceloss (utils/loss_fn.py:55-88) or IOULoss (:90-124) */
enum mis_loss_kind { MIS_LOSS_DICE = 0, MIS_LOSS_IOU = 1 };

/* library / build probe */
int         mis_version(void);              /* MAJOR*10000 + MINOR*100 + PATCH */
const char* mis_arch(void);                 /* "gfx950" */
const char* mis_last_error(void);

/* ------------------------------------------------------------------------- */
/* Loss + metrics: one pass over the logits.                                   */
/* Replaces utils/loss_fn.py:72-83 (softmax, one_hot, three pooled sums, Dice), */
/* utils/loss_fn.py:199-203 (per-pixel CE map, label_smoothing 1e-7,            */
/* ignore_index 0) and utils/metrics.py:31-34,44-46,87-97 (argmax, one_hot,     */
/* TP/FN/FP/TN counting).                                                       */
/*                                                                              */
/* logits element (b, c, pixel) is read at  b*stride_b + c*stride_c + pixel*stride_p */
/* (elements), so both channels-last (stride_c=1, stride_p=C) and the           */
/* reference's NCHW (stride_c=H*W, stride_p=1) layouts are accepted.            */
/*                                                                              */
/* sums    float32 [3][C]: I_c = sum p_c*[t==c], P_c = sum p_c, T_c = sum [t==c] */
/*         over (B,H,W) with p = softmax(logits) (class 0 included; the Dice     */
/*         loss ignores it).  May be NULL.                                       */
/* ce_map  float32 [B][H][W] or NULL.                                            */
/* cm      int64 [B][C][C], cm[b][t][p] = #pixels with target t and              */
/*         argmax(logits) = p (first maximum on ties; NaN counts as maximal).    */
/*         May be NULL.  Zeroed by the call.                                     */
/* status  int32 [1] device word: set non-zero if a target is outside [0, C).    */
/* 2 <= C <= 16.                                                                 */
size_t mis_loss_metrics_workspace_bytes(int B, int C, int HW);
int mis_loss_metrics_fwd(const void* logits, int dtype, int64_t stride_b, int64_t stride_c, int64_t stride_p,
                         const int64_t* target, int B, int C, int HW,
                         float* sums, float* ce_map, int64_t* cm, int32_t* status,
                         void* workspace, mis_stream_t stream);

/* mis_loss_metrics_fwd for callers that cannot allocate device memory.  It has   */
/* no counterpart in the reference; it is for code running under HIP graph        */
/* capture, tight serving loops and harnesses that forbid hipMalloc: same          */
/* arguments, same kernels and bit-identical results, minus the workspace — the    */
/* scratch (per-workgroup partial sums; the count matrices when cm == NULL) is a    */
/* fixed array inside the library's device image, one copy per GPU, held for the    */
/* life of the process.  The call allocates and frees nothing.  Problems whose      */
/* mis_loss_metrics_workspace_bytes(B, C, HW) exceeds                               */
/* MIS_LOSS_METRICS_STATIC_SCRATCH_BYTES are rejected with MIS_ERR_ARG (for C = 4 */
/* that allows batches of several thousand images of any size).  All calls on one GPU share */
/* that GPU's array: calls that may overlap on one GPU (different streams without    */
/* ordering, several host threads) race on it — serialise them or use the            */
/* workspace form.  The array's address is looked up on the first call on each GPU   */
/* (make that call outside graph capture).                                           */
#define MIS_LOSS_METRICS_STATIC_SCRATCH_BYTES ((size_t)1 << 20)
int mis_loss_metrics_fwd_static(const void* logits, int dtype, int64_t stride_b, int64_t stride_c, int64_t stride_p,
                                const int64_t* target, int B, int C, int HW,
                                float* sums, float* ce_map, int64_t* cm, int32_t* status, mis_stream_t stream);

/* Loss values from the pooled sums — utils/loss_fn.py:81-83 (Dice) and          */
/* :116-118 (IoU).  losses float32 [C]: for i < C-1, with c = i+1 and            */
/* U = P_c + T_c:  Dice  losses[i] = 1 - 2 I_c / (U + smooth);                   */
/*                 IoU   losses[i] = 1 - I_c / (U - I_c + smooth);               */
/* losses[C-1] = total = sum_i losses[i] / C.   kind: enum mis_loss_kind.        */
int mis_dice_from_sums(const float* sums, int C, float smooth, int kind, float* losses, mis_stream_t stream);

/* Gradient of  sum_i w[i]*losses[i]  (+ sum over pixels of dce*ce_map when dce  */
/* is not NULL) with respect to the logits; autograd of loss_fn.py:72-83 (or     */
/* :107-118 for kind = MIS_LOSS_IOU) and :199-203.                               */
/* w float32 [C] device (upstream gradient of each entry of `losses`);           */
/* dlogits has the logits' dtype and strides.                                    */
int mis_loss_bwd(const void* logits, int dtype, int64_t stride_b, int64_t stride_c, int64_t stride_p,
                 const int64_t* target, int B, int C, int HW,
                 const float* sums, const float* w, float smooth, int kind, const float* dce,
                 void* dlogits, mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Down / Up: nn.MaxPool2d(2,2) (model/unet.py:19,48,52,56,60) and             */
/* nn.Upsample(2,'bilinear',align_corners=True) + torch.cat([up, skip], 1)      */
/* (model/unet.py:20,66-67,71-72,76-77,80-81).  Channels-last, H and W even.    */
/*                                                                              */
/* nn.Dropout2d (model/unet.py:27-42; applied to every pooled tensor and to the  */
/* center / decoder1 / decoder2 outputs, unet.py:49-74) is folded into these     */
/* kernels: `noise` / `low_noise` (nullable) is its float32 [B][C] multiplier     */
/* per (sample, channel) — 0 or 1/(1-p), drawn by the caller — applied to the     */
/* pooled output (and to dy on the way back), respectively to the low-resolution  */
/* activation before interpolation (and to dlow on the way back).                 */
int mis_maxpool2x2_fwd(const void* x, const float* x_affine, const float* noise, int dtype, int B, int H, int W, int C,
                       void* y, mis_stream_t stream);
/* dx = gradient routed to the first maximum of each window (row-major scan,    */
/* strict '>'), zero elsewhere; x is the forward input.  dx_add (nullable, may    */
/* alias dx) is a second gradient of x — the encoder output also feeds the       */
/* decoder as a skip connection (model/unet.py:67,72,77,81) — summed in here      */
/* (autograd's accumulation of the two uses, without a separate pass).           */
/* bn_sums (nullable): x is the lazy output of conv+BatchNorm(train)+ReLU and      */
/* x_affine is [4][C] (scale, shift, mean, rstd); the kernel then also produces    */
/* float32 [2][C] = (sum dy, sum dy*xhat) with dy = dx where the activation is     */
/* positive — exactly what mis_bn_relu_bwd_reduce would compute from (x, dx) in a   */
/* separate pass (needs the workspace of mis_maxpool2x2_bwd_workspace_bytes).       */
size_t mis_maxpool2x2_bwd_workspace_bytes(int C);
int mis_maxpool2x2_bwd(const void* x, const float* x_affine, const float* noise, const void* dy, const void* dx_add,
                       int dtype, int B, int H, int W, int C, void* dx, float* bn_sums, void* workspace,
                       mis_stream_t stream);
/* out[b][2h][2w][0:Cup] = bilinear(low), out[...][Cup:Cup+Cskip] = skip.        */
int mis_upsample2x_cat_fwd(const void* low, const float* low_affine, const float* low_noise, const void* skip,
                           const float* skip_affine, int dtype, int B, int h, int w, int Cup, int Cskip, void* out,
                           mis_stream_t stream);
/* dlow[b][h][w][Cup] gathered from dout[..., 0:Cup]; dskip = dout[..., Cup:].   */
int mis_upsample2x_cat_bwd(const void* dout, const float* low_noise, int dtype, int B, int h, int w, int Cup, int Cskip,
                           void* dlow, void* dskip, mis_stream_t stream);

/* The multipliers of one nn.Dropout2d call (model/unet.py:49-74: F.dropout2d draws one     */
/* Bernoulli(1-p) per (sample, channel) and scales the kept maps by 1/(1-p)) from a           */
/* counter-based generator instead of a generator with state: noise[b*C + c] = 1/(1-p) when    */
/* the top 24 bits of word 0 of Philox4x32-10(counter = (b*C + c, 0, offset_lo, offset_hi),    */
/* key = (seed_lo, seed_hi)), as a fraction of 2^24, are below 1-p, else 0.  float32 [B][C],   */
/* consumed by the `noise` / `low_noise` arguments above.  The caller advances `offset` per    */
/* call; equal (seed, offset, b, c) give equal values on every device and launch shape.        */
int mis_dropout2d_noise(uint64_t seed, uint64_t offset, int B, int C, float p, float* noise, mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* DoubleConv building block: Conv2d(3x3, pad 1, bias) -> BatchNorm2d -> ReLU   */
/* (model/utils/modules.py:86-94,97-99).                                        */
/*                                                                              */
/* mis_pack_conv3x3_weights: float32 master weights [Cout_w][Cin_w][3][3] (nn.Conv2d  */
/* layout) -> the operand image the conv kernels read, in the activation dtype.     */
/* transposed = 0: operand of the forward conv (Cin_w -> Cout_w channels);           */
/* transposed = 1: operand of the data gradient  dX = conv(dZ, W^T, taps mirrored)   */
/*                 (Cout_w -> Cin_w channels).                                        */
/* The image layout is private to the library (fragment-ordered for the matrix-core  */
/* kernels, [9][Cout][Cin] for the vector-ALU ones) and depends only on the channel   */
/* counts, dtype and the conv path; size it with mis_conv3x3_packed_weight_bytes.     */
size_t mis_conv3x3_packed_weight_bytes(int Cout_w, int Cin_w, int transposed, int dtype);
int mis_pack_conv3x3_weights(const float* w, int Cout_w, int Cin_w, int transposed, int dtype, void* out,
                             mis_stream_t stream);
/* Both images of one weight tensor (transposed = 0 into out_fwd, transposed = 1 */
/* into out_t) in one launch where the kernel family allows.                     */
int mis_pack_conv3x3_weights_pair(const float* w, int Cout_w, int Cin_w, int dtype, void* out_fwd, void* out_t,
                                  mis_stream_t stream);
/* Test hook: 0 = pick the kernel per shape (default), 1 = vector-ALU kernels only.   */
/* Weights must be re-packed after a change.                                          */
int mis_set_conv_path(int mode);
/* Test / tuning hook: the number of workgroups the matrix-core conv kernels are launched  */
/* with.  0 = sized by the library (one round of resident workgroups, the default);        */
/* n >= 1 = forward / data-gradient launches use n workgroups along the tile axis and the  */
/* weight gradient n workgroups in all, whatever the shape — the tests use small values so  */
/* that every persistent workgroup walks many tiles of a small tensor.  Results do not      */
/* depend on it beyond the summation order of the per-workgroup partial sums.               */
int mis_set_conv_launch(int fwd_workgroups, int wgrad_workgroups);

/* z = conv3x3(x, w) + bias (bias may be NULL), channels-last, same H,W.        */
/* If stats is not NULL it receives float32 [2][Cout]: sum z and sum z^2 over   */
/* (B,H,W), taken from the float32 accumulators (before any rounding to bf16);  */
/* it needs the workspace.  Used for the forward conv (transposed=0 image,       */
/* Cin->Cout) and, with the transposed=1 image and the channel counts swapped,  */
/* for the data gradient.                                                       */
/* Passing stats == workspace skips the final reduction: the workspace then     */
/* starts with mis_conv3x3_stats_rows(...) rows of partial sums [rows][2][Cout]   */
/* for mis_bn_finalize to reduce (one launch fewer per layer).                    */
size_t mis_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* Both images of MANY weight tensors in ONE launch (the parameters change every   */
/* optimizer step, so every step re-packs all of them).  mis_pack_conv3x3_job       */
/* (host only, no GPU work) fills one job for a weight tensor and its two          */
/* destination images and returns the number of workgroups it needs (0: this       */
/* tensor is not handled by the multi-tensor kernel — pack it on its own);         */
/* first_block = sum of the counts returned for the jobs before it.  The caller     */
/* uploads the job array and calls mis_pack_conv3x3_weights_multi with the total.   */
typedef struct {
  const float* w;
  void* out_fwd;
  void* out_t;
  int64_t total_fwd, total_t;
  int Cw_out, Cw_in;
  int cpg_fwd, nt_fwd, cpg_t, nt_t;
  int first_block, n_blocks;
} mis_pack_job;
int mis_pack_conv3x3_job(const float* w, int Cout_w, int Cin_w, int dtype, void* out_fwd, void* out_t, int first_block,
                         mis_pack_job* job);
int mis_pack_conv3x3_weights_multi(const mis_pack_job* jobs, int n_jobs, int total_blocks, int dtype,
                                   mis_stream_t stream);
int mis_conv3x3_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout);
int mis_conv3x3_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype,
                    int B, int H, int W, int Cin, int Cout,
                    void* z, float* stats, void* workspace, mis_stream_t stream);

/* Data gradient of a conv whose INPUT was the lazy activation relu(bn(z_prev)) of   */
/* the previous conv+BatchNorm+ReLU (the second conv of a DoubleConv,                */
/* model/utils/modules.py:97-99): da = conv(dz, transposed weights) as               */
/* mis_conv3x3_fwd computes it (Cin = channels of dz, Cout = channels of da), and    */
/* in the same kernel the BatchNorm-backward sums of that previous layer,            */
/*   bn_sums[0][c] = sum da*[y>0],  bn_sums[1][c] = sum da*[y>0]*xhat                 */
/* (y = z_prev*scale+shift, xhat = (z_prev-mean)*rstd; prev_affine4 = [4][Cout]),     */
/* i.e. what mis_bn_relu_bwd_reduce(z_prev, da) would compute in a pass of its own.   */
int mis_conv3x3_dgrad_bn_sums_supported(int dtype, int B, int H, int W, int Cin, int Cout);
int mis_conv3x3_dgrad_bn_sums(const void* dz, const void* w_packed_t, int dtype, int B, int H, int W, int Cin, int Cout,
                              void* da, const void* z_prev, const float* prev_affine4, float* bn_sums,
                              void* workspace, mis_stream_t stream);

/* dW float32 [Cout][Cin][3][3] = sum_{b,y,x} dz[b,y,x,co] * x[b,y+ky-1,x+kx-1,ci] */
/* (autograd of nn.Conv2d wrt its weight).  Overwrites dw.                       */
/* Rider (dbias not NULL): dbias float32 [Cout] = column sums of bias_rows           */
/* ([n_bias_rows][Cout], the partial rows mis_bn_relu_bwd_dz leaves when its dbias    */
/* argument equals its workspace) — the conv-bias gradient, reduced by extra          */
/* workgroups of this call's own reduction launch instead of a launch of its own.     */
size_t mis_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int mis_conv3x3_wgrad(const void* x, const float* x_affine, const void* dz, int dtype, int B, int H, int W,
                      int Cin, int Cout, float* dw, const float* bias_rows, int n_bias_rows, float* dbias,
                      void* workspace, mis_stream_t stream);

/* The weight gradient in two parts, for a caller that reduces the partial results of SEVERAL   */
/* layers in ONE launch (autograd needs the weight gradients of a step only when the optimizer   */
/* or a gradient exchange reads them: utils/train_and_eval.py:109-118; one small reduction       */
/* launch per layer costs more than the work in it).  mis_conv3x3_wgrad_slabs writes *n_slabs     */
/* partial gradients [n_slabs][9][Cin][Cout] float32 into `slabs`                                 */
/* (mis_conv3x3_wgrad_workspace_bytes, Cin = Ca + Cb; the caller keeps them until the reduction)   */
/* for a one- (xb NULL, Cb 0) or two-tensor input and dilation 1..3; MIS_ERR_UNSUPPORTED when the  */
/* matrix-core kernels do not take the shape (*_supported tells; use mis_conv3x3_wgrad then).       */
/* mis_wgrad_reduce_multi                                                                        */
/* sums each job's slabs in a fixed order into dw [Cout][Cin][3][3] (and bias_rows into dbias, as  */
/* the rider of mis_conv3x3_wgrad); `jobs` is a HOST array.  Results are bit-identical to          */
/* mis_conv3x3_wgrad / mis_conv3x3_cat_wgrad / mis_conv3x3_dilated_wgrad.                          */
typedef struct mis_wgrad_reduce_job {
  const float* slabs;
  float* dw;
  const float* bias_rows;   /* [n_bias_rows][Cout] or NULL */
  float* dbias;             /* [Cout] or NULL */
  int n_slabs, Cin, Cout, n_bias_rows;
} mis_wgrad_reduce_job;
int mis_conv3x3_wgrad_slabs_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout, int dilation);
int mis_conv3x3_wgrad_slabs(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                            const void* dz, int dtype, int B, int H, int W, int Ca, int Cb, int Cout, int dilation,
                            void* slabs, int* n_slabs, mis_stream_t stream);
int mis_wgrad_reduce_multi(const mis_wgrad_reduce_job* jobs, int n_jobs, mis_stream_t stream);

/* The same conv with DILATED taps: nn.Conv2d(kernel_size=3, padding=d, dilation=d), d = 2, 3 */
/* — ResDConv.cbr2 / cbr3, model/utils/modules.py:440-445 (d = 1 is the plain conv).  The data  */
/* gradient of a dilated conv is mis_conv3x3_dilated_fwd on the transposed weight image with    */
/* the same d.  Matrix-core kernels only (no vector-ALU fallback): *_supported tells.           */
/* d = 0: the caller states that only the CENTRE tap of the packed 3x3 weights is non-zero — a */
/* 1x1 conv carried as a 3x3 (AMSFN's conv1x1 / final_conv, modules.py:801,838; ResDConv.c1,    */
/* modules.py:432): the kernels then read no halo and multiply one tap instead of nine (same    */
/* result); the weight gradient (also mis_conv3x3_wgrad_slabs with dilation 0) returns the      */
/* centre tap — the 1x1 gradient — and ZEROS in the other eight.                                */
int mis_conv3x3_dilated_supported(int dtype, int B, int H, int W, int Cin, int Cout, int dilation);
int mis_conv3x3_dilated_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout, int dilation);
int mis_conv3x3_dilated_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype,
                            int B, int H, int W, int Cin, int Cout, int dilation, void* z, float* stats,
                            void* workspace, mis_stream_t stream);
int mis_conv3x3_dilated_wgrad(const void* x, const float* x_affine, const void* dz, int dtype, int B, int H, int W,
                              int Cin, int Cout, int dilation, float* dw, const float* bias_rows, int n_bias_rows,
                              float* dbias, void* workspace, mis_stream_t stream);

/* The decoder's `torch.cat([up(x), skip], dim=1)` followed by a 3x3 conv         */
/* (model/unet.py:66-68, 71-73, 76-78, 80-82) without materialising the          */
/* concatenation: the conv reads its input channels [0,Ca) from xa and [Ca,Ca+Cb) */
/* from xb (two dense channels-last tensors, each with its own optional lazy      */
/* affine), the weight gradient does the same, and the data gradient is written   */
/* as two tensors (the gradient of `up` and the gradient of the skip connection). */
/* w_packed is the ordinary packed image of the [Cout][Ca+Cb][3][3] weights        */
/* (transposed=0 for cat_fwd; transposed=1, producing Ca+Cb channels, for          */
/* split_fwd).  mis_conv3x3_cat_supported returns 1 when all three take the shape */
/* (matrix-core kernels only: equal Ca == Cb that are whole 32-channel (bf16) /    */
/* 16-channel (f32) chunks); otherwise concatenate and use the one-tensor calls.  */
/* Workspaces: mis_conv3x3_workspace_bytes / mis_conv3x3_wgrad_workspace_bytes     */
/* with Cin = Ca+Cb.                                                               */
int mis_conv3x3_cat_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout);
int mis_conv3x3_cat_fwd(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                        const void* w_packed, const float* bias, int dtype, int B, int H, int W,
                        int Ca, int Cb, int Cout, void* z, float* stats, void* workspace, mis_stream_t stream);
int mis_conv3x3_split_fwd(const void* x, const float* x_affine, const void* w_packed, const float* bias, int dtype,
                          int B, int H, int W, int Cin, int Ca, int Cb, void* za, void* zb, void* workspace,
                          mis_stream_t stream);
int mis_conv3x3_cat_wgrad(const void* xa, const float* xa_affine, const void* xb, const float* xb_affine,
                          const void* dz, int dtype, int B, int H, int W, int Ca, int Cb, int Cout,
                          float* dw, const float* bias_rows, int n_bias_rows, float* dbias, void* workspace,
                          mis_stream_t stream);

/* Training-mode BatchNorm2d statistics -> per-channel affine + running stats.   */
/* stats float32 [stats_rows][2][C]: (sum, sum of squares over n = B*H*W values   */
/* per channel), either reduced (stats_rows = 1) or as the per-workgroup partial  */
/* rows a conv launch leaves in its workspace (summed here in a fixed order).     */
/* scale_shift float32 [2][C]: y = z*scale + shift with                          */
/*   scale = gamma / sqrt(var_biased + eps), shift = beta - mean*scale.           */
/* mean_rstd float32 [2][C] saved for the backward.  running_mean/var (may be    */
/* NULL) are updated with `momentum` and the UNBIASED variance, as               */
/* nn.BatchNorm2d does; num_batches_tracked (int64[1], may be NULL) += 1.        */
int mis_bn_finalize(const float* stats, int stats_rows, int C, int64_t n, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* scale_shift, float* mean_rstd, mis_stream_t stream);
/* Eval-mode affine from the running statistics.                                 */
int mis_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma,
                       const float* beta, int C, float eps, float* scale_shift, mis_stream_t stream);
/* a = relu(z*scale + shift); n_pix = B*H*W.                                     */
int mis_bn_relu_apply(const void* z, int dtype, int64_t n_pix, int C, const float* scale_shift,
                      void* a, mis_stream_t stream);
/* Backward of a = relu(bn(z)) in training mode; the ReLU mask is recomputed from  */
/* z (y = xhat*gamma + beta > 0), so the activation itself is not an input.       */
/* Pass 1 reduces  sums[0][c] = sum dy, sums[1][c] = sum dy*xhat  with             */
/*   dy = da*[y>0], xhat = (z-mean)*rstd;  dbeta = sums[0], dgamma = sums[1].      */
/* Pass 2 writes  dz = gamma*rstd*(dy - sums0/n - xhat*sums1/n)  and, when dbias is */
/* not NULL (needs the workspace), dbias[c] = sum over pixels of the stored dz —    */
/* the gradient of the conv bias that feeds this BatchNorm.  With dbias == workspace  */
/* the partial rows [mis_bn_relu_bwd_dz_rows()][C] are left in the workspace          */
/* un-reduced for mis_conv3x3_wgrad's rider (tensors must be 16-byte aligned).        */
size_t mis_bn_relu_bwd_workspace_bytes(int64_t n_pix, int C);
int mis_bn_relu_bwd_dz_rows(int dtype, int64_t n_pix, int C);
int mis_bn_relu_bwd_reduce(const void* z, const void* da, int dtype, int64_t n_pix, int C,
                           const float* mean_rstd, const float* gamma, const float* beta,
                           float* sums, void* workspace, mis_stream_t stream);
int mis_bn_relu_bwd_dz(const void* z, const void* da, int dtype, int64_t n_pix, int C,
                       const float* mean_rstd, const float* gamma, const float* beta, const float* sums,
                       void* dz, float* dbias, void* workspace, mis_stream_t stream);
/* Pass 2 for the layer whose only consumer is a 1x1 conv (UNet.decoder4 -> out_conv,        */
/* model/unet.py:82-84): the incoming gradient da is that conv's data gradient, formed on the  */
/* fly — da[p][c] = sum_j head_dy[p][j] * head_w[j][c], rounded to dtype as mis_conv1x1_bwd      */
/* stores it (bit-identical results) — instead of being written by mis_conv1x1_bwd (dx = NULL    */
/* there; its bn_sums are this call's `sums`) and read back here.  head_cout <= 8.               */
int mis_bn_relu_bwd_dz_head(const void* z, const void* head_dy, const float* head_w, int head_cout, int dtype,
                            int64_t n_pix, int C, const float* mean_rstd, const float* gamma, const float* beta,
                            const float* sums, void* dz, float* dbias, void* workspace, mis_stream_t stream);
/* Pass 2 for a branch layer of the AMSFN fusion (modules.py:846-868): the incoming  */
/* gradient is da[p][c] = g[p][c]*g_scale[b][c] + g_shift[b][c] (b = p / hw; tables    */
/* float32 with row stride coef_ld), formed on load in float32 from the fusion's      */
/* output gradient g; `sums` come from mis_amsfn_fuse_bwd_lazy.  n_pix < 2^31.         */
int mis_bn_relu_bwd_dz_scaled(const void* z, const void* g, const float* g_scale, const float* g_shift, int coef_ld,
                              int64_t hw, int dtype, int64_t n_pix, int C, const float* mean_rstd, const float* gamma,
                              const float* beta, const float* sums, void* dz, float* dbias, void* workspace,
                              mis_stream_t stream);
/* The same for all four branch layers of one block in ONE launch (g is then read from  */
/* memory once, not four times): z, mean_rstd, gamma, beta, dz, rows are HOST arrays of   */
/* 4 device pointers; g_scale / g_shift are the [B][4*C] tables of the block (branch k    */
/* at column k*C, coef_ld >= 4*C); sums is [4][2][C]; rows[k] is NULL or a                */
/* [mis_bn_relu_bwd_dz_rows(dtype, n_pix, C)][C] buffer that receives the partial rows    */
/* of the conv-bias gradient (summed by the caller or by the weight-gradient reduction).  */
/* Results are bit-identical to four mis_bn_relu_bwd_dz_scaled calls.                     */
int mis_bn_relu_bwd_dz_scaled4(const void* const* z, const void* g, const float* g_scale, const float* g_shift,
                               int coef_ld, int64_t hw, int dtype, int64_t n_pix, int C,
                               const float* const* mean_rstd, const float* const* gamma, const float* const* beta,
                               const float* sums, void* const* dz, float* const* rows, mis_stream_t stream);
/* The same two passes for a BatchNorm2d that is NOT followed by a ReLU of its own  */
/* (ResDConv's `c1` and `cbr3`, model/utils/modules.py:432-433,444-445): dy is the    */
/* gradient of the BatchNorm output as it is, no mask.                               */
int mis_bn_bwd_reduce(const void* z, const void* dy, int dtype, int64_t n_pix, int C,
                      const float* mean_rstd, const float* gamma, const float* beta,
                      float* sums, void* workspace, mis_stream_t stream);
int mis_bn_bwd_dz(const void* z, const void* dy, int dtype, int64_t n_pix, int C,
                  const float* mean_rstd, const float* gamma, const float* beta, const float* sums,
                  void* dz, float* dbias, void* workspace, mis_stream_t stream);
/* Per-channel sum over pixels of a channels-last tensor (conv bias gradient).   */
size_t mis_channel_sum_workspace_bytes(int64_t n_pix, int C);
int mis_channel_sum(const void* x, int dtype, int64_t n_pix, int C, float* out, void* workspace,
                    mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Output layer nn.Conv2d(base, n_classes, 1) — model/unet.py:44,84.            */
/* w float32 [Cout][Cin], bias float32 [Cout].                                   */
int mis_conv1x1_fwd(const void* x, const float* x_affine, int dtype, int64_t n_pix, int Cin, int Cout,
                    const float* w, const float* bias, void* y, mis_stream_t stream);
size_t mis_conv1x1_bwd_workspace_bytes(int64_t n_pix, int Cin, int Cout);
/* dx (same dtype), dw float32 [Cout][Cin], dbias float32 [Cout]; dx may be NULL. */
/* bn_sums (nullable): as for mis_maxpool2x2_bwd — x_affine is then [4][Cin] and   */
/* float32 [2][Cin] = (sum dy, sum dy*xhat) over (x, dx) is produced in the same    */
/* pass (from the values dx holds, or would hold when it is NULL: see               */
/* mis_bn_relu_bwd_dz_head).                                                       */
int mis_conv1x1_bwd(const void* x, const float* x_affine, const void* dy, int dtype, int64_t n_pix, int Cin,
                    int Cout, const float* w, void* dx, float* dw, float* dbias, float* bn_sums, void* workspace,
                    mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Layout / dtype helpers for the model boundary (NCHW float32 images in,       */
/* channels-last activations inside).                                           */
int mis_nchw_f32_to_nhwc(const float* x, int B, int C, int H, int W, int dtype, void* y, mis_stream_t stream);

/* One activation read by several consumers (the input of an AMSFN block feeds four branches,  */
/* model/utils/modules.py:846-856; an encoder output feeds its block and the decoder's skip,    */
/* model/a_unet.py:66-105): autograd adds the consumers' gradients pairwise, one full pass per  */
/* addition.  out = grads[0] + ... + grads[n-1] (2 <= n <= 5) in ONE pass, summed in float32 in */
/* slot order and rounded once.  With bn_sums (float32 [2][C]) the activation is the lazy       */
/* output relu(bn(z)) of a training-mode layer (z_affine4 = [4][C]: scale, shift, mean, rstd)   */
/* and the kernel also returns that layer's BatchNorm-backward sums of `out` as stored,         */
/*   bn_sums[0][c] = sum out*[y>0],  bn_sums[1][c] = sum out*[y>0]*xhat,                         */
/* i.e. what mis_bn_relu_bwd_reduce(z, out) would compute in a pass of its own.                  */
size_t mis_sum_gradients_workspace_bytes(int C);
int mis_sum_gradients(const void* const* grads, int n_grads, int dtype, int64_t n_pix, int C, const void* z,
                      const float* z_affine4, void* out, float* bn_sums, void* workspace, mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* AMSFN multi-scale fusion block of A_UNet (model/utils/modules.py:791-871,     */
/* model/a_unet.py:27-30,48-51): everything that is not a convolution.            */
/*                                                                               */
/* nn.MaxPool2d(3, stride=1, padding=1) (modules.py:799).  argmax: uint8, same    */
/* element count as y — the window position (0..8, row-major, first maximum)      */
/* that supplied each output; the backward gathers dy through it.                 */
int mis_maxpool3x3s1_fwd(const void* x, const float* x_affine, int dtype, int B, int H, int W, int C, void* y,
                         uint8_t* argmax, mis_stream_t stream);
int mis_maxpool3x3s1_bwd(const void* dy, const uint8_t* argmax, int dtype, int B, int H, int W, int C, void* dx,
                         mis_stream_t stream);
/* Test hook (process-global): 1 = every input goes through the generic row-strip kernels,  */
/* 0 (default) = bf16 tensors of 8-channel vectors take the integer-key forward (lazy input)  */
/* and the row-run backward.  Returns the previous setting.                                    */
int mis_set_maxpool3_path(int strip_kernels);
/* Fusion (modules.py:846-868): z / z_affine are HOST arrays of 4 device pointers  */
/* (the branch outputs c1..c4, [B][HW][C] each, lazy when the affine is not NULL).  */
/*   s[b][k*C+c] = mean_p act_k          (AdaptiveAvgPool2d(1) + cat)               */
/*   h = relu(W1 s + b1), w = sigmoid(W2 h + b2)   (W1 [C][4C], W2 [4C][C])         */
/*   out[b][p][c] = sum_k act_k[b][p][c] * w[b][k*C+c]                              */
/* s [B][4C], h [B][C], w [B][4C] float32 are kept for the backward, which returns  */
/* dact[k] = dout*w_k + ds_k/HW (gradient with respect to each branch activation)   */
/* and the MLP parameter gradients.  C <= 256.  Both calls need the workspace of     */
/* mis_amsfn_workspace_bytes (partial rows of the pixel reductions).                */
size_t mis_amsfn_workspace_bytes(int B, int HW, int C);
int mis_amsfn_fuse_fwd(const void* const* z, const float* const* z_affine, int dtype, int B, int HW, int C,
                       const float* W1, const float* b1, const float* W2, const float* b2, float* s, float* h,
                       float* w, void* out, void* workspace, mis_stream_t stream);
int mis_amsfn_fuse_bwd(const void* const* z, const float* const* z_affine, const void* dout, int dtype, int B, int HW,
                       int C, const float* W1, const float* W2, const float* s, const float* h, const float* w,
                       void* const* dact, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                       mis_stream_t stream);
/* The same backward (modules.py:846-868 under autograd) when every branch is the    */
/* lazy output of a training-mode conv + BatchNorm + ReLU layer (z_affine4[k] is its   */
/* [4][C] table scale, shift, mean, rstd; bn_gamma / bn_beta [k] its BatchNorm          */
/* parameters): the four gradients dact[k] = dout*w_k + ds_k/HW are NOT written.        */
/* Returned instead: dact_shift [B][4C] = ds/HW (with w, the two tables that turn dout  */
/* into dact[k] — consumed by mis_bn_relu_bwd_dz_scaled) and bn_sums [4][2][C] =         */
/* (sum dy, sum dy*xhat) of each branch layer's BatchNorm backward, collected in the     */
/* same pass over (dout, z_k) that forms the fusion-weight gradient.  Workspace:         */
/* mis_amsfn_lazy_workspace_bytes.                                                       */
size_t mis_amsfn_lazy_workspace_bytes(int B, int HW, int C);
int mis_amsfn_fuse_bwd_lazy(const void* const* z, const float* const* z_affine4, const float* const* bn_gamma,
                            const float* const* bn_beta, const void* dout, int dtype, int B, int HW, int C,
                            const float* W1, const float* W2, const float* s, const float* h, const float* w,
                            float* bn_sums, float* dact_shift, float* dW1, float* db1, float* dW2, float* db2,
                            void* workspace, mis_stream_t stream);

/* DWConv (modules.py:52-60) = pointwise 1x1 after depthwise 3x3 is one dense 3x3    */
/* conv: W[o][i][3][3] = P[o][i] * D[i][3][3], b[o] = bp[o] + sum_i P[o][i]*bd[i]      */
/* (P = pconv.weight [Cout][Cin], D = dconv.weight [Cin][9], bd/bp their biases).      */
/* The backward maps (dW, db) of the dense conv to (dP, dD, dbd); dbp = db.            */
int mis_dwconv_compose_fwd(const float* P, const float* D, const float* bd, const float* bp, int Cout, int Cin,
                           float* W, float* b, mis_stream_t stream);
int mis_dwconv_compose_bwd(const float* P, const float* D, const float* bd, const float* dW, const float* db, int Cout,
                           int Cin, float* dP, float* dD, float* dbd, mis_stream_t stream);
/* The forward composition of MANY layers in one launch (a network with 48 DWConv     */
/* layers otherwise pays 48 launches of a few microseconds per step): job j owns the   */
/* workgroups [first_block, first_block + n_blocks) of the launch; `jobs` is a DEVICE  */
/* array.  D == NULL: the layer is a plain 1x1 conv (AMSFN.conv1x1 / final_conv,        */
/* modules.py:801,838) and W is P at the centre tap, zero elsewhere; b is then not      */
/* written (the conv takes the 1x1 conv's own bias).  Same expressions as               */
/* mis_dwconv_compose_fwd: identical results.                                           */
typedef struct {
  const float* P;     /* [Cout][Cin] */
  const float* D;     /* [Cin][9] or NULL */
  const float* bd;    /* [Cin] (D != NULL) */
  const float* bp;    /* [Cout] (D != NULL) */
  float* W;           /* [Cout][Cin][3][3] */
  float* b;           /* [Cout] (D != NULL) */
  int Cout, Cin, first_block, n_blocks;
} mis_compose_job;
int mis_dwconv_compose_multi(const mis_compose_job* jobs, int n_jobs, int total_blocks, mis_stream_t stream);
/* The chain rule of many layers in one launch per 24 jobs; `jobs` is a HOST array (it travels  */
/* in the kernel arguments).  D == NULL: the 1x1 layer — dP is the centre tap of dW; the other  */
/* pointers of such a job are ignored.  Same arithmetic as mis_dwconv_compose_bwd.              */
typedef struct {
  const float* P;
  const float* D;
  const float* bd;
  const float* dW;    /* [Cout][Cin][3][3] */
  const float* db;    /* [Cout] */
  float* dP;          /* [Cout][Cin] */
  float* dD;          /* [Cin][9] */
  float* dbd;         /* [Cin] */
  int Cout, Cin;
} mis_compose_bwd_job;
int mis_dwconv_compose_bwd_multi(const mis_compose_bwd_job* jobs, int n_jobs, mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* RDAM_UNet (model/rdam_unet.py:7-115): what is not a convolution.              */
/*                                                                               */
/* ResDConv tail, model/utils/modules.py:447-453: out = relu(bn3(z3) + bn1(z1))   */
/* from the two conv outputs and their [2][C] (scale, shift): `x = x + re;        */
/* return self.relu(x)` and the two BatchNorm applications in one pass.           */
int mis_add_bn_relu_fwd(const void* za, const float* za_affine, const void* zb, const float* zb_affine, int dtype,
                        int64_t n_pix, int C, void* out, mis_stream_t stream);
/* Its backward's first step: g = dout where out > 0, else 0 (the gradient of both */
/* BatchNorm outputs).                                                             */
int mis_relu_mask_bwd(const void* out, const void* dout, int dtype, int64_t n_pix, int C, void* g,
                      mis_stream_t stream);
/* MDAM, model/utils/attention.py:171-190: out = x * sigmoid(w[b][c]) with          */
/* w = softmax(avgpool(GroupNorm(..))) + fc(avgpool(conv3x3(x))).  The first term is  */
/* the GroupNorm bias whatever x is (one channel per group: the average of a          */
/* normalised map is its bias), the second needs only the nine zero-padded window     */
/* sums of x per (sample, channel):                                                   */
/*   S float32 [B][C][9], S[b][c][ky*3+kx] = sum of x[b][.][.][c] over the pixels     */
/*   that tap (ky, kx) of a 3x3 / padding-1 conv reads (attention.py:184-185:         */
/*   avgpool(conv3x3(x))[co] = bias[co] + sum_ci,tap W[co][ci][tap]*S[ci][tap]/(H*W)).  */
/* mis_mdam_window_sums_bwd: dx[b][p][c] = sum of dS[b][c][tap] over the taps whose     */
/* window contains p.  mis_channel_scale: out = x * s[b][c] (attention.py:189);          */
/* mis_channel_dot: out[b][c] = sum_p x*d (its gradient with respect to s).              */
/* Workspace of both reductions: mis_mdam_workspace_bytes.  C a multiple of the          */
/* 16-byte channel vector, C <= 2048 (bfloat16) / 1024 (float32).                        */
size_t mis_mdam_workspace_bytes(int B, int C);
int mis_mdam_window_sums(const void* x, int dtype, int B, int H, int W, int C, float* S, void* workspace,
                         mis_stream_t stream);
int mis_mdam_window_sums_bwd(const float* dS, int dtype, int B, int H, int W, int C, void* dx, mis_stream_t stream);
int mis_channel_scale(const void* x, const float* s, int dtype, int B, int64_t HW, int C, void* out,
                      mis_stream_t stream);
int mis_channel_dot(const void* x, const void* d, int dtype, int B, int64_t HW, int C, float* out, void* workspace,
                    mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Sliding-window prediction — utils/slide_predict.py:60-117.                    */
/*                                                                               */
/* mis_slide_gather_tiles: crops n_tiles windows out of a uint8 [H][W][C] image  */
/* (tile_yx int32 [n_tiles][2] device = top-left corners) into a channels-last   */
/* activation [n_tiles][window][window][C] of value u8/255 — the                 */
/* `from_numpy(crop).permute(2,0,1).float()/255.0` of slide_predict.py:83-87.    */
int mis_slide_gather_tiles(const uint8_t* image, int H, int W, int C, const int32_t* tile_yx, int n_tiles,
                           int window, int dtype, void* out, mis_stream_t stream);
/* mis_argmax_classes: `torch.argmax(pred, dim=1)` (slide_predict.py:92) as      */
/* uint8 [B][HW]; first maximal class, NaN maximal.  logits element (b,c,p) at   */
/* b*stride_b + c*stride_c + p*stride_p.  1 <= C <= 256.                         */
int mis_argmax_classes(const void* logits, int dtype, int64_t stride_b, int64_t stride_c, int64_t stride_p, int B,
                       int C, int HW, uint8_t* out, mis_stream_t stream);
/* mis_slide_blend: the Gaussian-weighted accumulation, normalisation and        */
/* rounding of slide_predict.py:74-112.  Tile k = i*nx + j (the reference's      */
/* order, duplicates of the clamped last row/column included) has its top-left   */
/* corner at (ys[i], xs[j]) and its uint8 class map at                           */
/* class_maps[map_index[k]] ([window][window]); gauss = float64 [window][window] */
/* weights.  Per output pixel the covering tiles are visited in tile order with  */
/* float32 accumulators updated in float64, as numpy evaluates                   */
/* `out[...] += mask * weight`; out uint8 [out_h][out_w] (the unpadded image).   */
/* ny + nx <= 4096.                                                              */
int mis_slide_blend(const uint8_t* class_maps, const int32_t* map_index, const int32_t* ys, const int32_t* xs,
                    int ny, int nx, int window, const double* gauss, int out_h, int out_w, uint8_t* out,
                    mis_stream_t stream);

/* ------------------------------------------------------------------------- */
/* torch.nn.utils.clip_grad_norm_(params, max_norm) followed by                  */
/* torch.optim.AdamW.step() (utils/train_and_eval.py:115-118, train.py:221-226)  */
/* over all parameter tensors in two launches.  descs: device array of           */
/* mis_param_desc, one per tensor (float32, dense).  Work is cut into chunks of   */
/* MIS_OPT_CHUNK elements: chunk_prefix (device int[n_tensors+1]) holds each      */
/* tensor's first chunk, chunk_prefix[n_tensors] = n_chunks.                      */
/* mis_grad_sqsum writes one partial per chunk into the workspace;                */
/* mis_clip_adamw_step then (for the chunks [chunk_begin, chunk_end) — one call   */
/* per optimizer param group, each with its own hyper-parameters) forms           */
/* total_norm = sqrt(sum of partials) (written to total_norm[0] if not NULL, by   */
/* the call whose range contains chunk 0), clip = min(1, max_norm/(norm+1e-6))    */
/* (max_norm <= 0: no clipping), g *= clip, and applies the AdamW update with     */
/* bias corrections for `step` (1-based) exactly as torch's non-fused AdamW:      */
/* p *= 1-lr*wd; m = lerp(m,g,1-b1); v = b2*v+(1-b2)*g*g;                          */
/* p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).                            */
/* write_clipped_grads != 0 stores the clipped gradients back (as torch does).    */
#define MIS_OPT_CHUNK 4096
typedef struct {
  float* p;        /* parameter            */
  float* g;        /* gradient             */
  float* m;        /* exp_avg              */
  float* v;        /* exp_avg_sq           */
  int64_t n;       /* elements             */
} mis_param_desc;
size_t mis_clip_adamw_workspace_bytes(int n_chunks);
int mis_grad_sqsum(const void* descs, const int* chunk_prefix, int n_tensors, int n_chunks, void* workspace,
                   mis_stream_t stream);
int mis_clip_adamw_step(const void* descs, const int* chunk_prefix, int n_tensors, int chunk_begin, int chunk_end,
                        int n_chunks, const void* workspace, float max_norm, double lr, double beta1, double beta2,
                        double eps, double weight_decay, int64_t step, int write_clipped_grads, float* total_norm,
                        mis_stream_t stream);
/* The same update with its step-dependent constants read from device memory, for a launch that  */
/* is captured once in a HIP graph and replayed every step (the reference has no counterpart: it */
/* re-issues optimizer.step() from Python).  step_constants: device float[3] = (1-lr*wd,          */
/* lr/(1-b1^t), sqrt(1-b2^t)); the host forms them with mis_adamw_step_constants (host-only, no   */
/* GPU work: the same double arithmetic mis_clip_adamw_step uses) and copies them in before each  */
/* replay.                                                                                       */
void mis_adamw_step_constants(double lr, double beta1, double beta2, double weight_decay, int64_t step, float* out3);
int mis_clip_adamw_step_dyn(const void* descs, const int* chunk_prefix, int n_tensors, int chunk_begin, int chunk_end,
                            int n_chunks, const void* workspace, float max_norm, double beta1, double beta2, double eps,
                            const float* step_constants, int write_clipped_grads, float* total_norm,
                            mis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MISAMD_H */
