/*
 * unet_hip.h — C ABI of libunet_hip.so: the MI355X (gfx950) kernels behind the
 * Our_UNet train step of Ulixes-8/UNet-Implementations.
 *
 * The reference has no FFI: its operator API for this path is the PyTorch module
 * surface.  Every entry point below replaces the ATen op(s) the
 * reference reaches through torch.nn at the cited file:line (paths relative to
 * the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every function returns 0 on success or a negative UNET_E_* code;
 *     unet_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers owned by the caller (no ownership
 *     transfer, no hidden allocation); workspaces are caller-provided and sized
 *     by the matching *_workspace_bytes() query.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - activations are NHWC fp32 ("pixel-major": [N][H][W][C]); logits and the
 *     input image cross the module boundary as NCHW, exactly like the reference.
 *   - packed 3x3 weights, tap = ky*3+kx, the GEMM's reduction axis contiguous:
 *     wf[tap][co][ci] (forward: K = ci) and wd[tap][ci][co] (data gradient: K = co).
 */
#ifndef UNET_HIP_H_
#define UNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_ABI_VERSION 8

#define UNET_OK 0
#define UNET_E_INVALID (-1) /* bad argument / unsupported shape */
#define UNET_E_LAUNCH (-2)  /* HIP launch or runtime error */
#define UNET_E_WORKSPACE (-3) /* workspace too small */

typedef void* unet_stream_t; /* hipStream_t */

const char* unet_last_error(void);
int unet_abi_version(void);
/* number of HIP devices visible to the library (0 without a GPU); never fails */
int unet_device_count(void);

/* Test hook.  Operands are addressed through 2 GiB buffer descriptors; a batch whose tensor is
 * larger (the reference trains at bs 32: Our_UNet/src/train.py:748) is split over N inside the
 * convolution entry points.  This lowers the split threshold so the chunked path can be
 * exercised on small tensors; bytes <= 0 restores 2^31 - 1. */
int unet_debug_set_chunk_limit(int64_t bytes);

/* ---- layout helpers ------------------------------------------------------ */

/* NCHW -> NHWC (module boundary for the input image, Our_UNet/models/unet.py:399) */
int unet_nchw_to_nhwc(const float* x_nchw, float* y_nhwc, int N, int C, int H, int W,
                      unet_stream_t stream);
/* NHWC -> NCHW (test helper / hooks that want the reference layout) */
int unet_nhwc_to_nchw(const float* x_nhwc, float* y_nchw, int N, int C, int H, int W,
                      unet_stream_t stream);

/* OIHW [Cout][Cin][3][3] (state_dict layout, Our_UNet/models/unet.py:106-115)
 * -> wf[9][Cout][Cin] and wd[9][Cin][Cout]; either output may be NULL. */
int unet_pack_conv3x3_weights(const float* w_oihw, float* wf, float* wd, int Cout, int Cin,
                              unet_stream_t stream);

/* ---- 3x3 convolution (pad 1, stride 1 or 2) ------------------------------ */

/* y[N][Ho][Wo][Cout] = conv3x3(cat(x0[.., C0], x1[.., C1])) + bias.
 * Replaces nn.Conv2d forward inside ConvBlock (Our_UNet/models/unet.py:106-115,
 * stride rule :103) and the torch.cat of UpBlock (:228) via the two-source
 * input (x1 may be NULL with C1 = 0).  H, W are the INPUT spatial sizes;
 * Ho = (H-1)/stride+1.  C0 == 3 (RGB stem) or C0, C1 multiples of 32;
 * Cout multiple of 32.
 */
int unet_conv3x3_fwd(const float* x0, int C0, const float* x1, int C1, const float* wf,
                     const float* bias, float* y, int N, int H, int W, int Cout, int stride,
                     unet_stream_t stream);

/* bf16 mixed-precision variants (BASELINE.json config 4; the reference's AMP path,
 * Our_UNet/src/train.py:638-652, is fp16 autocast): identical arguments and fp32 tensors; both
 * operands are rounded to bf16 while staged on chip and contracted on the bf16 matrix cores with
 * fp32 accumulation.  The RGB stem (C0 == 3) stays fp32. */
int unet_conv3x3_fwd_bf16(const float* x0, int C0, const float* x1, int C1, const float* wf,
                          const float* bias, float* y, int N, int H, int W, int Cout, int stride,
                          unet_stream_t stream);
int unet_conv3x3_bwd_data_bf16(const float* dy, const float* wd, int Cin_total, int ci_offset,
                               float* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                               int accumulate, unet_stream_t stream);

/* dx[N][H][W][Ccols] (+)= conv3x3_transpose(dy[N][Ho][Wo][Cout], wd slice).
 * Replaces the data-gradient half of aten::convolution_backward reached from
 * loss.backward() (Our_UNet/src/train.py:663).  wd is the whole
 * wd[9][Cin_total][Cout]; the call produces input channels
 * [ci_offset, ci_offset + Ccols) into dx, whose channel count is Ccols (the two
 * halves of a concatenated input are two calls).  H, W are the spatial sizes of
 * dx (the conv INPUT).  accumulate != 0 adds into dx (skip tensors receive two
 * gradients). */
int unet_conv3x3_bwd_data(const float* dy, const float* wd, int Cin_total, int ci_offset,
                          float* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                          int accumulate, unet_stream_t stream);

/* dw_oihw[Cout][Cin_total][3][3] (columns ci_offset .. ci_offset+Cx) =
 *   sum over pixels of x[.., Cx] (x) dy[.., Cout]; db[Cout] = sum dy if db != NULL.
 * Replaces the weight/bias-gradient half of aten::convolution_backward.
 * H, W are the spatial sizes of x (the conv input). */
size_t unet_conv3x3_bwd_weight_workspace_bytes(int N, int H, int W, int Cx, int Cout, int stride);
int unet_conv3x3_bwd_weight(const float* x, int Cx, const float* dy, float* dw_oihw,
                            int ci_offset, int Cin_total, float* db, void* workspace,
                            size_t workspace_bytes, int N, int H, int W, int Cout, int stride,
                            unet_stream_t stream);

/* bf16 mixed-precision weight gradient (config 4): same arguments; stride-1 layers run on the
 * bf16 matrix cores (operands rounded on chip, fp32 sums), the rest fall back to fp32. */
int unet_conv3x3_bwd_weight_bf16(const float* x, int Cx, const float* dy, float* dw_oihw,
                                 int ci_offset, int Cin_total, float* db, void* workspace,
                                 size_t workspace_bytes, int N, int H, int W, int Cout, int stride,
                                 unet_stream_t stream);

/* Split-bf16 ("bf16x3") operand mode: fp32 operands are split on chip into three bf16 terms
 * (x = h + m + l, residual <= 2^-26 |x|) and every product is evaluated as the six bf16 x bf16
 * terms of weight >= 2^-16 on the bf16 matrix cores with fp32 accumulation; the dropped terms
 * are below one fp32 rounding of the product, so results agree with the fp32-MFMA entry points
 * to fp32 accuracy (tests/test_kernels_gpu.py compares both with fp64).  Arguments as the
 * plain entry points plus the weights pre-split by unet_pack_conv3x3_weights_bf16x3:
 * wf3 = [3][9][Cout][Cin] and wd3 = [3][9][Cin][Cout] bf16 (bit patterns as uint16_t), either
 * may be NULL in the pack call.  The RGB stem and the stride-2 weight gradient run on the fp32
 * path, which is why the fp32 wf / wd are still passed. */
int unet_pack_conv3x3_weights_bf16x3(const float* w_oihw, uint16_t* wf3, uint16_t* wd3, int Cout,
                                     int Cin, unet_stream_t stream);

/* Every layer's packing in ONE launch (the per-layer form costs 22 small launches per step).
 * `table_device` is an array of n entries in DEVICE memory, built once by the caller; all
 * pointers are device pointers, any destination may be NULL; Cout must be a multiple of 32.
 * A workgroup packs one 32 (co) x 32 (ci) x 9 tile: entry k owns the tiles
 * [tile_begin, tile_begin + (Cout/32) * ceil(Cin/32)), total_tiles = their sum.  Layouts as above. */
typedef struct unet_pack_entry {
  const float* w;   /* OIHW source [Cout][Cin][3][3] */
  float* wf;        /* [9][Cout][Cin] */
  float* wd;        /* [9][Cin][Cout] */
  uint16_t* wf3;    /* [3][9][Cout][Cin] bf16 planes ([1][9][Cout][Cin] when planes == 1) */
  uint16_t* wd3;    /* [3][9][Cin][Cout] bf16 planes */
  int Cout, Cin;
  int tile_begin;
  int reserved;     /* planes: 0 or 3 = the three planes of the split-bf16 form; 1 = only plane 0,
                       the bf16-rounded weight (what the mixed-precision kernels stage) */
} unet_pack_entry;
int unet_pack_conv3x3_weights_batched(const unet_pack_entry* table_device, int n, int total_tiles,
                                      unet_stream_t stream);
int unet_conv3x3_fwd_bf16x3(const float* x0, int C0, const float* x1, int C1, const float* wf,
                            const uint16_t* wf3, const float* bias, float* y, int N, int H, int W,
                            int Cout, int stride, unet_stream_t stream);
int unet_conv3x3_bwd_data_bf16x3(const float* dy, const float* wd, const uint16_t* wd3,
                                 int Cin_total, int ci_offset, float* dx, int N, int H, int W,
                                 int Cout, int Ccols, int stride, int accumulate,
                                 unet_stream_t stream);
int unet_conv3x3_bwd_weight_bf16x3(const float* x, int Cx, const float* dy, float* dw_oihw,
                                   int ci_offset, int Cin_total, float* db, void* workspace,
                                   size_t workspace_bytes, int N, int H, int W, int Cout,
                                   int stride, unet_stream_t stream);

/* ---- 1x1 convolution (CLIP fusion layer) ---------------------------------- */

/* y = conv1x1(cat(x0, x1)) + bias with w[Cout][C0+C1]; replaces clip_fusion_conv[0] =
 * nn.Conv2d(512 + clip_dim, 512, 1) and the torch.cat in front of it
 * (CLIP_UNet/models/unet.py:356-362, :477-478). */
int unet_conv1x1_fwd(const float* x0, int C0, const float* x1, int C1, const float* w,
                     const float* bias, float* y, int N, int H, int W, int Cout,
                     unet_stream_t stream);
/* dx[.., Ccols] (+)= dy . wT[ci_offset.., :] with wT[Cin_total][Cout] (unet_transpose2d of w) */
int unet_conv1x1_bwd_data(const float* dy, const float* wT, int Cin_total, int ci_offset,
                          float* dx, int N, int H, int W, int Cout, int Ccols, int accumulate,
                          unet_stream_t stream);
/* dw[Cout][Cin_total] (columns ci_offset .. ci_offset+Cx); workspace sized by
 * unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, Cx, Cout, 1) */
int unet_conv1x1_bwd_weight(const float* x, int Cx, const float* dy, float* dw, int ci_offset,
                            int Cin_total, void* workspace, size_t workspace_bytes, int N, int H,
                            int W, int Cout, unet_stream_t stream);
/* dst[C][R] = src[R][C]^T */
int unet_transpose2d(const float* src, float* dst, int R, int C, unet_stream_t stream);

/* ---- InstanceNorm2d(eps, affine) + LeakyReLU + SpatialDropout2d ----------- */

/* Per-(n,c) statistics of y[N][HW][C] and the folded affine coefficients
 *   alpha[n][c] = gamma[c]*rstd, beta2[n][c] = beta[c] - mean*alpha.
 * Replaces aten::native_batch_norm statistics as lowered from
 * nn.InstanceNorm2d(eps=1e-5, affine=True) (Our_UNet/models/unet.py:118-119).
 * mean, rstd, alpha, beta2 are [N][C]. */
size_t unet_instnorm_workspace_bytes(int N, int HW, int C);
int unet_instnorm_stats(const float* y, const float* gamma, const float* beta, float eps,
                        float* mean, float* rstd, float* alpha, float* beta2, void* workspace,
                        size_t workspace_bytes, int N, int HW, int C, unet_stream_t stream);

/* a = leaky_relu(y*alpha + beta2, slope) * mask ; mask [N][C] may be NULL.
 * Replaces InstanceNorm apply + nn.LeakyReLU (unet.py:122-123) +
 * SpatialDropout2d.forward (unet.py:13-35; mask already scaled by 1/(1-p)). */
int unet_instnorm_lrelu_drop_fwd(const float* y, const float* alpha, const float* beta2,
                                 const float* mask, float slope, float* a, int N, int HW, int C,
                                 unet_stream_t stream);

/* Backward of the block above: from ga = dL/da produce dy = dL/dy (may alias ga),
 * dgamma[C], dbeta[C] and dbias[C] = the gradient of the conv bias in front of the norm (may be
 * NULL) = sum over pixels of dy, evaluated in closed form from the reduction sums: it is
 * identically zero under InstanceNorm (the reference's autograd value is rounding noise). */
int unet_instnorm_lrelu_drop_bwd(const float* ga, const float* y, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta,
                                 const float* mask, float slope, float* dy, float* dgamma,
                                 float* dbeta, float* dbias, void* workspace,
                                 size_t workspace_bytes, int N, int HW, int C,
                                 unet_stream_t stream);

/* ---- bilinear 2x upsample (align_corners=False) --------------------------- */

/* y[N][2h][2w][C] from x[N][h][w][C]; replaces F.interpolate in UpBlock.forward
 * (Our_UNet/models/unet.py:219-225) for the exact-2x case. */
int unet_upsample2x_fwd(const float* x, float* y, int N, int h, int w, int C,
                        unet_stream_t stream);
/* gx[N][h][w][C] (+)= transpose-stencil of gy[N][2h][2w][C] (gather form, no atomics) */
int unet_upsample2x_bwd(const float* gy, float* gx, int N, int h, int w, int C, int accumulate,
                        unet_stream_t stream);

/* ---- 1x1 segmentation head ------------------------------------------------ */

/* logits_nchw[N][K][HW] = a[N][HW][C] . w[K][C] + b[K]; replaces
 * segmentation_output = nn.Conv2d(32, 3, 1) (Our_UNet/models/unet.py:374-381,:430).
 * C == 32, K <= 4. */
int unet_head1x1_fwd(const float* a, const float* w, const float* b, float* logits_nchw, int N,
                     int HW, int C, int K, unet_stream_t stream);
size_t unet_head1x1_bwd_workspace_bytes(int N, int HW, int C, int K);
int unet_head1x1_bwd(const float* a, const float* dlogits_nchw, const float* w, float* da,
                     float* dw, float* db, void* workspace, size_t workspace_bytes, int N, int HW,
                     int C, int K, unet_stream_t stream);

/* ---- deferred slab reductions of the weight gradients ----------------------------------------
 * Every *_bwd_weight* entry point ends with 2-3 small launches that sum its per-workgroup slabs
 * in a fixed order and scatter the result into the OIHW gradient (44 such launches per train
 * step).  Between unet_wgrad_defer_begin() and unet_wgrad_defer_end() (per calling thread) the
 * entry points only QUEUE those reductions; unet_wgrad_defer_flush() launches everything queued
 * so far as 2-3 batched launches (all layers side by side; the same arithmetic in the same
 * order: bit-identical gradients).  While a reduction is queued its `workspace` must stay
 * allocated and dw_oihw is NOT yet valid.  The stand-alone bias gradient (db != NULL) cannot be
 * deferred.  No reference counterpart (aten::convolution_backward returns finished gradients). */
int unet_wgrad_defer_begin(void);
int unet_wgrad_defer_pending(void);
int unet_wgrad_defer_flush(unet_stream_t stream);
int unet_wgrad_defer_end(unet_stream_t stream);

/* ---- SimpleLoss: dynamic-weighted CE + soft Dice, forward and gradient ----- */

/* loss_out[0] = w_ce*CE + w_dice*Dice, loss_out[1] = CE, loss_out[2] = Dice,
 * loss_out[3..5] = the class weights used; dlogits (NCHW, may be NULL) = dL/dlogits.
 * Replaces SimpleLoss.forward and its autograd backward
 * (Our_UNet/models/losses.py:24-62 class weights, :73-76 CE, :84-121 Dice).
 * class_weights: 3 floats on device used when dynamic_weights == 0 (NULL = unweighted).
 * global_counts (optional, may be NULL): when non-NULL the kernel pair is split
 * so the caller can all-reduce the 4 class/valid counts between the passes. */
size_t unet_dice_wce_loss_workspace_bytes(int N, int H, int W);
int unet_dice_wce_loss_fwd_bwd(const float* logits_nchw, const int64_t* target, float* loss_out,
                               float* dlogits_nchw, void* workspace, size_t workspace_bytes,
                               int N, int H, int W, float smooth, float w_dice, float w_ce,
                               int ignore_index, int dynamic_weights, const float* class_weights,
                               float grad_scale, unet_stream_t stream);

/* The gradient pass alone, for a caller that learns dL/d(loss) only later (autograd's
 * backward): run unet_dice_wce_loss_fwd_bwd / _shard_apply with dlogits == NULL, keep the SAME
 * workspace (it holds the class weights and Dice coefficients of that call) and call this with
 * `upstream` = a device float holding dL/d(loss) (NULL = 1).  dlogits = upstream * the gradient
 * the one-call form writes (bit-identical for upstream == 1).  Replaces the autograd backward
 * of SimpleLoss.forward (Our_UNet/models/losses.py:64-82) without an extra scaling pass. */
int unet_dice_wce_loss_grad(const float* logits_nchw, const int64_t* target,
                            const void* workspace, size_t workspace_bytes, const float* upstream,
                            float* dlogits_nchw, int N, int H, int W, int ignore_index,
                            unet_stream_t stream);

/* The same loss over a batch SHARDED across processes (data parallel, SURVEY.md 8e
 * "global-exact"): the value and gradient of SimpleLoss on the concatenated batch.
 * shard_stats leaves per-image sums in `workspace` and writes stats[10] (device, double) =
 * {class counts[3], per-class NLL sums[3], valid pixels, sum over images of Dice[3]}; the host
 * sums stats over the shards (one 80-byte all-reduce) and passes the result, with the global
 * image count, to shard_apply together with the SAME workspace.  loss_out is identical on every
 * shard; dlogits already carry the global normalisation, so parameter gradients are SUMMED
 * (not averaged) across shards. */
int unet_dice_wce_loss_shard_stats(const float* logits_nchw, const int64_t* target, double* stats,
                                   void* workspace, size_t workspace_bytes, int N, int H, int W,
                                   float smooth, int ignore_index, unet_stream_t stream);
int unet_dice_wce_loss_shard_apply(const float* logits_nchw, const int64_t* target,
                                   const double* global_stats, int N_global, float* loss_out,
                                   float* dlogits, void* workspace, size_t workspace_bytes, int N,
                                   int H, int W, float smooth, float w_dice, float w_ce,
                                   int ignore_index, int dynamic_weights,
                                   const float* class_weights, float grad_scale,
                                   unet_stream_t stream);

/* ---- validation metrics and input pipeline ------------------------- */

/* preds[N][H][W] (uint8, optional) = argmax over the class planes; counts[9] (uint64, device) =
 * per class {intersection, predicted, labelled} over the batch, ignore_index pixels excluded:
 * the integers behind the Dice scores of validate() (Our_UNet/src/train.py:556-577).
 * target == NULL and counts == NULL: prediction only (the argmax of src/evaluate.py:185-207). */
int unet_argmax_dice_counts(const float* logits_nchw, const int64_t* target, uint8_t* preds,
                            uint64_t* counts, int N, int H, int W, int ignore_index,
                            unet_stream_t stream);

/* uint8 HWC image (+ uint8 mask) -> ((v/255) - mean)/std NHWC fp32 (+ int64 target with values
 * > 2 other than 255 mapped to 0): PetSegmentationDataset.__getitem__, src/train.py:300-311.
 * mean3 / std3 are HOST pointers to 3 floats. */
int unet_preprocess_u8(const uint8_t* image_hwc, const uint8_t* mask, float* out_nhwc,
                       int64_t* target, int N, int H, int W, const float* mean3, const float* std3,
                       unet_stream_t stream);

/* ---- SGD with Nesterov momentum over a flat arena -------------------------- */

/* g' = g*grad_scale + wd*p; buf = first_step ? g' : mu*buf + g'; p -= lr*(g' + mu*buf).
 * Replaces optim.SGD(lr, momentum, nesterov=True, weight_decay).step()
 * (Our_UNet/src/train.py:445-451,:664). */
int unet_sgd_nesterov_step(float* params, const float* grads, float* momentum, int64_t n,
                           float lr, float mu, float weight_decay, int first_step,
                           float grad_scale, unet_stream_t stream);

/* The same with the hyper-parameters read from DEVICE memory, hyper = {lr, mu, weight_decay,
 * grad_scale}: a train step captured in a HIP graph keeps following the learning-rate schedule
 * (the reference steps it per epoch, Our_UNet/src/train.py:940) without being re-captured. */
int unet_sgd_nesterov_step_dev(float* params, const float* grads, float* momentum, int64_t n,
                               const float* hyper, int first_step, unet_stream_t stream);

/* out[i] = a[i] + b[i] (gradient accumulation of skip tensors, test helper) */
int unet_add_inplace(float* a, const float* b, int64_t n, unet_stream_t stream);

/* ---- fused layer pipeline -------------------------------------------------- */
/*
 * One ConvBlock unit of the reference is conv -> InstanceNorm2d -> LeakyReLU -> SpatialDropout2d
 * (Our_UNet/models/unet.py:101-134).  The stand-alone entry points above run it as four passes
 * over the layer tensor (conv, statistics, apply, next conv).  The fused pipeline keeps only
 * the RAW convolution output y_l of every layer in HBM:
 *   - the convolution that produces y_l emits the InstanceNorm statistics of y_l from its
 *     epilogue (per-tile (mean, M2) summaries, merged by a tiny finalize launch), and
 *   - every consumer of the activated tensor a_l = dropout(lrelu(IN(y_l))) -- the next
 *     convolution, the weight gradient of the next convolution, the bilinear up-sampling, the
 *     1x1 head -- applies  a = lrelu(y * alpha[n][c] + beta[n][c], slope)  to the operand while
 *     it stages it on chip (zero padding is applied after the activation).
 * alpha / beta are the folded coefficients the statistics finalize writes:
 *   alpha[n][c] = gamma[c] * rstd[n][c] * mask[n][c],
 *   beta [n][c] = (beta[c] - mean[n][c] * gamma[c] * rstd[n][c]) * mask[n][c]
 * (mask = the SpatialDropout2d factor 0 or 1/(1-p), folded in because lrelu(z)*m == lrelu(z*m)
 * for m >= 0).  a_l itself is never written.
 */
typedef struct unet_act_src {
  const float* x;     /* [N][H][W][C] raw convolution output, or a plain tensor (alpha == NULL) */
  int C;              /* channels: 3 (RGB image, plain only) or a multiple of 32 */
  const float* alpha; /* [N][C] folded InstanceNorm scale x dropout mask; NULL = use x as stored */
  const float* beta;  /* [N][C] folded shift; ignored when alpha == NULL */
} unet_act_src;

/* y[N][Ho][Wo][Cout] = conv_kxk(cat(act(s0), act(s1))) + bias  (ksize 3: pad 1, stride 1|2,
 * w = wf[9][Cout][Cin]; ksize 1: stride 1, w = [Cout][Cin]; s1 may be NULL).  The epilogue leaves
 * per-tile (mean, M2) summaries of y in `workspace`; *stats_px_out (HOST int) receives the pixels
 * per summary tile, or 0 when this shape has no statistics epilogue (tiny or ragged maps).
 * Replaces Conv2d (+ the previous unit's InstanceNorm apply / LeakyReLU / dropout) of ConvBlock
 * (Our_UNet/models/unet.py:101-134) and the torch.cat of UpBlock (:228).  H, W = input size. */
size_t unet_conv_in_fwd_workspace_bytes(int N, int H, int W, int Cout, int stride);
int unet_conv_in_fwd(const unet_act_src* s0, const unet_act_src* s1, float slope, const float* w,
                     const float* bias, int ksize, int stride, float* y, void* workspace,
                     size_t workspace_bytes, int* stats_px_out, int N, int H, int W, int Cout,
                     unet_stream_t stream);
/* The same in the split-bf16 ("bf16x3") operand mode: fp32 tensors, fp32-class accuracy on the
 * bf16 matrix cores.  w3 = the pre-split planes of unet_pack_conv3x3_weights_bf16x3 (forward
 * layout; may be NULL for ksize 1).  Stride-1 3x3 shapes that tile as 4 x 32 pixels run the
 * split patch kernel with the activation applied before the split; other shapes run the fp32
 * kernels (same results to fp32 rounding). */
int unet_conv_in_fwd_bf16x3(const unet_act_src* s0, const unet_act_src* s1, float slope,
                            const float* w, const uint16_t* w3, const float* bias, int ksize,
                            int stride, float* y, void* workspace, size_t workspace_bytes,
                            int* stats_px_out, int N, int H, int W, int Cout, unet_stream_t stream);
/* InstanceNorm statistics of y [N][HoWo][Cout] from the summaries unet_conv_in_fwd left in
 * `workspace` (stats_px > 0: a few-microsecond merge) or from y itself (stats_px == 0), and the
 * folded coefficients for y's consumers: mean, rstd, alpha_out, beta_out [N][Cout]
 * (gamma / beta = the norm's affine parameters, mask [N][Cout] = this layer's dropout factors
 * or NULL).  Replaces the statistics half of nn.InstanceNorm2d (unet.py:118-119). */
int unet_conv_in_stats_finalize(const float* y, void* workspace, size_t workspace_bytes,
                                int stats_px, const float* gamma, const float* beta, float eps,
                                const float* mask, float* mean, float* rstd, float* alpha_out,
                                float* beta_out, int N, int HoWo, int Cout, unet_stream_t stream);

/* First convolution of a decoder stage with the bilinear 2x up-sampling done in the loader:
 *   y[N][H][W][Cout] = conv3x3(cat(upsample2x(act(low)), act(skip))) + bias,
 * low->x = [N][H/2][W/2][C0] raw, skip->x = [N][H][W][C1] raw: UpBlock.forward
 * (Our_UNet/models/unet.py:215-231) without the up-sampled tensor and without the concatenation.
 * Covers the shapes the patch-staged kernel tiles (unet_conv_up_in_fwd_supported != 0; H % 4,
 * W % 32, channels % 32, >= 512 tiles); otherwise use unet_upsample2x_in_fwd + unet_conv_in_fwd.
 * Workspace, *stats_px_out and the statistics finalize as unet_conv_in_fwd (stride 1). */
int unet_conv_up_in_fwd_supported(int N, int H, int W, int C0, int C1, int Cout);
int unet_conv_up_in_fwd(const unet_act_src* low, const unet_act_src* skip, float slope,
                        const float* wf, const float* bias, float* y, void* workspace,
                        size_t workspace_bytes, int* stats_px_out, int N, int H, int W, int Cout,
                        unet_stream_t stream);

/* RGB stem straight from the dataset's uint8 HWC image [N][H][W][3]: the normalisation
 * ((v / 255) - mean[c]) / std[c] of PetSegmentationDataset.__getitem__ (Our_UNet/src/train.py:
 * 303-308) happens in the loaders of the first convolution and of its weight gradient, so the
 * fp32 image is never written (12x less input traffic).  mean3 / std3 are HOST pointers.
 * W % 128 == 0; other widths go through unet_preprocess_u8.  Workspaces / *stats_px_out as
 * unet_conv_in_fwd (stride 1) and unet_conv3x3_bwd_weight (Cx = 3). */
int unet_stem_u8_fwd(const uint8_t* image_hwc, const float* mean3, const float* std3,
                     const float* wf, const float* bias, float* y, void* workspace,
                     size_t workspace_bytes, int* stats_px_out, int N, int H, int W, int Cout,
                     unet_stream_t stream);
int unet_stem_u8_bwd_weight(const uint8_t* image_hwc, const float* mean3, const float* std3,
                            const float* dy, float* dw_oihw, void* workspace,
                            size_t workspace_bytes, int N, int H, int W, int Cout,
                            unet_stream_t stream);

/* Weight gradient with the activation applied to the input operand on load:
 * dw_oihw[Cout][Cin_total][k][k] (columns ci_offset .. +x->C) = sum_pixels act(x) (x) dy.
 * Same workspace query as unet_conv3x3_bwd_weight.  ksize 1 keeps the centre tap. */
int unet_conv_in_bwd_weight(const unet_act_src* x, float slope, const float* dy, float* dw_oihw,
                            int ci_offset, int Cin_total, int ksize, int stride, void* workspace,
                            size_t workspace_bytes, int N, int H, int W, int Cout,
                            unet_stream_t stream);
/* The same in the split-bf16 operand mode (fp32 tensors; stride-1 3x3 layers on the bf16 matrix
 * cores with the activation applied before the three-term split, other shapes on the fp32
 * kernels).  Workspace as unet_conv3x3_bwd_weight_workspace_bytes. */
int unet_conv_in_bwd_weight_bf16x3(const unet_act_src* x, float slope, const float* dy, float* dw_oihw,
                            int ci_offset, int Cin_total, int ksize, int stride, void* workspace,
                            size_t workspace_bytes, int N, int H, int W, int Cout,
                            unet_stream_t stream);

/* up[N][2h][2w][C] = bilinear2x(act(x)) (the activation is applied to each of the four taps;
 * UpBlock.forward, Our_UNet/models/unet.py:219-225). */
int unet_upsample2x_in_fwd(const unet_act_src* x, float slope, float* up, int N, int h, int w,
                           unet_stream_t stream);

/* Backward of conv3x3(upsample2x(a)) at LOW resolution.  The bilinear up-sampling U is linear:
 *   dW[tap]  = sum_p (U a)[p + tap] (x) dy[p]   =  sum_q a[q] (x) D_tap[q]
 *   dL/da[q] = (U^T sum_tap W_tap^T shift_tap dy)[q]  =  sum_tap W_tap^T D_tap[q]
 * with D_tap = U^T shift_tap(dy): both gradients of the up-sampled operand of UpBlock's first
 * convolution (Our_UNet/models/unet.py:219-231) become GEMMs over the low-resolution pixels q --
 * a quarter of the positions, so a quarter of the FLOPs of the 3x3 weight / data gradient on
 * the up-sampled grid, no up-sampled tensor and no upsample2x_bwd pass.
 *   unet_upsample2x_bwd_taps:   D[N][h][w][9*C] (tap-major channels) from dy[N][2h][2w][C]
 *   unet_conv3x3_up_bwd_weight: dw_oihw[Cout][Cin_total][3][3] (columns ci_offset .. +x->C)
 *                               from act(x)[N][h][w][Cx] and D
 *   unet_conv3x3_up_bwd_data:   g[N][h][w][Ccols] (+)= sum_tap D_tap . wd[tap][ci_offset..][:]
 */
int unet_upsample2x_bwd_taps(const float* dy, float* D, int N, int h, int w, int C,
                             unet_stream_t stream);
/* General bilinear resize of NCHW planes, align_corners = False (x: [planes][h][w] ->
 * y: [planes][H][W]) and its adjoint (gather form, deterministic): the F.interpolate of
 * SimpleLoss for logits whose size differs from the target's (Our_UNet/models/losses.py:66-68)
 * and of CLIP_UNet's bottleneck features (CLIP_UNet/models/unet.py:444-450). */
int unet_resize_bilinear_fwd(const float* x, float* y, int planes, int h, int w, int H, int W,
                             unet_stream_t stream);
int unet_resize_bilinear_bwd(const float* gy, float* gx, int planes, int h, int w, int H, int W,
                             unet_stream_t stream);
size_t unet_conv3x3_up_bwd_weight_workspace_bytes(int N, int h, int w, int Cx, int Cout);
int unet_conv3x3_up_bwd_weight(const unet_act_src* x, float slope, const float* D, float* dw_oihw,
                               int ci_offset, int Cin_total, void* workspace,
                               size_t workspace_bytes, int N, int h, int w, int Cout,
                               unet_stream_t stream);
int unet_conv3x3_up_bwd_data(const float* D, const float* wd, int Cin_total, int ci_offset,
                             float* g, int N, int h, int w, int Cout, int Ccols, int accumulate,
                             unet_stream_t stream);

/* Data gradient whose output g = dL/da_l is FINAL for layer l (not a partial sum that another
 * call still accumulates into): the epilogue can also emit the two reductions of layer l's
 * InstanceNorm + LeakyReLU + dropout backward, S1 = sum gz and S2 = sum gz * xhat per tile,
 * which saves the pass that would re-read g and y_l (unet_instnorm_lrelu_drop_bwd's first half).
 * `bs` describes layer l; on return bs->tiles_out = summaries per image (0: this shape / kernel
 * has no such epilogue - run unet_instnorm_lrelu_drop_bwd as usual).  bs->partial needs
 * N * ceil(H*W / 64) * C * 8 bytes.  Consume with unet_instnorm_lrelu_drop_bwd_partials. */
typedef struct unet_bwd_stats {
  const float* y;      /* raw conv output of layer l, same shape as the gradient being written */
  const float* mean;   /* [N][C] */
  const float* rstd;   /* [N][C] */
  const float* gamma;  /* [C] */
  const float* beta;   /* [C] */
  const float* mask;   /* [N][C] dropout factors or NULL */
  float slope;
  void* partial;
  size_t partial_bytes;
  int tiles_out;
} unet_bwd_stats;
int unet_conv3x3_bwd_data_bs(const float* dy, const float* wd, int Cin_total, int ci_offset,
                             float* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                             int accumulate, unet_bwd_stats* bs, unet_stream_t stream);
/* The same on the mixed-precision pipeline: dy, dx and bs->y are bf16 tensors (stride-1 shapes
 * of the patch kernel emit the reductions; bs->tiles_out = 0 otherwise). */
int unet_conv3x3_bwd_data_bs_b16(const uint16_t* dy, const float* wd, int Cin_total, int ci_offset,
                                 uint16_t* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                                 int accumulate, unet_bwd_stats* bs, unet_stream_t stream);
/* unet_conv3x3_bwd_data_bs in the split-bf16 mode (wd3 = pre-split planes, data-gradient
 * layout); bs may be NULL (no reductions wanted). */
int unet_conv3x3_bwd_data_bs_bf16x3(const float* dy, const float* wd, const uint16_t* wd3,
                                    int Cin_total, int ci_offset, float* dx, int N, int H, int W,
                                    int Cout, int Ccols, int stride, int accumulate,
                                    unet_bwd_stats* bs, unet_stream_t stream);
int unet_conv3x3_up_bwd_data_bs(const float* D, const float* wd, int Cin_total, int ci_offset,
                                float* g, int N, int h, int w, int Cout, int Ccols, int accumulate,
                                unet_bwd_stats* bs, unet_stream_t stream);
/* ---- Winograd F(2x2, 3x3) form of the stride-1 3x3 layers (csrc/conv_wino.hip) ------------
 * 2.25x fewer matrix-core FLOPs than the direct kernels for the layers it tiles (image as
 * 8 x 32 pixels, >= 64 reduction channels in multiples of 8, output channels in multiples of
 * 64, >= 256 workgroups).  The transformed weights U = G g G^T (16 floats per filter, stored in
 * the kernel's LDS image order) are produced once per step by unet_pack_wino_weights:
 *   uf (forward,       Cout % 64 == 0, Cin % 8 == 0)  and/or
 *   ud (data gradient, Cin % 64 == 0, Cout % 8 == 0),  unet_wino_weight_floats() floats each.
 * Same arithmetic as nn.Conv2d / its data gradient (Our_UNet/models/unet.py:106-115) up to
 * fp32 rounding of the transforms (<= 3e-6 of max |y| measured). */
int unet_conv_wino_supported(int N, int H, int W, int C0, int C1, int Cout);
/* The 32 -> 32 channel stride-1 layers (enc0 / dec4 at full resolution) have a Winograd form of
 * their own inside unet_conv_in_fwd / unet_conv3x3_bwd_data(_bs): it needs no extra weight form
 * (U = G g G^T is built in the kernel's prologue from the packed weights), so it is a process-wide
 * switch rather than an entry point: 1 (default) = Winograd when the launch fills the chip (at
 * least 512 tiles of 8 x 32 pixels: the persistent kernel runs two workgroups per CU), 2 =
 * Winograd for every shape the kernel tiles, 0 = the direct kernel.  Returns the previous
 * setting.  The switch is kept PER CALLING THREAD (ABI 7; it was process-wide): a caller sets it
 * around its own launches and restores it, so two models on two threads - or a framework's
 * backward thread - cannot change each other's dispatch.  The weight gradient of the same layers (unet_conv_in_bwd_weight /
 * unet_conv3x3_bwd_weight with Cx = Cout = 32, stride 1, H % 8 == 0, W % 32 == 0) follows the
 * same switch: Winograd F(3x3,2x2) when there is an 8 x 32 tile for every CU (1) / always (2).
 * (csrc/conv_c32.hip, csrc/conv_wgrad.hip) */
int unet_set_c32_winograd(int on);
/* 1 when a 3x3 fused forward / data gradient of this shape runs that Winograd form (for FLOP
 * accounting: it issues 16/36 of the direct kernel's matrix-core FLOPs). */
int unet_conv_c32_is_winograd(int N, int H, int W, int Cin, int Cout, int stride);
/* The same switch covers unet_conv_up_in_fwd of the last decoder stage, (64 up-sampled + 32 skip)
 * -> 32 channels with both sources activated on load: K = 96 as three register-resident Winograd
 * chunks, the bilinear up-sampling folded into the input transform (1: with an 8 x 32 tile for
 * every CU, 2: every shape with H % 8 == 0 and W % 32 == 0).  1 when this shape takes it: */
int unet_conv_up_c32_is_winograd(int N, int H, int W, int C0, int C1, int Cout);
/* unet_conv_in_bwd_weight of a 32 -> 32 channel layer with the layer's InstanceNorm + LeakyReLU +
 * dropout backward (Our_UNet/models/unet.py:118-127 under autograd) applied ON LOAD by the dy side
 * of the Winograd weight-gradient kernel, which reads every pixel exactly once: g = dL/da [N][H][W]
 * [Cout] (w.r.t. the layer's activated output), y = the layer's raw convolution output, coef5 /
 * sums from unet_instnorm_bwd_coefs.  Writes dz = dL/dy to dz_out (may alias g) for the layer's
 * data gradient, fills dgamma / dbeta / dbias (each may be null) and the weight gradient as
 * unet_conv_in_bwd_weight does: the elementwise unet_instnorm_lrelu_drop_bwd pass of such a
 * layer disappears.  x must be activated on load (alpha / beta).  (csrc/conv_wgrad.hip) */
int unet_conv_in_bwd_weight_dz_supported(int N, int H, int W, int Cx, int Cout);
int unet_conv_in_bwd_weight_dz(const unet_act_src* x, float slope, const float* g, const float* y,
                               const float* coef5, const float* sums, const float* gamma,
                               const float* rstd, float dz_slope, float* dz_out, float* dgamma,
                               float* dbeta, float* dbias, float* dw_oihw, int ci_offset,
                               int Cin_total, void* workspace, size_t workspace_bytes, int N, int H,
                               int W, int Cout, unet_stream_t stream);
size_t unet_wino_weight_floats(int Cout, int Cin);
int unet_pack_wino_weights(const float* w_oihw, float* uf, float* ud, int Cout, int Cin,
                           unet_stream_t stream);
/* The same for several layers in one launch.  Entry k owns the blocks [block_begin,
 * block_begin + ceil(Cout * Cin / 8 / 256)); uf / ud may be null per entry (form not wanted);
 * the shape rules of unet_pack_wino_weights apply per entry (checked by the caller). */
typedef struct unet_wino_pack_entry {
  const float* w;   /* OIHW source [Cout][Cin][3][3] */
  float* uf;        /* forward form or null */
  float* ud;        /* data-gradient form or null */
  int Cout, Cin;
  int block_begin, reserved;
} unet_wino_pack_entry;
int unet_pack_wino_weights_batched(const unet_wino_pack_entry* table_device, int n,
                                   int total_blocks, unet_stream_t stream);
/* unet_conv_in_fwd (ksize 3, stride 1) on the Winograd kernel: y = conv3x3(cat(act(s0),
 * act(s1))) + bias and the per-tile statistics of y (256 pixels per tile). */
int unet_conv_in_fwd_wino(const unet_act_src* s0, const unet_act_src* s1, float slope,
                          const float* uf, const float* bias, float* y, void* workspace,
                          size_t workspace_bytes, int* stats_px_out, int N, int H, int W, int Cout,
                          unet_stream_t stream);
/* 1 when unet_conv3x3_bwd_weight / unet_conv_in_bwd_weight (ksize 3, fp32 tensors) run this
 * shape on the Winograd F(3x3,2x2) weight-gradient kernel (bench.py: executed FLOPs). */
int unet_conv3x3_bwd_weight_is_winograd(int N, int H, int W, int Cx, int Cout, int stride);
/* unet_conv_up_in_fwd on the Winograd kernel (the bilinear gather of the low-resolution source
 * runs inside its loader): low = [N][H/2][W/2][C0], skip = [N][H][W][C1]. */
int unet_conv_up_wino_supported(int N, int H, int W, int C0, int C1, int Cout);
int unet_conv_up_in_fwd_wino(const unet_act_src* low, const unet_act_src* skip, float slope,
                             const float* uf, const float* bias, float* y, void* workspace,
                             size_t workspace_bytes, int* stats_px_out, int N, int H, int W,
                             int Cout, unet_stream_t stream);
/* unet_conv3x3_bwd_data_bs (stride 1, accumulate 0) on the Winograd kernel; ud covers the whole
 * weight [Cout][Cin_total], ci_offset % 64 == 0; bs may be NULL. */
int unet_conv3x3_bwd_data_bs_wino(const float* dy, const float* ud, int Cin_total, int ci_offset,
                                  float* dx, int N, int H, int W, int Cout, int Ccols,
                                  unet_bwd_stats* bs, unet_stream_t stream);
/* Apply-on-load InstanceNorm backward (round 3).  unet_instnorm_bwd_coefs turns the per-tile
 * reductions a data-gradient epilogue left (unet_bwd_stats.partial, `tiles` per image) into the
 * coefficient planes coef5 = [5][N][C] (a1, b1, P, Q, R) of
 *   dz = (z > 0 ? P : P slope) g + (Q y + R),  z = y a1 + b1,
 * and sums = [N][C][2] (S1, S2).  unet_conv3x3_bwd_data_dz_wino is unet_conv3x3_bwd_data_bs_wino
 * whose loader forms dz from (g, y) with them - the elementwise unet_instnorm_lrelu_drop_bwd
 * pass of the layer is not run - and which also writes dz (dz_out, must not alias g) for the
 * layer's weight gradient and the layer's dgamma / dbeta / dbias (each may be null).
 * Replaces the InstanceNorm2d / LeakyReLU / dropout backward of ConvBlock
 * (Our_UNet/models/unet.py:118-123 under autograd). */
int unet_instnorm_bwd_coefs(const void* partial, int tiles, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, const float* mask,
                            float* coef5, float* sums, int N, int HW, int C, unet_stream_t stream);
int unet_conv3x3_bwd_data_dz_wino(const float* g, const float* y, const float* coef5,
                                  const float* sums, const float* gamma, const float* rstd,
                                  float slope, float* dz_out, float* dgamma, float* dbeta,
                                  float* dbias, const float* ud, int Cin_total, int ci_offset,
                                  float* dx, int N, int H, int W, int Cout, int Ccols,
                                  unet_bwd_stats* bs, unet_stream_t stream);
/* unet_instnorm_lrelu_drop_bwd with the reductions already summarised per tile
 * (partial[(n * tiles + t) * C + c] = (S1, S2)). */
int unet_instnorm_lrelu_drop_bwd_partials(const float* ga, const float* y, const float* mean,
                                          const float* rstd, const float* gamma,
                                          const float* beta, const float* mask, float slope,
                                          float* dy, float* dgamma, float* dbeta, float* dbias,
                                          const void* partial, int tiles, void* workspace,
                                          size_t workspace_bytes, int N, int HW, int C,
                                          unet_stream_t stream);
/* bf16 storage (mixed-precision pipeline): ga, y, dy are bf16 tensors. */
int unet_instnorm_lrelu_drop_bwd_partials_b16(const uint16_t* ga, const uint16_t* y,
                                              const float* mean, const float* rstd,
                                              const float* gamma, const float* beta,
                                              const float* mask, float slope, uint16_t* dy,
                                              float* dgamma, float* dbeta, float* dbias,
                                              const void* partial, int tiles, void* workspace,
                                              size_t workspace_bytes, int N, int HW, int C,
                                              unet_stream_t stream);

/* ---- mixed precision with bf16 activations in HBM (BASELINE config 4) ----------------------
 * The reference's AMP path is fp16 autocast + GradScaler (Our_UNet/src/train.py:638-652); the
 * MI355X form is bf16 (fp32's exponent range: no loss scaling).  The *_b16 entry points are the
 * fused-pipeline entry points above with every layer tensor -- raw convolution outputs y,
 * activation gradients, dy, D -- stored as bf16 (uint16_t* here; `x` of a unet_act_src then
 * points to bf16 data) while the RGB image, logits, InstanceNorm statistics / coefficients,
 * weights, weight gradients and the optimizer state stay fp32.  Convolutions contract bf16
 * operands on the bf16 matrix cores with fp32 accumulation (stride-1 weight gradients too;
 * stride-2 weight gradients and the low-resolution tap GEMM use the fp32 matrix cores on bf16
 * storage); the activation and all InstanceNorm arithmetic are fp32 on values widened on load;
 * statistics come from the fp32 accumulators before y is rounded. */
int unet_conv_in_fwd_b16(const unet_act_src* s0, const unet_act_src* s1, float slope,
                         const float* w, const float* bias, int ksize, int stride, uint16_t* y,
                         void* workspace, size_t workspace_bytes, int* stats_px_out, int N, int H,
                         int W, int Cout, unet_stream_t stream);
/* unet_conv_in_fwd_b16 with the weights also given pre-rounded to bf16 (wb = [9][Cout][Cin],
 * plane 0 of unet_pack_conv3x3_weights_batched's wf3; may be null): the stride-1 patch kernel
 * stages its weight panels without the conversion, at half the L2 traffic; bit-identical. */
int unet_conv_in_fwd_b16_wb(const unet_act_src* s0, const unet_act_src* s1, float slope,
                            const float* w, const uint16_t* wb, const float* bias, int ksize,
                            int stride, uint16_t* y, void* workspace, size_t workspace_bytes,
                            int* stats_px_out, int N, int H, int W, int Cout, unet_stream_t stream);
/* unet_conv3x3_bwd_data_bs_b16 likewise (wdb = [9][Cin_total][Cout] bf16, plane 0 of wd3). */
int unet_conv3x3_bwd_data_bs_b16_wb(const uint16_t* dy, const float* wd, const uint16_t* wdb,
                                    int Cin_total, int ci_offset, uint16_t* dx, int N, int H,
                                    int W, int Cout, int Ccols, int stride, int accumulate,
                                    unet_bwd_stats* bs, unet_stream_t stream);
int unet_conv_in_stats_finalize_b16(const uint16_t* y, void* workspace, size_t workspace_bytes,
                                    int stats_px, const float* gamma, const float* beta, float eps,
                                    const float* mask, float* mean, float* rstd, float* alpha_out,
                                    float* beta_out, int N, int HoWo, int Cout,
                                    unet_stream_t stream);
int unet_conv_in_bwd_weight_b16(const unet_act_src* x, float slope, const uint16_t* dy,
                                float* dw_oihw, int ci_offset, int Cin_total, int ksize, int stride,
                                void* workspace, size_t workspace_bytes, int N, int H, int W,
                                int Cout, unet_stream_t stream);
int unet_conv3x3_bwd_data_b16(const uint16_t* dy, const float* wd, int Cin_total, int ci_offset,
                              uint16_t* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                              int accumulate, unet_stream_t stream);
int unet_instnorm_lrelu_drop_bwd_b16(const uint16_t* ga, const uint16_t* y, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta,
                                     const float* mask, float slope, uint16_t* dy, float* dgamma,
                                     float* dbeta, float* dbias, void* workspace,
                                     size_t workspace_bytes, int N, int HW, int C,
                                     unet_stream_t stream);
int unet_upsample2x_in_fwd_b16(const unet_act_src* x, float slope, uint16_t* up, int N, int h,
                               int w, unet_stream_t stream);
/* unet_conv_up_in_fwd on bf16 tensors: the first convolution of a decoder stage with the bilinear
 * 2x up-sampling of the (activated) low-resolution source done in the loader - no up-sampled
 * tensor (round 4; bit-identical to unet_upsample2x_in_fwd_b16 + unet_conv_in_fwd_b16_wb).
 * low [N][H/2][W/2][C0], skip [N][H][W][C1] (both activated on load), y [N][H][W][Cout] bf16;
 * w3 = the bf16-rounded plane of the packed weights (required).  Replaces UpBlock.forward
 * (Our_UNet/models/unet.py:215-231) under autocast. */
int unet_conv_up_in_fwd_b16_supported(int N, int H, int W, int C0, int C1, int Cout);
int unet_conv_up_in_fwd_b16(const unet_act_src* low, const unet_act_src* skip, float slope,
                            const float* wf, const uint16_t* w3, const float* bias, uint16_t* y,
                            void* workspace, size_t workspace_bytes, int* stats_px_out, int N,
                            int H, int W, int Cout, unet_stream_t stream);
int unet_upsample2x_bwd_taps_b16(const uint16_t* dy, uint16_t* D, int N, int h, int w, int C,
                                 unet_stream_t stream);
int unet_conv3x3_up_bwd_weight_b16(const unet_act_src* x, float slope, const uint16_t* D,
                                   float* dw_oihw, int ci_offset, int Cin_total, void* workspace,
                                   size_t workspace_bytes, int N, int h, int w, int Cout,
                                   unet_stream_t stream);
int unet_conv3x3_up_bwd_data_b16(const uint16_t* D, const float* wd, int Cin_total, int ci_offset,
                                 uint16_t* g, int N, int h, int w, int Cout, int Ccols,
                                 int accumulate, unet_stream_t stream);
/* ... with the BSTATS epilogue (g final for the layer bs describes; bs->y bf16), as
 * unet_conv3x3_up_bwd_data_bs: removes that layer's stand-alone reduction pass. */
int unet_conv3x3_up_bwd_data_bs_b16(const uint16_t* D, const float* wd, int Cin_total,
                                    int ci_offset, uint16_t* g, int N, int h, int w, int Cout,
                                    int Ccols, int accumulate, unet_bwd_stats* bs,
                                    unet_stream_t stream);
/* ... with the weights also pre-rounded to bf16 (wdb = [9][Cin_total][Cout] bf16, plane 0 of
 * wd3; NULL = as above; the backward of UpBlock's first convolution with respect to its
 * up-sampled operand, Our_UNet/models/unet.py:219-231, under autocast,
 * Our_UNet/src/train.py:638-652): the contraction runs as a plain bf16 GEMM over the 9 * Cout contiguous
 * values of a D row (64-wide K steps, no conversion of the weights).  Same result as the form
 * above up to the summation order of the fp32 accumulators; bs may be NULL. */
int unet_conv3x3_up_bwd_data_bs_b16_wb(const uint16_t* D, const float* wd, const uint16_t* wdb,
                                       int Cin_total, int ci_offset, uint16_t* g, int N, int h,
                                       int w, int Cout, int Ccols, int accumulate,
                                       unet_bwd_stats* bs, unet_stream_t stream);
int unet_head1x1_in_fwd_b16(const unet_act_src* x, float slope, const float* w, const float* b,
                            float* logits_nchw, int N, int HW, int K, unet_stream_t stream);
int unet_head1x1_in_bwd_b16(const unet_act_src* x, float slope, const float* dlogits_nchw,
                            const float* w, uint16_t* da, float* dw, float* db, void* workspace,
                            size_t workspace_bytes, int N, int HW, int K, unet_stream_t stream);

/* Head on an activated-on-load operand: logits = act(x) . w + b, and its backward
 * (da = dL/d act(x), dw, db); x->C == 32. */
int unet_head1x1_in_fwd(const unet_act_src* x, float slope, const float* w, const float* b,
                        float* logits_nchw, int N, int HW, int K, unet_stream_t stream);
int unet_head1x1_in_bwd(const unet_act_src* x, float slope, const float* dlogits_nchw,
                        const float* w, float* da, float* dw, float* db, void* workspace,
                        size_t workspace_bytes, int N, int HW, int K, unet_stream_t stream);
/* ... with the reductions of the InstanceNorm + LeakyReLU + dropout backward of the layer whose
 * raw output x->x is (bs->y == x->x; Our_UNet/models/unet.py:128-134 in front of :427): da is
 * that layer's final dL/da, so the kernel that writes it also leaves S1 = sum gz and
 * S2 = sum gz * xhat per workgroup in bs->partial (bs->tiles_out summaries per image; 0 = the
 * shape does not split evenly - run unet_instnorm_lrelu_drop_bwd as usual).  _b16: x, da and
 * bs->y are bf16 tensors. */
int unet_head1x1_in_bwd_bs(const unet_act_src* x, float slope, const float* dlogits_nchw,
                           const float* w, float* da, float* dw, float* db, void* workspace,
                           size_t workspace_bytes, int N, int HW, int K, unet_bwd_stats* bs,
                           unet_stream_t stream);
int unet_head1x1_in_bwd_bs_b16(const unet_act_src* x, float slope, const float* dlogits_nchw,
                               const float* w, uint16_t* da, float* dw, float* db,
                               void* workspace, size_t workspace_bytes, int N, int HW, int K,
                               unet_bwd_stats* bs, unet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H_ */
