/*
 * cvhip.h — C ABI of libcvhip.so, the MI355X (gfx950 / CDNA4) engine for CvPytorch's
 * data-parallel CNN training hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes + a `hipStream_t` (as void*),
 * is NON-ALLOCATING and ASYNCHRONOUS on the caller's stream (so a whole train step can be
 * captured in a hipGraph), and returns an int status (CVHIP_OK == 0; negative = refused).
 * No torch types appear here; the host side (cvpytorch_amd python modules, or any other language with an
 * FFI) owns every buffer.
 *
 * The reference (shanglianlm0525/CvPytorch) is pure Python: its hot path bottoms out in
 * ATen/cuDNN/torchvision/NCCL calls, so there is no reference FFI to mirror symbol-for-symbol.
 * Each entry point below therefore cites the reference CALL SITE whose native op it replaces
 * (paths relative to the reference root).
 *
 * Layout contract
 *   activations : NHWC ("channels_last"), bf16, addressed as rows of `ld` elements per pixel so
 *                 that a channel slice of a wider concat buffer is a first-class operand
 *                 (ld >= C, base pointer already offset to the slice's first channel).
 *   weights     : master copy fp32 KRSC (= torch OIHW tensor in channels_last memory format);
 *                 cvhip_conv2d_prep_weights() derives the bf16 operand images.
 *   BN stats    : fp32.
 */
/* (cvhip_f16.h includes this file a second time with CVHIP_REDECLARE_F16 defined: only the prototypes are seen again then) */
#if !defined(CVHIP_H_) || defined(CVHIP_REDECLARE_F16)
#ifndef CVHIP_H_
#define CVHIP_H_
#endif

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVHIP_VERSION 100

/* status codes */
#define CVHIP_OK 0
#define CVHIP_ERR_INVALID (-1)     /* bad argument (null pointer, misaligned, negative size) */
#define CVHIP_ERR_UNSUPPORTED (-2) /* shape/feature not implemented by the HIP path */
#define CVHIP_ERR_LAUNCH (-3)      /* hipLaunch / runtime error (see cvhip_last_error) */

/* activation ids — src/models/bricks/activation.py:13-30 (ReLU/LeakyReLU/SiLU),
 * src/models/bricks/swish.py:8-25 (Swish == SiLU arithmetic) */
#define CVHIP_ACT_NONE 0
#define CVHIP_ACT_RELU 1
#define CVHIP_ACT_SILU 2
#define CVHIP_ACT_LEAKY 3 /* negative slope passed separately */
#define CVHIP_ACT_SIGMOID 4
#define CVHIP_ACT_HSWISH 5

/* ------------------------------------------------------------------------------------------
 * Convolution descriptor (POD). Logical tensor shapes follow nn.Conv2d:
 *   x (N,C,H,W)  w (K,C/groups,R,S)  y (N,K,P,Q),
 *   P = (H + 2*pad_h - dil_h*(R-1) - 1)/stride_h + 1, likewise Q.
 * Replaces aten::convolution / convolution_backward reached from
 *   src/models/bricks/conv_module.py:209 (ConvModule.forward -> self.conv(x)),
 *   src/models/bricks/conv.py:8-46 (build_conv_layer -> nn.Conv2d),
 *   src/models/detects/yolov5_detect.py:25,42, src/models/heads/seg/base_seg_head.py:30.
 * ------------------------------------------------------------------------------------------ */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_conv_desc {
  int32_t N, C, H, W;     /* input: batch, channels, height, width                        */
  int32_t K, R, S;        /* output channels, kernel height, kernel width                  */
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;
  int32_t dil_h, dil_w;
  int32_t groups;         /* 1 (dense, MFMA implicit GEMM) or C==K (depthwise)             */
  int32_t x_ld;           /* elements between consecutive input pixels  (>= C)             */
  int32_t y_ld;           /* elements between consecutive output pixels (>= K)             */
  /* Channel padding: the engine works on C and K that are multiples of 8 (16-byte vectors). When the
   * layer's real channel counts are smaller (3-channel image stem, 255-channel detect head) the tensors are
   * padded and these fields give the REAL counts of the fp32 master weight [k_valid][R][S][c_valid] and of
   * `bias` (k_valid entries); 0 means "same as K / C". Padded rows/columns of the operand images are zero. */
  int32_t k_valid, c_valid;
} cvhip_conv_desc;
#endif

int cvhip_version(void);
/* last HIP runtime error string seen by the library on this thread (never NULL) */
const char* cvhip_last_error(void);

/* output spatial size for a descriptor (pure host arithmetic, usable without a GPU) */
int cvhip_conv2d_out_hw(const cvhip_conv_desc* d, int32_t* P, int32_t* Q);

/* Plan queries (pure host arithmetic, usable without a GPU; exercised by the CPU test-suite).
 *   cvhip_conv2d_fprop_stats_rows : number of per-M-tile partial rows the fprop epilogue writes
 *                                   into `stats_partial` ([rows][2][K] fp32) when BN statistics
 *                                   are requested.
 *   cvhip_conv2d_dgrad_weight_elems: number of bf16 elements of the dgrad weight image.
 *   cvhip_conv2d_dgrad_plan       : the stride-parity decomposition used by dgrad. For class
 *                                   (ph,pw) in [0,stride_h)x[0,stride_w) writes 8 int32:
 *                                   {TR, TS, r0, r_step, dh0, dh_step(neg), s0/... see DESIGN.md}
 */
int cvhip_conv2d_fprop_stats_rows(const cvhip_conv_desc* d);
/* grid size of the persistent streaming kernel (conv1x1_stream.hip) that cvhip_conv2d_fprop / _dgrad pick for a 1x1, stride-1,
 * unpadded pass with `nout` result channels, `cin` reduced channels and `m` pixel rows (`with_stats`: the pass also writes
 * BatchNorm partial sums); 0 = the general implicit-GEMM kernel runs (profiling labels / tests; pure host arithmetic). */
int cvhip_conv1x1_stream_blocks(int nout, int cin, int64_t m, int with_stats);
/* grid size of the direct stem kernel (conv_stem.hip: 8 input channels, <= 32 output channels, kernel <= 7x7, stride 1 or 2) that
 * cvhip_conv2d_fprop picks for this descriptor; 0 = another kernel runs. */
int cvhip_conv_stem_blocks(const cvhip_conv_desc* d);
int64_t cvhip_conv2d_dgrad_weight_elems(const cvhip_conv_desc* d);
/* per class: {TR, TS, r0, r_step, dh0, dh_step, s0, s_step, dw0, dw_step, w_offset(elems, 2 x int32 lo/hi)} = 12 int32 */
#define CVHIP_DGRAD_CLASS_INTS 12
int cvhip_conv2d_dgrad_plan(const cvhip_conv_desc* d, int32_t* out_classes, int max_classes);
/* The multiply-shift constants the conv kernels use to split a pixel index into (image, row, column) without integer division:
 * for 0 <= n < 2^31,  n / d == (mul ? umulhi(n, mul) >> shift : n)  exactly (d == 1 -> mul = 0). */
int cvhip_div31_consts(int32_t d, uint32_t* mul, uint32_t* shift);

/* Derive the bf16 operand images from the fp32 KRSC master weights.
 *   w_fprop : bf16 [K][R*S*C]              (also the wgrad output layout)
 *   w_dgrad : bf16, per stride-parity class [C][taps(class)][K], classes concatenated
 *             (may be NULL when the layer's input needs no gradient)
 * Stride-1 3x3 layers whose channel counts the row-band kernel accepts (conv_band.hip) carry a second, FRAGMENT-ORDERED copy of the
 * same values behind each image (csrc/conv_plan.h "band image": 1 KB per MFMA weight fragment, so that a wave's fetch is contiguous):
 * the buffers must hold cvhip_conv2d_weight_image_elems(d, 0 / 1) elements — K*R*S*C, or twice that where the copy exists
 * (stride-1 3x3 layers: both images; 3x3 / stride 2 / padding 1 layers: the forward image).
 * Replaces the implicit fp32->half weight cast autocast performs at trainer.py:179-184. */
int64_t cvhip_conv2d_weight_image_elems(const cvhip_conv_desc* d, int which /* 0 = w_fprop, 1 = w_dgrad */);
int cvhip_conv2d_prep_weights(const cvhip_conv_desc* d, const float* w_master_krsc,
                              void* w_fprop_bf16, void* w_dgrad_bf16, void* stream);

/* Batched operand preparation (weights_optim.hip): ONE launch re-packs the bf16 fprop and dgrad images of many layers from
 * their fp32 masters — what cvhip_conv2d_prep_weights does per layer (autocast's per-forward weight casts in the reference,
 * trainer.py:179-184). `cvhip_prep_plan_build` fills a host table of `n` x cvhip_prep_plan_item_bytes() bytes from the entries
 * (descriptors + fixed operand addresses; w_dgrad may be NULL) and returns the grid size; the caller copies the table to the
 * device once and calls `cvhip_prep_plan_run` after every optimizer step. */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_prep_entry {
  cvhip_conv_desc desc;
  const float* master;
  void* w_fprop;
  void* w_dgrad;
} cvhip_prep_entry;
#endif
int cvhip_prep_plan_item_bytes(void);
int cvhip_prep_plan_build(const cvhip_prep_entry* entries, int32_t n, void* table_host, int32_t* total_blocks);
int cvhip_prep_plan_run(const void* table_device, int32_t n, int32_t total_blocks, void* stream);

/* fprop: y = conv(x, w) (+ bias). If stats_partial != NULL the epilogue also emits per-M-tile
 * partial sums  stats_partial[tile][0][k] = sum_m acc, [tile][1][k] = sum_m acc^2  (fp32
 * accumulators, before rounding to bf16) for training-mode BatchNorm; `bias` must be NULL then. */
int cvhip_conv2d_fprop(const cvhip_conv_desc* d, const void* x_bf16, const void* w_fprop_bf16,
                       const float* bias, void* y_bf16, float* stats_partial, void* stream);

/* dgrad: dx = conv_transpose(dy, w) — implicit GEMM over stride-parity classes (no wasted taps). */
int cvhip_conv2d_dgrad(const cvhip_conv_desc* d, const void* dy_bf16, const void* w_dgrad_bf16,
                       void* dx_bf16, void* stream);
/* dx = dgrad(dy) + addend: `addend` is a bf16 NHWC tensor on dx's pixel grid with d->C channels and pitch addend_ld — the gradient
 * arriving over a skip connection (x feeds both this convolution and a residual add: yolo_modules.py:102, torchvision Bottleneck).
 * Folding it into the epilogue replaces autograd's gradient-accumulation add (one read of dx + addend and one write less). */
int cvhip_conv2d_dgrad_add(const cvhip_conv_desc* d, const void* dy_bf16, const void* w_dgrad_bf16, const void* addend_bf16,
                           int32_t addend_ld, void* dx_bf16, void* stream);

/* Fused backward of a 1x1 / stride-1 / unpadded Conv-BN-act layer (conv1x1_bwd.hip): ONE streaming launch replaces
 * cvhip_bn_act_bwd_apply + cvhip_conv2d_dgrad(_add) + cvhip_conv2d_wgrad — aten::native_batch_norm_backward +
 * silu_backward + convolution_backward reached from trainer.py:189 (loss.backward()) for the kernel_size-1 layers of
 * src/models/bricks/conv_module.py:201-214. The BN/activation derivative is applied while the gradient is loaded, so the
 * gradient at the convolution output never exists in memory:
 *   dy = scale * (dz * act'(scale*y + shift) - dbeta/M - xhat * dgamma/M),  xhat = (y - mean) * invstd     (M = N*H*W)
 *   dx = dy . W (+ addend)            dw += dy^T . x      (fp32, KRSC == [K][C]; ACCUMULATED: zero it first when needed)
 * dz arrives as one tensor (k_split == K) or as two channel ranges [0,k_split) / [k_split,K) with their own pitches (sibling
 * pairs); y has pitch d->y_ld, x pitch d->x_ld. scale/shift NULL = no BatchNorm (activation only); mean/invstd/dgamma/dbeta
 * NULL = BatchNorm in eval mode. The kernel takes K in {32,64,128}, C % 32 == 0 (<= 1024), pitches % 8 == 0 (otherwise
 * CVHIP_ERR_UNSUPPORTED: the three-pass form applies); `cvhip_conv1x1_bwd_fused_ok` is the POLICY query (pure host arithmetic):
 * 1 when the geometry fits and the layer is large enough for the fused form to be the faster one (>= 2400 64-row trips;
 * CVHIP_BWD1X1=0 never, =2 whenever it fits). All tensor pointers must be 16-byte aligned. */
int cvhip_conv1x1_bwd_fused_ok(const cvhip_conv_desc* d);
int cvhip_conv1x1_bwd_fused(const cvhip_conv_desc* d, const void* dz0_bf16, int32_t dz0_ld, const void* dz1_bf16, int32_t dz1_ld,
                            int32_t k_split, const void* y_bf16, const void* x_bf16, const void* w_dgrad_bf16, const float* scale,
                            const float* shift, const float* mean, const float* invstd, const float* dgamma, const float* dbeta,
                            int32_t act, float act_param, const void* addend_bf16, int32_t addend_ld, void* dx_bf16, int32_t dx_ld,
                            float* dw, void* stream);

/* wgrad: dw[K][R][S][C] (fp32) (+)= sum over pixels dy^T * im2col(x); split-K over pixels with
 * fp32 atomics. accumulate==0 zero-fills dw first (hipMemsetAsync on `stream`). */
int cvhip_conv2d_wgrad(const cvhip_conv_desc* d, const void* x_bf16, const void* dy_bf16,
                       float* dw_krsc_f32, int accumulate, void* stream);
/* Deterministic weight gradient: the pixel splits store their partial tiles into private slabs of `workspace` (plain stores) and a
 * second kernel adds the slabs in split order — no floating-point atomics, two runs give bit-identical gradients. Same operands and
 * result as cvhip_conv2d_wgrad (accumulate != 0: dw += ...). cvhip_conv2d_wgrad_det_workspace_bytes(desc) sizes the workspace
 * (16-byte aligned). */
int64_t cvhip_conv2d_wgrad_det_workspace_bytes(const cvhip_conv_desc* d);
int cvhip_conv2d_wgrad_det(const cvhip_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* workspace, int64_t ws_bytes,
                           void* stream);


/* depthwise conv (groups == C == K), direct, bandwidth-bound.
 * src/models/bricks/depthwise_separable_conv_module.py:76-94 (DeepLabv3+ ASPP / fuse convs).
 * weights: fp32 [C][R][S] master is used directly. */
int cvhip_dwconv2d_fprop(const cvhip_conv_desc* d, const void* x_bf16, const float* w_crs,
                         const float* bias, void* y_bf16, void* stream);
/* depthwise fprop with the layer's activation applied in the same pass (inference: after deploy.fuse_model the BatchNorm of a
 * DepthwiseSeparableConvModule's depthwise half is folded into w / bias — utils/fuse.py:32-54 — and only the activation remains) */
int cvhip_dwconv2d_fprop_act(const cvhip_conv_desc* d, const void* x, const float* w_crs, const float* bias, int32_t act, float act_param,
                             void* y, void* stream);
/* Depthwise forward + training-mode BatchNorm sums in ONE pass (round 6): the sums of the fp32 outputs (before their rounding to 16 bits,
 * as the dense convolutions' epilogues take them) leave as partial rows [rows][2][C] in the layout cvhip_bn_finalize reads — the
 * reduction pass over the stored output (aten::native_batch_norm's statistics half, conv_module.py:209-211 behind
 * depthwise_separable_conv_module.py:10-99) is not run. cvhip_dwconv2d_fprop_stats_rows: the number of rows for this problem and these
 * operand addresses, 0 when the strip kernel does not run it (then: cvhip_dwconv2d_fprop + cvhip_bn_stats_partial). The partial buffer
 * needs rows + CVHIP_REDUCE_SCRATCH_ROWS rows like every partial buffer. */
int64_t cvhip_dwconv2d_fprop_stats_rows(const cvhip_conv_desc* d, const void* x, const void* y);
int cvhip_dwconv2d_fprop_stats(const cvhip_conv_desc* d, const void* x_bf16, const float* w_crs, const float* bias, void* y_bf16,
                               float* stats_partial, void* stream);

int cvhip_dwconv2d_dgrad(const cvhip_conv_desc* d, const void* dy_bf16, const float* w_crs,
                         void* dx_bf16, void* stream);
int cvhip_dwconv2d_wgrad(const cvhip_conv_desc* d, const void* x_bf16, const void* dy_bf16,
                         float* dw_crs, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Column reductions over a [M][C] bf16 matrix with row pitch ld (NHWC activations, M=N*H*W).
 * Two-stage and deterministic: stage 1 writes `rows` partial rows, finalize reduces them.
 * ------------------------------------------------------------------------------------------ */
/* Every `partial` buffer ([rows][2][C] fp32) handed to a *_finalize entry point must be allocated with
 * CVHIP_REDUCE_SCRATCH_ROWS extra rows behind its payload: finalize pre-reduces many-row partials
 * (one row per conv M-tile: 25,600 for the YOLOv5-s stem at batch 64) into that scratch in parallel. */
#define CVHIP_REDUCE_SCRATCH_ROWS 64
/* BatchNorm statistic ACCUMULATORS (round 3): instead of one partial row per tile + a finalize launch, producers add their tile sums
 * with fp64 atomics into acc[CVHIP_BN_ACC_SHARDS][2][ld] (zeroed by the caller once per step) and consumers fold the shards in their
 * prologue — entry points with the suffix _acc. Arrival order changes the fp64 sum by <= ~1e-16 relative, so the fp32 statistics
 * derived from it are run-to-run identical for practical purposes. */
#define CVHIP_BN_ACC_SHARDS 16
/* number of partial rows stage-1 reductions produce for an M-row matrix */
int cvhip_colreduce_rows(int64_t M, int32_t C);

/* partial[rows][2][C] = per-block (sum x, sum x^2).  aten::native_batch_norm statistics —
 * src/models/bricks/conv_module.py:211 (self.norm(x)), bricks/norm.py:74-123. */
int cvhip_bn_stats_partial(const void* x_bf16, int64_t M, int32_t C, int32_t ld,
                           float* partial, void* stream);

/* Finalize training-mode BN statistics from partial rows:
 *   mean, invstd = 1/sqrt(var_biased + eps)            (saved for backward)
 *   scale = gamma*invstd, shift = beta - mean*scale     (consumed by cvhip_bn_act_fwd)
 *   running_mean/var updated with `momentum` (unbiased variance), as torch.nn.BatchNorm2d.
 * gamma/beta may be NULL (affine=False); running_* may be NULL (track_running_stats=False). */
int cvhip_bn_finalize(const float* partial, int32_t rows, int32_t C, int64_t count,
                      const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, float* mean, float* invstd,
                      float* scale, float* shift, void* stream);

/* eval-mode BN folded into scale/shift from running statistics */
int cvhip_bn_eval_scale_shift(int32_t C, const float* gamma, const float* beta,
                              const float* running_mean, const float* running_var, float eps,
                              float* scale, float* shift, void* stream);

/* z = act(y*scale + shift) (+ residual).  scale/shift may be NULL (pure activation).
 * Fuses aten::batch_norm apply + aten::silu_/relu_ (+ DarknetBottleneck shortcut add,
 * src/models/modules/yolo_modules.py:102) into one pass. */
int cvhip_bn_act_fwd(const void* y_bf16, int32_t ld_y, void* z_bf16, int32_t ld_z, int64_t M,
                     int32_t C, const float* scale, const float* shift, int32_t act,
                     float act_param, const void* residual_bf16, int32_t ld_res, void* stream);

/* z = act(y*scale + shift + residual): BatchNorm apply, skip-connection add and activation of a ResNet bottleneck tail
 * (torchvision resnet.Bottleneck.forward: `out += identity; out = self.relu(out)`) in one pass. */
int cvhip_bn_add_act_fwd(const void* y_bf16, int32_t ld_y, void* z_bf16, int32_t ld_z, int64_t M,
                         int32_t C, const float* scale, const float* shift, int32_t act,
                         float act_param, const void* residual_bf16, int32_t ld_res, void* stream);

/* backward stage 1: partial[rows][2][C] = (sum du, sum du*xhat),  du = dz*act'(u),
 * u = y*scale+shift, xhat = (y-mean)*invstd. */
int cvhip_bn_act_bwd_partial(const void* dz_bf16, int32_t ld_dz, const void* y_bf16, int32_t ld_y,
                             int64_t M, int32_t C, const float* scale, const float* shift,
                             const float* mean, const float* invstd, int32_t act,
                             float act_param, float* partial, void* stream);
/* backward finalize: dbeta = sum du, dgamma = sum du*xhat (sums over partial rows), consumed by
 * cvhip_bn_act_bwd_apply; if acc_dgamma / acc_dbeta are non-NULL the same sums are also ADDED there
 * (the parameter's slot in a flat gradient arena). */
int cvhip_bn_bwd_finalize(const float* partial, int32_t rows, int32_t C, float* dgamma,
                          float* dbeta, float* acc_dgamma, float* acc_dbeta, void* stream);
/* backward stage 2: dy = gamma*invstd*(du - dbeta/M - xhat*dgamma/M)   (training BN)
 *   if mean==NULL (no BN / eval BN): dy = scale*du (scale may be NULL -> dy = du). */
int cvhip_bn_act_bwd_apply(const void* dz_bf16, int32_t ld_dz, const void* y_bf16, int32_t ld_y,
                           void* dy_bf16, int32_t ld_dy, int64_t M, int32_t C, const float* scale,
                           const float* shift, const float* mean, const float* invstd,
                           const float* dgamma, const float* dbeta, int32_t act, float act_param,
                           void* stream);

/* column sum of a bf16 matrix into fp32 (conv bias gradient): out[c] = sum_m x[m][c] */
int cvhip_colsum_partial(const void* x_bf16, int64_t M, int32_t C, int32_t ld, float* partial,
                         void* stream);
int cvhip_colsum_finalize(const float* partial, int32_t rows, int32_t C, float* out, int accumulate,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / resampling / layout glue (all NHWC bf16 with row pitches)
 * ------------------------------------------------------------------------------------------ */
/* max_pool2d forward (+ uint8 argmax window offset for backward). First maximum in row-major
 * window scan order wins ties (ATen CPU rule).  src/models/modules/yolo_modules.py:176-192
 * (SPPF), src/models/modules/yolov7_modules.py:39,53,130, src/models/backbones/seg/resnet.py:80 */
int cvhip_maxpool2d_fwd(const void* x_bf16, int32_t ld_x, void* y_bf16, int32_t ld_y,
                        uint8_t* argmax, int32_t N, int32_t C, int32_t H, int32_t W, int32_t k,
                        int32_t stride, int32_t pad, void* stream);
int cvhip_maxpool2d_bwd(const void* dy_bf16, int32_t ld_dy, const uint8_t* argmax, void* dx_bf16,
                        int32_t ld_dx, int32_t N, int32_t C, int32_t H, int32_t W, int32_t k,
                        int32_t stride, int32_t pad, int accumulate, void* stream);

/* out[:, :Ca] = nearest_upsample_x2(a); out[:, Ca:Ca+Cb] = b   — the FPN/PAN "up + cat"
 * (src/models/modules/yolo_modules.py:147,152 UpsamplingModule). b may be NULL (Cb=0). */
int cvhip_upsample2x_cat_fwd(const void* a_bf16, int32_t ld_a, int32_t Ca, const void* b_bf16,
                             int32_t ld_b, int32_t Cb, void* out_bf16, int32_t ld_out, int32_t N,
                             int32_t Ha, int32_t Wa, void* stream);
/* da = 2x2 sum-pool of dout[:, :Ca]  (db is the channel slice dout[:, Ca:], a view) */
int cvhip_upsample2x_bwd(const void* dout_bf16, int32_t ld_dout, void* da_bf16, int32_t ld_da,
                         int32_t Ca, int32_t N, int32_t Ha, int32_t Wa, void* stream);

/* zero-fill by a kernel. (hipMemsetAsync nodes proved unreliable under hipGraph replay on ROCm 7.2; the
 * library never issues memset/memcpy calls, so a captured step contains kernel nodes only.) */
int cvhip_zero_fill(void* ptr, int64_t bytes, void* stream);
/* dst[k][t][c] += src[k][t][c] for k < K_valid, c < C_valid; src is [K][T][C] fp32 (padded wgrad result),
 * dst is [K_valid][T][C_valid] fp32 (the parameter's gradient). */
int cvhip_f32_unpad_add(const float* src, float* dst, int32_t K_valid, int32_t T, int32_t C,
                        int32_t C_valid, void* stream);

/* strided 2-D copy dst[m][0:C] = src[m][0:C] (channel concat / slice materialisation) */
int cvhip_copy2d(const void* src_bf16, int32_t ld_src, void* dst_bf16, int32_t ld_dst, int64_t M,
                 int32_t C, void* stream);
/* dst = a + b  (gradient accumulation at fan-out points, residual adds) */
int cvhip_add2d(const void* a_bf16, int32_t ld_a, const void* b_bf16, int32_t ld_b, void* dst_bf16,
                int32_t ld_dst, int64_t M, int32_t C, void* stream);

/* out = act(a + b): the ResNet bottleneck tail relu(bn3(conv3) + identity) (torchvision Bottleneck, used through
 * src/models/backbones/seg/resnet.py:91-94). Backward: du = dz*act'(out) via cvhip_bn_act_bwd_apply on `out`. */
int cvhip_add_act_fwd(const void* a_bf16, int32_t ld_a, const void* b_bf16, int32_t ld_b,
                      void* out_bf16, int32_t ld_out, int64_t M, int32_t C, int32_t act,
                      float act_param, void* stream);

/* y[n][hw][c] = x[n][hw][c] * scale[n][c] — Dropout2d apply and its backward
 * (src/models/heads/seg/base_seg_head.py:25-37; the Bernoulli mask/(1-p) is drawn by the host). */
int cvhip_scale_nc(const void* x_bf16, int32_t ld_x, const float* scale_nc, void* y_bf16,
                   int32_t ld_y, int32_t N, int32_t C, int32_t HW, void* stream);

/* Per-pixel softmax cross-entropy with ignore_index, reduction 'mean' over non-ignored pixels
 * (nn.CrossEntropyLoss: src/losses/seg/cross_entropy_loss.py:32-40). logits: [M][ld] bf16 NHWC (M = N*H*W),
 * target int64 [M]. fwd: partial = float[2*cvhip_seg_ce_rows(M)] scratch, out2 = device float[2] {mean loss,
 * #valid}. bwd: dlogits = grad_scale[0] * (softmax - onehot)/#valid (zeros for ignored pixels and pad channels);
 * grad_scale is a DEVICE scalar (the upstream gradient of the loss), NULL = 1. */
int cvhip_seg_ce_rows(int64_t M);
int cvhip_seg_ce_fwd(const void* logits_bf16, int32_t ld, const int64_t* target, int64_t M,
                     int32_t C, int32_t ignore_index, float* partial, float* out2, void* stream);
int cvhip_seg_ce_bwd(const void* logits_bf16, int32_t ld, const int64_t* target, int64_t M,
                     int32_t C, int32_t ignore_index, const float* out2, const float* grad_scale,
                     void* dlogits_bf16, int32_t ld_d, void* stream);

/* Bilinear resize of the logits to label size + the cross-entropy above in ONE pass (encoder_decoder.py:93-107 does
 * F.interpolate(seg_logit, size=label.shape, mode='bilinear', align_corners=...) and then the loss): the label-resolution logits and
 * their gradient never exist in memory. x: [N][Hi][Wi][ld_x] low-resolution logits, target int64 [N][Ho][Wo]; partial / out2 /
 * grad_scale as for cvhip_seg_ce_fwd / _bwd with M = N*Ho*Wo; bwd writes dx [N][Hi][Wi][ld_dx] = d loss / d x (pad channels zero),
 * deterministically (a gather, no atomics). The interpolated logits are kept in fp32 (the two-op form rounds them to 16 bits).
 * cvhip_seg_ce_bilinear_ok: 1 when the geometry is supported (upsampling, C <= 32, footprint tile fits the LDS), else the entry
 * points return CVHIP_ERR_UNSUPPORTED and the caller composes cvhip_resize_bilinear_* with cvhip_seg_ce_*. */
int cvhip_seg_ce_bilinear_ok(int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t align_corners);
int cvhip_seg_ce_bilinear_fwd(const void* x_bf16, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                              int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index, float* partial, float* out2,
                              void* stream);
int cvhip_seg_ce_bilinear_bwd(const void* x_bf16, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                              int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* out2,
                              const float* grad_scale, void* dx_bf16, int32_t ld_dx, void* stream);

/* Per-pixel forms of the fused resize + cross-entropy (round 6) for OHEM (src/losses/seg/cross_entropy_loss.py:51-69:
 * F.cross_entropy(..., reduction='none') on the label-resolution logits, then a data-dependent selection of the hard pixels):
 *   fwd_px : loss_px[N*Ho*Wo] = -log p_target of every label pixel (0 where ignored); nothing is reduced
 *   bwd_px : dx = grad_scale[0] * sum_m w_px[m] * d(-log p_target(m)) / dx with per-pixel weights w_px in [0, 1] (the selection
 *            mask, kept in 16 bits inside the kernel: put the common 1 / count into the DEVICE scalar grad_scale, NULL = 1)
 * Same geometry limits as cvhip_seg_ce_bilinear_ok. */
int cvhip_seg_ce_bilinear_fwd_px(const void* x_bf16, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                 int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index, float* loss_px, void* stream);
int cvhip_seg_ce_bilinear_bwd_px(const void* x_bf16, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                 int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* w_px,
                                 const float* grad_scale, void* dx_bf16, int32_t ld_dx, void* stream);

/* OHEM selection on the device (cross_entropy_loss.py:51-69: sort the per-pixel losses, branch on loss_sorted[min_kept] > thresh, average
 * either the losses above the threshold or the min_kept largest): the (min_kept + 1)-th largest value of loss_px * loss_weight by a
 * three-pass radix select, the masked sums by one two-stage reduction — no sort, no host read.
 *   sel8 (DEVICE float[8]): [0] the loss, [1] d loss / d(selected per-pixel CE) incl. loss_weight, [2] common weight of the pixels tied
 *   at the cut, [3] 1 = threshold branch, [4] the cut value v, [5] thresh_nlog = -log(thresh), [6] #pixels above v, [7] #pixels tied.
 * cvhip_seg_ce_bilinear_bwd_ohem: the weighted backward of the fused resize + cross-entropy with the weights derived from loss_px and
 * sel8 inside the kernel (times the DEVICE scalar grad_scale, NULL = 1). workspace: cvhip_ohem_select_workspace_bytes(). */
int64_t cvhip_ohem_select_workspace_bytes(void);
int cvhip_ohem_select(const float* loss_px, int64_t M, int32_t min_kept, float thresh_nlog, float loss_weight, void* workspace,
                      float* sel8, void* stream);
int cvhip_seg_ce_bilinear_bwd_ohem(const void* x_bf16, int32_t ld_x, const int64_t* target, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                   int32_t Ho, int32_t Wo, int32_t align_corners, int32_t ignore_index, const float* loss_px,
                                   float loss_weight, const float* sel8, const float* grad_scale, void* dx_bf16, int32_t ld_dx,
                                   void* stream);

/* Boundary targets of the STDC detail loss (src/losses/seg/detail_loss.py:37-79): Laplacian pyramid of the int64 label map
 * (3x3 kernel, padding 1, strides 1 / 2 / 4, clamp(min=0), nearest up-sampling, threshold), fused with weights 0.6 / 0.3 / 0.1 and
 * thresholded again -> out fp32 [N][H][W] in {0, 1}. Replaces three F.conv2d + two F.interpolate + cat + a 1x1 F.conv2d per step. */
int cvhip_detail_boundary_targets(const int64_t* labels, int32_t N, int32_t H, int32_t W, float threshold, float* out, void* stream);

/* nearest-neighbour resize to an arbitrary size: F.interpolate(x, size, mode="nearest") of the STDC neck (src/models/necks/seg:
 * stdc neck `F.interpolate(..., mode='nearest')` calls; torch's index rule src = min(floor(dst * in/out), in-1)). Forward is an
 * exact copy (bit-exact); backward a deterministic gather-sum over the output pixels of each input pixel. */
int cvhip_resize_nearest_fwd(const void* x, int32_t ld_x, void* y, int32_t ld_y, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, void* stream);
int cvhip_resize_nearest_bwd(const void* dy, int32_t ld_dy, void* dx, int32_t ld_dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                             int32_t Ho, int32_t Wo, void* stream);

/* bilinear resize, align_corners = 0/1 (F.interpolate).
 * src/models/heads/seg/deeplabv3plus_head.py:56-66, segmentors/encoder_decoder.py:99 */
int cvhip_resize_bilinear_fwd(const void* x_bf16, int32_t ld_x, void* y_bf16, int32_t ld_y,
                              int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                              int32_t align_corners, void* stream);
int cvhip_resize_bilinear_bwd(const void* dy_bf16, int32_t ld_dy, void* dx_bf16, int32_t ld_dx,
                              int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                              int32_t align_corners, void* stream);
/* The same backward in separable form for upsampling ratios >= 4 on both axes (encoder_decoder.py / the DeepLabv3+ head upsample
 * x8): a vertical pass reads dy ONCE into an fp32 workspace [N][Hi][Wo][round8(C)], a horizontal pass reduces it — the gather form
 * fetches every dy element four times. workspace_bytes returns 0 when the gather form is the better one; _ws then falls back to it
 * (also when the workspace is NULL / too small), so callers may always use _ws. Deterministic (no atomics), fp32 intermediate. */
int64_t cvhip_resize_bilinear_bwd_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo);
int cvhip_resize_bilinear_bwd_ws(const void* dy_bf16, int32_t ld_dy, void* dx_bf16, int32_t ld_dx, int32_t N, int32_t C, int32_t Hi,
                                 int32_t Wi, int32_t Ho, int32_t Wo, int32_t align_corners, void* workspace, int64_t ws_bytes,
                                 void* stream);

/* global average pool (AdaptiveAvgPool2d(1)) fwd/bwd: y[n][c] = mean_hw x */
int cvhip_global_avgpool_fwd(const void* x_bf16, int32_t ld_x, void* y_bf16, int32_t N, int32_t C,
                             int32_t HW, void* stream);
/* ds[n][c] = sum_p dy[n][p][c] * x[n][p][c] (fp32, [N][C]): the gate's gradient of the channel scaling y = x * s[n][c] of STDC's
 * attention-refinement / feature-fusion modules (necks/seg/stdc_neck.py:53-58,110-114; aten mul + sum in the reference's backward).
 * C % 8 == 0, pitches % 8 == 0, 16-byte-aligned bases; else CVHIP_ERR_UNSUPPORTED (the host falls back to torch). */
int cvhip_channel_scale_bwd_ds(const void* dy, int32_t ld_dy, const void* x, int32_t ld_x, float* ds, int32_t N, int32_t C, int32_t HW, void* stream);
int cvhip_global_avgpool_bwd(const void* dy_bf16, void* dx_bf16, int32_t ld_dx, int32_t N,
                             int32_t C, int32_t HW, void* stream);

/* uint8 NHWC images (as the CPU loader produces them before ToTensor) -> normalised bf16 NHWC, channels zero-padded to ld:
 * y = x * scale[c] + shift[c] with scale = 1/(255*std), shift = -mean/std == ToTensor + Normalize
 * (src/data/transforms/det_transforms.py:102-109, conf/coco_yolov5_s.yml:36-37) fused with the stem's layout. C <= 8. */
int cvhip_u8_nhwc_to_bf16_norm(const void* x_u8, int64_t npix, int32_t C, void* y_bf16, int32_t ld, const float* scale,
                               const float* shift, void* stream);
/* fp32 NCHW image batch -> bf16 NHWC with channels zero-padded to Cpad (trainer.py:157-175 H2D
 * boundary + channels_last relayout); and Focus space-to-depth (yolo_modules.py:30-36,
 * concat order TL, BL, TR, BR) fused with the same relayout. */
int cvhip_nchw_f32_to_nhwc_bf16(const float* x, void* y_bf16, int32_t N, int32_t C, int32_t H,
                                int32_t W, int32_t Cpad, void* stream);
int cvhip_focus_nchw_f32_to_nhwc_bf16(const float* x, void* y_bf16, int32_t N, int32_t C,
                                      int32_t H, int32_t W, int32_t Cpad, void* stream);
/* bf16 NHWC (pitch ld) -> fp32 NCHW (logits handed to torch-side losses) and its adjoint */
int cvhip_nhwc_bf16_to_nchw_f32(const void* x_bf16, int32_t ld, float* y, int32_t N, int32_t C,
                                int32_t H, int32_t W, void* stream);
int cvhip_nchw_f32_to_nhwc_bf16_ld(const float* x, void* y_bf16, int32_t ld, int32_t N, int32_t C,
                                   int32_t H, int32_t W, void* stream);

/* YOLO head boundary: bf16 NHWC (N,H,W,ld >= A*NO) <-> fp32 (N,A,H,W,NO) contiguous; fuses the
 * reference's view/permute/contiguous (src/models/detects/yolov5_detect.py:43-44) with the fp32 cast
 * the loss needs. The backward writes all `ld` channels (pad channels zero). */
int cvhip_head_permute_fwd(const void* x_bf16, int32_t ld, float* y, int32_t N, int32_t A,
                           int32_t NO, int32_t H, int32_t W, void* stream);
int cvhip_head_permute_bwd(const float* dy, void* dx_bf16, int32_t ld, int32_t N, int32_t A,
                           int32_t NO, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Detection post-processing (bit-exact index semantics)
 * ------------------------------------------------------------------------------------------ */
/* YOLOv5 eval decode: src/models/detects/yolov5_detect.py:48-55.
 * p: bf16 NHWC head output (N, H, W, ld>=A*NO) ; out fp32 (N, A*H*W, NO) with the reference's
 * anchor-major ordering; xy=(sig*2-0.5+grid)*stride, wh=(sig*2)^2*anchor_px, rest=sigmoid. */
int cvhip_yolov5_decode(const void* p_bf16, int32_t ld, float* out, int32_t N, int32_t A,
                        int32_t NO, int32_t H, int32_t W, float stride, const float* anchors_px,
                        int64_t out_image_stride, int64_t out_level_offset, void* stream);

/* Greedy NMS on boxes ALREADY sorted by descending score (stable) — torchvision.ops.nms contract
 * (models/yolov5.py:137): keep i, suppress later j with IoU(i,j) > thr, IoU=inter/(a_i+a_j-inter),
 * area=(x2-x1)*(y2-y1). Two kernels: 64x64-tile suppression bitmask (ballot), then a single-wave
 * sequential scan. mask: uint64 [n][ceil(n/64)] workspace. keep_idx: int32[n], keep_count: int32[1]. */
int64_t cvhip_nms_workspace_bytes(int32_t n);
int cvhip_nms_sorted(const float* boxes_xyxy, int32_t n, float iou_thr, void* workspace,
                     int32_t* keep_idx, int32_t* keep_count, void* stream);

/* Batched detection post-processing (post_batch.hip): confidence filter -> top-`cap` selection by score (descending, ties by
 * ascending row = a stable sort) -> class-offset boxes -> greedy NMS -> fixed-capacity outputs, for ALL images of a batch in one
 * launch set with no host round trip. Replaces the per-image python loops (boolean-mask indexing, torch.sort, torchvision.ops.nms)
 * of src/models/yolov5.py:62-153 non_max_suppression (mode 0: best-class path, class offset `class_offset` = max_wh 4096, or 0
 * for agnostic) and src/models/yolox.py:48-68 yolox_post_process (mode 1: torchvision.ops.batched_nms, offset = idx * (max box
 * coordinate of the image + 1)).
 *   pred     : fp32 [B][n][no] decoded rows {cx, cy, w, h, obj, cls[nc], ...}
 *   multi_label (mode 0): every (row, class) with obj*cls > conf_thres is a detection of its own (yolov5.py:106-108, the val
 *              path); `cand_cap` = per-image size of the candidate buffer in the workspace (>= n; n*nc bounds the multi_label case —
 *              candidates beyond it are dropped and flagged in `overflow`)
 *   dets     : fp32 [B][max_det][6] {x1,y1,x2,y2,conf,cls} (mode 0) / [B][max_det][7] {x1,y1,x2,y2,obj,class_conf,cls} (mode 1);
 *              rows >= counts[b] are zero
 *   overflow : 1 when an image had more than `cap` candidates (the `cap` best were kept: the reference's rule with
 *              max_nms := cap; the reference's own max_nms is 30000). cap: a power of two in [64, 8192].
 * Arithmetic (xywh->xyxy, offsets, IoU predicate) is the reference's fp32 arithmetic: results are bit-exact. */
int64_t cvhip_detect_postprocess_workspace_bytes(int32_t B, int32_t cand_cap, int32_t cap);
int cvhip_detect_postprocess(const float* pred, int32_t B, int32_t n, int32_t no, int32_t nc, float conf_thres, float iou_thres,
                             float class_offset, int32_t mode, int32_t multi_label, int32_t cand_cap, int32_t cap, int32_t max_det,
                             void* workspace, float* dets, int32_t* counts, int32_t* overflow, void* stream);
/* device argsort, descending by score, ties by ascending index (stable): order[i] = index of the i-th best. Any n <= 2^29
 * (LDS bitonic blocks + global merge steps). The sort behind batched_nms / multiclass_nms (src/models/modules/nms.py:5-132)
 * instead of torch.sort. workspace: cvhip_sort_workspace_bytes(n). */
int64_t cvhip_sort_workspace_bytes(int64_t n);
int cvhip_argsort_desc_f32(const float* scores, int64_t n, void* workspace, int64_t* order, void* stream);

/* pairwise IoU matrix (N x M) fp32 — models/yolov5.py:27-49 box_iou, losses/det/yolox_loss.py:14-31 */
int cvhip_box_iou(const float* a_xyxy, int32_t n, const float* b_xyxy, int32_t m, float* out,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused optimizer: SGD(momentum, nesterov, weight decay) + EMA over a flat fp32 parameter arena.
 * torch.optim.SGD semantics (src/optimizers/__init__.py:60-68) + ModelEMA.update
 * (src/utils/ema.py:30-39). wd/lr are per-element-range via a segment table.
 *   seg: int64 [nseg][2] = {begin, end} element ranges; seg_lr/seg_wd: float[nseg].
 * ------------------------------------------------------------------------------------------ */
/* AdamW (decoupled weight decay; torch.optim.AdamW, src/optimizers/__init__.py:71-73 — conf/mini-imagenet.yml:91-99 trains config 1
 * with it) + ModelEMA over the same flat arenas and segment table. step_state: ONE float in device memory holding the step count t
 * (zero-initialised by the caller; the call increments it before use — not on a GradScaler skip — so a captured step replays with the
 * right bias corrections). scaler2 (optional): {1/scale, skip} of the dynamic loss scaler, as in cvhip_sgd_nesterov_ema_scaled. */
int cvhip_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t n, const int64_t* seg_bounds,
                    const float* seg_lr, const float* seg_wd, int32_t nseg, float beta1, float beta2, float eps, float* step_state,
                    float ema_decay, float grad_scale, const float* dyn_decay_lrscale, const float* scaler2, void* stream);
int cvhip_sgd_nesterov_ema(float* param, const float* grad, float* momentum_buf, float* ema,
                           int64_t n, const int64_t* seg_bounds, const float* seg_lr,
                           const float* seg_wd, int32_t nseg, float momentum, int32_t nesterov,
                           int32_t first_step, float ema_decay, float grad_scale,
                           const float* dyn_decay_lrscale, void* stream);

/* Dynamic loss scaling for fp16 storage — torch.cuda.amp.GradScaler on device (trainer.py:189-201: scaler.scale(loss).backward(),
 * scaler.step(optimizer), scaler.update(); defaults init 65536, growth 2, backoff 0.5, interval 2000). No host round trip: the
 * whole step, scaler included, stays one hipGraph.
 *   state4  : fp32 {scale, growth_tracker, found_inf, skipped_steps}; the loss gradient is seeded with state4[0]
 *   check   : found_inf = 1 when any of the n gradient-arena values is inf / NaN (run after the gradient all-reduce)
 *   update  : scaler2 = {found_inf ? 0 : 1/scale, found_inf}; scale *= backoff on overflow, *= growth after `growth_interval`
 *             clean steps in a row; found_inf cleared
 *   cvhip_sgd_nesterov_ema_scaled : the fused optimizer with scaler2: gradients are multiplied by scaler2[0]; when scaler2[1] != 0
 *             parameters and momentum are left untouched (the step is skipped), the EMA still follows the parameters. */
int cvhip_loss_scale_check(const float* grad_arena, int64_t n, float* state4, void* stream);
int cvhip_loss_scale_update(float* state4, float* scaler2, float growth_factor, float backoff_factor, int32_t growth_interval,
                            void* stream);
int cvhip_sgd_nesterov_ema_scaled(float* param, const float* grad, float* momentum_buf, float* ema, int64_t n,
                                  const int64_t* seg_bounds, const float* seg_lr, const float* seg_wd, int32_t nseg,
                                  float momentum, int32_t nesterov, int32_t first_step, float ema_decay, float grad_scale,
                                  const float* dyn_decay_lrscale, const float* scaler2, void* stream);
/* dyn_decay_lrscale: optional DEVICE float[2] = {ema_decay, lr_scale}; when non-NULL it overrides
 * `ema_decay` and multiplies every segment lr, so the values can change between hipGraph replays. */
/* ema[i] = d*ema[i] + (1-d)*src[i] over a flat fp32 range (buffers: BN running stats);
 * dyn_decay: optional DEVICE float[1] overriding `decay`. */
/* v[i] += delta for an int64 array (the BatchNorm num_batches_tracked counters of a model, kept in one buffer by the flat
 * training state: one launch per step instead of one add per layer) */
int cvhip_i64_add(int64_t* v, int64_t n, int64_t delta, void* stream);
int cvhip_ema_update(float* ema, const float* src, int64_t n, float decay, const float* dyn_decay,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * YOLOv5 loss on device, one detection level at a time.
 * Replaces src/losses/yolov5_loss.py:173-278 (YOLOv5Loss.__call__ + build_targets) and :12-54 (bbox_iou/CIoU): reads the
 * bf16 NHWC head map (N, H, W, ld) with channel a*NO + o directly (no (N,A,H,W,NO) fp32 copy) and writes its bf16
 * gradient directly. targets: (T, 6) fp32 [img, cls, cx, cy, w, h] normalised, img < 0 = padding row.
 *   level_fwd : build_targets + per-candidate CIoU / class BCE (+ their unscaled gradients, kept in `ws`) + objectness BCE
 *               -> sums4 = {n_matched, sum(1 - ciou), sum(cls bce), sum(obj bce)}  (deterministic reductions)
 *   finalize  : lbox = hyp_box * sum_l lbox_l / n_l ...; total = (lbox + lobj + lcls) * batch; stats3 = {lbox, lobj, lcls}
 *   level_bwd : d total / d head map, scaled by the DEVICE scalar gout (NULL = 1):
 *               k_box = hyp_box*batch, k_cls = hyp_cls*batch/nc, k_obj = hyp_obj*balance_l*batch/(N*A*H*W)
 * `ws` (cvhip_yolov5_loss_workspace_bytes) must stay untouched between level_fwd and level_bwd.
 * ------------------------------------------------------------------------------------------ */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_yolo_loss_desc {
  int32_t N, A, NO, H, W, ld, T;
  float anchor_t;       /* hyp anchor_t = 4.0 (yolov5_loss.py:247-248) */
  float anchors[16];    /* A x (w, h) in grid units of this level */
} cvhip_yolo_loss_desc;
#endif
int64_t cvhip_yolov5_loss_workspace_bytes(const cvhip_yolo_loss_desc* d);
int cvhip_yolov5_loss_level_fwd(const cvhip_yolo_loss_desc* d, const void* raw_bf16, const float* targets, void* ws,
                                float* sums4, void* stream);
int cvhip_yolov5_loss_finalize(const float* sums, int32_t levels, const float* ncell, const float* balance, float hyp_box,
                               float hyp_obj, float hyp_cls, int32_t nc, float batch, float* total, float* stats3,
                               void* stream);
int cvhip_yolov5_loss_level_bwd(const cvhip_yolo_loss_desc* d, const void* raw_bf16, const float* targets, void* ws,
                                const float* sums4, const float* gout, float k_box, float k_cls, float k_obj,
                                void* draw_bf16, void* stream);
/* level_bwd that also produces the COLUMN SUMS of the gradient map — the bias gradient of the detect convolution in front of the
 * loss (yolov5_head.py: nn.Conv2d(ch, na * no, 1) with bias; aten::convolution_backward's bias output reached from trainer.py:189) —
 * from the loss's own compact state instead of a pass over the map: bias_partial = fp32 [CVHIP_YOLO_BIAS_ROWS][2][A*NO] partial rows
 * (first half of every row written) for cvhip_colsum_finalize(bias_partial, CVHIP_YOLO_BIAS_ROWS, A*NO, out, accumulate). The sums are
 * taken BEFORE the 16-bit rounding of the map's entries; deterministic. */
#define CVHIP_YOLO_BIAS_ROWS 256
int cvhip_yolov5_loss_level_bwd_bias(const cvhip_yolo_loss_desc* d, const void* raw_bf16, const float* targets, void* ws,
                                     const float* sums4, const float* gout, float k_box, float k_cls, float k_obj,
                                     void* draw_bf16, float* bias_partial, void* stream);

/* ------------------------------------------------------------------------------------------
 * YOLOv7 OTA label assignment on device (ota_assign.hip) + the loss on that assignment.
 * Replaces src/losses/yolov7_loss.py:217-365 (build_targets: per-image python loop, boolean-mask compaction, torch.topk, per-gt
 * `.item()` loops over the find_3_positive candidates :367-420). raws[l]: 16-bit NHWC head map of level l (N, H_l, W_l, ld_l) with
 * channel a*NO + o; targets (T, 6) fp32 [img, cls, cx, cy, w, h] normalised, rows of one image contiguous, img < 0 = padding.
 *   cvhip_ota_assign : assign[l][c], c = (offset*A + anchor)*T + target (the candidate ordinal of the YOLOv5 loss kernels) =
 *                      matched flat target row, or -1. G = capacity of targets per image (images with more: the extra targets get
 *                      no candidates, flagged via cvhip_ota_read_overflow); L*5*A*G <= 4096.
 *   cvhip_yolov5_loss_level_fwd_assigned : cvhip_yolov5_loss_level_fwd with the positives taken from `assign` (one level's slice):
 *                      the candidate keeps its cell, box / class targets come from the matched row. finalize / level_bwd unchanged
 *                      (hyp 0.05 / 0.7 / 0.3, balance 4 / 1 / 0.4: yolov7_loss.py:129-215).
 * ------------------------------------------------------------------------------------------ */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_ota_desc {
  int32_t L, N, A, NO, T, G;
  int32_t H[4], W[4], ld[4];
  float stride[4];
  float anchors[4][16]; /* per level: A x (w, h) in grid units */
  float anchor_t;       /* 4.0 */
  float img_size;       /* the reference's imgs[b].shape[1] */
} cvhip_ota_desc;
#endif
int64_t cvhip_ota_workspace_bytes(const cvhip_ota_desc* d);
int cvhip_ota_assign(const cvhip_ota_desc* d, const void* const* raws, const float* targets, void* ws, int32_t* assign, void* stream);
int cvhip_ota_read_overflow(const cvhip_ota_desc* d, const void* ws, int32_t* out_device, void* stream);
int cvhip_yolov5_loss_level_fwd_assigned(const cvhip_yolo_loss_desc* d, const void* raw_bf16, const float* targets, const int32_t* assign,
                                         void* ws, float* sums4, void* stream);

/* ------------------------------------------------------------------------------------------
 * YOLOX loss on device: SimOTA assignment + 5*IoU^2 + objectness + class BCE, all levels at once.
 * Replaces src/losses/det/yolox_loss.py:73-435 (per-image python loop, per-gt `.item()` top-k loop). raws[l]: bf16 NHWC head
 * map of level l, (B, H_l, W_l, ld_l) with channels [reg 4, obj 1, cls nc] (heads/det/yolox_head.py:94); targets: (B, G, 5) fp32
 * [cls, cx, cy, w, h] in pixels, all-zero rows = padding (models/yolox.py:112-139). A = sum_l H_l*W_l anchors.
 *   loss_fwd : out5 = {loss, conf_loss, cls_loss, 5*iou_loss, num_fg/num_gts}; assignment + intermediates stay in `ws`
 *   loss_bwd : draws[l] (same shape/pitch as raws[l]) = d loss / d raws[l], scaled by the DEVICE scalar gout (NULL = 1)
 * ------------------------------------------------------------------------------------------ */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_simota_desc {
  int32_t L, B, A, G, nc;
  int32_t H[4], W[4], ld[4];
  float stride[4];
} cvhip_simota_desc;
#endif
int64_t cvhip_simota_workspace_bytes(const cvhip_simota_desc* d);
int cvhip_simota_loss_fwd(const cvhip_simota_desc* d, const void* const* raws, const float* targets, void* ws, float* out5,
                          void* stream);
int cvhip_simota_loss_bwd(const cvhip_simota_desc* d, const void* const* raws, const float* targets, void* ws,
                          const float* gout, void* const* draws, void* stream);
int cvhip_simota_read_assignment(const cvhip_simota_desc* d, void* ws, int32_t* matched_out, float* miou_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel collectives (comm.hip): an RCCL communicator behind the C ABI, one process per GPU over xGMI.
 * Replaces torch.nn.parallel.DistributedDataParallel's reducer + ProcessGroupNCCL reached from trainer.py:312-313 (DDP wrap:
 * bucketed gradient all-reduce overlapped with backward, buffer broadcast) and src/utils/distributed.py:82-98
 * (init_process_group('nccl')); SyncBatchNorm's statistics exchange (trainer.py:126-127) uses cvhip_comm_allreduce too.
 * librccl is bound at run time (dlopen; the copy already mapped into the process wins, then $CVHIP_RCCL_PATH, librccl.so,
 * librccl.so.1). Rendezvous is the host's business: rank 0 calls cvhip_comm_get_unique_id and hands the
 * cvhip_comm_unique_id_bytes() opaque bytes to the other ranks over any channel; every rank then calls
 * cvhip_comm_init_rank with ITS HIP device current. All collectives are in place, asynchronous on `stream`, non-allocating
 * and capturable in a hipGraph.
 * ------------------------------------------------------------------------------------------ */
#define CVHIP_DTYPE_F32 0
#define CVHIP_DTYPE_F64 1
#define CVHIP_DTYPE_I32 2
#define CVHIP_DTYPE_BF16 3
#define CVHIP_DTYPE_U8 4
#define CVHIP_RED_SUM 0
#define CVHIP_RED_MAX 1
#define CVHIP_RED_MIN 2
int cvhip_comm_available(void);     /* 1 when librccl could be bound in this process */
int cvhip_comm_rccl_version(void);  /* ncclGetVersion code, 0 when unavailable */
int cvhip_comm_unique_id_bytes(void);
int cvhip_comm_get_unique_id(void* id_out);
int cvhip_comm_init_rank(void** comm_out, int32_t world, int32_t rank, const void* unique_id);
int cvhip_comm_destroy(void* comm);
int cvhip_comm_world(void* comm);
int cvhip_comm_rank(void* comm);
/* gradient bucket: in-place SUM all-reduce of `count` fp32 values of the gradient arena (the 1/world average is folded into
 * the optimizer kernel's grad_scale) */
int cvhip_allreduce_bucket(void* comm, void* buf_f32, int64_t count, void* stream);
int cvhip_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream);
/* DDP broadcast_buffers / initial parameter broadcast: `bytes` raw bytes from rank `root` */
int cvhip_comm_broadcast(void* comm, void* buf, int64_t bytes, int32_t root, void* stream);
/* the all-reduce split in its two ring phases (count % world == 0): after reduce_scatter rank r owns the complete sum of
 * chunk r; all_gather then publishes every chunk. Equivalent to cvhip_allreduce_bucket; lets the host overlap the
 * all-gather with the next forward or shard the optimizer between the phases. */
int cvhip_comm_reduce_scatter_f32(void* comm, void* buf_f32, int64_t count, void* stream);
int cvhip_comm_all_gather_f32(void* comm, void* buf_f32, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm through statistic accumulators (no partial rows, no finalize launches); see CVHIP_BN_ACC_SHARDS.
 * Replaces the same reference calls as the partial-row forms (bricks/conv_module.py:209-213, native_batch_norm(_backward)).
 * ------------------------------------------------------------------------------------------ */
int cvhip_bn_acc_shards(void);
/* "tail" of a dgrad-producing call: the tensor whose gradient the call writes (its dx) is the OUTPUT z = act(bn(y)) of a
 * Conv-BN-act layer P. With a tail the call's epilogue also adds P's BatchNorm-backward sums (sum du, sum du*xhat; du = dx *
 * act'(scale*y + shift), xhat = (y - mean) * invstd, taken over the bf16-rounded dx it stores) into P's accumulator, so P's backward
 * runs no reduction pass over (dz, y). y: P's raw convolution output — same pixel grid and channel count as dx, 16-byte aligned,
 * y_ld % 8 == 0; scale / shift / mean / invstd: P's saved statistics (channel count floats each); act: none / ReLU / LeakyReLU /
 * SiLU; acc: P's backward accumulator [CVHIP_BN_ACC_SHARDS][2][acc_ld]. */
typedef struct cvhip_bn_tail {
  const void* y;
  int32_t y_ld;
  const float *scale, *shift, *mean, *invstd;
  int32_t act;
  float act_param;
  double* acc;
  int32_t acc_ld;
} cvhip_bn_tail;
/* cvhip_conv2d_dgrad / _dgrad_add (addend may be NULL) with a tail; CVHIP_ERR_UNSUPPORTED when the geometry has no packed
 * epilogue (C % 8, dx pitch / alignment) — the caller then runs the plain dgrad and the layer's own reduction pass */
int cvhip_conv2d_dgrad_tail(const cvhip_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, int32_t addend_ld, void* dx,
                            const cvhip_bn_tail* tail, void* stream);
/* convolution (no bias) whose epilogue adds (sum y, sum y^2) of the fp32 accumulators into bn_acc[shards][2][K] */
int cvhip_conv2d_fprop_acc(const cvhip_conv_desc* d, const void* x, const void* w, void* y, double* bn_acc, void* stream);
/* z = act(bn(y)) (+ residual; res_pre: before the activation). Every block derives scale / shift of its channels from `acc`
 * (element count `count` per channel); block 0 stores mean | invstd | scale | shift (4 arrays of C floats) for backward and updates
 * the running statistics exactly as cvhip_bn_finalize does. C <= 2048. */
int cvhip_bn_act_fwd_acc(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const double* acc, int32_t acc_ld,
                         int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, float* mean, float* invstd, float* scale, float* shift, int32_t act, float act_param,
                         const void* residual, int32_t ld_res, int32_t res_pre, void* stream);
/* (sum du, sum du*xhat) of the BN+activation backward into acc[shards][2][acc_ld] (du = dz * act'(scale*y + shift)) */
int cvhip_bn_act_bwd_sums_acc(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, int64_t M, int32_t C, const float* scale,
                              const float* shift, const float* mean, const float* invstd, int32_t act, float act_param, double* acc,
                              int32_t acc_ld, void* stream);
/* (round 5) residual TAIL z = act(bn(y) + identity) of a ResNet bottleneck (torchvision Bottleneck.forward through
 * src/models/backbones/seg/resnet.py:91-94): du = dz * act'(z_out) is STORED (it is the identity branch's gradient and the input of
 * the layer's BN backward) and (sum du, sum du*xhat) are added to `acc`, in one pass over (dz, z_out, y) — what cvhip_bn_act_bwd_apply
 * on z_out followed by cvhip_bn_act_bwd_sums_acc (activation none) computes in two. none / ReLU / LeakyReLU. */
int cvhip_bn_tail_bwd_sums_acc(const void* dz, int32_t ld_dz, const void* z_out, int32_t ld_z, const void* y, int32_t ld_y, void* du, int32_t ld_du,
                               int64_t M, int32_t C, const float* mean, const float* invstd, int32_t act, float act_param, double* acc,
                               int32_t acc_ld, void* stream);
/* dy from (dz, y) with the two sums taken from `acc`; block 0 stores (accumulate != 0: adds) dgamma / dbeta (either may be NULL) */
int cvhip_bn_act_bwd_apply_acc(const void* dz, int32_t ld_dz, const void* y, int32_t ld_y, void* dy, int32_t ld_dy, int64_t M, int32_t C,
                               const float* scale, const float* shift, const float* mean, const float* invstd, const double* acc,
                               int32_t acc_ld, float* dgamma, float* dbeta, int32_t accumulate, int32_t act, float act_param, void* stream);
/* cvhip_conv1x1_bwd_fused with the two sums taken from `acc` (K channels: both siblings'); `acc` may be NULL for a layer
 * without training-mode BatchNorm that still wants a tail */
int cvhip_conv1x1_bwd_fused_acc(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                const void* y, const void* x, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const double* acc, int32_t acc_ld, float* dgamma_out, float* dbeta_out, int32_t accumulate,
                                int32_t act, float act_param, const void* addend, int32_t addend_ld, void* dx, int32_t dx_ld, float* dw,
                                const cvhip_bn_tail* tail /* optional: the layer that produced x */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused convolution (round 4). `ConvModule.forward` is act(norm(conv(x))) (src/models/bricks/conv_module.py:201-214); in eval
 * mode / after utils/fuse.py:32-54 the BatchNorm is a per-channel scale + shift, so the whole module is ONE pass:
 *   EPILOGUE  (every convolution kernel)    y = act((conv(x, w) + bias) * ep_scale + ep_shift)
 *   PROLOGUE  (patch-resident kernel only)  the convolution reads act_p(pro_scale * x + pro_shift) instead of x — x is then the RAW
 *             convolution output of the producing Conv-BN-act layer, whose BN-apply + activation pass disappears; the zero padding
 *             is applied AFTER the activation. z_out (optional, stride-1 "same" geometry): the activated tensor is also stored,
 *             once, for the weight-gradient pass of training.
 * stats_partial / bn_acc: training-mode BatchNorm sums of the raw accumulators as in cvhip_conv2d_fprop / _fprop_acc (then no
 * bias / epilogue). Unused members must be zero. CVHIP_ERR_UNSUPPORTED: the geometry has no kernel with the requested prologue
 * (cvhip_conv2d_fprop_prologue_ok tells beforehand); epilogues are available for every dense geometry.
 * ------------------------------------------------------------------------------------------ */
#ifndef CVHIP_REDECLARE_F16
typedef struct cvhip_conv_fuse {
  const float* bias;
  float* stats_partial;
  double* bn_acc;
  const float* ep_scale;
  const float* ep_shift;
  int32_t ep_act;
  float ep_act_param;
  const float* pro_scale;
  const float* pro_shift;
  int32_t pro_act;
  float pro_act_param;
  void* z_out;
  int32_t z_ld;
  const void* residual; /* optional addend, pitch residual_ld: applied AFTER the activation (Darknet shortcut x + act(bn(conv))) ... */
  int32_t residual_ld;
  int32_t residual_pre; /* ... or, non-zero, BEFORE it: act(bn(conv) + residual), the ResNet bottleneck tail (torchvision Bottleneck) */
  /* image stems (round 4): != NULL = the input is the dataloader's own tensor, fp32 NCHW [N][x_image_planes][H][W] (1..4 real channels;
   * d->C stays 8, d->c_valid = x_image_planes, `x` may be NULL) — read plane by plane and rounded to 16 bits inside the stem kernel, so
   * the layout / precision pass (cvhip_nchw_f32_to_nhwc_bf16) disappears from the step. Only for descriptors the image-stem kernel
   * runs (cvhip_conv_stem_blocks(d) > 0), else CVHIP_ERR_UNSUPPORTED; no prologue, no residual. */
  const float* x_image;
  int32_t x_image_planes;
  /* (round 5) LAZY ACTIVATIONS in training: with pro_scale / pro_shift set and the streaming 1x1 kernel running the problem
   * (cvhip_conv1x1_stream_prologue_ok), the prologue covers the input-channel range [pro_lo, pro_hi) only (0, 0 = every channel):
   * channels outside it — slices of a concatenation that were materialised — are read as they are. pro_scale / pro_shift are indexed
   * by the ABSOLUTE input channel (only [pro_lo, pro_hi) is dereferenced). Combines with stats_partial / bn_acc (the consumer's own
   * training-mode BatchNorm sums); no z_out, residual or epilogue in that form. */
  int32_t pro_lo;
  int32_t pro_hi;
  /* (round 5) SPLIT STORE, streaming 1x1 kernel only (CVHIP_ERR_UNSUPPORTED elsewhere): output channels [y_split, K) are written to y2
   * (pitch y2_ld elements) instead of y — the second layer of a sibling pair (ops.ConvBnActPair: CSP conv1 / conv2,
   * modules/yolo_modules.py:131-139) puts its raw output straight into its slice of the concat buffer. 8-aligned split, 16-byte
   * aligned destinations. */
  void* y2;
  int32_t y2_ld;
  int32_t y_split;
} cvhip_conv_fuse;
/* a lazy input operand of a backward kernel: x' = act(scale[c] * x + shift[c]) for channels [c_lo, c_hi) (c_hi 0 = all), as above */
typedef struct cvhip_lazy_in {
  const float* scale;
  const float* shift;
  int32_t act;
  float act_param;
  int32_t c_lo;
  int32_t c_hi;
} cvhip_lazy_in;
#endif
int cvhip_conv2d_fprop_fused(const cvhip_conv_desc* d, const void* x, const void* w, void* y, const cvhip_conv_fuse* f, void* stream);
/* weight gradient of an image stem straight from the fp32 NCHW image (see cvhip_conv_fuse.x_image): dw = [K][R*S][8] fp32, ACCUMULATED
 * (atomics) like cvhip_conv2d_wgrad with accumulate != 0. CVHIP_ERR_UNSUPPORTED unless cvhip_conv_stem_blocks(d) > 0. */
int cvhip_conv2d_wgrad_image(const cvhip_conv_desc* d, const float* x_nchw, int32_t planes, const void* dy, float* dw, void* stream);
/* (round 5) weight gradient of an image STEM Conv-BN-act layer straight from dz, the gradient at the layer's OUTPUT: the BN + activation
 * backward (what cvhip_bn_act_bwd_apply_acc computes) is applied on load inside the stem weight-gradient kernel, so that pass and the
 * dy tensor do not exist — an image stem has no input gradient, the weight gradient is dy's only consumer
 * (src/models/backbones/det/yolov5_csp_darknet.py:38-45 under trainer.py:189). x: bf16 NHWC 8-channel image, or x_nchw: the fp32 NCHW
 * batch (cvhip_conv_fuse.x_image). (sum du, sum du*xhat) are taken from `acc`; dgamma / dbeta are stored (accumulate != 0: added).
 * dw = [K][R*S][8] fp32, ACCUMULATED. CVHIP_ERR_UNSUPPORTED unless cvhip_conv_stem_blocks(d) > 0. */
int cvhip_conv2d_wgrad_stem_bn(const cvhip_conv_desc* d, const void* x, const float* x_nchw, int32_t planes, const void* dz, const void* y,
                               const float* scale, const float* shift, const float* mean, const float* invstd, const double* acc, int32_t acc_ld,
                               float* dgamma_out, float* dbeta_out, int32_t accumulate, int32_t act, float act_param, float* dw, void* stream);
/* 1 when cvhip_conv2d_fprop_fused accepts a prologue (and, with_z_out != 0, the z_out side output) for this descriptor */
int cvhip_conv2d_fprop_prologue_ok(const cvhip_conv_desc* d, int with_z_out);
/* (round 5) 1 when the STREAMING 1x1 kernel runs this fprop descriptor (with_stats != 0: with training-mode BatchNorm sums) and takes
 * a prologue on its input: the lazy-activation form of `ConvModule.forward` (src/models/bricks/conv_module.py:201-214) in training —
 * the producing layer's BN-apply + activation pass is not run, its consumers read the raw convolution output */
int cvhip_conv1x1_stream_prologue_ok(const cvhip_conv_desc* d, int with_stats);
/* statistics of a training-mode BatchNorm from its fp64 accumulator (what block 0 of cvhip_bn_act_fwd_acc does in its prologue), for
 * a layer whose apply pass is deferred into its consumers: mean | invstd | scale | shift stored, running statistics updated */
int cvhip_bn_finalize_acc(const double* acc, int32_t acc_ld, int32_t C, int64_t count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                          float* shift, void* stream);
/* cvhip_bn_act_fwd_acc whose (post-activation) residual is itself LAZY: z = act(bn(y)) + act(res_scale * residual_raw + res_shift),
 * the residual's activation being the layer's own (DarknetBottleneck shortcut whose input is a lazy CSP branch:
 * src/models/modules/yolo_modules.py:102). ReLU / LeakyReLU / SiLU. */
int cvhip_bn_act_fwd_acc_lazyres(const void* y, int32_t ld_y, void* z, int32_t ld_z, int64_t M, int32_t C, const double* acc, int32_t acc_ld,
                                 int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int32_t act,
                                 float act_param, const void* residual_raw, int32_t ld_res, const float* res_scale, const float* res_shift,
                                 void* stream);
/* cvhip_conv1x1_bwd_fused_acc whose input operand x is LAZY (raw output of the producing layer, `xin` describes the transform the
 * weight gradient needs); K <= 128, no tail */
/* cvhip_conv1x1_bwd_fused_lazy for a layer whose raw output lives in TWO buffers (split store above): y holds channels [0, k_split),
 * y1 (pitch y1_ld) channels [k_split, K); xin may be NULL (no lazy input) */
int cvhip_conv1x1_bwd_fused_split(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                  const void* y, const void* y1, int32_t y1_ld, const void* x, const void* w_dgrad, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, const double* acc, int32_t acc_ld,
                                  float* dgamma_out, float* dbeta_out, int32_t accumulate, int32_t act, float act_param, const void* addend,
                                  int32_t addend_ld, void* dx, int32_t dx_ld, float* dw, const cvhip_lazy_in* xin, void* stream);
int cvhip_conv1x1_bwd_fused_lazy(const cvhip_conv_desc* d, const void* dz0, int32_t dz0_ld, const void* dz1, int32_t dz1_ld, int32_t k_split,
                                 const void* y, const void* x_raw, const void* w_dgrad, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const double* acc, int32_t acc_ld, float* dgamma_out, float* dbeta_out, int32_t accumulate,
                                 int32_t act, float act_param, const void* addend, int32_t addend_ld, void* dx, int32_t dx_ld, float* dw,
                                 const cvhip_lazy_in* xin, void* stream);
/* Plan query of the patch-resident kernel (conv_patch.hip; pure host arithmetic, exercised by the CPU test-suite against a numpy
 * interpreter of the kernel's addressing): per class CVHIP_PATCH_CLASS_INTS int32
 *   {TR, TS, dh0, dh_step, dw0, dw_step, out_oh, out_ow, OHi, OWi, lo_h, lo_w, TH, TW, PH, PW, PWh, PWc, vho, tiles_w, tile_begin,
 *    w_off lo, w_off hi, n_tiles, total_tiles, BN, CK, patch capacity (pixels), per_image (tiles never span images), reserved}.
 * Returns the class count, 0 when another kernel runs this problem. flags bit 0: the plan of cvhip_conv2d_dgrad instead of fprop;
 * bit 1: geometry only (skip the launcher's "too much padding" policy — what the kernel WOULD do on a small problem). */
#define CVHIP_PATCH_CLASS_INTS 30
int cvhip_conv2d_patch_plan(const cvhip_conv_desc* d, int flags, int32_t* out_classes, int max_classes);

/* Plan query of the row-band 3x3 kernel (conv_band.hip; pure host arithmetic): returns 1 when launch_igemm hands this problem to the band
 * kernel under the current policy (CVHIP_BAND / CVHIP_BAND_NF / CVHIP_BAND_PF are read per call), 0 when another kernel runs it.
 * out (may be NULL): {NF (16-channel weight fragments per wave: 2 narrow / 4 wide), WN (waves across the channel tile), MFW (pixel
 * fragments per wave), PPS (patch DMA instructions per wave and K step), PF (pixel fragments read one K step ahead), TH (output rows per
 * band), bands per image, channel tiles, blocks, LDS bytes, patch rows, patch row pitch (pixels), NW (waves per block: 8, or 4 = two
 * co-resident blocks per CU)}. flags bit 0: cvhip_conv2d_dgrad's plan. */
#define CVHIP_BAND_PLAN_INTS 13
int cvhip_conv2d_band_plan(const cvhip_conv_desc* d, int flags, int32_t* out);

/* Plan query of the tap-resident 3x3 weight-gradient kernel (conv_wgrad_band.hip; pure host arithmetic): returns 1 when
 * cvhip_conv2d_wgrad hands this problem to it under the current policy (CVHIP_WGRAD_BAND is read per call), 0 when the general
 * kernel runs. out (may be NULL): {KF (16-channel dY fragments per wave: block tile = 16*KF output channels x 32 input channels x 9
 * taps), tiles, pixel splits, 256-pixel ranges per split, blocks, LDS bytes, patch row pitch (pixels), patch KB pieces per buffer}. */
#define CVHIP_WGRAD_BAND_PLAN_INTS 8
int cvhip_conv2d_wgrad_band_plan(const cvhip_conv_desc* d, int32_t* out);

/* The hardware probes (lane-layout known-answer kernels, machine-ceiling micro-benchmarks) are NOT part of this library: they live in
 * libcvhip_probes.so, declared in include/cvhip_probes.h. */

#ifdef __cplusplus
}
#endif
#endif /* CVHIP_H_ */
