/*
 * mtbt_hip.h -- C ABI of libmtbt_hip.so: the MI355X (gfx950) kernels behind the drop-in
 * `ConvNeXtBiFPNYOLO` module and its post-process.
 *
 * The reference has no FFI of its own: its operator API for this path is the `nn.Module`
 * (`/root/reference/src/main_model.py:300-393`) calling torch.nn operators.  Each entry point below
 * names the torch operator call sites (reference file:line) it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's storage);
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); launches are
 *     asynchronous, nothing here synchronises, allocates, reads environment variables or keeps mutable
 *     global state (the only process-wide effect is the idempotent, per-device, one-time opt-in of a few
 *     kernels to more than 64 KiB of dynamic LDS), so the library is re-entrant (autograd worker
 *     threads) and graph-capturable;
 *   - activations are NHWC ("channels last"): element (n,y,x,c) of a tensor with C channels lives at
 *     base + n*batch_stride + (y*W + x)*pixel_stride + c   (strides in ELEMENTS);
 *     pixel_stride >= C lets a tensor be a channel slice of a wider buffer (concat-free C2f);
 *   - return value: 0 = launched, <0 = MTBT_E* (nothing launched).  No exceptions cross the ABI.
 */
#ifndef MTBT_HIP_H
#define MTBT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTBT_OK 0
#define MTBT_EINVAL (-1)   /* bad argument / unsupported shape */
#define MTBT_EALIGN (-2)   /* pointer or stride not aligned as the kernel requires */
#define MTBT_ELAUNCH (-3)  /* hipLaunchKernel reported an error */
#define MTBT_EWORKSPACE (-4) /* caller workspace too small */

/* storage / arithmetic types */
#define MTBT_F32 0
#define MTBT_BF16 1
#define MTBT_F16 2 /* IEEE binary16 storage, v_mfma_f32_16x16x32_f16, fp32 accumulate, SATURATING stores (+-65504): the inference
                      kernels (conv, depthwise, stem, LayerNorm, fusion, GAP+FC, fused MLP, casts) -- BASELINE configs[4] */

/* fused epilogue activations */
#define MTBT_ACT_NONE 0
#define MTBT_ACT_SILU 1 /* main_model.py:136 ; ultralytics Conv */
#define MTBT_ACT_ELU 2  /* main_model.py:96 */
#define MTBT_ACT_GELU 3 /* timm Mlp act (erf form), main_model.py:21-26 [upstream] */
#define MTBT_ACT_DSILU 5 /* backward epilogues of mtbt_conv2d_nhwc: y = (conv * scale + shift) * act'(res), res = the kept PRE-activation */
#define MTBT_ACT_DELU 6
#define MTBT_ACT_DGELU 7 /* e.g. the ConvNeXt fc2 input gradient lands directly as d(fc1 pre-activation) */
#define MTBT_ACT_GELU_POLY 4 /* the same GELU as x * Phi(x) with Phi an odd degree-13 polynomial on [-4,4]: |error| <= 2.3e-4,
                              below bf16 resolution; no exp / rcp.  For bf16 outputs (the fp32 parity mode uses MTBT_ACT_GELU). */
#define MTBT_ACT_DGELU_POLY 8 /* backward epilogue for MTBT_ACT_GELU_POLY: the EXACT derivative of that polynomial form (no erf / exp) */

/* conv output addressing */
#define MTBT_OUT_NHWC 0
#define MTBT_OUT_CONVT2X2 1 /* ConvTranspose2d(k=2,s=2): GEMM row q*Cout+co -> pixel (2y+q/2, 2x+q%2), channel co */

/* Bumped whenever an argument struct or a signature below changes (round 1 = 1; round 2 added fields / positional arguments without
 * bumping it; round 3 starts at 3).  The library travels prebuilt: a binding built against another header must refuse to load. */
#define MTBT_ABI_VERSION 5
int mtbt_abi_version(void);
/* sizeof() of the argument structs as the LIBRARY was compiled: which = 0 mtbt_conv_args, 1 mtbt_fuse_args, 2 mtbt_decode_args,
 * 3 mtbt_mask_args, 4 mtbt_loss_args, 5 mtbt_prep_desc, 6 mtbt_raw_image, 7 mtbt_upconv_args, 8 mtbt_node_args; -1 for any other value.  A binding compares them with its
 * own layout at load time (a stale prebuilt .so would otherwise read pointers from the wrong offsets). */
int mtbt_sizeof_args(int which);
/* "gfx950" */
const char* mtbt_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on MFMA, fused affine + activation (+ residual) epilogue.
 *   y = act(conv(x, w) * scale[k] + shift[k]) (+ res)
 * Replaces every `nn.Conv2d(groups=1)` / `nn.Linear` / `nn.ConvTranspose2d(2,2)` + folded BatchNorm
 * + activation on the path: ConvBlock (main_model.py:113-141), C2f/Bottleneck (:42-59, :144-173),
 * DepthwiseConvBlock.pointwise (:84-93), BiFPN projections (:263-265), ConvNeXt downsample 2x2/2 and
 * MLP fc1/fc2 (+GELU, +layer-scale residual) (:21-26 [timm]), ultralytics Conv / Conv2d / Proto
 * (:324-328 [ultralytics]).
 * dtype MTBT_BF16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  MTBT_F32: v_mfma_f32_16x16x4_f32
 * (exact fp32 products and sums) -- the parity mode.
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_conv_args {
  const void* x;       /* [N,H,W,C] dtype */
  const void* w;       /* [K][R][S][C] dtype (KRSC, C contiguous) */
  void* y;             /* [N,Ho,Wo,K] out_dtype (or ConvT scatter target) */
  const float* scale;  /* [K] or NULL (=1) */
  const float* shift;  /* [K] or NULL (=0) */
  const void* res;     /* residual, addressed like y, dtype; or NULL */
  int64_t x_batch_stride, y_batch_stride, res_batch_stride; /* elements */
  int32_t x_pixel_stride, y_pixel_stride, res_pixel_stride; /* elements */
  int32_t N, H, W, C;  /* input; C % (64 bytes / sizeof(dtype)) == 0 */
  int32_t K;           /* GEMM rows = output channels (4*Cout for CONVT2X2) */
  int32_t R, S;        /* filter */
  int32_t stride, pad; /* same in y and x */
  int32_t Ho, Wo;      /* conv output size (before the CONVT scatter) */
  int32_t dtype;       /* MTBT_F32 | MTBT_BF16 | MTBT_F16: x, w, res */
  int32_t out_dtype;   /* dtype of y: == dtype, or MTBT_F32 */
  int32_t act;         /* MTBT_ACT_* */
  int32_t out_mode;    /* MTBT_OUT_* */
  int32_t tile_hint;   /* 0 = heuristic; else (TC<<16)|TP to force a tile (tests / tuning); bit 25 = row-reuse direct 3x3
                          kernel, bit 26 = keep a 3x3 on the implicit-GEMM kernel, bit 27 = 64-byte K-steps, bits 28-30 = stages */
  void* y2;            /* optional second output (training forward): the PRE-activation conv * scale + shift, addressed and typed
                          like y (no residual added); NULL = not written.  MTBT_OUT_NHWC only. */
  int32_t policy;      /* 0 = default kernel-selection policy; else 0x100 | bits (bit0 small 1x1 tiles, bit1 64x64 tiles for small k x k,
                          bit2 direct 3x3 kernel, bit3 row-reuse 3x3 everywhere, bit4 never, bit5 64-channel direct tiles): the host's
                          A/B knob, passed per call -- the library reads no environment variables and keeps no mutable global state */
  int32_t debug;       /* ablation bits, honoured by -DMTBT_CONV_ABLATION builds only */
  /* Optional per-channel COLUMN SUMS of the output the call stores (values as rounded to out_dtype), from the conv epilogue itself:
   *   colsum[k] (+)= sum_p (y[p][k] - colsum_shift[k]);   colsum_sq: colsum[K + k] (+)= sum_p (y[p][k] - colsum_shift[k])^2
   * Replaces a separate pass over y for nn.BatchNorm2d's batch statistics (main_model.py:95,126-136: the conv in front of the BatchNorm;
   * shift = the running mean keeps sum / sum of squares well conditioned) and for bias gradients (d fc1.bias = sum_p of the
   * fc2-dgrad * GELU' output, main_model.py:21-26 [timm Mlp]).  MTBT_OUT_NHWC only; colsum NULL = off.  Deterministic (fixed-order
   * partial rows in colsum_ws -- rows x pitch floats as reported by mtbt_conv_colsum_layout, never more than
   * mtbt_conv_colsum_workspace_bytes -- then one wave per channel). */
  float* colsum;             /* [K] or [2K] f32 */
  const float* colsum_shift; /* [K] f32 or NULL (= 0) */
  int32_t colsum_sq;         /* != 0: also the sum of squares */
  int32_t colsum_accumulate; /* != 0: add to colsum instead of overwriting */
  void* colsum_ws;
  int64_t colsum_ws_bytes;
} mtbt_conv_args;

int mtbt_conv2d_nhwc(const mtbt_conv_args* a, void* stream);
int64_t mtbt_conv_colsum_workspace_bytes(int64_t pixels /* N*Ho*Wo */, int K, int with_squares);
/* colsum may be NULL with colsum_ws set: only the partial rows are written (rows x pitch floats, a row = [sums (K) | sums of squares (K)]
 * of one pixel tile's wave row) for a consumer that reduces them itself; this reports the layout a call with the same arguments produces. */
int mtbt_conv_colsum_layout(const mtbt_conv_args* a, int64_t* rows, int32_t* pitch);
/* Which kernel and tile mtbt_conv2d_nhwc would run for these arguments (nothing is launched, no pointer is dereferenced):
 * choice[0] = 0 implicit GEMM / 1 direct 3x3 (LDS-resident halo) / 2 streaming head conv; [1] channel tile; [2] pixel tile (256 = the
 * 16 x 16 halo tile); [3] = 128-byte K-steps (implicit GEMM) or first formulation (direct).  For tests of the tile rules and tools. */
int mtbt_conv_kernel_choice(const mtbt_conv_args* a, int32_t* choice);

/* ---------------------------------------------------------------------------------------------
 * ConvNeXt stem: Conv2d(3,Cout,4,stride 4,bias) on the caller's NCHW fp32 image + LayerNorm2d.
 * Replaces timm `stem_0`/`stem_1` (main_model.py:21-26,34 [timm]).
 *   x [N,3,H,W] f32 NCHW contiguous;  w [Cout][48] f32 (c,ky,kx order = torch layout flattened);
 *   y [N,H/4,W/4,Cout] out_dtype NHWC dense.
 * ------------------------------------------------------------------------------------------- */
int mtbt_stem_conv4x4_ln(const float* x, const float* w, const float* bias, const float* ln_w,
                         const float* ln_b, float ln_eps, void* y, int N, int H, int W, int Cout,
                         int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise k x k convolution (k in {3,7}, stride 1, pad k/2), NHWC dense, two epilogues:
 *   ln_w != NULL : + bias, then LayerNorm over C (ConvNeXt block conv_dw + norm, [timm])
 *   ln_w == NULL : * scale[c] + shift[c], then activation (ultralytics DWConv + BN + SiLU,
 *                  Detect.cv3, main_model.py:324 [ultralytics])
 *   w [k*k][C] (tap-major) in the activation dtype;  C % 8 == 0, C <= 768.
 * ------------------------------------------------------------------------------------------- */
int mtbt_dwconv_nhwc(const void* x, const void* w, const float* bias, const float* ln_w,
                     const float* ln_b, float ln_eps, const float* scale, const float* shift, int act,
                     void* y, int N, int H, int W, int C, int ksize, int dtype, void* stream);

/* LayerNorm over C of an NHWC dense tensor (timm LayerNorm2d in `stages_i.downsample.0`). */
int mtbt_layernorm_nhwc(const void* x, const float* w, const float* b, float eps, void* y,
                        int64_t pixels, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BiFPN fusion node (main_model.py:211-213, :217-219, :228-232, :236-240):
 *   y = sum_i wgt[i] * resample_i(x_i), i < n_in <= 3, accumulated left to right in fp32.
 * resample: 0 identity, 1 bilinear x2 up (align_corners=False), 2 bilinear x0.5 (== 2x2 mean),
 *           3 nearest x2 up, 4 max-pool 2x2  (3,4: src/model.py:58-70).
 * add_weight_bug != 0 reproduces src/model.py:33-36 `sum(w_i + f_i)`.
 * All tensors NHWC dense with C channels; y is [N,H,W,C].
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_fuse_args {
  const void* x[3];
  float wgt[3];
  int32_t resample[3];
  int32_t n_in;
  void* y;
  int32_t N, H, W, C; /* OUTPUT size; input i has H/2,W/2 (up), 2H,2W (down) or H,W */
  int32_t dtype;
  int32_t add_weight_bug;
  const float* wgt_dev; /* optional: n_in weights in DEVICE memory used instead of wgt[] (training: the fusion weights are
                           parameters that change every step, main_model.py:191-196) */
} mtbt_fuse_args;

int mtbt_bifpn_fuse(const mtbt_fuse_args* a, void* stream);

/* The whole BiFPN node in one launch (inference, 16-bit storage): the weighted sum above as the B-operand staging of the
 * DepthwiseConvBlock's 1x1 GEMM (main_model.py:62-102: depthwise k = 1 scale folded into the pointwise weight, BatchNorm folded, ELU):
 *   y[p][k] = act( sum_c w[k][c] * fuse(p)[c] + shift[k] )
 * fuse.y is ignored (the fused map never reaches memory; it is rounded to the storage type exactly as mtbt_bifpn_fuse stores it, so the
 * result equals the two-launch form up to the GEMM's accumulation order).  K == fuse.C in {128, 256}, fuse.dtype MTBT_BF16 | MTBT_F16,
 * add_weight_bug must be 0; the inputs' modes must be (identity, bilinear x2) or (identity, identity, 2x2 mean) -- the two node shapes of
 * BiFPNUnit.forward; anything else returns MTBT_EINVAL (use mtbt_bifpn_fuse + mtbt_conv2d_nhwc).  y [N,H,W,K] NHWC with pixel stride y_pixel_stride (batch stride H*W*y_pixel_stride). */
typedef struct mtbt_node_args {
  mtbt_fuse_args fuse;
  const void* w;       /* [K][C] dtype */
  const float* shift;  /* [K] */
  void* y;
  int32_t y_pixel_stride;
  int32_t K;
  int32_t act;         /* MTBT_ACT_NONE .. MTBT_ACT_GELU_POLY */
} mtbt_node_args;
int mtbt_bifpn_node_nhwc(const mtbt_node_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d forward with BATCH statistics + activation (module in train mode): the reference flips the
 * Detect/Segment heads to train mode inside forward(mode="train") (main_model.py:358-359), so their
 * BatchNorms use batch statistics and update running_mean/var (momentum, unbiased variance) -- SURVEY F14.
 *   y = act((x - mean_batch) / sqrt(var_batch + eps) * gamma + beta);  x, y dense NHWC [pixels][C] (y may alias x).
 * Deterministic two-pass statistics.  workspace >= mtbt_bn_train_workspace_bytes(pixels, C); on return its
 * last 2*C floats hold (mean, biased var).  running_mean / running_var may be NULL.
 * ------------------------------------------------------------------------------------------- */
int64_t mtbt_bn_train_workspace_bytes(int64_t pixels, int C);
int mtbt_bn_train_nhwc(const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Global average pool over H*W then Linear(C, nout) (main_model.py:333-334, :364). y [N,nout] f32. */
int mtbt_gap_fc(const void* x, const float* w, const float* b, float* y, int N, int HW, int C,
                int nout, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Box decode (running_main_v3.py:264-290, :510-533 ; ultralytics Detect._inference/DFL/dist2bbox).
 * Per level l: map [N,h_l,w_l,no] f32 NHWC with pixel stride `map_pixel_stride[l]`, no = 4*reg_max+nc.
 *   ltrb = softmax(16 bins) . arange(16);  anchor = (x+.5, y+.5)
 *   xyxy : boxes = (anchor -/+ ltrb) * stride[l]          (trainer decode)
 *   xywh : boxes = ((x1y1+x2y2)/2, x2y2-x1y1) * stride[l]  (Detect eval; stride may be 0)
 * Outputs (any may be NULL): boxes [N,A,4], scores = sigmoid(cls) [N,A,nc], best score [N,A],
 * best label [N,A] (first max, as torch.max), and `preds_cat` [N, A, cat_stride] row-major with
 * (box4, sigmoid cls) written at columns 0..4+nc-1 (the caller views it as [N,4+nc(+nm),A]).
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_decode_args {
  const float* map[3];
  int32_t h[3], w[3];
  int32_t map_pixel_stride[3];
  float stride[3];
  int32_t n_levels, N, nc, reg_max;
  int32_t xywh; /* 0: xyxy, 1: xywh */
  float* boxes;
  float* scores;
  float* best_score;
  int32_t* best_label;
  float* preds_cat;
  int32_t cat_stride;
} mtbt_decode_args;

int mtbt_decode_boxes(const mtbt_decode_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-image confidence filter + clamp + greedy NMS + top-k (running_main_v3.py:535-552 calling
 * torchvision.ops.nms [upstream]).  One workgroup per image.  For image n:
 *   cand = { a : best_score[n,a] > conf_th } in ascending a   (the reference's boolean-mask order)
 *   boxes clamped to [0, clamp_max]; order = stable descending sort of scores;
 *   greedy: keep i; suppress j if inter/(area_i+area_j-inter) > iou_th  (fp32, no FMA contraction);
 *   stop after top_k kept.
 * Outputs per image (row stride top_k): keep_idx = index into `cand` (== torchvision's return value),
 * keep_anchor = a, out_boxes [top_k,4] (clamped), out_scores, out_labels; counts[n] = #kept;
 * n_cand[n] = |cand|.  workspace: >= mtbt_nms_workspace_bytes(N, A) bytes.
 * ------------------------------------------------------------------------------------------- */
int64_t mtbt_nms_workspace_bytes(int N, int A);
int mtbt_nms_batched(const float* boxes, const float* best_score, const int32_t* best_label, int N, int A,
                     float conf_th, float iou_th, float clamp_max, int top_k, int64_t* keep_idx,
                     int32_t* keep_anchor, float* out_boxes, float* out_scores, int64_t* out_labels,
                     int32_t* counts, int32_t* n_cand, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Prototype x coefficient mask assembly (test_model.py:80-85 intended form) and the trainer's
 * proto projector (running_main_v3.py:186, :251-257; evaluate_model.py:160-171), one kernel:
 *   low[n,k,y,x] = sum_c coeff[n,k,c] * protos[n,y,x,c] (+ bias)
 *   up = bilinear resize of low to (Hout,Wout), align_corners=False
 *   logits (f32, may be NULL) = up ;  masks (u8, may be NULL) = sigmoid(up) > 0.5
 * protos [N,hp,wp,nm] f32 NHWC dense.  coeff element (n,k,c) at coeff + n*coeff_batch_stride +
 * k*coeff_k_stride + c*coeff_c_stride (so mc[N,A,nm] rows gathered through `gather_idx[n,k]`
 * (anchor index, may be NULL => k itself) need no copy; projector: batch stride 0, K=1).
 * Only k < counts[n] (counts may be NULL => K) is computed; padded slots k >= counts[n] are written as zeros.
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_mask_args {
  const float* protos;
  const float* coeff;
  int64_t coeff_batch_stride, coeff_k_stride, coeff_c_stride;
  const int32_t* gather_idx; /* [N,K] or NULL */
  const int32_t* counts;     /* [N] or NULL */
  float bias;
  int32_t N, K, nm, hp, wp, Hout, Wout;
  float* logits;  /* [N,K,Hout,Wout] or NULL */
  uint8_t* masks; /* [N,K,Hout,Wout] or NULL */
} mtbt_mask_args;

int mtbt_mask_assemble(const mtbt_mask_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multitask loss VALUE (forward only), MultiTaskLitModel._multitask_loss, running_main_v3.py:232-387:
 *   per (image, anchor): trainer decode (:268-290), IoU against the image's GT boxes (:316), positives = max IoU >
 *   iou_thresh (:319-321), sum(1 - IoU) (:331), BCE-with-logits(sum) of the class logits against one-hot /
 *   label-smoothed targets (:334-346), two-bin DFL cross-entropy (:351-367); segmentation BCE-with-logits (mean over
 *   seg_n logits, :257); image-classification cross-entropy (:237); normalisation by the batch's positive count (batch
 *   size if none, :371) and the weighted total (:377-383).
 * map[l]: raw Detect maps [N,h_l,w_l,4*reg_max+nc] f32 NHWC (pixel stride map_pixel_stride[l]).  GT boxes grouped by image:
 * gt_xyxy [G][4] pixels, gt_cls [G], gt_off [N+1] (image n owns [gt_off[n], gt_off[n+1])).  seg_logits / seg_targets:
 * seg_n floats each (may be NULL with seg_n = 0); *seg_bias (device scalar, optional) is added to every seg logit.  img_logits [N][n_img_classes] f32, img_gt [N] int64.
 * out[8] = total, seg, box, dfl, cls_det, img_cls, #positives, mean matched IoU.  Deterministic; no host synchronisation.
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_loss_args {
  const float* map[3];
  int32_t h[3], w[3];
  int32_t map_pixel_stride[3];
  int32_t n_levels, N, nc, reg_max;
  float img_size;
  const float* gt_xyxy;
  const int32_t* gt_cls;
  const int32_t* gt_off;
  float iou_thresh, label_smoothing;
  int32_t training; /* label smoothing applies in training mode only (:337) */
  const float* seg_logits;
  const float* seg_targets;
  const float* seg_bias; /* optional device scalar added to every seg logit (the projector's bias), may be NULL */
  int64_t seg_n;
  const float* img_logits;
  const int64_t* img_gt;
  int32_t n_img_classes;
  float w_seg, w_box, w_dfl, w_cls, w_img;
  float* workspace;
  int64_t workspace_bytes; /* >= mtbt_loss_workspace_bytes(N, A, seg_n) */
  float* out;
} mtbt_loss_args;

int64_t mtbt_loss_workspace_bytes(int N, int A, int64_t seg_n);
int mtbt_multitask_loss(const mtbt_loss_args* a, void* stream);

/* Gradient of out[0] (the weighted total) with respect to the head outputs the loss reads -- what `total_loss.backward()` hands
 * to the heads in `training_step` (running_main_v3.py:421-447), the first operator of the backward pass.  Call after
 * mtbt_multitask_loss with the SAME args on the same stream (the batch's positive count is read from a->out[6]).
 * d_map[l]: [N,h_l,w_l,4*reg_max+nc] f32 NHWC, pixel stride d_map_pixel_stride[l]; every element is written (zeros for anchors
 * that are not positives).  d_seg_logits: seg_n floats (required when seg_n > 0) = w_seg / seg_n * (sigmoid - target);
 * d_img_logits [N][n_img_classes] (may be NULL) = w_img / N * (softmax - onehot).  The positives mask, the matched GT index and
 * the DFL targets carry no gradient, exactly as in autograd. */
int mtbt_multitask_loss_grad(const mtbt_loss_args* a, float* const* d_map, const int32_t* d_map_pixel_stride, float* d_seg_logits,
                             float* d_img_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ConvTranspose2d(C, Cm, 2, stride 2, bias) -> Conv 3x3 (Cm -> K, pad 1) + per-channel shift + activation as ONE direct convolution
 * over the LOW-resolution map (inference).  Replaces ultralytics `Proto.upsample` followed by `Proto.cv2` (conv + folded BatchNorm +
 * SiLU): main_model.py:326-328 [ultralytics Proto], SURVEY 8a row 10.  Both operators are linear with nothing in between: per output
 * parity q = 2 * (Y & 1) + (X & 1) the pair is a 2 x 2-tap convolution of the source map with composed weights (4/10 of the MACs; the
 * upsampled tensor is never written).
 *   x [N,H,W,C] dtype NHWC (H % 16 == 0, W % 16 == 0);  y [N,2H,2W,K] dtype (K % 128 == 0)
 *   w [4][K][2][2][C] dtype:  w[q][k][rho][sigma][ci] multiplies source pixel (i + a - 1 + rho, j + b - 1 + sigma), a = q >> 1, b = q & 1,
 *       for output pixel (2i + a, 2j + b) (zero outside the source map)
 *   shift [9][K] f32: shift[rc * 3 + cc][k], rc / cc = border class of the OUTPUT row / column: 0 = first, 2 = last, 1 = interior
 *       (the transposed conv's bias enters through the taps that lie inside the upsampled map only: the 3x3 conv pads with zeros)
 *   y = act(conv2x2_q(x) + shift[class])           act: MTBT_ACT_NONE | MTBT_ACT_SILU
 * ------------------------------------------------------------------------------------------- */
typedef struct mtbt_upconv_args {
  const void* x;
  const void* w;
  void* y;
  const float* shift;
  int64_t x_batch_stride, y_batch_stride; /* elements */
  int32_t x_pixel_stride, y_pixel_stride; /* elements */
  int32_t N, H, W, C, K;
  int32_t dtype; /* MTBT_F32 | MTBT_BF16 | MTBT_F16: x, w, y */
  int32_t act;
} mtbt_upconv_args;
int mtbt_convt2x2_conv3x3_nhwc(const mtbt_upconv_args* a, void* stream);

/* Fused ConvNeXt MLP (timm Mlp fc1 -> GELU -> fc2 with the layer-scale folded, + residual) for d in {96, 192}, bf16 only:
 *   y[p][:] = res[p][:] + W2' . GELU(W1 . t[p][:] + b1) + b2'      (the 4d-wide hidden tensor never leaves the chip)
 * t, res, y: dense [M][d] bf16; w1 [4d][d] bf16; b1 [4d] f32; b2 [d] f32; w2p [d][4d] bf16 whose columns are reordered
 * inside every group of 32 hidden units: slot 8g+j holds hidden 4g+j (j < 4) or 16+4g+(j-4) (j >= 4), g = 0..3. */
int mtbt_convnext_mlp_fused(const void* t, const void* res, const void* w1, const float* b1, const void* w2p,
                            const float* b2, void* y, int64_t M, int D, void* stream);
/* the same with the 16-bit storage type given: MTBT_BF16 or MTBT_F16 */
int mtbt_convnext_mlp_fused_dt(const void* t, const void* res, const void* w1, const float* b1, const void* w2p,
                               const float* b2, void* y, int64_t M, int D, int dtype, void* stream);
/* Training forward of the same block (bf16, d in {96, 192}): y as above PLUS hpre [M][4d] bf16 = W1 . t + b1, the fc1 PRE-activation in
 * natural hidden order -- all the backward needs (GELU' in the fc2 input gradient, GELU re-applied by mtbt_conv_wgrad_xact); the activated
 * hidden tensor is never written.  Weight order of THIS entry: w1_perm / b1_perm have their rows permuted inside every group of 32 hidden
 * units -- row 16 b + 4 g + e holds hidden unit 8 g + 4 b + e (b = 0..1, g = 0..3, e = 0..3) -- and w2 [d][4d] keeps the natural column
 * order (layer scale folded into its rows, b2 = gamma * fc2.bias).  Replaces timm Mlp.forward under model.train(), main_model.py:21-26. */
int mtbt_convnext_mlp_fused_train(const void* t, const void* res, const void* w1_perm, const float* b1_perm, const void* w2, const float* b2,
                                  void* y, void* hpre, int64_t M, int D, void* stream);

/* Pairwise IoU of xyxy boxes (running_main_v3.py:71-97, `batch_bbox_iou`): out[i][j] = inter / (area1_i + area2_j - inter + eps),
 * inter = clamp(min(x2)-max(x1), 0) * clamp(min(y2)-max(y1), 0); fp32, the reference's operation order without FMA
 * contraction (bit-exact with the torch CPU result).  boxes1 [n,4], boxes2 [m,4], out [n,m] row-major; n or m == 0 is a
 * no-op (the reference returns an empty/zero matrix). */
int mtbt_bbox_iou_pairwise(const float* boxes1, int n, const float* boxes2, int m, float eps, float* out, void* stream);

/* Input pipeline of one batch (dataset_btxrdv2.py:109-166: `_letterbox`, BGR->RGB, /255, HWC->CHW, mask binarise), SURVEY §8f N2.
 * images: HOST array of `count` descriptors of DEVICE buffers (decoded 8-bit BGR image as cv2.imread returns it, optional
 * 8-bit grayscale mask of the same size).  Per image: scale = S / max(H0, W0), new = max(1, int(dim * scale)),
 * image resized like cv2.resize(INTER_LINEAR) (OpenCV's 8-bit fixed-point path), mask like INTER_NEAREST, both placed
 * top-left; image padded with 114, mask with 0.  out_images [count][3][S][S] f32 RGB in [0,1]; out_masks
 * [count][1][S][S] f32 in {0,1} (NULL = skip; a NULL descriptor mask gives zeros); out_scales HOST [count] (NULL = skip):
 * the letterbox scale the caller applies to its YOLO-txt boxes (:200-203).  S % 4 == 0. */
typedef struct {
  const uint8_t* bgr;
  const uint8_t* mask;
  int32_t height, width;
  int64_t row_stride;       /* bytes between image rows (>= 3 * width) */
  int64_t mask_row_stride;  /* bytes between mask rows (>= width) */
} mtbt_raw_image;
int mtbt_letterbox_batch(const mtbt_raw_image* images, int count, int img_size, float* out_images, float* out_masks,
                         double* out_scales, void* stream);

/* Segmentation metric accumulators (running_main_v3.py:466-498 feeding the torchmetrics objects of :198-203), SURVEY §8f N3.
 * logits, gt: [B][n_per_image] f32 (n % 4 == 0); prediction = sigmoid(logit) > 0.5, target = int(gt) >= 1.
 * counts [B][4] int64 = TP, FP, FN, TN per image; prob_sum [B] = sum of sigmoid(logit) over predicted-foreground pixels
 * (numerator of the per-image mask score, :483).  Asynchronous, deterministic. */
int64_t mtbt_seg_confusion_workspace_bytes(int B);
int mtbt_seg_confusion(const float* logits, const float* gt, int B, int64_t n_per_image, int64_t* counts, float* prob_sum,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradient of a k x k convolution (any stride / padding; 1x1 and the 2x2 stride-2 downsample included), bf16 (MFMA) or fp32 operands:
 *   dw[k][r][s][c] (fp32, packed [K][R*S*C] like the forward weight) (+)= sum_p dy[p][k] * x[n][y*stride + r - pad][x*stride + s - pad][c]
 * x [N,H,W,C], dy [N,Ho,Wo,K] (Ho = (H + 2 pad - R) / stride + 1) NHWC with pixel / batch strides in elements (multiples of 8; C % 8 == K % 8 == 0;
 * fp32 operands -- the parity mode, a VALU kernel -- multiples of 4).  accumulate != 0
 * adds to dw (gradient accumulation into a flat bucket).  Deterministic: per-slice fp32 partials in `workspace`
 * (>= mtbt_conv_wgrad_workspace_bytes) summed in a fixed order.  This is what autograd computes for `Conv2d.weight.grad`. */
int64_t mtbt_conv_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S);
int mtbt_conv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int pad, int stride,
                    int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                    int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with an activation applied to x while it is staged (x_act = MTBT_ACT_GELU_POLY; bf16, 1 x 1): x is a kept PRE-activation,
 * dw[k][c] (+)= sum_p dy[p][k] * gelu(x[p][c]) -- the fc2 weight gradient behind mtbt_convnext_mlp_fused_train. */
int mtbt_conv_wgrad_xact(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int pad, int stride,
                         int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride, int dtype,
                         int x_act, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* The same plus the bias gradient dbias[k] (+)= sum_p dy[p][k] (`Conv2d.bias.grad`, `Linear.bias.grad`, and the sum_p dy the layer-scale
 * gradient needs) from the dY fragments the kernel holds anyway: the workgroups of the first input-channel tile and tap multiply them
 * with a fragment of ones -- no separate pass over dy. */
int mtbt_conv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int K, int R, int S, int pad,
                         int stride, int64_t x_batch_stride, int32_t x_pixel_stride, int64_t dy_batch_stride, int32_t dy_pixel_stride,
                         int dtype, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Pointwise pieces of the backward pass.  mtbt_act_backward: dz[i] = dy[i] * act'(z[i]) for MTBT_ACT_* (z = the PRE-activation the
 * training forward keeps; n % 8 == 0; dtype f32 or bf16 for all three arrays).  mtbt_channel_sum: out[c] (+)= sum_p x[p][c] (* x2[p][c]
 * when x2 != NULL) over `pixels` rows of pixel_stride elements (C % 8 == 0): the gradient of a conv bias / BatchNorm shift, and
 * with x2 of a BatchNorm scale / layer-scale; deterministic. */
int mtbt_act_backward(const void* dy, const void* z, void* dz, int64_t n, int act, int dtype, void* stream);
int64_t mtbt_channel_sum_workspace_bytes(int64_t pixels, int C);
int mtbt_channel_sum(const void* x, const void* x2, int64_t pixels, int C, int32_t pixel_stride, int32_t pixel_stride2, int dtype,
                     float* out, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* out[p][c] = a[c] * x1[p][c] + b[c] * x2[p][c] + d[c] over dense [pixels][C] tensors (C % 8 == 0; f32 or bf16): the elementwise pass
 * of a batch-statistic BatchNorm backward, dx = (gamma/sigma) (dy - mean(dy) - xhat mean(dy xhat)), written on (dy, pre-activation). */
int mtbt_channel_affine2(const void* x1, const void* x2, const float* a, const float* b, const float* d, void* out, int64_t pixels, int C,
                         int dtype, void* stream);

/* LayerNorm backward over the channels of every pixel (timm ConvNeXt block `norm`, downsample LayerNorm2d): x, dy, dx (and the
 * optional xhat output) dense [pixels][C], f32 or bf16; w = gamma [C] f32.  dx = rstd (g - mean_c g - xhat mean_c(g xhat)), g = dy gamma.
 * d gamma = mtbt_channel_sum(dy, xhat), d beta = mtbt_channel_sum(dy). */
int mtbt_layernorm_backward_nhwc(const void* x, const void* dy, const float* w, float eps, void* dx, void* xhat, int64_t pixels, int C,
                                 int dtype, int accumulate /* != 0: dx += (dx already holds another consumer's gradient) */, void* stream);

/* The same in ONE pass together with the parameter gradients (the training plan's form): dx (+)= ...; dgamma (+)= sum_p dy * xhat;
 * dbeta (+)= sum_p dy.  Per-wave partial rows in `workspace` (>= mtbt_layernorm_backward_params_workspace_bytes), summed in a fixed
 * order.  No xhat tensor, no separate channel-sum passes over dy. */
int64_t mtbt_layernorm_backward_params_workspace_bytes(int64_t pixels, int C);
int mtbt_layernorm_backward_params_nhwc(const void* x, const void* dy, const float* w, float eps, void* dx, int64_t pixels, int C, int dtype,
                                        int accumulate_dx, float* dgamma, float* dbeta, int accumulate_params, void* workspace,
                                        int64_t workspace_bytes, void* stream);

/* Weight gradient of a depthwise k x k convolution (stride 1, pad k/2; k = 3 or 7): dw[tap][c] (fp32, the forward tap layout [k*k][C])
 * (+)= sum_p dy[p][c] * x[p shifted by the tap][c]; x, dy dense [N,H,W,C], f32 or bf16.  Deterministic.  (Persistent workgroups over
 * 8 x 8-pixel tiles: input halo and dy tile staged in LDS, the k*k tap accumulators in registers across tiles.) */
int64_t mtbt_dwconv_wgrad_workspace_bytes(int N, int H, int W, int C, int ksize);
int mtbt_dwconv_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int ksize, int dtype, int accumulate, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* the same plus dbias[c] (+)= sum_p dy[p][c] from the same launch */
int mtbt_dwconv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int ksize, int dtype, int accumulate,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* Fused AdamW step over a flat fp32 bucket: torch.optim.AdamW as the reference trainer configures it
 * (running_main_v3.py:732-734: lr, weight_decay 0.0005, default betas / eps), torch's single-tensor operation order, in place.
 * step >= 1 is the number of the step being taken (bias corrections use beta^step).  n need not be a multiple of 4.
 * grad_scale: optional DEVICE scalar multiplied into every gradient first (the clip coefficient, see mtbt_clip_coef). */
int mtbt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale, void* stream);

/* torch.optim.SGD over a flat fp32 bucket (BASELINE configs[2] names SGD; momentum / dampening / weight decay / Nesterov as torch's
 * single-tensor form; momentum_buf may be NULL when momentum == 0; step 1 initialises the buffer with the gradient). */
int mtbt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                  float weight_decay, int nesterov, int64_t step, const float* grad_scale, void* stream);

/* Gradient clipping by global norm (Trainer(gradient_clip_val=10), running_main_v3.py:826 -> torch.nn.utils.clip_grad_norm_):
 * mtbt_sumsq adds the sum of squares of one flat bucket to *out (deterministic two-level reduction; workspace >=
 * mtbt_sumsq_workspace_bytes()); after an all-reduce-free sum over the buckets, mtbt_clip_coef writes
 * coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)) -- the device scalar the optimiser kernels take as `grad_scale` (NULL = 1),
 * so clipping costs no pass over the gradients and no host synchronisation. */
int64_t mtbt_sumsq_workspace_bytes(void);
int mtbt_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
int mtbt_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream);

/* =============================================================================================================================
 * Training step (BASELINE configs[2]-[3]; running_main_v3.py:393-445 training_step -> total_loss.backward() -> clip -> optimizer).
 * The entry points below, with mtbt_conv2d_nhwc (dgrad = the forward kernel on dY with re-laid-out weights; y2 / MTBT_ACT_D*),
 * mtbt_conv_wgrad, mtbt_dwconv_wgrad, mtbt_layernorm_backward_nhwc and mtbt_multitask_loss(_grad), are what the backward plan
 * of `ConvNeXtBiFPNYOLO.forward(x, "train")` launches (multitask_bonetumor_yolo_amd/train.py).
 * ============================================================================================================================= */

/* Training variants of the stem and the depthwise kernel: they also write what the backward pass needs.
 * stem: `raw` [N,H/4,W/4,Cout] (out_dtype) = the LayerNorm2d INPUT (conv + bias).
 * dwconv, LayerNorm form: `raw` = conv + bias (the LayerNorm input); scale / shift form: `res` is added after the activation and
 * may alias y -- the depthwise input gradient accumulating into a buffer that already holds the residual branch's gradient. */
int mtbt_stem_conv4x4_ln_train(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float ln_eps,
                               void* y, void* raw, int N, int H, int W, int Cout, int out_dtype, void* stream);
int mtbt_dwconv_nhwc_train(const void* x, const void* w, const float* bias, const float* ln_w, const float* ln_b, float ln_eps,
                           const float* scale, const float* shift, int act, void* y, void* raw, const void* res, int N, int H, int W,
                           int C, int ksize, int dtype, void* stream);

/* BatchNorm2d forward, general form (every ConvBlock / DepthwiseConvBlock / ultralytics Conv BatchNorm under model.train(),
 * main_model.py:95,126-136): x dense [pixels][C]; y rows of y_pixel_stride elements (a channel slice of a C2f concat buffer,
 * main_model.py:144-173, or dense).  use_running != 0: module in eval mode (running statistics, nothing updated).  stats [2*C]
 * receives the (mean, biased variance) used -- the input of mtbt_bn_backward_nhwc.  workspace >= mtbt_bn_train_workspace_bytes. */
int mtbt_bn_forward_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype, int use_running,
                         float* stats, void* workspace, int64_t workspace_bytes, void* stream);

/* The same with the batch statistics taken from the column sums the producing conv accumulated in its epilogue (mtbt_conv_args.colsum
 * with colsum_sq): sums [2C] = sum (x - shift), sum (x - shift)^2.  shift [C] or NULL; it may alias running_mean.  One pass over x
 * instead of two. */
int mtbt_bn_forward_sums_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype, const float* sums,
                              const float* shift, float* stats, void* stream);

/* ... and straight from the conv's partial rows (second level and statistics in one launch).  The partial rows are CONSUMED: a tall
 * matrix (> 768 rows) is folded in place before the per-channel second level. */
int mtbt_bn_forward_partials_nhwc(const void* x, void* y, int32_t y_pixel_stride, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int act, int64_t pixels, int C, int dtype, float* partial,
                                  int64_t rows, int32_t pitch, const float* shift, float* stats, void* stream);

/* Backward of activation + BatchNorm2d in one operator: dy = gradient of the ACTIVATED output (rows of dy_pixel_stride elements),
 * x = the conv output the forward normalised (dense), stats as written by mtbt_bn_forward_nhwc.
 *   du = dy * act'(xhat * gamma + beta);  d beta (+)= sum du;  d gamma (+)= sum du * xhat;
 *   dx = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat))     (use_running: dx = gamma * rstd * du)
 * Two passes over (dy, x), deterministic.  dgamma / dbeta may be NULL. */
int64_t mtbt_bn_backward_workspace_bytes(int64_t pixels, int C);
int mtbt_bn_backward_nhwc(const void* dy, int32_t dy_pixel_stride, const void* x, const float* stats, const float* gamma, const float* beta,
                          float eps, int act, int use_running, void* dx, float* dgamma, float* dbeta, int accumulate, int64_t pixels, int C,
                          int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* Per-step weight preparation, ONE launch for the whole network: descriptor j turns a master parameter (fp32, any strides) into the
 * dense row-major [dim0][dim1][dim2][dim3] tensor a kernel reads, in the compute dtype:
 *   dst[a][b][c][d] = src[ia*sstride0 + ib*sstride1 + ic*sstride2 + id*sstride3] * scale0[index of dim scale0_dim] * scale1[...]
 * with ix = flip[x] ? dim[x]-1-x : x.  Forward weights: KRSC; dgrad weights: C,R,S,K with R,S flipped; ConvNeXt fc2 with the layer
 * scale, DepthwiseConvBlock with its k=1 depthwise scale: per-row / per-column scale vectors.  table_dev / block_start_dev live in
 * DEVICE memory; block_start[j] = first workgroup of descriptor j (mtbt_weight_prep_blocks(elements) workgroups each). */
typedef struct mtbt_prep_desc {
  const float* src;
  void* dst;
  const float* scale0;
  const float* scale1;
  int64_t sstride[4];
  int32_t dim[4];
  int32_t flip[4];
  int32_t scale0_dim, scale1_dim;
  int32_t dst_dtype;
  int32_t src_dim3; /* > 0: the source has only src_dim3 entries along dim 3; dst[..][d >= src_dim3] = 0 (channel padding) */
} mtbt_prep_desc;
int mtbt_weight_prep_blocks(int64_t elements);
int mtbt_weight_prep(const mtbt_prep_desc* table_dev, const int32_t* block_start_dev, int n_desc, int total_blocks, void* stream);

/* BiFPN fusion weights on the device (main_model.py:194-196): out[j][i] = ELU(w[i][j]) / (sum_i ELU(w[i][j]) + eps), w [n][2]
 * (n = 2: w1, n = 3: w2); out / dout are TRANSPOSED [2][n] so that the n weights of fusion node j are contiguous (wgt_dev of
 * mtbt_bifpn_fuse).  The backward of that normalisation: dw [n][2] (+)= J^T dout. */
int mtbt_bifpn_norm_weights(const float* w, int n, float eps, float* out, void* stream);
int mtbt_bifpn_norm_weights_backward(const float* w, int n, float eps, const float* dout, float* dw, int accumulate, void* stream);

/* One input of a BiFPN fusion node y = sum_i w_i * resample_i(x_i) (main_model.py:211-240), backward:
 *   dx (+)= wgt * resample^T(dy);   *dwgt (+)= <dy, resample(x_in)>
 * dy [N,H,W,C]; x_in / dx [N,H,W,C] (mode 0), [N,H/2,W/2,C] (mode 1, bilinear x2) or [N,2H,2W,C] (mode 2, 2x2 mean); wgt, dwgt DEVICE
 * scalars; dx or dwgt may be NULL.  Dense NHWC in `dtype`; workspace >= mtbt_bifpn_fuse_backward_workspace_bytes(). */
int64_t mtbt_bifpn_fuse_backward_workspace_bytes(void);
int mtbt_bifpn_fuse_backward(const void* dy, const void* x_in, int mode, const float* wgt, void* dx, int accumulate_dx, float* dwgt,
                             int accumulate_dwgt, int N, int H, int W, int C, int dtype, void* workspace, int64_t workspace_bytes,
                             void* stream);

/* The oldest variant's WeightedAdd node (reference src/model.py:27-37: `w = relu(w); w = w / (w.sum() + eps); sum(w_i + f_i)`; its inputs
 * src/model.py:60-74: identity, F.interpolate(scale_factor=2, mode="nearest"), F.max_pool2d(., 2)), training side:
 *   mtbt_wadd_norm_weights            out[i] = relu(w[i]) / (sum_j relu(w[j]) + eps), n <= 8 (the forward fusion reads them through wgt_dev)
 *   mtbt_wadd_norm_weights_backward   dw[j] (+)= [w[j] > 0] * eps / (s + eps)^2 * sum(dy); dy_colsum [C] = per-channel sums of dy
 *   mtbt_resample_backward            dx (+)= resample^T(dy) for ONE input: mode 0 identity, 3 nearest x2 up (x_in / dx [N,H/2,W/2,C]),
 *                                     4 max pooling 2x2 (x_in / dx [N,2H,2W,C]; x_in = the forward input, required: dy goes to each window's
 *                                     first maximum in row-major order, as torch's max_pool2d backward does).  Dense NHWC, f32 / bf16. */
int mtbt_wadd_norm_weights(const float* w, int n, float eps, float* out, void* stream);
int mtbt_wadd_norm_weights_backward(const float* w, int n, float eps, const float* dy_colsum, int C, float* dw, int accumulate, void* stream);
int mtbt_resample_backward(const void* dy, const void* x_in, int mode, void* dx, int accumulate, int N, int H, int W, int C, int dtype,
                           void* stream);

/* Backward of the trainer's proto projector + bilinear resize (running_main_v3.py:251-255): dseg [N,Hout,Wout] f32 = d loss / d (resized
 * logits); protos [N,hp,wp,nm] f32 (forward output); w [nm] the Conv2d(nm,1,1) weight.  d_protos [N,hp,wp,nm] (dprotos_dtype) (+)=;
 * dw [nm], db [1] fp32 (+)= (dw NULL to skip both). */
int64_t mtbt_projector_backward_workspace_bytes(int N, int hp, int wp, int nm);
int mtbt_projector_backward(const float* dseg, const float* protos, const float* w, void* d_protos, int dprotos_dtype, int accumulate_dprotos,
                            float* dw, float* db, int accumulate_dw, int N, int hp, int wp, int nm, int Hout, int Wout, void* workspace,
                            int64_t workspace_bytes, void* stream);

/* AdaptiveAvgPool2d(1) + Linear backward (main_model.py:333-334, :364): x [N,HW,C]; dlogits [N,nout]; w [nout][C];
 * dx (+)= (W^T dlogits) / HW broadcast over the pixels; dw [nout][C] (+)= dlogits^T pool; db [nout] (+)= sum_n dlogits (may be NULL).
 * pool_ws: [N][C] floats of scratch. */
int mtbt_gap_fc_backward(const void* x, const float* dlogits, const float* w, void* dx, int accumulate_dx, float* dw, float* db,
                         int accumulate_dw, float* pool_ws, int N, int HW, int C, int nout, int dtype, void* stream);

/* dst[n][p][c] = c < C ? src[n][p][c] : 0, c < C_pad, with dtype conversion; element strides, no alignment requirement (the fp32
 * Detect gradient maps are 66 floats wide: their 64- and nc-channel slices become dense, zero-padded operands of dgrad / wgrad). */
int mtbt_copy_strided(const void* src, int src_dtype, int64_t src_batch_stride, int32_t src_pixel_stride, void* dst, int dst_dtype,
                      int64_t dst_batch_stride, int32_t dst_pixel_stride, int N, int64_t pixels, int C, int C_pad, void* stream);

/* Parameter gradients of a weight the forward folded with a vector, from the raw GEMM weight gradient G [K][C] (fp32, dense):
 *   mode 0, rows (ConvNeXt y = x + gamma * (W h + b), timm layer scale): dW (+)= gamma[k] G;  dvec = d gamma[k] (+)= sum_c W G + bias[k] s[k];
 *           dbias[k] (+)= gamma[k] s[k], with s[k] = sum_p dy[p][k]  (bias / s / dbias may be NULL)
 *   mode 1, columns (DepthwiseConvBlock y = W (v * x), main_model.py:84-93): dW (+)= G v[c];  dvec = d v[c] (+)= sum_k G W. */
int mtbt_scale_grad(int mode, const float* G, const float* W, const float* vec, const float* bias, const float* s, float* dW, float* dvec,
                    float* dbias, int K, int C, int accumulate, void* stream);

/* dst += src over [N][pixels][C] views (element strides, multiples of 8): gradient accumulation where the producer cannot. */
int mtbt_add_nhwc(void* dst, int64_t dst_batch_stride, int32_t dst_pixel_stride, const void* src, int64_t src_batch_stride,
                  int32_t src_pixel_stride, int N, int64_t pixels, int C, int dtype, void* stream);

/* Weight gradient of the ConvNeXt stem conv (4x4, stride 4, 3 -> K) on the caller's NCHW fp32 image: dw [K][48] (torch's [K,3,4,4])
 * (+)= sum_p d[p][k] * patch(p); d dense [N*(H/4)*(W/4)][K] in `dtype`. */
int64_t mtbt_stem_wgrad_workspace_bytes(int K);
int mtbt_stem_wgrad(const float* x, const void* d, float* dw, int N, int H, int W, int K, int dtype, int accumulate, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* dtype / layout helpers on the boundary */
int mtbt_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTBT_HIP_H */
