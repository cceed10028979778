/* rfi_hip.h -- C ABI of librfi_hip.so, the MI355X (gfx950) implementation of the
 * rfi_toolbox segmentation hot path.
 *
 * The reference (preshanth/rfi_toolbox v0.2.0) is pure Python and has no FFI
 * layer of its own; its boundary for this path is the torch.nn.Module object
 * protocol plus two plain functions.  Each group below names the reference
 * interface it replaces (paths relative to the reference root).  The Python
 * binding that a maintainer would add is shown in INTEGRATION.md and shipped in
 * rfi_toolbox_amd/_lib.py (ctypes; the header is also cffi-ABI-mode parsable).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message of
 *     the last failure on the calling thread is rfi_last_error().
 *   - plain pointers + sizes only.  `mem` arguments say where a buffer lives:
 *     RFI_HOST (pageable/pinned host memory) or RFI_DEVICE (HBM on the ctx's GPU).
 *     Buffers are caller-owned and never retained past the call.
 *   - images are NHWC float32 (what Preprocessor emits, preprocessor.py:380-404);
 *     the NCHW entry points exist because UNet.forward takes NCHW (unet.py:60).
 *   - a handle is bound to one GPU and one HIP stream and is not thread-safe:
 *     one host thread per handle, one process per GPU.
 */
#ifndef RFI_HIP_H
#define RFI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFI_HIP_ABI_VERSION 1

enum { RFI_HOST = 0, RFI_DEVICE = 1 };

typedef struct rfi_ctx rfi_ctx;
typedef struct rfi_model rfi_model;

/* ---- library / context ------------------------------------------------------------ */
int         rfi_abi_version(void);
const char* rfi_last_error(void);
int rfi_device_count(int* count);
int rfi_ctx_create(int device_id, rfi_ctx** out);
int rfi_ctx_destroy(rfi_ctx* ctx);
/* backward-pass overlap (weight-gradient kernels on a side stream next to the dgrad / batch-norm
 * chain; results are bit-identical either way).  On by default; off = every kernel alone on the
 * main stream, which is what per-kernel timings and rocprofv3 comparisons want. */
int rfi_ctx_set_overlap(rfi_ctx* ctx, int enabled);
int rfi_ctx_synchronize(rfi_ctx* ctx);
/* the hipStream_t every kernel of this ctx is launched on (for event timing / interop) */
int rfi_ctx_stream(rfi_ctx* ctx, void** hip_stream);
int rfi_ctx_device_name(rfi_ctx* ctx, char* buf, size_t buflen);

/* device memory owned by the ctx (freed at rfi_ctx_destroy if still live) */
int rfi_malloc(rfi_ctx* ctx, size_t bytes, void** dptr);
int rfi_free(rfi_ctx* ctx, void* dptr);
int rfi_memcpy(rfi_ctx* ctx, void* dst, int dst_mem, const void* src, int src_mem, size_t bytes);
int rfi_memset(rfi_ctx* ctx, void* dptr, int value, size_t bytes);

/* HIP-event stopwatch on the ctx stream: start; ...launches...; stop -> elapsed ms (syncs) */
int rfi_timer_start(rfi_ctx* ctx);
int rfi_timer_stop(rfi_ctx* ctx, float* elapsed_ms);

/* per-kernel-family HIP-event profile of everything launched while enabled.
 * families: see rfi_profile_family_name(); flops/bytes are ALGORITHMIC counts. */
int rfi_profile_enable(rfi_ctx* ctx, int on);
int rfi_profile_reset(rfi_ctx* ctx);
int rfi_profile_family_count(void);
const char* rfi_profile_family_name(int family);
int rfi_profile_get(rfi_ctx* ctx, int family, int64_t* launches, double* total_ms,
                    double* flops, double* bytes);
/* every profiled launch in stream order as CSV (family, shape label, ms, GFLOP, TFLOP/s, MB, GB/s) */
int rfi_profile_dump(rfi_ctx* ctx, const char* csv_path);

/* ---- model: replaces rfi_toolbox.models.UNet (models/unet.py:41-77; UNetBigger :79-118
 *      is depth=5) as constructed by scripts/train_model.py:111, evaluate_model.py:34 ---- */
int rfi_unet_create(rfi_ctx* ctx, int in_channels, int out_channels, int init_features,
                    int depth, rfi_model** out);
/* the "3-layer CNN segmenter" of BASELINE.json configs[0]/[1] (not a reference
 * class -- nearest text is the elided example README.md:379-398): Conv3x3(in->width,p1)+ReLU ->
 * Conv3x3(width->width,p1)+ReLU -> Conv1x1(width->out) logits.  Entries: encoder.0.weight/bias,
 * encoder.2.weight/bias, decoder.0.weight/bias.  Every rfi_model_* / rfi_train_* call below applies. */
int rfi_cnn3_create(rfi_ctx* ctx, int in_channels, int out_channels, int width, rfi_model** out);
/* "U-Net with a ResNet-18 encoder" of BASELINE.json configs[2] (not a reference class and
 * no torchvision / segmentation_models_pytorch here: builder-defined, oracle/resnet_unet_ref.py).  stem Conv3x3(bias
 * =False)+BN+ReLU at full resolution; layer1..4 of two BasicBlocks each (widths f, 2f, 4f, 8f; stride 2 + 1x1
 * projection entering layers 2-4); then the reference's bottleneck / DecoderBlocks / final_conv (models/unet.py:30-77)
 * with the four stage outputs as skips.  H and W must be multiples of 16; init_features a multiple of 4.
 * Entries: stem.0.weight, stem.1.*, layer{l}.{b}.conv1.weight, .bn1.*, .conv2.weight, .bn2.*, .downsample.0.weight,
 * .downsample.1.*, bottleneck.conv.*, decoder{l}.up.*, decoder{l}.conv.conv.*, final_conv.*.  Compute modes:
 * 2 (float32 by 3 x bf16, default), 0 (native float32 MFMA), 4 (bf16 operands); 1 / 3 run as 4 / 2 (the plane
 * data flow exists for the plain U-Net only). */
int rfi_unet_resnet_create(rfi_ctx* ctx, int in_channels, int out_channels, int init_features, rfi_model** out);
/* The per-RoI mask branch of Mask R-CNN (BASELINE.json configs[3], north_star "per-pixel mask head"; SURVEY.md 8a row A11;
 * not in the reference and no torchvision here: builder-defined as the published head, oracle/mask_head_ref.py):
 * conv_layers x [Conv3x3(C->C, p1)+bias -> ReLU] -> ConvTranspose2d(C->C, k2, s2)+bias -> ReLU -> Conv1x1(C->K).  Input:
 * RoIAlign-ed features [R, h, w, C] (rfi_op_roi_align), output logits [R, 2h, 2w, K]; labels of the training calls are
 * [R, 2h, 2w] uint8 and the loss is the mean BCE-with-logits over them (K = 1; rfi_model_set_loss(m, 1, alpha, gamma)
 * switches to a focal loss).  Entries: mask_fcn{1..L}.weight/bias, conv5_mask.weight/bias, mask_fcn_logits.weight/bias.
 * Every rfi_model_* / rfi_train_* call applies with n = R; rfi_model_input_grad returns the gradient w.r.t. the input
 * features of the last backward pass ([R, h, w, C]) for rfi_op_roi_align_backward. */
int rfi_mask_head_create(rfi_ctx* ctx, int in_channels, int conv_layers, int out_channels, rfi_model** out);
int rfi_model_input_grad(rfi_model* m, float* dx, int dx_mem);
/* The RPN head (Faster R-CNN; SURVEY 8a A11, not in the reference): conv_layers x [Conv3x3(C->C, p1)+bias -> ReLU] -> ONE
 * Conv1x1(C -> 5 A): per pixel A objectness logits followed by A x 4 box deltas (anchor-major), i.e. cls_logits and
 * bbox_pred of the usual implementation stacked.  Entries: conv.{i}.0.weight/bias, head.weight/bias.  forward gives
 * [N, H, W, 5 A]; its loss needs per-anchor targets and lives outside the model: rfi_op_rpn_loss produces
 * d(loss)/d(head output), rfi_model_backward_dlogits (after a forward pass on the same input) turns it into parameter
 * gradients and the input gradient (rfi_model_input_grad), rfi_train_apply steps the optimiser.  Also valid for the
 * mask head. */
int rfi_rpn_head_create(rfi_ctx* ctx, int in_channels, int conv_layers, int anchors_per_pixel, rfi_model** out);
/* The ResNet-50-FPN backbone of the Mask R-CNN path (not in the reference: the published networks with the
 * layer names and the FROZEN BatchNorm of the usual detection backbone, oracle/backbone_ref.py).  base_width 64 and
 * fpn_channels 256 give ResNet-50; both must be multiples of 4, H and W multiples of 64.  Entries: body.conv1.weight,
 * body.bn1.{weight,bias,running_mean,running_var}, body.layer{1..4}.{b}.conv{1,2,3}.weight / .bn{1,2,3}.* /
 * .downsample.{0.weight,1.*}, fpn.inner_blocks.{i}.0.{weight,bias}, fpn.layer_blocks.{i}.0.{weight,bias}; BatchNorm entries
 * are buffers (never updated), every conv weight and FPN bias is a parameter of rfi_train_apply.
 * backbone_forward: feats[0..4] = P2..P6, [n, h >> (2 + i), w >> (2 + i), fpn_channels] each (null entries are skipped);
 * backbone_backward (after a forward pass on the same input): dfeats[i] = d(loss)/d(P_{2+i}) (null = zero) -> parameter
 * gradients. */
/* The box head (not in the reference: two FC layers + ReLU on the flattened RoI features, then class scores and
 * per-class box deltas -- TwoMLPHead + FastRCNNPredictor of the usual implementation, oracle/mask_head_ref.py): input
 * [R, in_features] (n = R, h = w = 1), output [R, num_outputs] with num_outputs = 5 K1 = K1 class logits followed by K1 x 4
 * deltas (cls_score and bbox_pred stacked).  Entries: fc6.weight [hidden, in_features, 1, 1] / .bias, fc7..., head.weight /
 * .bias.  Loss outside the model: rfi_op_fastrcnn_loss (labels int32 in [0, K1), 0 = background; targets [R][4]; mean
 * cross-entropy + smooth L1 (beta) of the ground-truth class's deltas over the foreground RoIs, both / R) ->
 * rfi_model_backward_dlogits -> rfi_model_input_grad / rfi_train_apply. */
int rfi_box_head_create(rfi_ctx* ctx, int in_features, int hidden, int fc_layers, int num_outputs, rfi_model** out);
/* x += y on the device (n % 4 == 0): sums the feature-map gradients of several branches */
int rfi_op_add_inplace(rfi_ctx* ctx, float* x, const float* y, int64_t n);
int rfi_op_fastrcnn_loss(rfi_ctx* ctx, const float* head, int64_t rois, int num_classes, const int32_t* labels, const float* targets,
                         float beta, float* dhead, float* loss_classifier, float* loss_box_reg);
int rfi_resnet50_fpn_create(rfi_ctx* ctx, int in_channels, int base_width, int fpn_channels, rfi_model** out);
int rfi_backbone_forward(rfi_model* m, const float* x, int x_mem, int n, int h, int w, float* const feats[5], int feats_mem);
int rfi_backbone_backward(rfi_model* m, const float* x, int x_mem, int n, int h, int w, const float* const dfeats[5],
                          int dfeats_mem);
int rfi_model_backward_dlogits(rfi_model* m, const float* x, int x_mem, const float* dlogits, int dlogits_mem, int n, int h,
                               int w);
/* A backward pass OVERWRITES the model's gradient buffer.  To sum the gradients of several passes (a head shared by the
 * pyramid levels, several micro-batches): phase 0 zeroes an accumulator, phase 1 adds the current gradients to it (call
 * after each backward pass), phase 2 copies the sum into the gradient buffer, ready for rfi_train_apply. */
int rfi_model_grad_accumulate(rfi_model* m, int phase);
int rfi_model_destroy(rfi_model* m);
/* variants of models/unet.py:120-268 on the same graph: UNetDifferentActivation's activation
 * (0 = ReLU, 0 < s < 1 = LeakyReLU(negative_slope=s), after every BatchNorm) and UNetOverfit's head
 * (forward returns sigmoid(logits), :196; the training step then applies BCE-with-logits + dice to THAT
 * output, as scripts/train_model.py:120,146 does with whatever the model returns). */
int rfi_model_set_activation(rfi_model* m, float negative_slope);
/* arithmetic of the conv / transposed-conv / weight-gradient contractions (tensors in HBM, BatchNorm,
 * loss and the optimiser are float32 in every mode):
 *   2 (default) float32 by splitting: every float32 operand is the exact sum of three bfloat16 pieces and a
 *     product block is six v_mfma_f32_32x32x16_bf16 (float32 accumulate; the three piece products below
 *     2^-24 of the product are dropped).  Error against float64 is that of the native float32 MFMA path
 *     (tests/test_gpu_ops.py runs both against the same tolerances), at 2.7x its matrix-pipe rate.
 *   0 native float32 MFMA (v_mfma_f32_32x32x2_f32, an exact fmaf chain).
 *   1 bfloat16: operands rounded to bfloat16 (RNE), float32 accumulate -- a builder-chosen reduced-precision mode,
 *     NOT the reference's arithmetic (float32 on CPU; float16 autocast + GradScaler on a GPU,
 *     scripts/train_model.py:131,144). */
int rfi_model_set_compute_dtype(rfi_model* m, int dtype);
int rfi_model_set_head_sigmoid(rfi_model* m, int enabled);
/* training loss: 0 = mean BCE-with-logits + dice, the reference's (scripts/train_model.py:120-128,146; default);
 * 1 = sigmoid focal loss, mean over elements (NOT in the reference; builder-defined as
 * Lin et al. 2017 / torchvision.ops.sigmoid_focal_loss: alpha_t (1 - p_t)^gamma BCEwithLogits; alpha < 0
 * disables the alpha weighting). */
int rfi_model_set_loss(rfi_model* m, int kind, float alpha, float gamma);
/* deterministic init with torch's default distributions (kaiming-uniform(a=sqrt5) conv
 * weights/biases, BN gamma=1 beta=0, running stats 0/1) from a 64-bit seed */
int rfi_model_init(rfi_model* m, uint64_t seed);

/* state_dict surface (train_model.py:179, evaluate_model.py:35): entries are in the
 * reference's state_dict order with the reference's key names and shapes; float32
 * except num_batches_tracked (int64).  Values cross the boundary in the reference's
 * layouts (Conv2d OIHW, ConvTranspose2d IOHW); the library converts. */
int rfi_model_entry_count(rfi_model* m, int* n);
int rfi_model_entry_info(rfi_model* m, int index, const char** name, int* ndim,
                         int64_t dims[4], int* is_int64, int* is_parameter);
int rfi_model_load_entry(rfi_model* m, const char* name, const void* host, size_t bytes);
int rfi_model_store_entry(rfi_model* m, const char* name, void* host, size_t bytes);
int rfi_model_param_count(rfi_model* m, int64_t* n_scalars);   /* == sum(p.numel()) */

/* .train() / .eval()  (train_model.py:136,157) */
int rfi_model_set_training(rfi_model* m, int training);

/* model(x) -> logits (N,out,H,W)  (unet.py:60-77; train_model.py:145).  In training mode
 * BatchNorm uses batch statistics and updates the running buffers exactly as the
 * reference does, including the encoder's double EMA update (unet.py:28). */
/* Synchronisation of the model entry points that take tensors with a memory-space argument (forward_nhwc / forward_nchw,
 * backward_dlogits, input_grad, backbone_forward / backward): with a HOST pointer among the arguments the call returns when the
 * data is in place; when every tensor argument is a DEVICE pointer it returns as soon as the work is enqueued on the
 * context's stream (the next call that hands data to the host -- rfi_memcpy, a loss scalar, rfi_ctx_synchronize --
 * waits).  RFI_SYNC_ALWAYS=1 restores a synchronisation at the end of every call. */
int rfi_model_forward_nhwc(rfi_model* m, const float* x, int x_mem, int n, int h, int w,
                           float* logits, int logits_mem);
int rfi_model_forward_nchw(rfi_model* m, const float* x, int x_mem, int n, int h, int w,
                           float* logits, int logits_mem);

/* ---- optimisation step: replaces the body of the loop in scripts/train_model.py:139-154
 *      zero_grad -> forward -> BCEWithLogits(mean)+dice (:120-128,146) -> backward ->
 *      clip_grad_norm_(max_norm) (:149) -> Adam(lr, betas, eps, coupled L2 weight_decay)
 *      (:130,150).  labels are uint8 (N,H,W), non-zero == RFI.  float32 results throughout (the
 *      reference's CPU path; autocast/GradScaler are off there, :131,144); how the contractions
 *      reach them is rfi_model_set_compute_dtype's business. ---- */
typedef struct rfi_hyper {   /* doubles: the reference's hyper-parameters are python floats */
    double lr, beta1, beta2, eps, weight_decay, max_grad_norm;
} rfi_hyper;

/* Data parallel: when the model's context holds a communicator of more than one rank (rfi_comm_init),
 * rfi_train_step and rfi_train_step_async all-reduce(sum) the flat gradient buffer over the ranks before
 * clipping and apply the MEAN gradient (grad_scale = 1/world), so every rank makes the identical update.
 * BatchNorm batch statistics, running buffers and the dice term stay LOCAL to each rank's batch (torch DDP
 * broadcasts rank 0's buffers every step; here they are only taken from rank 0 at checkpoint time). */
int rfi_train_step(rfi_model* m, const float* x_nhwc, int x_mem, const uint8_t* labels,
                   int labels_mem, int n, int h, int w, const rfi_hyper* hp, float* loss_out);
/* same step split in two so a data-parallel caller can all-reduce the gradients in between */
int rfi_train_forward_backward(rfi_model* m, const float* x_nhwc, int x_mem,
                               const uint8_t* labels, int labels_mem, int n, int h, int w,
                               float* loss_out);
int rfi_train_apply(rfi_model* m, const rfi_hyper* hp, float grad_scale, float* grad_norm_out);
/* loss only, no update (validation loop, train_model.py:157-167) */
int rfi_model_loss(rfi_model* m, const float* x_nhwc, int x_mem, const uint8_t* labels,
                   int labels_mem, int n, int h, int w, float* loss_out);
/* non-blocking variant used by bench loops: enqueue one full step, no host sync */
int rfi_train_step_async(rfi_model* m, const float* x_dev, const uint8_t* labels_dev,
                         int n, int h, int w, const rfi_hyper* hp);
int rfi_model_last_loss(rfi_model* m, float* loss_out, float* grad_norm_out);   /* syncs */

/* flat gradient / parameter buffers (device pointers, library layout) + gradient access
 * by reference name in reference layout (what p.grad would hold after backward) */
int rfi_model_grad_buffer(rfi_model* m, float** dptr, int64_t* n_floats);
int rfi_model_param_buffer(rfi_model* m, float** dptr, int64_t* n_floats);
int rfi_model_store_grad(rfi_model* m, const char* name, void* host, size_t bytes);
int rfi_model_store_adam(rfi_model* m, const char* name, void* host_m, void* host_v, size_t bytes,
                         int64_t* step);
/* resume: load Adam moments of one parameter (reference layout) / set the step counter
 * (the optimizer_state_dict the reference saves, train_model.py:180) */
int rfi_model_load_adam(rfi_model* m, const char* name, const void* host_m, const void* host_v,
                        size_t bytes);
int rfi_model_set_adam_step(rfi_model* m, int64_t step);
int rfi_model_algorithmic_flops(rfi_model* m, int n, int h, int w, double* fwd, double* step);
/* read an internal activation / gradient buffer of the last prepared shape (parity debugging):
 * "encY1.<l>" "encY2.<l>" "decY1.<l>" "decY2.<l>" "concat.<l>" "pool.<l>" "bottY1" "bottY2" "logits"
 * "dlogits" "gA.<l>" "gB.<l>" "dconcat.<l>" "dpool.<l>" "gBottA" "gBottB" (raw NHWC fp32), and
 * "chan.<conv index>" = [running_mean|running_var|mean|invstd|scale|shift|c1|c2] x Cout.
 * host == NULL only reports the element count. */
int rfi_model_debug_tensor(rfi_model* m, const char* name, float* host, size_t host_floats,
                           int64_t* n_floats);

/* one batch of scripts/evaluate_model.py:40-51 entirely on device: forward in the CURRENT mode,
 * sigmoid > threshold, confusion counts against the uint8 labels (non-zero == positive) */
int rfi_model_eval_batch(rfi_model* m, const float* x_nhwc, int x_mem, const uint8_t* labels,
                         int labels_mem, int n, int h, int w, float threshold, int64_t* tp,
                         int64_t* fp, int64_t* fn);

/* ---- data-parallel gradient exchange (new; the reference has no multi-GPU path) -------
 * RCCL over xGMI: ncclAllReduce(sum) of the flat gradient buffer on the ctx stream.
 * librccl.so is dlopen()ed on first use.  id_buf: 128 bytes (ncclUniqueId). */
int rfi_comm_unique_id(void* id_buf128);
int rfi_comm_init(rfi_ctx* ctx, const void* id_buf128, int rank, int world_size);
int rfi_comm_destroy(rfi_ctx* ctx);
int rfi_comm_allreduce_sum_f32(rfi_ctx* ctx, float* dptr, int64_t count);
int rfi_model_allreduce_grads(rfi_model* m);   /* all-reduce(sum) of the grad buffer */
/* Inside rfi_train_step / rfi_train_step_async the exchange is BUCKETED and overlapped with the backward pass:
 * contiguous ranges of the flat gradient buffer (head + decoder1, decoder2, ..., bottleneck, encoderD, ..., encoder1
 * -- the order the backward pass finishes them) are all-reduced on a separate HIP stream as soon as their last
 * producer kernel has been enqueued.  rfi_comm_emulate(ctx, W) (tests on ONE GPU; W >= 2, 0 = off) replaces every
 * bucket's all-reduce by "multiply the range by W" and the step applies grad_scale = 1/W: the step then equals the
 * plain step bit for bit iff every element is exchanged exactly once, after its producers and before clip + Adam. */
int rfi_comm_emulate(rfi_ctx* ctx, int world);

/* ---- preprocessing: replaces the per-patch hot loop of Preprocessor.create_dataset
 *      (preprocessing/preprocessor.py:366-384: _extract_channels_from_complex :562-606 /
 *      _from_real :608-644, then _apply_sam2_normalization :765-783) for a stack of
 *      patches already tiled/rotated by the host.  in: (n,ps_h,ps_w) complex128/complex64
 *      (interleaved re,im) or float64/float32 real; out: NHWC float32 (n,ps_h,ps_w,3). ---- */
enum { RFI_C128 = 0, RFI_C64 = 1, RFI_F64 = 2, RFI_F32 = 3 };
int rfi_preprocess_patches(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype,
                           int n, int ps_h, int ps_w, float* out_nhwc, int out_mem);

/* ---- preprocessing, gather form: the augmentation views (preprocessor.py:413-446), the zero-padded
 *      tiling (:478-560, patchify :22-42) and the blank-patch test (:746-756) resolved ON DEVICE from
 *      the waterfall itself.  `planes`: n_planes x C x T (all baselines x polarisations), complex or
 *      real as above; `flags`: uint8, same shape, non-zero == RFI.  A patch is named by a table entry;
 *      the host only shuffles / truncates the table (the reference's global-RNG permutation, :758-763). */
typedef struct rfi_patch_src {
    int32_t plane;        /* (baseline, polarisation) plane index */
    int32_t view;         /* 0 plane, 1 plane[::-1,:], 2 plane.T, 3 plane.T[::-1,:] */
    int32_t row0, col0;   /* tile origin in view coordinates; pixels past the view's edge are zero padding */
} rfi_patch_src;
/* any_out[i] = 1 when patch i holds a flagged pixel (host array of n bytes) */
int rfi_patch_any_flag(rfi_ctx* ctx, const uint8_t* flags, int flags_mem, int n_planes, int c, int t,
                       const rfi_patch_src* table_host, int n, int ps, uint8_t* any_out_host);
/* images (n,ps,ps,3) float32 NHWC and, when flags != NULL, labels (n,ps,ps) uint8 of the table's patches */
int rfi_preprocess_gather(rfi_ctx* ctx, const void* planes, int planes_mem, int dtype, int n_planes,
                          int c, int t, const uint8_t* flags, int flags_mem,
                          const rfi_patch_src* table_host, int n, int ps, float* out_nhwc, int out_mem,
                          uint8_t* out_labels, int labels_mem);

/* ---- preprocessing, order-statistic branches (preprocessor.py:646-745) on device, for a stack of
 *      patches (n, ps_h, ps_w) already cut by the host:
 *      rfi_preprocess_real: REAL float64 or float32 input (float32: every result rounded to float32, i.e. NumPy's
 *      float32 arithmetic on a float32 array): optional median normalise (:646-670), stretch
 *      (0 none, 1 SQRT, 2 LOG10; infinities <- MAD of the patch's finite values, :672-706), optional second
 *      normalise, then the 3-channel extraction of :608-644 + ImageNet normalisation -> out_nhwc; when
 *      flags_out != NULL also the MAD flags of the PROCESSED patches (:708-745, |x - med| > sigma * MAD).
 *      rfi_mad_flags: the same flags for any input dtype (complex: of |z|), nothing else. ---- */
int rfi_preprocess_real(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype, int n, int ps_h,
                        int ps_w, int stretch, int normalize_before, int normalize_after, double flag_sigma,
                        float* out_nhwc, int out_mem, uint8_t* flags_out, int flags_mem);
int rfi_mad_flags(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype, int n, int ps_h, int ps_w,
                  double flag_sigma, uint8_t* flags_out, int flags_mem);

/* ---- synthetic data on device (the step in front of the path): the sample model of
 *      SyntheticDataGenerator._generate_single_sample (data_generation/synthetic_generator.py:520-815)
 *      with a counter-based per-pixel random stream (Philox4x32-10) instead of NumPy's sequential global
 *      one -- distribution-level parity, exact for bandpass / signal / mask given the event table.
 *      Events are drawn by the host (a few dozen per sample); kind 0 fills channels [r0,r1) x times
 *      [c0,c1) with `amp`; kind 1 is a frequency sweep from channel r0 to r1 of width c0 and power-law
 *      order c1 (1 or 2).  Output: planes (n_samples, n_pol, C, T) complex128/complex64 and uint8 flags. */
typedef struct rfi_event {
    int32_t kind, r0, r1, c0, c1;
    int32_t pad_;
    double amp;
} rfi_event;
int rfi_generate_waterfalls(rfi_ctx* ctx, uint64_t seed, int n_samples, int n_pol, int c, int t,
                            double noise_mjy, int bandpass, int bandpass_order, double pol_corr,
                            const rfi_event* events_host, const int32_t* event_offsets_host,
                            int out_dtype, void* planes_out, int planes_mem, uint8_t* flags_out, int flags_mem);

/* ---- metrics: replaces the reductions of evaluation/metrics.py:25-172.  pred/true are
 *      uint8 or float32 arrays of `count` elements, non-zero == positive (:36-37). ---- */
enum { RFI_U8 = 0, RFI_FLOAT32 = 1 };
int rfi_confusion_counts(rfi_ctx* ctx, const void* pred, int pred_dtype, int pred_mem,
                         const void* truth, int truth_dtype, int truth_mem, int64_t count,
                         int64_t* tp, int64_t* fp, int64_t* fn);
/* sigmoid(logit) > threshold on device (evaluate_model.py:44-47), u8 out */
int rfi_threshold_logits(rfi_ctx* ctx, const float* logits_dev, int64_t count, float threshold,
                         uint8_t* mask_dev);

/* ---- kernel-level entry points (device pointers only).  Used by the parity tests to
 *      check each HIP kernel against the oracle in isolation.  impl: 0 auto, 1 direct VALU,
 *      2 MFMA implicit GEMM in native float32 (v_mfma_f32_32x32x2_f32), 3 MFMA implicit GEMM with bfloat16
 *      operands rounded in registers, 4 MFMA implicit GEMM, float32 by 3 x bf16 splitting in registers (the
 *      models' default arithmetic), 5 / 6 the plane kernels (bf16 pieces staged by LDS-DMA from plane tensors;
 *      3x3 convolutions) in the 3 x bf16 / bf16 arithmetic -- the kernels of the bfloat16 compute mode. ---- */
int rfi_op_conv3x3(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin,
                   const float* w_oihw, const float* bias, int cout,
                   const float* in_scale, const float* in_shift, int in_relu, float* y);
/* 1x1 stride-1 conv (a GEMM over the pixels): the Bottleneck / pyramid / fully connected layers of the detector (A11). */
int rfi_op_conv1x1(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin, const float* w_oihw, const float* bias,
                   int cout, const float* in_scale, const float* in_shift, int in_relu, float* y);
int rfi_op_conv3x3_dgrad(rfi_ctx* ctx, int impl, const float* dy, int n, int h, int w, int cout,
                         const float* w_oihw, int cin, float* dx);
int rfi_op_conv3x3_wgrad(rfi_ctx* ctx, int impl, const float* x, const float* dy, int n, int h,
                         int w, int cin, int cout, const float* in_scale, const float* in_shift,
                         int in_relu, float* dw_oihw);
/* stride-2 convolutions of the ResNet-style encoder (ksize 3: Conv2d(k3, s2, p1, bias=False) run as a 2x2 stride-1
 * convolution on the space-to-depth input; ksize 1: Conv2d(k1, s2, bias=False), the projection shortcut).  h, w =
 * INPUT size (even), cin % 4 == 0; y / dy are n x h/2 x w/2 x cout, dx is n x h x w x cin. */
int rfi_op_conv_s2(rfi_ctx* ctx, int impl, int ksize, const float* x, int n, int h, int w, int cin,
                   const float* w_oihw, int cout, float* y);
int rfi_op_conv_s2_dgrad(rfi_ctx* ctx, int impl, int ksize, const float* dy, int n, int h, int w, int cout,
                         const float* w_oihw, int cin, float* dx);
int rfi_op_conv_s2_wgrad(rfi_ctx* ctx, int impl, int ksize, const float* x, const float* dy, int n, int h, int w,
                         int cin, int cout, float* dw_oihw);
int rfi_op_convt2x2(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin,
                    const float* w_iohw, const float* bias, int cout, float* y);
int rfi_op_convt2x2_dgrad(rfi_ctx* ctx, int impl, const float* dy, int n, int h, int w, int cout,
                          const float* w_iohw, int cin, float* dx);
int rfi_op_convt2x2_wgrad(rfi_ctx* ctx, int impl, const float* x, const float* dy, int n, int h,
                          int w, int cin, int cout, float* dw_iohw);
/* Building blocks of the Mask R-CNN path of BASELINE.json configs[3].  NOT in the reference
 * (no detector code exists there, docs/API.md:180 and README.md:90 only name one); defined by the published
 * algorithms: RoIAlign (He et al. 2017; the sampling rules of the public torchvision.ops.roi_align: rois =
 * r x (batch index, x1, y1, x2, y2), bilinear samples, sampling_ratio <= 0 -> ceil(roi extent / bins), `aligned`
 * shifts by half a pixel) over an NHWC float32 feature map, out [r][ph][pw][c]; and the FPN top-down merge (Lin et
 * al. 2017): out = lateral + nearest-neighbour 2x upsampling of top [n][ceil(h/2)][ceil(w/2)][c].  Device pointers. */
int rfi_op_roi_align(rfi_ctx* ctx, const float* x, int n, int h, int w, int c, const float* rois, int r,
                     float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, float* out);
int rfi_op_roi_align_backward(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, const float* rois,
                              int r, float spatial_scale, int ph, int pw, int sampling_ratio, int aligned,
                              float* dx);
/* roi_align_backward_sorted: the same gradient by gather -- RoIs sorted by batch index (ascending); every element of dx is
 * WRITTEN (no prior zeroing, no accumulation), no float atomics: bit-reproducible.  Device pointers. */
int rfi_op_roi_align_backward_sorted(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, const float* rois_sorted, int r,
                                     float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, float* dx);
/* mask_targets: the training targets of the mask branch -- RoIAlign (rules above, scale 1, not aligned) of one-channel
 * uint8 instance masks [g][h][w], thresholded at 0.5: rois[r] = (instance index, x1, y1, x2, y2) -> out uint8 [r][ph][pw].
 * Device pointers. */
int rfi_op_mask_targets(rfi_ctx* ctx, const uint8_t* masks, int g, int h, int w, const float* rois, int r, int ph, int pw,
                        int sampling_ratio, uint8_t* out);
/* Region-proposal pieces (not in the reference: Faster R-CNN's box parameterisation and RPN loss, greedy IoU
 * NMS, oracle/detection_ref.py).  All tensors device pointers unless named *_host.
 * box_decode: boxes[n][4] = decode(anchors[n_anchors][4] repeating, deltas[n][4]) (weights 1, dw/dh <= log(1000/16)),
 *   clipped to [0, clip_w] x [0, clip_h] when clip_w > 0.
 * nms: boxes sorted by descending score -> indices kept by greedy suppression at IoU > iou_threshold (host array of n ints).
 * rpn_loss: head output [pixels][5 A] (A objectness logits, then A x 4 deltas), labels int8 [pixels A] in {1, 0, -1 =
 *   not sampled}, regression targets [pixels A][4]; objectness = sum BCE over sampled / num_sampled, box = sum smooth-L1
 *   (beta) over positives / num_sampled; dhead = gradient of (objectness + box) w.r.t. the head output. */
int rfi_op_box_decode(rfi_ctx* ctx, const float* anchors, int64_t n_anchors, const float* deltas, int64_t n, float clip_h,
                      float clip_w, float* boxes);
/* anchor_match: the Matcher + BoxCoder.encode of RPN training on device -- per anchor the first-argmax ground truth,
 *   label 1 (IoU >= fg_iou, or, with allow_low_quality, the anchor attains some ground truth's best IoU), 0 (IoU < bg_iou),
 *   -1 (between); matched[n] = ground-truth index for labels 1 (else -1); targets[n][4] (may be null) = encoded deltas of the
 *   positives, zeros elsewhere.  The random 256-anchor sampling that follows stays on the host. */
int rfi_op_anchor_match(rfi_ctx* ctx, const float* anchors, int64_t n, const float* gt_boxes, int n_gt, float fg_iou, float bg_iou,
                        int allow_low_quality, int8_t* labels, int32_t* matched, float* targets);
int rfi_op_nms(rfi_ctx* ctx, const float* boxes_sorted, int n, float iou_threshold, int32_t* keep_host, int* n_keep);
/* Batched forms -- one launch for every image of a batch, all tensors device pointers.
 * anchor_match_batched: anchors shared by the images (anchor_stride 0: [n][4]) or per image (anchor_stride = n:
 *   [images][n][4] with anchor_count[b] valid rows -- the proposals of the RoI stage; null: all n); gt_boxes
 *   [images][gt_max][4] with gt_count[b] valid rows; labels / matched [images][n], targets [images][n][4] (may be null).  A row
 *   beyond anchor_count[b] gets label -2.
 * nms_batched: `sets` independent sets of at most k <= 256 boxes, each sorted by descending score ([sets][k][4], count[s]
 *   valid rows) -> keep uint8 [sets][k] (1 kept, 0 suppressed or beyond count). */
int rfi_op_anchor_match_batched(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int32_t* anchor_count,
                                const float* gt_boxes, int images, int gt_max, const int32_t* gt_count, float fg_iou, float bg_iou,
                                int allow_low_quality, int8_t* labels, int32_t* matched, float* targets);
int rfi_op_nms_batched(rfi_ctx* ctx, const float* boxes_sorted, const int32_t* count, int sets, int k, float iou_threshold, uint8_t* keep);
int rfi_op_rpn_loss(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                    const float* targets, int64_t num_sampled, float beta, float* dhead, float* loss_objectness,
                    float* loss_box);
/* rpn_loss_dev: rpn_loss without the host round trip -- the two loss scalars land in loss2_dev[0..1] (device), nothing
 * synchronises; workspace: rfi_op_rpn_loss_ws_bytes() bytes of device memory the caller keeps until the stream has passed. */
int rfi_op_rpn_loss_dev(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                        const float* targets, int64_t num_sampled, float beta, float* dhead, void* workspace, float* loss2_dev);
size_t rfi_op_rpn_loss_ws_bytes(void);
/* ---- The detector's box bookkeeping on the device (csrc/detect_sample.hip): the work rfi_toolbox_amd.models.MaskRCNN did in
 * NumPy between the GPU stages (the reference has no detector, so no reference interface is replaced: SURVEY 8a A11).  All
 * pointers are device pointers unless named *_host; nothing here synchronises or allocates.
 * rpn_loss_devcount: rfi_op_rpn_loss_dev with the normaliser read from device memory (the sampler's count).
 * fastrcnn_loss_dev: rfi_op_fastrcnn_loss leaving (classification, box) in loss2_dev; workspace as rpn_loss_dev.
 * anchor_match_batched_ws: rfi_op_anchor_match_batched with a caller-held workspace of images x gt_max floats.
 * segsort_u64: n_segs segments of `stride` (a power of two <= 8192) 64-bit keys, each sorted ascending in place.
 * sample_keys: labels [images][n] (1 positive, 0 negative) -> keys [images][stride] = class << 48 | r << 16 | i with
 *   r = Philox4x32-10(counter (i, image, stream0 + class, step), key seed).x; rows >= count[image] (null: n) get ~0.
 * rpn_sample_apply: from the SORTED keys, `batch` anchors per image (at most max_pos positive, smallest keys first) keep their
 *   label, other labels >= 0 become -1; labels / targets are written level by level (level l holds anchors
 *   [level_off[l], level_off[l + 1]) of every image: [images][count_l] and [..][4]); *n_sampled += anchors sampled.
 * topk_keys / topk_decode: keys of the objectness logits of head [images][pixels][5 A] (descending, ties by anchor index);
 *   from the sorted keys the k best are decoded against `anchors`, clipped, boxes under min_size moved behind the rest
 *   (score -inf) -> slot `level` of boxes [images][levels][k][4], scores [images][levels][k], counts [images][levels].
 * proposals_select: per image the post_nms best kept candidates (descending score, ties level-major) followed by its
 *   ground-truth boxes -> props [images][pmax][4], pcount [images].
 * roi_sample: `batch` proposals per image, at most max_pos with matcher label 1 -> sel [images][batch] (positives first, -1
 *   padding), nsel / npos [images].  roi_compact: the batch's RoIs, image-major and compact: rois [R][5], class labels
 *   (gt_labels [images][gt_max] of the matched instance; 0 background), targets, matched instance (-1), pyramid level
 *   0 + [area >= t1] + [area >= t2] + [area >= t3], img_start [images + 1]; the foreground rows again (rois_fg, rois_gt with
 *   the global instance index gt_base[image] + matched as column 0, level_fg, fg_start); counts = (R, Rf).
 * roi_align_ml / _backward: RoIAlign where RoI r uses map level[r] of four ([n][h0 >> k][w0 >> k][c], scale scale0 / 2^k; host
 *   arrays of device pointers); the row count is read from count_dev, max_rois sizes the launch; backward ADDS into dmaps.
 * readback_begin / _end: a copy of <= 2048 bytes to the host that waits for the work enqueued BEFORE begin only. */
int rfi_op_rpn_loss_devcount(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                             const float* targets, const int32_t* num_sampled_dev, float beta, float* dhead, void* workspace,
                             float* loss2_dev);
int rfi_op_fastrcnn_loss_dev(rfi_ctx* ctx, const float* head, int64_t rois, int num_classes, const int32_t* labels, const float* targets,
                             float beta, float* dhead, void* workspace, float* loss2_dev);
int rfi_op_anchor_match_batched_ws(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int32_t* anchor_count,
                                   const float* gt_boxes, int images, int gt_max, const int32_t* gt_count, float fg_iou, float bg_iou,
                                   int allow_low_quality, float* best_ws, int8_t* labels, int32_t* matched, float* targets);
int rfi_op_segsort_u64(rfi_ctx* ctx, uint64_t* keys, int n_segs, int stride);
int rfi_op_sample_keys(rfi_ctx* ctx, const int8_t* labels, int images, int n, const int32_t* count, uint64_t seed, uint32_t step,
                       uint32_t stream0, uint64_t* keys, int stride);
int rfi_op_rpn_sample_apply(rfi_ctx* ctx, const uint64_t* keys_sorted, int images, int n, int stride, int batch, int max_pos,
                            const int8_t* labels, const float* targets, int levels, const int32_t* level_off_host,
                            int8_t* const* level_labels_host, float* const* level_targets_host, int32_t* n_sampled);
int rfi_op_topk_keys(rfi_ctx* ctx, const float* head, int images, int pixels, int anchors_per_pixel, uint64_t* keys, int stride);
int rfi_op_topk_decode(rfi_ctx* ctx, const uint64_t* keys_sorted, int images, int stride, int pixels, int anchors_per_pixel, int k,
                       const float* head, const float* anchors, float clip_h, float clip_w, float min_size, float* boxes, float* scores,
                       int32_t* counts, int levels, int level);
int rfi_op_proposals_select(rfi_ctx* ctx, const float* boxes, const float* scores, const uint8_t* keep, int images, int levels, int k,
                            int post_nms, const float* gt_boxes, int gt_max, const int32_t* gt_count, int pmax, float* props,
                            int32_t* pcount);
int rfi_op_roi_sample(rfi_ctx* ctx, const int8_t* labels, const int32_t* pcount, int images, int pmax, int batch, int max_pos,
                      uint64_t seed, uint32_t step, uint32_t stream0, int32_t* sel, int32_t* nsel, int32_t* npos);
int rfi_op_roi_compact(rfi_ctx* ctx, const int32_t* sel, const int32_t* nsel, const int32_t* npos, int images, int batch, int pmax,
                       const float* props, const int32_t* matched, const float* targets, const int32_t* gt_labels, int gt_max,
                       const int32_t* gt_base, float t1, float t2, float t3, float* rois, int32_t* cls, float* tgt, int32_t* gt,
                       int32_t* level, int32_t* img_start, float* rois_fg, float* rois_gt, int32_t* level_fg, int32_t* fg_start,
                       int32_t* counts);
int rfi_op_roi_align_ml(rfi_ctx* ctx, const float* const* maps_host, int n, int h0, int w0, int c, float scale0, const float* rois,
                        const int32_t* level, const int32_t* count_dev, int max_rois, int ph, int pw, int sampling_ratio, float* out);
int rfi_op_roi_align_ml_backward(rfi_ctx* ctx, float* const* dmaps_host, int n, int h0, int w0, int c, float scale0, const float* dout,
                                 const float* rois, const int32_t* level, const int32_t* img_start, int max_rois, int ph, int pw,
                                 int sampling_ratio);
/* ---- elementwise kernels of the bfloat16 data flow (ResNet-style encoder, BatchNorm backward), for the kernel-level parity
 *      tests.  bfloat16 tensors are dense [m][c] arrays of uint16 bit patterns, c % 8 == 0, 16-byte aligned.  No counterpart
 *      in the reference (its tensors are float32): these are the builder's reduced-precision storage forms of
 *      relu(BN(y) + shortcut) (torchvision-style BasicBlock tail), of the masked sum of gradient terms at a block output,
 *      and of nn.BatchNorm2d's training-mode backward in front of a (Leaky)ReLU with slope `slope` (1: no activation).
 *      bn_backward16: dgamma / dbeta / dbias (dbias may be null) are float32 [c]; dy is bfloat16. ---- */
int rfi_op_bn_add_relu16(rfi_ctx* ctx, const uint16_t* y, const float* scale, const float* shift, const uint16_t* s, const float* s_scale,
                         const float* s_shift, int64_t m, int c, uint16_t* out);
int rfi_op_relu_mask_sum16(rfi_ctx* ctx, const uint16_t* g0, const uint16_t* g1, const float* g2_f32, const uint16_t* g2_bf16,
                           int64_t g2_stride, const uint16_t* a, int64_t m, int c, uint16_t* dz);
int rfi_op_bn_backward16(rfi_ctx* ctx, const uint16_t* da, const uint16_t* y, int64_t m, int c, const float* gamma, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float slope, uint16_t* dy, float* dgamma,
                         float* dbeta, float* dbias);
int rfi_readback_begin(rfi_ctx* ctx, const void* src_dev, size_t bytes);
int rfi_readback_end(rfi_ctx* ctx, void* dst_host, size_t bytes);
int rfi_op_fpn_merge(rfi_ctx* ctx, const float* lateral, const float* top, int n, int h, int w, int c, float* out);
int rfi_op_fpn_merge_backward(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, float* dtop);
int rfi_op_bn_stats(rfi_ctx* ctx, const float* y, int64_t m, int c, float* mean, float* var_biased);
/* a = relu(y*scale+shift): skip (n,h,w,c) and 2x2 max-pooled (n,h/2,w/2,c) */
int rfi_op_bn_relu_pool(rfi_ctx* ctx, const float* y, int n, int h, int w, int c, const float* scale,
                        const float* shift, float* skip, float* pooled);
/* da = dskip + max-pool routing of dpool (first maximum in row-major window order wins) */
int rfi_op_pool_bwd_merge(rfi_ctx* ctx, const float* y, int n, int h, int w, int c, const float* scale,
                          const float* shift, const float* dskip, const float* dpool, float* da);
/* train-mode BatchNorm+ReLU backward of y (m,c) with affine gamma/beta: da (grad w.r.t. the
 * activated output) is overwritten with dy; dgamma, dbeta, dbias(=sum dy) are written */
int rfi_op_bn_relu_backward(rfi_ctx* ctx, const float* y, int64_t m, int c, const float* gamma,
                            const float* beta, float* da_inout, float* dgamma, float* dbeta, float* dbias);

#ifdef __cplusplus
}
#endif
#endif /* RFI_HIP_H */
