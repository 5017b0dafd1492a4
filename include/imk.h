/* imk.h -- C ABI of libimk.so: the MI355X (gfx950) implementation of the Inconsistency-Mask hot path.
 *
 * The reference (MichaelVorndran/InconsistencyMasks) is pure Python and has no FFI/plugin layer; the
 * boundary it offers for this path is a set of Python functions.  Each entry
 * point below names the reference call site(s) whose arithmetic it replaces (file:line relative to the
 * reference checkout).  The Python host layer (inconsistencymasks_amd/) binds these with ctypes and
 * presents the reference's own function names on top (functions.py / unet.py mirrors).
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative IMK_E* code for bad arguments, a positive value = hipError_t;
 *   - never throw, never allocate or free caller memory, never synchronise: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and the caller owns every buffer until
 *     it has synchronised that stream;
 *   - all pointers are DEVICE pointers unless a parameter says "host";
 *   - compute entry points keep no mutable global state; a PLAN owns its fork / join events, so one plan must not be
 *     driven from two host threads at the same time (use one plan per thread; plans are cheap).  The side streams the
 *     training step and the ensemble forward fork onto are created once per device and shared by every plan of the
 *     process (more streams than the runtime has hardware queues serialise behind each other); they are never destroyed.
 *     The debug switches at the end of this file act on ONE plan (imk_unet_plan_debug); the measurement context
 *     (imk_prof_*) is created and owned by the caller and bound to the launching thread.
 *   - images and masks are uint8, NHWC; probabilities float32 NHWC; sizes int64.
 */
#ifndef IMK_H
#define IMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMK_VERSION 100 /* 0.1.0 */

/* libimk.so is built with -fvisibility=hidden: only the entry points declared here are exported. */
#define IMK_API __attribute__((visibility("default")))

enum {
    IMK_OK = 0,
    IMK_EINVAL = -1,      /* null pointer / non-positive size / unsupported combination */
    IMK_EUNSUPPORTED = -2,/* shape outside what the kernels cover (e.g. more than 64 classes) */
    IMK_EWORKSPACE = -3,  /* workspace too small */
};

IMK_API int imk_version(void);
IMK_API const char *imk_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Inconsistency-mask kernels
 * ---------------------------------------------------------------------------------------------- */

/* Binary / HeLa IM.  Replaces get_im_prediction_binary (functions.py:3140-3162: `> thr`, Kb = 1),
 * get_im_prediction_hela (functions.py:3165-3202: `>= thr`, Kb = 3, combined IM = max, size = sum),
 * pred_masks_to_im_binary (functions.py:3104-3120) and the blocking of functions.py:2867-2874, for a
 * whole batch in one launch.
 *   preds      [N,B,H,W,Kb] float32   probabilities of the N ensemble members
 *   img        [B,H,W,C] uint8 or NULL (no image blocking)
 *   img_out    [B,H,W,C] uint8 (ignored when img == NULL); = img with IM pixels zeroed if block_in
 *   masks_out  [B,Kb,H,W] uint8 {0,255}; IM pixels zeroed if block_out
 *   im_out     [B,H,W]   uint8 {0,255}  (max over the Kb channel IMs)
 *   im_size    [B,Kb] int64, pred_size [B,Kb] int64   -- counted BEFORE blocking, like the reference
 * NaN votes 0 under both comparison operators.                                                        */
IMK_API int imk_im_binary(const float *preds, int n_models, int batch, int h, int w, int kb,
                  float thr, int cmp_ge,
                  const uint8_t *img, int c, int block_in, int block_out,
                  uint8_t *img_out, uint8_t *masks_out, uint8_t *im_out,
                  int64_t *im_size, int64_t *pred_size, void *stream);

/* Multi-class IM.  Replaces get_im_prediction_multiclass (functions.py:3206-3238: argmax, first
 * maximum wins), pred_masks_to_im_multiclass (functions.py:3123-3137) and the blocking of
 * functions.py:3055-3062.
 *   probs      [N,B,H,W,K] float32, K <= 64
 *   final_out  [B,H,W] uint8 class ids (0 where the models disagree; zeroed on IM pixels if block_out --
 *              the same thing, kept for symmetry with the reference's order of operations)
 *   im_out     [B,H,W] uint8 {0,255}
 *   im_size    [B] int64
 *   presence   [N,B,K] uint8 or NULL: 1 where model n predicts class k somewhere in image b (the
 *              `np.unique` sets of functions.py:3226, for the unique-set filter of :3231-3234)        */
IMK_API int imk_im_multiclass(const float *probs, int n_models, int batch, int h, int w, int k,
                      const uint8_t *img, int c, int block_in, int block_out,
                      uint8_t *img_out, uint8_t *final_out, uint8_t *im_out,
                      int64_t *im_size, uint8_t *presence, void *stream);

/* k x k all-ones erosion (op = 0) / dilation (op = 1) of [B,H,W] uint8 masks, out-of-image taps ignored.
 * Replaces cv2.erode / cv2.dilate at functions.py:2858-2864 (dead in every shipped config: EK = DK = 0). */
IMK_API int imk_morph(const uint8_t *src, uint8_t *dst, int batch, int h, int w, int ksize, int op, void *stream);

/* image[im>0] = 0 (all C channels) and mask[im>0] = 0 for n_masks [B,H,W] masks packed as [B,n_masks,H,W]:
 * the blocking step of functions.py:2867-2874 when it has to run AFTER morphology changed the IM.
 * In place.  img or masks may be NULL.                                                                */
IMK_API int imk_block_apply(const uint8_t *im, uint8_t *img, int c, uint8_t *masks, int n_masks,
                    int batch, int h, int w, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Tiny U-Net (unet.py:4-67): plan, parameters, forward, fused forward+IM, training step
 * ---------------------------------------------------------------------------------------------- */

typedef struct imk_unet_cfg {
    int h, w;        /* input height / width; multiples of 16 (4 poolings); h * w < 2^24 and w < 2^16, else IMK_EUNSUPPORTED */
    int c_in;        /* image channels (1 or 3)                              */
    int n_out;       /* output maps K (unet.py:63)                           */
    int ch[5];       /* int(16a), int(32a), int(64a), int(128a), int(256a)   (unet.py:49-56) */
    int act_out;     /* 0 = sigmoid, 1 = softmax                             */
} imk_unet_cfg;

typedef struct imk_unet_plan imk_unet_plan; /* opaque, host memory (U-Net and EvalNet plans) */

/* EvalNet (evalnet.py:24-73), see the section at the end of this file */
typedef struct imk_evalnet_cfg {
    int h, w;            /* input height / width: even, >= 64 (6 poolings; odd rows / columns are dropped like Keras does);
                            h * w < 2^24 and w < 2^16, else IMK_EUNSUPPORTED */
    int ca, cb;          /* channels of input A (image) and input B (mask stack); 1..4 each, cb up to 64 with b_onehot */
    int n_out;           /* units per Dense head: 1 (get_evalnet) or inputB_channels (get_evalnet_miou) */
    int two_heads;       /* 0: one sigmoid head (evalnet.py:45); 1: 'iou' + 'detection' (evalnet.py:70-71) */
    int normalize_a, normalize_b;   /* the x/255 Lambda of each input block (evalnet.py:5-6)            */
    int ch[5];           /* int(16a), int(32a), int(64a), int(128a), int(256a)  (evalnet.py:28-41)      */
    int b_onehot;        /* 1: input B is a class-id map [B,H,W] u8, expanded on the device to the one-hot stack over cb
                            classes that the reference feeds (functions.py:4978, 5990); no x/255 on it */
} imk_evalnet_cfg;

IMK_API int imk_unet_plan_create(const imk_unet_cfg *cfg, imk_unet_plan **out);
IMK_API void imk_unet_plan_destroy(imk_unet_plan *plan);

/* Flat fp32 parameter vector.  Trainable section first (what AdamW and the gradient all-reduce see),
 * in unet.py creation order: per Conv2D kernel [kh][kw][cin][cout] (Keras HWIO) then bias [cout]; per
 * BatchNormalization gamma then beta.  Then the non-trainable section: per BN moving_mean, moving_var. */
IMK_API int imk_unet_param_count(const imk_unet_plan *plan, int64_t *total, int64_t *trainable);

typedef struct imk_layer_info {
    char name[16];   /* "in.c", "e1.c3", "e1.c1", "e1.bn", "b.c3", ..., "d6.ca", "d6.bna", ..., "out" */
    int kind;        /* 0 = conv, 1 = batch-norm */
    int ksize, cin, cout;
    int64_t off_w, off_b;         /* conv: kernel, bias;   bn: gamma, beta   (offsets in floats) */
    int64_t off_mean, off_var;    /* bn only */
} imk_layer_info;
IMK_API int imk_unet_num_layers(const imk_unet_plan *plan);
IMK_API int imk_unet_layer_info(const imk_unet_plan *plan, int idx, imk_layer_info *out);

/* fp16 copies of the conv kernels in MFMA-fragment order (forward and transposed/flipped for dgrad),
 * folded BN scale/shift.  Re-run after every change of `params`.  train = 0 folds the moving statistics
 * into scale/shift (inference); in training the batch statistics are produced by the step itself. */
IMK_API int64_t imk_unet_packed_bytes(const imk_unet_plan *plan);
IMK_API int imk_unet_pack_weights(const imk_unet_plan *plan, const float *params, void *packed, void *stream);

/* Workspace (activations, gradients, reduction scratch).  mode 0 = inference, 1 = training. */
IMK_API int64_t imk_unet_workspace_bytes(const imk_unet_plan *plan, int batch, int mode);

/* Batched inference: replaces model.predict at functions.py:3157, 3184, 3224 (and the batch-64 form at
 * :1120).  x [B,H,W,c_in] uint8 (the model divides by 255 itself, unet.py:5) -> probs [B,H,W,n_out] f32. */
IMK_API int imk_unet_forward(const imk_unet_plan *plan, const float *params, const void *packed,
                     const uint8_t *x, int batch, float *probs, void *workspace, int64_t workspace_bytes,
                     void *stream);

/* Debug/parity: after imk_unet_forward or a training step, byte offset/shape of a stored intermediate
 * inside the workspace (fp16, NHWC with the channel count padded to a multiple of 8).
 * which: 0 = conv output of layer `layer_idx` (post-ReLU, pre-BN).  Returns IMK_EINVAL if not stored. */
IMK_API int imk_unet_tensor_info(const imk_unet_plan *plan, int batch, int mode, int layer_idx, int which,
                         int64_t *byte_offset, int *h, int *w, int *c, int *c_stride);

/* Ensemble inference fused with the IM chain: N models' forward passes, then head -> threshold/argmax ->
 * agreement -> IM -> blocking in ONE kernel that reads the N last decoder activations (fp16) and never writes the
 * probability stack: the head's fp32 arithmetic is the same code as imk_unet_forward's, so the outputs are bit-identical
 * to imk_unet_forward + imk_im_binary / imk_im_multiclass.  One call = functions.py:2844-2887 minus file I/O, for a batch.
 * `params`/`packed` are arrays (host) of n_models device pointers.
 * binary heads (act_out = 0): outputs as imk_im_binary;  softmax heads: as imk_im_multiclass
 * (masks_out = final_out [B,H,W], pred_size unused, presence optional).
 * workspace: imk_unet_forward_im_workspace_bytes(plan, n_models, B, k), 1 <= k <= min(n_models, 3): with k > 1 the models
 * run on k streams side by side (forked from and joined to `stream` with events; the call stays asynchronous), with
 * k = 1 back to back.  Shapes the fused kernel does not cover (sigmoid heads with more than 4 maps, H*W not a multiple
 * of 16, more than 8 models) and plans with the materialize debug switch take the unfused route through the fp32 probability stack
 * (n_models * align256(B*H*W*n_out*4) + k * imk_unet_workspace_bytes(plan, B, 0) bytes are enough for that one).      */
IMK_API int64_t imk_unet_forward_im_workspace_bytes(const imk_unet_plan *plan, int n_models, int batch, int n_streams);
IMK_API int imk_unet_forward_im(const imk_unet_plan *plan, int n_models,
                        const float *const *params, const void *const *packed,
                        const uint8_t *x, int batch, float thr, int cmp_ge,
                        const uint8_t *img, int block_in, int block_out,
                        uint8_t *img_out, uint8_t *masks_out, uint8_t *im_out,
                        int64_t *im_size, int64_t *pred_size, uint8_t *presence,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* Training (functions.py:207-218, one step of model.fit): forward with batch statistics, loss,
 * backward, optimizer.  Split in two so that a data-parallel caller can all-reduce `grads` in between.
 *   loss_kind 0 = 'mse' on {0,1} targets y u8 [B,H,W,n_out]   (ISIC, HeLa)
 *             1 = categorical cross-entropy, y u8 [B,H,W] class ids (one-hot formed on the fly)
 *   grads     [trainable] float32, UNSCALED gradient of the mean loss
 *   stats     device float[4]: {loss, found_inf (0/1), loss_scale used, reserved}
 *   state     device buffer of imk_unet_state_bytes(): Adam m, v, step counter, dynamic loss scale
 * imk_unet_fwd_bwd also updates the BN moving statistics in `params` (momentum 0.99).                  */
IMK_API int64_t imk_unet_state_bytes(const imk_unet_plan *plan);
IMK_API int imk_unet_state_init(const imk_unet_plan *plan, void *state, void *stream);
IMK_API int imk_unet_fwd_bwd(const imk_unet_plan *plan, float *params, void *packed, void *state,
                     const uint8_t *x, const uint8_t *y, int batch, int loss_kind,
                     float *grads, float *stats, void *workspace, int64_t workspace_bytes, void *stream);

/* tensorflow_addons AdamW (functions.py:215): var -= wd*var; Adam(b1,b2,eps) with bias-corrected lr.
 * Skips the update (and halves the dynamic loss scale) when stats[1] != 0; re-packs the conv weights for the next
 * training step.  It does NOT refresh the folded inference BatchNorm statistics (training never reads them):
 * call imk_unet_pack_weights before the next imk_unet_forward / imk_unet_forward_im.
 * grad_scale multiplies grads first (1/world_size after a sum all-reduce).                            */
IMK_API int imk_unet_adamw_step(const imk_unet_plan *plan, float *params, void *packed, void *state,
                        const float *grads, const float *stats, float grad_scale,
                        float lr, float wd, float beta1, float beta2, float eps, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Noisy-Student augmentation (IM+ / AIM+ drivers)
 * ---------------------------------------------------------------------------------------------- */
typedef struct imk_aug_params {   /* one per image, device array */
    int flip_v, flip_h;           /* cv2.flip(.., 0) / cv2.flip(.., 1)                (functions.py:2796-2803) */
    int rot;                      /* 0 none, 1 = 90 CW, 2 = 180, 3 = 90 CCW           (functions.py:2805-2818) */
    int bright_on;                /* apply convertScaleAbs(alpha, beta)               (functions.py:2820-2824) */
    float alpha, beta;
    int blur_k;                   /* 0/1 none, 3, 5, 7: GaussianBlur((k,k), 0)        (functions.py:1495-1501) */
    int noise_max;                /* uniform integer noise in [-m, m), then clip      (functions.py:1463-1478) */
    uint32_t seed;                /* per-image seed of the counter-based noise generator */
} imk_aug_params;

/* Epoch assembly of a training set that lives in HBM: row r of the outputs = row idx[r] of the inputs -- the shuffle + batch of
 * the reference's tf.data pipeline (functions.py:207-209) with the mask normalisation of its parsers folded in
 * (parse_image_ISIC_2018: mask / 255, functions.py:975; parse_image_hela: / 255 and position x Position_weight,
 * functions.py:1001-1011).
 *   img   [n_src, row_img] uint8 (any layout; a row is copied as it is)            -> img_out  [n, row_img]     (img may be NULL)
 *   mask  [n_src, planes, hw] uint8, planar (as imk_im_binary / imk_unet_forward_im write label maps)
 *                                                                                   -> mask_out [n, hw, planes], interleaved,
 *         every value v -> (div255 ? v / 255 : v) * (mul ? mul[plane] : 1)  (mul: device, [planes])   (mask may be NULL)
 *   idx   device, int64 [n]; n <= 65535.  Outputs must not alias inputs. */
IMK_API int imk_gather_pairs(const uint8_t *img, int64_t row_img, const uint8_t *mask, int planes, int64_t hw, int div255,
                             const uint8_t *mul, const int64_t *idx, int64_t n, uint8_t *img_out, uint8_t *mask_out,
                             void *stream);

/* Replaces augment_image_and_mask (functions.py:2779-2826) + add_noise_and_blur (:1481-1506) for a batch.
 *   img  [B,H,W,C] u8 -> img_out;  mask [B,H,W,Cm] u8 (or NULL) -> mask_out (geometric part only).
 * 90-degree turns need h == w (set any_quarter_turn if any params[i].rot is 1 or 3).  Not in place. */
IMK_API int imk_augment(const uint8_t *img, const uint8_t *mask, int batch, int h, int w, int c, int cm,
                const imk_aug_params *params, uint8_t *img_out, uint8_t *mask_out, int any_quarter_turn, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation reductions (benchmark_ISIC2018 / benchmark_multiclass)
 * ---------------------------------------------------------------------------------------------- */
/* Replaces the threshold + per-image numpy metric loop of benchmark_ISIC2018 (functions.py:1120-1140) with
 * get_IoU_binary (:1767-1788) and dice_score_numpy_binary (:1837-1861):
 *   probs [B,H,W] f32, gt [B,H,W] u8 -> pred_out [B,H,W] u8 in {0,255} (may be NULL), counts [B,5] int64 =
 *   { #(gt!=0 & pred), #(gt!=0 | pred), #(gt>=128), #pred, #(gt>=128 & pred) };  pred = p > thr (cmp_ge: >=). */
IMK_API int imk_eval_binary(const float *probs, float thr, int cmp_ge, const uint8_t *gt, int batch, int h, int w,
                    uint8_t *pred_out, int64_t *counts, void *stream);

/* Replaces argmax + pixel_accuracy (functions.py:1820-1834) + get_IoU_multi_unique (:1791-1816) of
 * benchmark_multiclass (:1308-1330):  probs [B,H,W,K] f32, gt [B,H,W] u8 class ids -> pred_out [B,H,W] u8 (may be
 * NULL), counts [B,4,256] int64: [0][v] = #(gt==v), [1][v] = #(pred==v), [2][v] = #(gt==v & pred==v),
 * [3][0] = #(pred==gt).  k <= 256. */
IMK_API int imk_eval_multiclass(const float *probs, const uint8_t *gt, int batch, int h, int w, int k, uint8_t *pred_out,
                        int64_t *counts, void *stream);

/* Reductions of the validation monitors of train_multiclass / train_hela (the ModelCheckpoint criteria, functions.py:
 * 255-258, 303-306), deterministic (fixed summation order):
 *   mode 0 -- MeanIoU.update_state (functions.py:75-86): probs [n_pix,K] f32, gt [n_pix] u8 class ids ->
 *             out[0..K) = sum_p [gt == k] p_k,  out[K..2K) = sum_p [gt == k],  out[2K..3K) = sum_p p_k
 *   mode 1 -- squared error of Keras' val_loss for 'mse': gt [n_pix,K] u8 targets -> out[0] = sum (p - y)^2
 * out: device buffer of imk_eval_soft_out_doubles(K) doubles (results first, block partials behind them); K <= 64. */
IMK_API int64_t imk_eval_soft_out_doubles(int k);
IMK_API int imk_eval_soft_sums(const float *probs, const uint8_t *gt, int64_t n_pix, int k, int mode, double *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * EvalNet (evalnet.py:24-73): the quality-scoring network of IM++ / AIM++
 * ----------------------------------------------------------------------------------------------
 * Replaces get_evalnet / get_evalnet_miou + model.predict([A, B]) (functions.py:5837-5941 call sites) and the
 * model.fit step of train_evalnet_ISIC_2018 / train_evalnet_miou_model_hela (functions.py:4464-4506, 4673-4722).
 * Two uint8 inputs A [B,H,W,ca] (image) and B [B,H,W,cb] (mask stack, or [B,H,W] class ids with b_onehot); each tower is input_block + conv_block, the
 * towers are concatenated, five conv_blocks follow, then GlobalAvgPool2D and the Dense sigmoid head(s).
 * The plan is an imk_unet_plan: imk_unet_param_count / _num_layers / _layer_info / _packed_bytes / _pack_weights /
 * _state_bytes / _state_init / _adamw_step / _plan_destroy apply unchanged.  Layer names: "a.in.c", "a.in.bn", "a.c3",
 * "a.c1", "a.bn", the same with "b.", "m1.c3", "m1.c1", "m1.bn" ... "m5.bn", then "dense" or "iou", "detection"
 * (Dense kernels [C, n_out] + bias, reported as 1x1 convs); parameter order = Keras creation order. */
IMK_API int imk_evalnet_plan_create(const imk_evalnet_cfg *cfg, imk_unet_plan **out);
IMK_API int64_t imk_evalnet_workspace_bytes(const imk_unet_plan *plan, int batch, int mode /* 0 inference, 1 training */);

/* out [B, n_heads*n_out] f32 sigmoid outputs (two heads: iou units first, then detection units); inference-mode BN */
IMK_API int imk_evalnet_forward(const imk_unet_plan *plan, const float *params, const void *packed, const uint8_t *xa,
                        const uint8_t *xb, int batch, float *out, void *workspace, int64_t workspace_bytes, void *stream);

/* Training pass: forward with batch statistics (moving statistics updated in params), losses, backward.
 *   y [B, n_heads*n_out] f32 targets.  Loss: head 0 mean squared error; head 1 binary cross-entropy (functions.py:4708
 *   loss=['mse','binary_crossentropy'], summed);  a single head: mean squared error (functions.py:4492).
 *   out (may be NULL): the training-mode outputs;  grads [n_trainable] f32 (unscaled);
 *   stats [8] f32: {total loss, overflow flag, loss scale, step, loss of head 0, loss of head 1, -, -}.
 * Follow with imk_unet_adamw_step. */
IMK_API int imk_evalnet_fwd_bwd(const imk_unet_plan *plan, float *params, void *packed, void *state, const uint8_t *xa,
                        const uint8_t *xb, const float *y, int batch, float *out, float *grads, float *stats,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* Debug/parity: like imk_unet_tensor_info, for EvalNet plans */
IMK_API int imk_evalnet_tensor_info(const imk_unet_plan *plan, int batch, int mode, int layer_idx, int which,
                            int64_t *byte_offset, int *h, int *w, int *c, int *c_stride);

/* Momentum of the BatchNorm moving statistics in the training steps of ONE plan (U-Net or EvalNet).  Default 0.99 = Keras'
 * BatchNormalization default, which the reference never overrides (unet.py:7).  Data-parallel runs with per-GPU batch 32 make
 * N times fewer optimizer steps per epoch: 0.99^N keeps the moving average's memory the same in SAMPLES (functions.py of this
 * repository: IMK_DP_BN_MOMENTUM=scaled; off by default, it is a deviation from the reference's recipe). */
IMK_API int imk_unet_plan_set_bn_momentum(imk_unet_plan *plan, float momentum);
IMK_API int imk_unet_plan_get_bn_momentum(const imk_unet_plan *plan, float *momentum);

/* Debug / measurement switches of ONE plan (U-Net or EvalNet); -1 leaves a switch as it is.  The library keeps no global state:
 * the caller owns the plan and these two flags in it.
 *   materialize = 1: inference also stores the intermediates that fused kernels normally keep on chip (the Conv3x3 output
 *     inside a fused Conv3x3 -> Conv1x1 kernel, the input block's output), so that imk_unet_tensor_info works on every layer.
 *     Training always stores them (the backward pass needs them).
 *   single_stream = 1: every kernel runs on the caller's stream (no side stream for the weight gradients, the ensemble's models
 *     back to back), so that per-kernel timings are exclusive.  Results are identical either way. */
IMK_API int imk_unet_plan_debug(imk_unet_plan *plan, int materialize, int single_stream);

/* ------------------------------------------------------------------------------------------------
 * Host-side PNG codec of the directory API (csrc/imk_png.cpp).  Replaces cv2.imread / cv2.imwrite of the
 * reference's writers and parsers (functions.py:2846, 2885-2887, 955-1048) for the files this path reads and writes.  ALL
 * pointers are HOST pointers; no GPU call is made; thread-safe and re-entrant (one file per call: callers parallelise over files,
 * ctypes drops the interpreter lock around the call).  Decoder: 8-bit greyscale / RGB / palette / greyscale + alpha / RGBA,
 * non-interlaced; anything else returns IMK_EUNSUPPORTED (the Python layer then uses Pillow).  want_c = 3: RGB (alpha dropped,
 * palette expanded, grey replicated); want_c = 1: Pillow's convert("L").  Encoder: 8-bit greyscale (c = 1) or RGB (c = 3),
 * deflate level 0-9.
 * ---------------------------------------------------------------------------------------------- */
IMK_API int imk_png_info(const char *path, int *h, int *w, int *color_type, int *bit_depth);
IMK_API int imk_png_read_file(const char *path, int want_c, uint8_t *out, int64_t out_cap, int *h_out, int *w_out);
IMK_API int imk_png_decode(const uint8_t *data, int64_t len, int want_c, uint8_t *out, int64_t out_cap, int *h_out, int *w_out);
IMK_API int imk_png_encode(const uint8_t *pixels, int h, int w, int c, int level, uint8_t *out, int64_t out_cap, int64_t *out_len);
IMK_API int imk_png_write_file(const char *path, const uint8_t *pixels, int h, int w, int c, int level);

/* ------------------------------------------------------------------------------------------------
 * Host-side geometry of the HeLa position masks (csrc/imk_geom.cpp).  HOST pointers, no GPU call,
 * thread-safe and re-entrant (one image per call).  Replace the reference's OpenCV calls:
 *   imk_pos_contours  functions.py:6181-6218 get_pos_contours: cv2.erode (erode_kernel x erode_kernel, odd or 0/1 for none) ->
 *                     threshold > 10 -> cv2.findContours(RETR_TREE) -> cv2.moments -> (int(m10 / m00) + 1, int(m01 / m00) + 1)
 *                     for every contour with m00 != 0 (blobs in raster order of their first pixel, each followed by its holes).
 *                     Writes up to `cap` (x, y) pairs, returns the number found (> cap: call again with a larger buffer) or a
 *                     negative IMK_E* code.
 *   imk_mod_pos_size  functions.py:6255-6292 mod_pos_size: blobs (3 x 3 erosion) re-drawn with cv2.circle of radius
 *                     clamp(min_dist // 4, min_r, max_r); blur2 != 0: cv2.blur (2, 2), < 254 -> 0.  lone_dist: the distance
 *                     used when the mask holds exactly one position (0 in mod_pos_size; 99 and blur2 = 0 in the pseudo-label
 *                     writer, functions.py:2952-2966).  out: h x w uint8 in {0, 255}.
 *   imk_cell_count    functions.py:6298-6371 get_cell_count: counts = {alive, dead, unclear} over the positions.
 * ---------------------------------------------------------------------------------------------- */
IMK_API int imk_pos_contours(const uint8_t *img, int h, int w, int erode_kernel, int32_t *xy, int cap);
IMK_API int imk_mod_pos_size(const uint8_t *img, int h, int w, int max_r, int min_r, int lone_dist, int blur2, uint8_t *out);
IMK_API int imk_cell_count(const int32_t *xy, int n, const uint8_t *alive, const uint8_t *dead, int h, int w,
                           int measuring_range, int32_t counts[3]);

/* Runtime environment checks.  imk_runtime_warnings() returns a bit mask of conditions the library has noticed so far in this
 * process (it also prints each once to stderr):
 *   IMK_WARN_HW_QUEUES  a side stream was requested (training step, ensemble forward) while GPU_MAX_HW_QUEUES is unset or
 *                       below 8: the HIP runtime's default of 4 hardware queues lets a side stream share the queue of the
 *                       stream it should run beside (measured: -14 % per generation).  Export GPU_MAX_HW_QUEUES=8 before the
 *                       first HIP call of the process (the Python package does so at import).
 * imk_unet_plan_side_stream returns side stream i (0 .. 1) of `plan` as a hipStream_t, creating it if needed -- the streams
 * belong to ONE pool per device shared by every plan (U-Net or EvalNet) of the process. */
#define IMK_WARN_HW_QUEUES 1
IMK_API int imk_runtime_warnings(void);
IMK_API int imk_unet_plan_side_stream(const imk_unet_plan *plan, int i, void **stream_out);

/* ------------------------------------------------------------------------------------------------
 * Measurement hook (bench.py): per-launch HIP-event timing of every kernel family of the path, on the stream the
 * kernel is launched on.  A context is created and owned by the caller and bound to the calling thread: while bound, every
 * period-th hooked launch made by THAT thread records an event pair into it (period 0 = off).  imk_prof_collect synchronises
 * those events and returns, per family v (0..IMK_PROF_VARIANTS-1), the number of sampled launches, their summed duration in
 * ms, their summed ALGORITHMIC bytes (every input tensor read once + every output tensor written once) and -- flops may be
 * NULL -- the summed FLOPs (2 x multiply-adds, logical channel counts; 0 for families that are not convolutions), then
 * resets the records.  Families:
 *   0..5  conv_mfma_kernel<TH,MT>: v = 3*(TH==8) + log2(MT)      6  conv_pipe_kernel
 *   7  wgrad_mfma_kernel      8  bn_bwd_prep(_pool)_kernel       9  bn_bwd_coef_kernel     10  bn_finalize_kernel
 *   11 wgf_stage1 + wgf_stage2 (one bracket)                     12 head_kernel            13  head_loss_kernel / head_cce_fused_kernel
 *   14 loss_finalize / adamw / pack_conv_batched / bn_fold_batched                         15  im_binary_* / im_multi_kernel
 *   16 conv_gemm_kernel (wide layers, forward / dgrad)           17 wgrad_gemm_kernel (wide layers, weight gradient)
 * A context must not be shared between threads that launch concurrently (two hipEventRecord per sampled launch while bound).
 * ---------------------------------------------------------------------------------------------- */
#define IMK_PROF_VARIANTS 18
typedef struct imk_prof imk_prof;   /* opaque, host memory */
IMK_API int imk_prof_create(int period, imk_prof **out);
IMK_API void imk_prof_destroy(imk_prof *ctx);
IMK_API int imk_prof_bind(imk_prof *ctx);              /* ctx or NULL (unbind) for the calling thread */
IMK_API int imk_prof_unbind(imk_prof *ctx);            /* unbinds the calling thread only if `ctx` is what it has bound (1 if it did, else 0) */
IMK_API int imk_prof_set_period(imk_prof *ctx, int period);
/* Totals for rocprofv3 cross-checks: while enabled, EVERY hooked launch of the bound thread adds its launch count, algorithmic
 * bytes and flops to a per-kernel-name sum (the conv_pipe / conv_wide families by template variant, spelled as rocprofv3 prints
 * them; the others by family).  imk_prof_totals_enable(ctx, on) clears the sums and switches them on / off;
 * imk_prof_totals_dump writes "name;launches;bytes;flops" lines into buf (cap bytes) and returns the size needed.
 * imk_prof_mark enqueues an empty marker kernel (imk_mark_kernel, 64 * id work-items) on `stream`: bench.py brackets its timed
 * region with ids 1 and 2 so that profiles/summarize.py can cut a kernel trace there. */
IMK_API int imk_prof_totals_enable(imk_prof *ctx, int on);
IMK_API int64_t imk_prof_totals_dump(imk_prof *ctx, char *buf, int64_t cap);
IMK_API int imk_prof_mark(int id, void *stream);
IMK_API int imk_prof_collect(imk_prof *ctx, int64_t *count, double *ms, double *bytes, double *flops);

#ifdef __cplusplus
}
#endif
#endif /* IMK_H */
