/*
 * sqdet.h -- C ABI of libsqdet_hip.so: the MI355X (gfx950) SqueezeDet hot path.
 *
 * The reference (BichenWuUCB/squeezeDet, TF 1.0 / Python 2.7) has NO FFI or
 * operator-plugin interface: its boundary is Python-level -- builder methods on
 * ModelSkeleton plus the model-object contract used by demo.py/eval.py/train.py
 * (SURVEY.md 8b).  This header is the C-ABI a maintainer would bind in place of
 * the TF graph ops those builders emit; every entry point cites the reference
 * interface it replaces (paths relative to the reference's src/).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP C++ types.
 *   - every data pointer is a CALLER-OWNED DEVICE pointer unless the name says
 *     "host"; nothing here allocates device memory.
 *   - sqdet_stream_t is a hipStream_t passed as void* (NULL = default stream);
 *     calls enqueue work on that stream and return without synchronising.
 *   - return value: SQDET_OK (0) or a negative SQDET_E* code; never throws.
 *     sqdet_last_error() returns a thread-local message for the last failure.
 *   - activations are NHWC; conv kernels are HWIO [kh,kw,Cin,Cout] float32
 *     (the reference's '<layer>/kernels' variables, nn_skeleton.py:531-533) and
 *     are re-laid-out once into MFMA fragment order by sqdet_conv_pack_weights.
 *   - dtype is the STORAGE type of activations/weights (SQDET_F16 or SQDET_F32);
 *     accumulation is always float32; biases are always float32.
 */
#ifndef SQDET_H
#define SQDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sqdet_stream_t;

enum { SQDET_OK = 0, SQDET_EINVAL = -1, SQDET_EUNSUPPORTED = -2, SQDET_EHIP = -3, SQDET_ESTATE = -4 };
enum { SQDET_F32 = 0, SQDET_F16 = 1 };
enum { SQDET_PAD_SAME = 0, SQDET_PAD_VALID = 1 };
enum { SQDET_ARCH_SQUEEZEDET = 0, SQDET_ARCH_SQUEEZEDET_PLUS = 1, SQDET_ARCH_RESNET50 = 2 };

const char* sqdet_version(void);
const char* sqdet_last_error(void);
/* Tuning knobs (process-wide).  "conv_algo": 0 = auto (specialised kernels when eligible,
 * default), 1 = generic implicit-GEMM kernels only (also env SQDET_CONV_ALGO=generic).
 * "fire_fuse": 0 = plan heuristic (default), 1 = one launch per fire module wherever the kernels cover it, 2 = never
 * fuse, 3 = no streaming kernel, 4 = fire modules and the pools behind them stay apart, 5 = no fire-module chains, 6 = chains on
 * the late (small) maps only, 7 = a run's first module as squeeze conv + chain launch instead of one streaming launch,
 * 8 = no streaming expand + next-squeeze launches (a pooled module then ends its run). */
int sqdet_set_option(const char* name, int value);

/* ------------------------------------------------------------------ conv --
 * Replaces ModelSkeleton._conv_layer (nn_skeleton.py:471-563):
 *   relu?(conv2d(x, W, [1,s,s,1], padding) + b), TF SAME/VALID semantics
 *   (asymmetric SAME: the extra pad cell goes bottom/right).
 */

/* Bytes of the packed (MFMA fragment order) form of a [k,k,cin,cout] kernel. */
size_t sqdet_conv_packed_bytes(int k, int cin, int cout, int dtype);

/* w_hwio_f32: device float32 [k,k,cin,cout] -> packed (dtype storage). */
int sqdet_conv_pack_weights(const float* w_hwio_f32, void* packed, int k, int cin, int cout, int dtype,
                            sqdet_stream_t stream);

/* x: [n,h,w,cin] -> y: [n,ho,wo,*] written at channel offset y_coffset of rows
 * y_cstride channels wide (y_cstride=cout, y_coffset=0 for a plain conv; a fire
 * module's two expand convs write the two halves of one concat tensor,
 * nets/squeezeDet.py:106).  bias: float32 [cout].  relu: 0/1. */
int sqdet_conv2d_nhwc_fwd(const void* x, const void* w_packed, const float* bias, void* y,
                          int n, int h, int w, int cin, int cout, int k, int stride, int pad_mode, int relu,
                          int dtype, int y_cstride, int y_coffset, sqdet_stream_t stream);

/* The ConvDet head (nets/squeezeDet.py:76-79: conv12, 3x3 / SAME, no ReLU) TOGETHER WITH the score half of
 * _add_interpretation_graph (nn_skeleton.py:150-170, 274-283): preds [n,h,w,apg*(classes+5)] as sqdet_conv2d_nhwc_fwd would
 * write them, and scores float32 [n, h*w*apg] = det_probs (max over classes of softmax(class logits) * sigmoid(confidence)),
 * computed in the conv's epilogue from the float16-rounded preds with the float expressions of sqdet_interpret_output --
 * bitwise the det_probs that call returns.  float16, anchors_per_grid 9, classes 3, Cin a multiple of 128 (the split-K
 * ConvDet kernel); SQDET_EUNSUPPORTED otherwise (sqdet_convdet_scores_supported tells in advance).  Follow with
 * sqdet_detect_filter_scored: the whole post-processing is then one 32-workgroup launch. */
int sqdet_convdet_fwd(const void* x, const void* w_packed, const float* bias, void* preds, float* scores, int n, int h, int w,
                      int cin, int anchors_per_grid, int classes, int dtype, sqdet_stream_t stream);
int sqdet_convdet_scores_supported(int cin, int anchors_per_grid, int classes, int dtype);

/* Residual form used by ResNet50ConvDet (nets/resnet50_convDet.py:55, `tf.nn.relu(branch1+branch2)`):
 *   y = relu?(conv2d(x, W) + b + y)   -- y holds the shortcut branch on entry, the block output on exit
 * (the branch2c 1x1 conv of a bottleneck adds its result to the shortcut in its own epilogue, so
 * the sum never makes an extra HBM round trip).  Same arguments as sqdet_conv2d_nhwc_fwd. */
int sqdet_conv2d_add_nhwc_fwd(const void* x, const void* w_packed, const float* bias, void* y_inout,
                              int n, int h, int w, int cin, int cout, int k, int stride, int pad_mode, int relu,
                              int dtype, int y_cstride, int y_coffset, sqdet_stream_t stream);
/* The same with the shortcut in a tensor of its own (rows of y_cstride channels, like y), left untouched:
 *   y = relu?(conv2d(x, W) + b + residual)
 * -- the training forward keeps every block input for the backward pass (nn_skeleton.py:329-361 differentiates through
 * resnet50_convDet.py:55), so the sum must not overwrite the shortcut.  1x1 convs read the residual tile in their epilogue
 * (conv1x1_pipe); other shapes copy it into y first. */
int sqdet_conv2d_res_nhwc_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                              int n, int h, int w, int cin, int cout, int k, int stride, int pad_mode, int relu,
                              int dtype, int y_cstride, int y_coffset, sqdet_stream_t stream);

/* ------------------------------------------------------------- batch norm --
 * Replaces the frozen-statistics batch norm of ModelSkeleton._conv_bn_layer (nn_skeleton.py:374-468:
 * tf.nn.batch_normalization(conv [+ biases], mean, var, offset=beta, scale=gamma, eps) with mean/var
 * non-trainable constants) by folding it into the conv it follows:
 *   inv = gamma / sqrt(var + eps);  w_folded[..., c] = w[..., c] * inv[c];
 *   b_folded[c] = (conv_bias[c] - mean[c]) * inv[c] + beta[c]          (conv_bias may be NULL = 0)
 * w_hwio / w_folded: float32 [k,k,cin,cout] (may alias); the per-channel vectors float32 [cout]. */
int sqdet_fold_batchnorm(const float* w_hwio, const float* conv_bias, const float* gamma, const float* beta,
                         const float* mean, const float* var, float eps, float* w_folded, float* b_folded,
                         int k, int cin, int cout, sqdet_stream_t stream);

/* Training of a _conv_bn_layer conv (the trainable res4* blocks, resnet50_convDet.py:94-118): the conv
 * backward kernels produce the gradients of the FOLDED kernel / bias; this turns them into the
 * gradients of the variables (float32):  dw = dw_folded * gamma/sqrt(var+eps)  (may alias dw_folded),
 *   dgamma = (sum over k,k,cin of dw_folded * w + (conv_bias - mean) * db_folded) / sqrt(var+eps),
 *   dbeta = db_folded.   conv_bias may be NULL.  workspace: sqdet_fold_batchnorm_bwd_workspace_bytes(...) of device
 *   scratch (per-row-block column sums, added in a fixed order: deterministic). */
size_t sqdet_fold_batchnorm_bwd_workspace_bytes(int k, int cin, int cout);
int sqdet_fold_batchnorm_bwd(const float* w_hwio, const float* dw_folded, const float* db_folded,
                             const float* conv_bias, const float* gamma, const float* mean, const float* var, float eps,
                             float* dw, float* dgamma, float* dbeta, float* workspace, int k, int cin, int cout,
                             sqdet_stream_t stream);
/* The same for MANY convs in two launches (per conv the results are bitwise sqdet_fold_batchnorm_bwd's): prepare() fills
 * a host table of sqdet_fold_batchnorm_bwd_many_table_bytes(n_items) bytes from per-item pointer arrays (conv_bias[i] may
 * be NULL; workspace[i]: that conv's sqdet_fold_batchnorm_bwd_workspace_bytes) and reports the two grids; the caller copies
 * the table to the device once. */
size_t sqdet_fold_batchnorm_bwd_many_table_bytes(int n_items);
int sqdet_fold_batchnorm_bwd_many_prepare(const float* const* w_hwio, const float* const* dw_folded,
                                          const float* const* db_folded, const float* const* conv_bias,
                                          const float* const* gamma, const float* const* mean, const float* const* var,
                                          float* const* dw, float* const* dgamma, float* const* dbeta, float* const* workspace,
                                          const int* k, const int* cin, const int* cout, int n_items, void* table_host,
                                          int* blocks, int* finish_blocks);
int sqdet_fold_batchnorm_bwd_many(const void* table_dev, int n_items, int blocks, int finish_blocks, float eps,
                                  sqdet_stream_t stream);

/* y[n,oy,ox,:] = x[n,oy*stride,ox*stride,:], y: [n,ceil(h/stride),ceil(w/stride),c] -- the pixels a 1x1
 * stride-s SAME conv reads (res3a/res4a branch1 and branch2a), so that their filter gradient can use
 * sqdet_conv2d_nhwc_bwd_filter (stride 1) on the gathered tensor. */
int sqdet_subsample_nhwc(const void* x, void* y, int n, int h, int w, int c, int stride, int dtype,
                         sqdet_stream_t stream);

/* ------------------------------------------------------------------ pool --
 * Replaces ModelSkeleton._pooling_layer (nn_skeleton.py:565-586): tf.nn.max_pool,
 * SAME-padded cells never win.  x: [n,h,w,c] -> y: [n,ho,wo,c]. */
int sqdet_maxpool_nhwc_fwd(const void* x, void* y, int n, int h, int w, int c, int k, int stride, int pad_mode,
                           int dtype, sqdet_stream_t stream);
/* The training forward's pool (k = 3): also records, per output element, which window cell won -- one byte,
 * 3 * row + column of the FIRST maximum in row-major order (tf.nn.max_pool's gradient convention), 255 if none did --
 * for sqdet_maxpool_nhwc_bwd_idx.  window_index: [n,ho,wo,c] uint8. */
int sqdet_maxpool_nhwc_fwd_idx(const void* x, void* y, unsigned char* window_index, int n, int h, int w, int c, int k,
                               int stride, int pad_mode, int dtype, sqdet_stream_t stream);

/* ------------------------------------------------------------------ stem --
 * conv1 + pool1 in one launch: relu(conv2d(x, W, stride 2) + b) followed by max_pool 3x3/s2
 * (nets/squeezeDet.py:40-44: k=3, 64 filters, SAME/SAME; nets/squeezeDetPlus.py:40-44: k=7,
 * 96 filters, VALID/VALID).  x: [n,h,w,3]; y: [n,hp,wp,cout].  Only those two stems are fused. */
int sqdet_stem_conv_pool_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w,
                             int cout, int k, int conv_pad_mode, int pool_pad_mode, int dtype, sqdet_stream_t stream);
/* The same ending with the NEXT layer's squeeze1x1 (fire2/squeeze1x1 of SqueezeDet: 64 -> 16 couts, ReLU,
 * nets/squeezeDet.py:46 via :95-97): pool1's tensor is never written, only the squeeze tensor sq_out [n, Hp, Wp, next_s]
 * (120 MB -> 30 MB at batch 32).  float16, k = 3, cout = 64, even w; w_next_s_packed from sqdet_conv_pack_weights(1, cout,
 * next_s).  Same values as the separate launches (the pooled pixels are rounded to float16 before the squeeze, the
 * squeeze accumulates its two K-chunks in ascending order). */
int sqdet_stem_conv_pool_squeeze_supported(int h, int w, int cout, int k, int conv_pad_mode, int pool_pad_mode, int next_s,
                                           int dtype, int n);
int sqdet_stem_conv_pool_squeeze_fwd(const void* x, const void* w_packed, const float* bias, const void* w_next_s_packed,
                                     const float* b_next_s, void* sq_out, int n, int h, int w, int cout, int k,
                                     int conv_pad_mode, int pool_pad_mode, int next_s, int dtype, sqdet_stream_t stream);

/* ------------------------------------------------------------------ fire --
 * Replaces SqueezeDet._fire_layer (nets/squeezeDet.py:81-106):
 *   sq = relu(conv1x1(x)); y = concat(relu(conv1x1(sq)), relu(conv3x3(sq))).
 * w_* are packed kernels; sq_scratch: [n,h,w,s1x1] scratch in dtype storage. */
int sqdet_fire_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                   const void* w_e3, const float* b_e3, void* sq_scratch, void* y,
                   int n, int h, int w, int cin, int s1x1, int e1x1, int e3x3, int dtype, sqdet_stream_t stream);
/* sqdet_fire_fwd that ALSO leaves the module's squeeze tensor relu(conv1x1(x)) in sq_out [n,h,w,s1x1] (training: the
 * module's backward reads it); the fused kernels write it from their squeeze epilogue -- still one launch. */
int sqdet_fire_fwd_keep(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                        const void* w_e3, const float* b_e3, void* sq_out, void* y, int n, int h, int w,
                        int cin, int s1x1, int e1x1, int e3x3, int dtype, sqdet_stream_t stream);

/* Fire module followed by max_pool 3x3 / stride 2 / SAME (fire3 -> pool3, fire5 -> pool5: nets/squeezeDet.py:49-57)
 * in one launch where the streaming kernel covers the shape: the pool is taken in registers and only the pooled
 * tensor y [n, ceil(h/2), ceil(w/2), e1x1+e3x3] is written.  fire_scratch [n,h,w,e1x1+e3x3] and sq_scratch (as in
 * sqdet_fire_fwd) are only used by the unfused fallback.  Results are bitwise those of fire -> pool. */
int sqdet_fire_maxpool_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                           const void* w_e3, const float* b_e3, void* sq_scratch, void* fire_scratch, void* y,
                           int n, int h, int w, int cin, int s1x1, int e1x1, int e3x3, int dtype, sqdet_stream_t stream);

/* The expand half of a fire module from its squeeze tensor sq_in [n,h,w,s1x1] (produced by a chain launch, below):
 *   y = concat(relu(conv1x1(sq_in, W_e1) + b_e1), relu(conv3x3(sq_in, W_e3) + b_e3))     (nets/squeezeDet.py:92-106),
 * pool != 0: followed by max_pool 3x3 / stride 2 / SAME (fire3 -> pool3, fire5 -> pool5: nets/squeezeDet.py:49-57) taken
 * in registers -- y is then the pooled tensor [n, ceil(h/2), ceil(w/2), e1x1+e3x3] (float16 shapes of the streaming
 * kernel only).  w_e1 / w_e3: packed by sqdet_conv_pack_weights.  Bitwise sqdet_fire_fwd / sqdet_fire_maxpool_fwd.
 * Deep squeezes (SqueezeDet+: 192 / 384 channels, nets/squeezeDetPlus.py:46-73), pool == 0: ONE launch of the 3x3 tile kernel that
 * also runs the expand1x1 on the staged squeeze tile where sqdet_fire_expand_pair_supported says 1 (float16, e1x1 == e3x3 a multiple
 * of 64 >= 128, the halo tile resident in LDS); two conv launches otherwise.  Bitwise the two convs either way. */
int sqdet_fire_expand_pair_supported(int n, int h, int w, int s1x1, int e1x1, int e3x3, int dtype);
int sqdet_fire_expand_fwd(const void* sq_in, const void* w_e1, const float* b_e1, const void* w_e3, const float* b_e3,
                          void* y, int n, int h, int w, int s1x1, int e1x1, int e3x3, int pool, int dtype,
                          sqdet_stream_t stream);

/* A whole fire module from its input x [n,h,w,cin] whose concat tensor is replaced by the NEXT module's squeeze tensor
 * sq_out [n,h,w,next_s1x1] = relu(conv1x1(fire(x), W_next_s) + b_next_s)  (fire2 -> fire3's squeeze, fire4 -> fire5's squeeze:
 * nets/squeezeDet.py:46-53, 81-106): the streaming kernel keeps the module's rounded float16 results in an LDS tile and
 * runs the next squeeze on it.  All kernels packed by sqdet_conv_pack_weights.  Bitwise sqdet_fire_fwd followed by the
 * squeeze conv.  sqdet_fire_squeeze_next_supported: 1 when the shape is covered (float16, SqueezeDet's two pairs). */
int sqdet_fire_squeeze_next_supported(int cin, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype);
int sqdet_fire_squeeze_next_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                                const void* w_e3, const float* b_e3, const void* w_next_s, const float* b_next_s, void* sq_out,
                                int n, int h, int w, int cin, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype,
                                sqdet_stream_t stream);

/* The expand half of a module from its squeeze tensor (+ max_pool 3x3/s2/SAME when pool != 0) whose output is the NEXT
 * module's squeeze tensor sq_out [n, h', w', next_s1x1] (h', w' = pooled dims when pool): fire3+pool3 -> fire4's squeeze,
 * fire4 -> fire5's, fire5+pool5 -> fire6's (nets/squeezeDet.py:49-58).  Bitwise sqdet_fire_expand_fwd + the squeeze conv. */
int sqdet_fire_expand_squeeze_next_supported(int s1x1, int e1x1, int e3x3, int next_s1x1, int pool, int dtype);
int sqdet_fire_expand_squeeze_next_fwd(const void* sq_in, const void* w_e1, const float* b_e1, const void* w_e3, const float* b_e3,
                                       const void* w_next_s, const float* b_next_s, void* sq_out, int n, int h, int w, int s1x1,
                                       int e1x1, int e3x3, int next_s1x1, int pool, int dtype, sqdet_stream_t stream);

/* Fire-module CHAIN (float16): the expand half of one fire module and the squeeze of the NEXT module in one launch.
 * Replaces, for consecutive fire modules on one feature map (fire6 .. fire11, nets/squeezeDet.py:58-69), the pair
 *   e = concat(relu(conv1x1(sq_in, W_e1) + b_e1), relu(conv3x3(sq_in, W_e3) + b_e3))      (nets/squeezeDet.py:92-106)
 *   sq_out = relu(conv1x1(e, W_next_s) + b_next_s)                                          (the next module's :86-90)
 * sq_in: [n,h,w,s1x1] = the module's own squeeze tensor; the concat tensor e [n,h,w,e1x1+e3x3] is written only when
 * y != NULL; sq_out [n,h,w,next_s1x1] only when next_s1x1 > 0 (at least one of the two).  The three kernels travel as
 * ONE packed weight stream (sqdet_fire_chain_pack: float32 HWIO in, any of the three may be NULL = that part of the
 * stream is left as it is); sqdet_fire_chain_stream_bytes returns 0 for shapes the kernel does not cover
 * (float16 only; s1x1 in 8..96 step 8, e1x1 / e3x3 multiples of 64, next_s1x1 in {0,16,32,48} behind a squeeze of up to
 * 32 channels, {0,48,64,96} behind a wider one).  Results are bitwise
 * those of sqdet_fire_fwd followed by the next module's squeeze conv. */
size_t sqdet_fire_chain_stream_bytes(int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype);
int sqdet_fire_chain_pack(const float* w_e1_hwio, const float* w_e3_hwio, const float* w_next_s_hwio, void* stream_buf,
                          int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype, sqdet_stream_t stream);
int sqdet_fire_chain_fwd(const void* sq_in, const void* stream_buf, const float* b_e1, const float* b_e3,
                         const float* b_next_s, void* y, void* sq_out, int n, int h, int w, int s1x1, int e1x1,
                         int e3x3, int next_s1x1, int dtype, sqdet_stream_t stream);

/* ---------------------------------------------------- interpret_output --
 * Replaces ModelSkeleton._add_interpretation_graph (nn_skeleton.py:142-283) +
 * util.safe_exp / bbox_transform / bbox_transform_inv (utils/util.py:167-231).
 * preds: [n,gh,gw,apg*(classes+1+4)] (dtype storage); anchors: float32
 * [gh*gw*apg,4] = float32(mc.ANCHOR_BOX).  Outputs (all [n,A,...], A=gh*gw*apg):
 * det_boxes float32 [n,A,4] (cx,cy,w,h), det_probs float32 [n,A], det_class
 * int64 [n,A]; optional (may be NULL) pred_class_probs float32 [n,A,classes],
 * pred_conf float32 [n,A].  Separate mul/add float32 ops (no FMA contraction). */
int sqdet_interpret_output(const void* preds, const float* anchors, float* det_boxes, float* det_probs,
                           int64_t* det_class, float* pred_class_probs, float* pred_conf,
                           int n, int gh, int gw, int apg, int classes, float img_w, float img_h, float exp_thresh,
                           int dtype, sqdet_stream_t stream);

/* --------------------------------------------------- filter_prediction --
 * Replaces ModelSkeleton.filter_prediction (nn_skeleton.py:696-734) +
 * util.nms / util.batch_iou (utils/util.py:32-76), batched over n images.
 * Inputs: boxes float32 [n,A,4] (cx,cy,w,h), probs float32 [n,A], cls int64 [n,A].
 * top_n > 0 and < A: the top_n highest probs (ties: higher anchor index first)
 *   enter NMS, prob_thresh is ignored (nn_skeleton.py:711-715);
 * otherwise: entries with prob > prob_thresh enter NMS (nn_skeleton.py:716-720).
 * NMS is the reference's NON-greedy rule: j is dropped iff some same-class i
 * with higher prob has (double)IoU(i,j) > nms_thresh.
 * Outputs, capacity max_out per image (>= top_n in the top-N branch), ordered
 * by class then descending prob: out_boxes float32 [n,max_out,4], out_probs
 * float32 [n,max_out], out_cls int32 [n,max_out], out_index int32 [n,max_out]
 * (anchor index), out_count int32 [n] (number of valid rows; if the threshold
 * branch yields more than max_out candidates the call reports it through
 * out_count[i] = -(number of candidates)). */
int sqdet_filter_prediction(const float* boxes, const float* probs, const int64_t* cls,
                            float* out_boxes, float* out_probs, int32_t* out_cls, int32_t* out_index,
                            int32_t* out_count, int n, int num_anchors, int classes, int top_n, int max_out,
                            double nms_thresh, float prob_thresh, sqdet_stream_t stream);

/* interpret_output + filter_prediction in ONE call (two launches: scores chip-wide, then one workgroup per image) for the top-N branch (0 < top_n <= 64 < A <= 20480: every reference
 * config): the scores of all anchors are computed on the fly, boxes and classes are decoded for the <= top_n selected
 * anchors only -- det_boxes / det_class (0.47 MB per image in the reference's sess.run) never exist.  Same float
 * expressions as sqdet_interpret_output, same selection / NMS as sqdet_filter_prediction: identical outputs.
 * scratch_probs: float32 [n, A] device scratch (det_probs; read back only when more than 2048 anchors tie at the top-N
 * boundary).  Outputs as sqdet_filter_prediction. */
int sqdet_detect_filter(const void* preds, const float* anchors, float* scratch_probs, float* out_boxes, float* out_probs,
                        int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh, int gw, int apg, int classes,
                        float img_w, float img_h, float exp_thresh, int top_n, int max_out, double nms_thresh, int dtype,
                        sqdet_stream_t stream);

/* sqdet_detect_filter with the scores already computed (sqdet_convdet_fwd / sqdet_net_set_scores: det_probs float32 [n, A]):
 * the filter launch only.  Identical outputs.  max_workgroups > 0: the n images are walked by at most that many workgroups
 * (<= 0: one per image) -- a serving loop that overlaps this launch with the next batch's forward sizes it to the CUs that
 * forward leaves idle (16 during SqueezeDet's fire6..fire11 launches at batch 32). */
int sqdet_detect_filter_scored(const void* preds, const float* anchors, const float* scores, float* out_boxes, float* out_probs,
                               int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh, int gw, int apg,
                               int classes, float img_w, float img_h, float exp_thresh, int top_n, int max_out,
                               double nms_thresh, int dtype, int max_workgroups, sqdet_stream_t stream);

/* ------------------------------------------------------------ training --
 * Replaces the gradient half of the reference's TF graph for the trainable convs (stride 1,
 * SAME: every conv but the frozen conv1, nets/squeezeDet.py:40-42), the loss graph
 * (ModelSkeleton._add_loss_graph, nn_skeleton.py:285-327, on top of _add_interpretation_graph
 * :142-283) and the train graph (ModelSkeleton._add_train_graph, nn_skeleton.py:329-361).
 * float32 storage (SQDET_F32) is the reference's training dtype; the activation-side kernels also take
 * SQDET_F16 (mixed precision: float16 activations / activation gradients with loss scaling, float32
 * master weights, weight gradients and optimizer -- BASELINE.json configs[4] "fp16 training").
 */

/* MANY kernels packed in ONE launch (a training step re-packs every trainable conv kernel twice after the optimizer step:
 * forward fragment order = sqdet_conv_pack_weights, backward-data order = sqdet_conv_pack_weights_bwd_data; 62 launches in
 * SqueezeDet's step).  sqdet_conv_pack_many_prepare fills a HOST table of sqdet_conv_pack_many_table_bytes(n) bytes from
 * per-item arrays (device pointers of the float32 HWIO kernels and of the packed outputs, k / cin / cout, bwd_data flags)
 * and reports the grid; the caller copies the table to the device once; sqdet_conv_pack_many runs it (same bytes as the
 * per-kernel entry points produce). */
size_t sqdet_conv_pack_many_table_bytes(int nitems);
int sqdet_conv_pack_many_prepare(const float* const* w_hwio_f32, void* const* packed, const int* k, const int* cin, const int* cout,
                                 const int* bwd_data, int nitems, int dtype, void* table_host, int* total_blocks);
/* With the batch norm of _conv_bn_layer convs folded on the way (items whose gamma[i] != NULL): what is packed is
 * W[..., c] * gamma[c] / sqrt(var[c] + eps) -- sqdet_fold_batchnorm followed by the packer, bit for bit, without the folded
 * float32 kernel ever being written -- and b_folded[i] (when not NULL) receives the folded bias.  The pointer arrays may be
 * NULL altogether (= sqdet_conv_pack_many_prepare). */
int sqdet_conv_pack_many_prepare_bn(const float* const* w_hwio_f32, void* const* packed, const int* k, const int* cin,
                                    const int* cout, const int* bwd_data, const float* const* gamma, const float* const* beta,
                                    const float* const* mean, const float* const* var, const float* const* conv_bias,
                                    float* const* b_folded, float eps, int nitems, int dtype, void* table_host, int* total_blocks);
int sqdet_conv_pack_many(const void* table_dev, int nitems, int total_blocks, int dtype, sqdet_stream_t stream);

/* Backward-data: dx = conv(dy, rot180(W)^T).  pack: float32 HWIO [k,k,cin,cout] -> fragment order
 * of the [k,k,cout,cin] kernel (same size as sqdet_conv_packed_bytes(k, cout, cin, dtype)).
 * dy is channels [dy_coffset, +cout) of rows dy_cstride wide (a fire module's concat gradient);
 * accumulate != 0: dx += result (the squeeze tensor receives expand1x1's and expand3x3's dgrad). */
int sqdet_conv_pack_weights_bwd_data(const float* w_hwio_f32, void* packed, int k, int cin, int cout, int dtype,
                                     sqdet_stream_t stream);
int sqdet_conv2d_nhwc_bwd_data(const void* dy, const void* w_packed_bwd, void* dx, int n, int h, int w, int cin,
                               int cout, int k, int dtype, int dy_cstride, int dy_coffset, int accumulate,
                               sqdet_stream_t stream);
/* The same with the ReLU backward of the layer below in the epilogue: dx is the gradient w.r.t. the ReLU output
 * relu_of [n,h,w,cin] (the conv's input in the forward pass) and is zeroed where relu_of <= 0 -- after the
 * accumulation when accumulate != 0 (tf.nn.relu's gradient, nn_skeleton.py:547 through tf.gradients) -- instead of a
 * separate sqdet_relu_bwd pass over the tensor. */
int sqdet_conv2d_nhwc_bwd_data_relu(const void* dy, const void* w_packed_bwd, void* dx, const void* relu_of, int n, int h,
                                    int w, int cin, int cout, int k, int dtype, int dy_cstride, int dy_coffset,
                                    int accumulate, sqdet_stream_t stream);

/* Backward-filter (+ bias): dW[kh,kw,ci,co] = grad_scale * sum_pixels x@tap[ci]*dy[co] (+ weight_decay*W
 * when w_hwio_for_decay != NULL: the gradient of wd*l2_loss(W), nn_skeleton.py:66-69), dbias[co] =
 * grad_scale * sum dy (dbias may be NULL).  x / dy (dtype SQDET_F32 or SQDET_F16) may be channel slices;
 * dW / dbias are float32 either way; grad_scale = 1, or 1/loss_scale in mixed-precision training.
 * workspace: device scratch of sqdet_conv2d_bwd_filter_workspace_bytes(...).  Deterministic (two-pass
 * slab reduction, no atomics). */
size_t sqdet_conv2d_bwd_filter_workspace_bytes(int n, int h, int w, int cin, int cout, int k);
int sqdet_conv2d_nhwc_bwd_filter(const void* x, const void* dy, float* dw_hwio, float* dbias,
                                 const float* w_hwio_for_decay, float weight_decay, float grad_scale, float* workspace,
                                 int n, int h, int w, int cin, int cout, int k, int x_cstride, int x_coffset,
                                 int dy_cstride, int dy_coffset, int dtype, sqdet_stream_t stream);
/* The two halves apart, for a step that takes MANY weight gradients: sqdet_conv2d_nhwc_bwd_filter_partial writes only the
 * partial slabs of one conv into its own workspace (same size query), and ONE sqdet_slab_reduce_many launch at the end of
 * the backward pass sums the slabs of all of them (a SqueezeDet step: 31 reductions, each a 10 us launch behind its
 * gradient kernel).  prepare() fills a host table (sqdet_slab_reduce_many_table_bytes(n_items) bytes; the caller copies it
 * to the device once) from the per-item workspace / dW / dbias (NULL: none) / decay-weight (NULL: none) pointers and the
 * conv shapes the partial launches used; per item the arithmetic is sqdet_conv2d_nhwc_bwd_filter's, bit for bit. */
int sqdet_conv2d_nhwc_bwd_filter_partial(const void* x, const void* dy, float* workspace, int want_bias, int n, int h, int w,
                                         int cin, int cout, int k, int x_cstride, int x_coffset, int dy_cstride,
                                         int dy_coffset, int dtype, sqdet_stream_t stream);
size_t sqdet_slab_reduce_many_table_bytes(int n_items);
int sqdet_slab_reduce_many_prepare(const float* const* workspaces, float* const* dws, float* const* dbiases,
                                   const float* const* w_for_decay, const float* decays, const int* n, const int* h,
                                   const int* w, const int* cin, const int* cout, const int* k, int n_items,
                                   void* table_host, int* total_blocks);
int sqdet_slab_reduce_many(const void* table_dev, int n_items, int total_blocks, float grad_scale, sqdet_stream_t stream);

/* dy *= (y > 0)  (tf.nn.relu gradient; count elements, a multiple of 16 bytes). */
int sqdet_relu_bwd(const void* y, void* dy_inout, size_t count, int dtype, sqdet_stream_t stream);
/* y = x * mask * scale: tf.nn.dropout forward (mask = floor(keep_prob + U) in {0,1}, scale =
 * 1/keep_prob; nets/squeezeDet.py:74) and its backward.  x, mask, y share dtype. */
int sqdet_scale_mask(const void* x, const void* mask, void* y, float scale, size_t count, int dtype,
                     sqdet_stream_t stream);
/* The same with the ReLU backward of the layer below in the same pass: y = relu_of > 0 ? x * mask * scale : 0 (relu_of: the
 * ReLU output x is the gradient of -- fire11's output under the dropout in front of conv12). */
int sqdet_scale_mask_relu(const void* x, const void* mask, const void* relu_of, void* y, float scale, size_t count, int dtype,
                          sqdet_stream_t stream);
/* dst[i] = (dst_dtype)(src[i] * scale): the float16 <-> float32 hand-offs of mixed-precision training (float16
 * preds -> the float32 loss kernel; its float32 dpreds * loss_scale -> float16).  count a multiple of 4. */
int sqdet_convert_scale(const void* src, int src_dtype, void* dst, int dst_dtype, float scale, size_t count,
                        sqdet_stream_t stream);
/* tf.nn.max_pool gradient: dx[cell] = sum of dy over the windows whose first maximum the cell is. */
int sqdet_maxpool_nhwc_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k, int stride,
                           int pad_mode, int dtype, sqdet_stream_t stream);
/* The same for a pool whose input x is a ReLU output (every pool of the reference's nets): dx is also zeroed where
 * x <= 0, i.e. the ReLU backward of the layer below is taken here instead of in a pass of its own. */
int sqdet_maxpool_nhwc_bwd_relu(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k, int stride,
                                int pad_mode, int dtype, sqdet_stream_t stream);
/* Both from the window index of sqdet_maxpool_nhwc_fwd_idx (k = 3, stride 2) instead of x: reads three quarter-size
 * maps (index, dy and -- relu != 0 -- the pooled y, whose sign is the sign of x at every cell that receives anything)
 * and writes dx [n,h,w,c]; bitwise the results of the two functions above. */
int sqdet_maxpool_nhwc_bwd_idx(const unsigned char* window_index, const void* y, const void* dy, void* dx, int n, int h,
                               int w, int c, int k, int stride, int pad_mode, int dtype, int relu, sqdet_stream_t stream);

/* Loss forward + backward.  Inputs as the reference's placeholders (nn_skeleton.py:86-97):
 * input_mask [B,A], box_delta_input [B,A,4], box_input [B,A,4] (cx,cy,w,h), labels [B,A,C];
 * num_objects = sum(input_mask) over the batch.  Outputs: dpreds = d(class+conf+bbox loss)/dpreds
 * [B,gh,gw,K*(C+5)], ious [B,A] (the assign'ed IoU target, no gradient), losses3 = {class_loss,
 * conf_loss, bbox_loss}.  workspace: sqdet_loss_workspace_bytes() of device scratch.
 * global_batch: the divisor of the confidence term's reduce_mean over the batch (nn_skeleton.py:304-312); <= 0 = `batch`.
 * Data-parallel replicas that reproduce ONE graph of batch world*B (num_objects all-reduced, gradients SUMMED) pass
 * world*B here: the class / bbox terms divide by num_objects only, the confidence term also by the batch. */
size_t sqdet_loss_workspace_bytes(void);
int sqdet_loss_fwd_bwd(const float* preds, const float* anchors, const float* input_mask, const float* box_delta_input,
                       const float* box_input, const float* labels, float* dpreds, float* ious, float* losses3,
                       float* workspace, int batch, int gh, int gw, int apg, int classes, float img_w, float img_h,
                       float exp_thresh, float epsilon, float coef_class, float coef_conf_pos, float coef_conf_neg,
                       float coef_bbox, float num_objects, int global_batch, sqdet_stream_t stream);

/* Same, with num_objects read from the DEVICE (float32 scalar, e.g. sqdet_sum_f32 of input_mask -- nn_skeleton.py:180 --
 * optionally SUM-all-reduced over the replicas first: the exact global-batch normalisation of SURVEY.md 8e option b):
 * no device -> host round trip, so the whole step can be captured in a hipGraph. */
int sqdet_loss_fwd_bwd_dev(const float* preds, const float* anchors, const float* input_mask, const float* box_delta_input,
                           const float* box_input, const float* labels, float* dpreds, float* ious, float* losses3,
                           float* workspace, int batch, int gh, int gw, int apg, int classes, float img_w, float img_h,
                           float exp_thresh, float epsilon, float coef_class, float coef_conf_pos, float coef_conf_neg,
                           float coef_bbox, const float* num_objects_dev, int global_batch, sqdet_stream_t stream);

/* Mixed-precision form: preds are FLOAT16 (read as their float32 values) and, beside the float32 dpreds, the loss-scaled
 * float16 gradient the float16 backward starts from is written in the same pass -- (float16)(dpreds * loss_scale), the
 * expression of sqdet_convert_scale -- instead of a conversion launch either side of the loss.  num_objects_dev != NULL:
 * the count is read from the device (num_objects ignored). */
int sqdet_loss_fwd_bwd_mixed(const void* preds_f16, const float* anchors, const float* input_mask, const float* box_delta_input,
                             const float* box_input, const float* labels, float* dpreds, void* dpreds_scaled_f16,
                             float loss_scale, float* ious, float* losses3, float* workspace, int batch, int gh, int gw, int apg,
                             int classes, float img_w, float img_h, float exp_thresh, float epsilon, float coef_class,
                             float coef_conf_pos, float coef_conf_neg, float coef_bbox, float num_objects,
                             const float* num_objects_dev, int global_batch, sqdet_stream_t stream);
/* out[0] = sum(x[0..count)) in a fixed order (deterministic): tf.reduce_sum(self.input_mask), nn_skeleton.py:180. */
int sqdet_sum_f32(const float* x, size_t count, float* out, sqdet_stream_t stream);
/* y = max(a + b, 0): tf.nn.relu(shortcut + branch) (nets/resnet50_convDet.py:55) where the producing conv could not take
 * the add in its epilogue.  count elements, a multiple of 16 bytes; y may alias a or b. */
int sqdet_add_relu(const void* a, const void* b, void* y, size_t count, int dtype, sqdet_stream_t stream);
/* y[p, y_coffset .. +c) = x[p, 0 .. c) for `pixels` rows: one input of tf.concat(values, 3) (nets/squeezeDet.py:106) whose
 * producer could not write its channel range directly.  c, y_cstride, y_coffset multiples of 16 bytes. */
int sqdet_copy_channels(const void* x, void* y, size_t pixels, int c, int y_cstride, int y_coffset, int dtype,
                        sqdet_stream_t stream);
/* mask[i] = floor(keep_prob + u_i), u_i ~ U[0,1) from a counter-based generator of (seed, i): the keep mask of
 * tf.nn.dropout (nets/squeezeDet.py:74), to be applied with sqdet_scale_mask(x, mask, 1/keep_prob). */
int sqdet_dropout_mask(void* mask, size_t count, float keep_prob, uint64_t seed, int dtype, sqdet_stream_t stream);

/* Momentum + per-variable clip_by_norm over flat parameter / gradient / momentum buffers.
 * Variable v = elements [offsets[v], +counts[v]); decays[v] = weight decay added to its gradient
 * BEFORE clipping (0 for biases).  step: g = g*grad_scale (1/world_size after a SUM all-reduce) + decay*w; g *= max_norm/max(||g||,max_norm);
 * accum = momentum*accum + g; w -= lr*accum.  Deterministic (fixed-order norm reduction), so
 * data-parallel replicas stay bit-identical.  found_inf (device int32, may be NULL): set to 1 when any
 * variable's gradient norm is inf / NaN -- the step is then skipped entirely (params and accum untouched;
 * the overflow case of loss-scaled float16 training) -- else 0. */
typedef struct sqdet_optimizer sqdet_optimizer_t;
int sqdet_optimizer_create(sqdet_optimizer_t** out, const long* offsets, const long* counts, const float* decays,
                           int nvars);
void sqdet_optimizer_destroy(sqdet_optimizer_t* opt);
size_t sqdet_optimizer_workspace_bytes(const sqdet_optimizer_t* opt);
int sqdet_optimizer_step(sqdet_optimizer_t* opt, float* params, float* grads, float* accum, void* workspace, float lr,
                         float momentum, float max_grad_norm, float grad_scale, int32_t* found_inf,
                         sqdet_stream_t stream);

/* ------------------------------------------------------------- network --
 * Replaces SqueezeDet.__init__/_add_forward_graph (nets/squeezeDet.py:19-79,
 * nets/squeezeDetPlus.py:19-79) + the sess.run([det_boxes,det_probs,det_class])
 * call shape of demo.py:193-195 / eval.py:75-77.  A net is a host-side plan; its
 * device memory (packed parameters + activation workspace) is caller-owned. */
typedef struct sqdet_net sqdet_net_t;

int sqdet_net_create(sqdet_net_t** out, int arch, int dtype, int batch, int img_h, int img_w, int classes,
                     int anchors_per_grid);
void sqdet_net_destroy(sqdet_net_t* net);

/* Parameters, in graph order, named like the reference's variables
 * ('conv1/kernels', 'fire2/squeeze1x1/biases', ...; nn_skeleton.py:531-536). */
int sqdet_net_num_params(const sqdet_net_t* net);
int sqdet_net_param_info(const sqdet_net_t* net, int index, char* name, size_t name_cap, int shape[4], int* ndim);

size_t sqdet_net_param_bytes(const sqdet_net_t* net);      /* packed kernels + float32 biases */
size_t sqdet_net_workspace_bytes(const sqdet_net_t* net);  /* activation buffers */
int sqdet_net_bind(sqdet_net_t* net, void* param_mem, void* workspace_mem);

/* value: device float32, HWIO for kernels / [cout] for biases.  SQDET_ARCH_RESNET50
 * (nets/resnet50_convDet.py:20-169) also lists '<conv>/gamma', '/beta', '/mean', '/var' for its
 * _conv_bn_layer convs (nn_skeleton.py:427-439); those layers keep the float32 values and fold
 * them into the packed kernel + bias (sqdet_fold_batchnorm) on the next sqdet_net_forward. */
int sqdet_net_set_param(sqdet_net_t* net, const char* name, const float* value_f32, sqdet_stream_t stream);
/* mc.BATCH_NORM_EPSILON (config/config.py:131; default 1e-5). */
int sqdet_net_set_bn_epsilon(sqdet_net_t* net, float eps);

int sqdet_net_output_dims(const sqdet_net_t* net, int* gh, int* gw, int* channels);

/* image_input: [batch,img_h,img_w,3] (dtype storage, BGR mean-subtracted:
 * demo.py:187-190) -> preds [batch,gh,gw,channels] (dtype storage). */
int sqdet_net_forward(sqdet_net_t* net, const void* image_input, void* preds, sqdet_stream_t stream);
/* Binds (NULL: unbinds) a float32 [batch, gh*gw*anchors_per_grid] device buffer: every following sqdet_net_forward also
 * writes interpret_output's det_probs there, from the ConvDet launch's epilogue (see sqdet_convdet_fwd).
 * SQDET_EUNSUPPORTED when the plan's last layer has no score epilogue (sqdet_net_scores_supported). */
int sqdet_net_set_scores(sqdet_net_t* net, float* scores);
int sqdet_net_scores_supported(const sqdet_net_t* net);
/* Serving-loop hook: every following sqdet_net_forward records `hip_event` (a hipEvent_t; NULL: none) on its stream right
 * before the launch of layer `layer_index` -- side work of the PREVIOUS batch (its filter launch, the copy of its rows) that
 * waits for the event on another stream then runs beside the launches behind that point instead of beside the stem.
 * sqdet_net_overlap_layer: the index where side work is cheapest -- the first fire_chain launch (those launches occupy 240
 * of the 256 CUs at batch 32) -- or -1 when the plan has none. */
int sqdet_net_set_signal(sqdet_net_t* net, int layer_index, void* hip_event);
int sqdet_net_overlap_layer(const sqdet_net_t* net);
/* Serving loop without a second stream: the decode + filter of the PREVIOUS batch (what sqdet_detect_filter_scored would
 * launch: same arguments, identical outputs) is handed to the plan and rides in the NEXT sqdet_net_forward as extra "rider"
 * workgroups of its fire_chain launches -- those occupy 240 of the 256 CUs at batch 32, a rider takes an idle CU and one
 * image.  No side stream, no events, no extra launches: stream order alone orders the previous batch's preds / scores
 * (written by its ConvDet launch) before the riders and the riders before the next overwrite.  The out_* rows may be
 * pinned host memory (device-accessible): the rows then need no copy either.  One-shot (consumed by the next forward;
 * preds == NULL cancels).  SQDET_EUNSUPPORTED when the plan cannot carry n images (sqdet_net_rider_capacity: the idle CUs
 * summed over its fire_chain launches; 0 for plans without such launches) -- run sqdet_detect_filter_scored instead. */
int sqdet_net_set_post_job(sqdet_net_t* net, const void* preds, const float* scores, const float* anchors, float* out_boxes,
                           float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh, int gw,
                           int apg, int classes, float img_w, float img_h, float exp_thresh, int top_n, int max_out,
                           double nms_thresh, int dtype);
int sqdet_net_rider_capacity(const sqdet_net_t* net);

/* Layer table for measurement: name, 2*MAC flops and algorithmic bytes (every
 * tensor touched once) of each launch of sqdet_net_forward. */
int sqdet_net_num_layers(const sqdet_net_t* net);
int sqdet_net_layer_info(const sqdet_net_t* net, int index, char* name, size_t name_cap, double* flops,
                         double* bytes);
/* Same as sqdet_net_forward but brackets every launch with HIP events on
 * `stream` and, after synchronising, writes per-launch milliseconds to
 * host_ms[num_layers]. */
int sqdet_net_forward_timed(sqdet_net_t* net, const void* image_input, void* preds, float* host_ms,
                            sqdet_stream_t stream);

/* Live measurement inside the normal forward: after sqdet_net_set_probe(net, i, cap) every
 * sqdet_net_forward records a HIP event pair around layer i's launch on the launch stream
 * (up to cap records); sqdet_net_read_probe synchronises those events, returns the per-launch
 * milliseconds and resets the record count.  layer_index -1 disables the probe. */
int sqdet_net_set_probe(sqdet_net_t* net, int layer_index, int max_records);
int sqdet_net_read_probe(sqdet_net_t* net, float* host_ms, int capacity, int* count);

/* ------------------------------------------------------- training labels --
 * Replaces the per-image Python of imdb.read_batch (dataset/imdb.py:195-239: every ground-truth box, in order,
 * claims the free anchor of highest IoU, or the nearest free anchor when nothing overlaps) and the dense
 * placeholder build of train.py:163-224 (sparse_to_dense).  anchors_f64: [num_anchors,4] float64 =
 * mc.ANCHOR_BOX; gt_boxes_f64: [batch,max_objects,4] (cx,cy,w,h, already scaled to the network input);
 * gt_classes / gt_counts: int32 [batch,max_objects] / [batch].  Outputs (all device, float32, fully
 * written): input_mask [batch,A], box_delta_input / box_input [batch,A,4], labels [batch,A,classes];
 * anchor_index int32 [batch,max_objects] (-1 beyond gt_counts). */
int sqdet_build_labels(const double* anchors_f64, const double* gt_boxes_f64, const int* gt_classes,
                       const int* gt_counts, float* input_mask, float* box_delta_input, float* box_input, float* labels,
                       int* anchor_index, int batch, int num_anchors, int max_objects, int classes,
                       sqdet_stream_t stream);

/* -------------------------------------------------------- pre-processing --
 * Replaces the caller-side image preparation of demo.py:186-190 / imdb.py:101-118:
 *   im = cv2.imread(f).astype(float32); im = cv2.resize(im, (dst_w, dst_h)); input = im - BGR_MEANS
 * src_bgr_u8: device uint8 [n,src_h,src_w,3] (BGR, as cv2.imread delivers) -> dst [n,dst_h,dst_w,3] in
 * dtype storage, ready to be image_input.  Bilinear with cv2 INTER_LINEAR coordinates, float32. */
int sqdet_preprocess_bgr(const uint8_t* src_bgr_u8, void* dst, int n, int src_h, int src_w, int dst_h, int dst_w,
                         float mean_b, float mean_g, float mean_r, int dtype, sqdet_stream_t stream);

/* ------------------------------------------------------------ utilities --
 * Device -> pinned-host copy issued as a KERNEL: dst is host memory mapped into the device's address space
 * (hipHostMalloc); nbytes a multiple of 16.  Used by the serving loop to hand the <= 64 filtered rows per image
 * (the return value of the reference's filter_prediction, nn_skeleton.py:696-734) to the host without a blocking
 * memcpy call.  The data is complete on the host once `stream` has been synchronised. */
int sqdet_copy_to_mapped_host(const void* src_device, void* dst_pinned_host, size_t nbytes, sqdet_stream_t stream);

/* Hardware self-test used by the GPU test-suite: runs one MFMA of each shape
 * the kernels rely on with index-encoded operands and writes the observed
 * (row, col) of every accumulator register to host_out (see csrc/probe.hip). */
int sqdet_probe_mfma_layout(int32_t* host_out, int capacity);

/* Box calibration (bench.py's `box_mfma_tflops` / `box_copy_gbs`; no reference counterpart -- the reference has no
 * device code).  sqdet_calib_mfma enqueues a fixed MFMA microkernel (512 workgroups x 4 waves x iters x 8 independent
 * 16x16x32 float16 MFMAs; scratch: >= 131072 device floats it may overwrite) and returns the flops it performs in
 * *flops; sqdet_calib_copy enqueues a plain 16-byte-per-lane device copy of `bytes` bytes.  The caller times them with
 * events on `stream`. */
int sqdet_calib_mfma(float* scratch, size_t scratch_floats, int iters, double* flops, sqdet_stream_t stream);
int sqdet_calib_copy(const void* src, void* dst, size_t bytes, sqdet_stream_t stream);
/* The MFMA loop in the shapes that settle what the box's ceiling is: shape 0 = mfma_f32_16x16x32_f16 (8 independent accumulators
 * per wave), shape 1 = mfma_f32_32x32x16_f16 (4 independent accumulators; the instruction the guide's 2495 TF/s was measured
 * with); `waves_per_simd` co-resident 256-thread workgroups per CU (grid = CUs x waves_per_simd, returned in *workgroups);
 * zero_operands != 0: all-zero A / B (power management gives clock back on them).  ticks[2 * wg] = shader cycles (s_memtime),
 * ticks[2 * wg + 1] = 100 MHz ticks (s_memrealtime) of workgroup wg's loop: effective clock = 100 MHz x cycles / ticks,
 * cycles per MFMA = cycles / (iters x accumulators).  scratch >= workgroups x 256 floats, ticks >= workgroups x 2. */
int sqdet_calib_mfma2(float* scratch, size_t scratch_floats, unsigned long long* ticks, size_t ticks_count, int iters, int shape,
                      int waves_per_simd, int zero_operands, double* flops, int* workgroups, sqdet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SQDET_H */
