/*
 * oneshotdet_hip.h — C-ABI of liboneshotdet_hip.so: the MI355X (gfx950) kernels of the siamese-FCOS hot path.
 *
 * This is the drop-in boundary.  The reference's native plugin is the pybind11 module
 * `maskrcnn_benchmark._C` (csrc/vision.cpp:7-15) plus the stock ATen ops its Python modules call; every entry
 * point below names the reference interface it replaces (paths relative to /root/reference/maskrcnn_benchmark/).
 *
 * Conventions (all entry points):
 *   - plain C types only: raw DEVICE pointers, ints, floats, an opaque stream handle (hipStream_t passed as void*);
 *   - the CALLER allocates every output and workspace; the library never allocates, frees or synchronises;
 *   - every call is asynchronous on `stream`; no host synchronisation anywhere (NMS returns a device-side count);
 *   - return value 0 = OK; negative = error (OSD_ERR_*); osd_last_error_string() describes the last failure of the
 *     calling thread;
 *   - activations are NHWC ("channels last"), element type selected by `dtype`: OSD_F32 or OSD_BF16
 *     (accumulation is always fp32); weights are packed by osd_pack_* into K-contiguous [Cout][R][S][Cin] rows.
 */
#ifndef ONESHOTDET_HIP_H
#define ONESHOTDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSD_F32 0
#define OSD_BF16 1

#define OSD_OK 0
#define OSD_ERR_INVALID_ARG (-1)
#define OSD_ERR_UNSUPPORTED (-2)
#define OSD_ERR_LAUNCH (-3)
#define OSD_ERR_WORKSPACE (-4)

/* epilogue activation */
#define OSD_ACT_NONE 0
#define OSD_ACT_RELU 1
#define OSD_ACT_EXP_SCALE 2 /* y = exp(act_scale * x): FCOSHead bbox_reg, modeling/rpn/fcos/fcos.py:95-97 */

/* residual operand of the conv epilogue */
#define OSD_RES_NONE 0
#define OSD_RES_SAME 1   /* y += res[n,ho,wo,c]            Bottleneck `out += identity`, modeling/backbone/resnet.py:312 */
#define OSD_RES_UP2X 2   /* y += res[n,ho/2,wo/2,c]        FPN top-down nearest 2x + add, modeling/backbone/fpn.py:59-64 */
#define OSD_RES_DOWN2X 3 /* y += res[n,2ho,2wo,c]          the identity of a bottleneck computed on every other pixel only: layer1's last block,
                            whose output nothing but layer2.0's stride-2 1x1 convs reads (resnet.py:295-315, 138-145); res_h >= 2 ho - 1, res_w >= 2 wo - 1
                            (osd_conv2d_fwd_grouped: exactly twice the output size) */

const char* osd_last_error_string(void);
int osd_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on MFMA).  Replaces layers.Conv2d / nn.Conv2d (ATen conv2d) + FrozenBatchNorm2d
 * (layers/batch_norm.py:19-24, folded into weight/bias by osd_pack_conv_weight) + relu_ + residual add as used by
 * modeling/backbone/resnet.py:295-315,332-337, modeling/backbone/fpn.py:51-75,95-99 and modeling/rpn/fcos/fcos.py:89-97.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct osd_conv_desc {
  int32_t dtype;          /* OSD_F32 | OSD_BF16: element type of x, w, res, y */
  int32_t n, h, w;        /* input batch and spatial size (as stored) */
  int32_t cin;            /* K elements per filter tap (multiple of 16 (f32) / 32 (bf16)) */
  int32_t in_stride_n, in_stride_h, in_stride_w; /* input strides in ELEMENTS (in_stride_w = cin for dense NHWC) */
  int32_t ho, wo, cout;   /* output spatial size and channels (cout multiple of 4) */
  int32_t r, s;           /* filter taps */
  int32_t stride_h, stride_w, pad_h, pad_w;
  int32_t w_rows;         /* rows in the packed weight matrix (>= cout, zero padded) */
  int32_t out_stride;     /* elements between consecutive output pixels (>= cout) */
  int32_t res_mode;       /* OSD_RES_* */
  int32_t res_h, res_w, res_stride; /* residual spatial size and pixel stride in elements */
  int32_t act;            /* OSD_ACT_* */
  float act_scale;
  int32_t relu_in;        /* 1: apply ReLU to x while staging (P7 = conv(relu(P6)), fpn.py:98) */
  int32_t algo;           /* 0 = library heuristic; otherwise 1 + impl*32 + variant*8 + tile (see osd_conv_algo_count /
                             DESIGN.md 4.1): lets the host autotune per layer shape by measurement.  Ids are only
                             meaningful for ONE osd_abi_version(): 41 (the persistent pointwise kernel of ABI 3's first
                             builds) is retired and returns OSD_ERR_UNSUPPORTED; tile 5 / tile 6 variant 0 name other kernels
                             than they did before ABI 4 — a cached id must be stored with the ABI version it was tuned under.
                             New in ABI 4: 57 - 60 = deep-ring small tiles (64 x 32 x 8 stages, 64 x 64 x 5, 64 x 64 x 8, 32 x 64 x 8;
                             bf16, cin in 64s: the latency-sized launches), 51 = the prediction convs' patch kernel (bf16, 3x3 / 1 / 1,
                             cout <= 4, no residual / mask) */
  int32_t reserved0;      /* (keeps the pointer below 8-byte aligned; set to 0) */
  void* ordered_ws;       /* weight-gradient entries only (ABI 3): NULL = partial tiles are added with fp32 atomics; otherwise
                             a caller-owned scratch buffer of ordered_ws_bytes bytes — the launch STORES its partial tiles
                             there and sums them in a fixed order (bit-reproducible dW).  Passed per call: the library keeps
                             no state between calls.  osd_conv2d_wgrad_mixed reads it from descs[0]. */
  int64_t ordered_ws_bytes;
} osd_conv_desc;

/* second pixel source of a 1x1 convolution (osd_conv2d_fwd): dense NHWC [n][h][w][cin2]; output pixel (ho, wo) reads
 * x[n][ho * stride][wo * stride].  cin and cin2 multiples of 64.  w2 (nullable): the second part's own packed weights
 * [w_rows][cin2] (osd_pack_conv_weight of its conv) — the `w` argument then holds the first part only, so two separately
 * packed (and separately trained) convs run as one GEMM; NULL: `w` holds both parts side by side, K = cin + cin2. */
typedef struct osd_conv_src2 {
  const void* x;
  const void* w2;
  int32_t cin2, h, w, stride;
} osd_conv_src2;

/* number of selectable algorithms for osd_conv_desc.algo (valid values 1..count); unsupported combinations for a
 * given shape return OSD_ERR_UNSUPPORTED */
int osd_conv_algo_count(void);
/* mask (optional, same geometry as y, pixel stride = out_stride): y = mask > 0 ? y : 0 after the residual add — the
 * ReLU backward of the layer whose forward output is `mask`, fused into the data-gradient convolution.
 * act_scale_dev (optional device scalar) overrides d->act_scale for OSD_ACT_EXP_SCALE: the learnable Scale of fcos.py:81
 * read on the device, so training needs no host round trip.
 * src2 (nullable): a SECOND pixel source of a 1x1 conv — y = act(W[:, :cin] x + W[:, cin:cin+cin2] x2 + bias ...), the
 * weights packed as one 1x1 conv over cin + cin2 input channels.  This is conv3 + downsample of a bottleneck's first block
 * (resnet.py:295-315: out = bn3(conv3(out)); identity = downsample(x); out += identity) as ONE GEMM over the concatenated
 * K: the downsample output is never stored and never re-read as a residual. */
int osd_conv2d_fwd(const osd_conv_desc* d, const void* x, const void* w, const float* bias, const void* res,
                   const void* mask, const float* act_scale_dev, const osd_conv_src2* src2, void* y, void* stream);

/* The same convolution applied to n_seg <= OSD_CONV_MAX_SEG dense NHWC tensors of different batch / spatial size in ONE
 * launch: the FPN levels that share an FCOS tower or prediction conv (fcos.py:83-99 loops `for l, feature in enumerate(x)`
 * over the same modules).  d gives dtype, cin, cout, r, s, strides, pads, w_rows, out_stride, res_mode (NONE, SAME, DOWN2X from a map of exactly twice the size, or UP2X
 * with an addend of exactly half the output size), res_stride, act, act_scale, algo (LDS-DMA algorithms 1..32, and 51);
 * xs / ys / residuals / masks / act_scale_devs are HOST arrays of n_seg device pointers (the last three nullable as a
 * whole; act_scale_devs[l] = the level's Scale), ns / hs / ws HOST arrays with each tensor's batch, height and width. */
#define OSD_CONV_MAX_SEG 12
int osd_conv2d_fwd_grouped(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys,
                           const void* const* residuals, const void* const* masks, const float* const* act_scale_devs,
                           const int32_t* ns, const int32_t* hs, const int32_t* ws, const void* w, const float* bias,
                           void* stream);

/* The general form: every pair also has its OWN packed weights and bias (wts / biases: HOST arrays of n_seg device
 * pointers; all of geometry d).  Convs that walk the same graph with different parameters go out as one launch: the cls
 * and the bbox tower of FCOSHead (fcos.py:27-49, 2 x 5 levels), and every layer of the query backbone beside the same layer
 * of the target backbone (generalized_rcnn.py:69-71,270-272: SIAMESE_BACKBONE = two separately parameterised R-50-FPN),
 * whose latency-sized launches (M = 8..8192 pixels) thereby disappear into the target's. */
int osd_conv2d_fwd_multi(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys,
                         const void* const* residuals, const void* const* masks, const float* const* act_scale_devs,
                         const int32_t* ns, const int32_t* hs, const int32_t* ws, const void* const* wts,
                         const float* const* biases, void* stream);

/* osd_conv2d_fwd_multi (no residual, mask or activation) whose epilogue also gathers GroupNorm statistics from the values it
 * stores (rounded to the conv's dtype: what a separate pass over the stored tensor would read), added by fp32 atomics into
 * buffers the caller has zeroed (the slab index is picked per workgroup to spread them).  Requirements, stated once: the
 * software-pipelined 3x3 kernel only (d->algo = 15: bf16, 3x3 / stride 1 / pad 1, map widths 64 / 128 / 256), and only pairs
 * made of whole 256-pixel tiles whose images are whole 128-pixel runs; anything else returns OSD_ERR_UNSUPPORTED and the caller
 * uses the plain pair of calls (conv, then the statistics pass).  gn_n = images per pair, gn_groups = GroupNorm groups.
 *
 * BACKWARD statistics — pair i with gn_us[i] != NULL: the conv is the data-gradient conv whose outputs dt are the gradients
 * w.r.t. the outputs of a GroupNorm + ReLU (the FCOS towers, fcos.py:29-39).  With z = a u + b, dz = z > 0 ? dt : 0 and
 * xhat = xa u + xb per pixel and channel,
 *   gn_wss[i] [gn_n][OSD_GN_SPLITS][gn_groups][2] += sum dz * gamma, sum dz * gamma * xhat     (per image, slab, group)
 *   gn_pws[i] [gn_n][OSD_GN_SPLITS][2][cout]      += sum dz * xhat, sum dz                       (d gamma / d beta partials)
 * i.e. the two halves of the ws of osd_groupnorm_relu_bwd_levels_fused, which then skips its own statistics pass for that
 * level.  gn_us[i]: the GroupNorm's INPUT (the forward conv's output) at this conv's output pixels, dtype and row stride of
 * ys[i]; gn_abs[i]: level i's [4][gn_n][cout] block (a, b, xa, xb) of osd_groupnorm_relu_fwd_levels' ab; gn_gammas[i]: [cout].
 * All of gn_us[i], gn_abs[i], gn_gammas[i], gn_wss[i], gn_pws[i] must be non-NULL for such a pair.
 *
 * FORWARD statistics — pair i with gn_us[i] == NULL (or gn_us == NULL) and gn_wss[i] != NULL: the conv is the tower conv whose
 * output feeds the GroupNorm;
 *   gn_wss[i] [gn_n][OSD_GN_SPLITS][gn_groups][2] += sum y, sum y * y                             (per image, slab, group)
 * = level i's part of the ws of osd_groupnorm_relu_fwd_levels_fused.  gn_abs, gn_gammas and gn_pws are not read (may be NULL).
 *
 * A pair with gn_wss[i] == NULL gathers nothing.  gn_wss itself must not be NULL. */
int osd_conv2d_fwd_multi_gn(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys, const int32_t* ns,
                            const int32_t* hs, const int32_t* ws, const void* const* wts, const float* const* biases,
                            const void* const* gn_us, const float* const* gn_abs, const float* const* gn_gammas,
                            float* const* gn_wss, float* const* gn_pws, int gn_n, int gn_groups, void* stream);

/* OIHW fp32 conv weight (+ optional per-Cout scale = FrozenBN weight*rsqrt(var), layers/batch_norm.py:20) ->
 * packed [w_rows][r][s][cin_pad] rows of `dtype`, zero padded. */
int osd_pack_conv_weight(const float* w_oihw, const float* scale, void* dst, int cout, int cin, int r, int s,
                         int w_rows, int cin_pad, int dtype, void* stream);
/* 7x7 stem weight [64][3][7][7] (BaseStem conv1, resnet.py:323-325) -> [w_rows][7][32] with (s, c) at s*4+c. */
/* same with the source in [cout][r][s][cin] order (src_orsi != 0): the layout the training engine keeps its fp32
 * master weights in, so the optimiser, the weight gradient and the packers share one layout */
int osd_pack_conv_weight_ex(const float* w, const float* scale, void* dst, int cout, int cin, int r, int s, int w_rows,
                            int cin_pad, int src_orsi, int dtype, void* stream);
/* Repack MANY convs in one launch (training: every step after the optimiser).  table: device array of
 * struct { int64 src_off, dst_off, scale_off; int32 cout, cin, r, s, rows, kpad, first_block, n_blocks; } (56 bytes),
 * src = flat fp32 masters in [cout][r][s][cin] order, scales = flat per-Cout factors (scale_off -1: none), dst = flat
 * packed buffer of `dtype`; block_entry[b] = table index served by workgroup b; dgrad != 0 writes the flipped /
 * transposed data-gradient form ([rows >= cin][r][s][kpad >= cout]). */
int osd_pack_multi(const void* table, const int32_t* block_entry, int n_blocks, const float* src, const float* scales,
                   void* dst, int dgrad, int dtype, void* stream);
int osd_pack_stem_weight(const float* w_oihw, const float* scale, void* dst, int cout, int w_rows, int dtype,
                         void* stream);
/* NCHW fp32 image batch -> zero-padded NHWC4 [n][hp][wp][4] of `dtype` with the image at (pad_t, pad_l). Replaces the
 * implicit zero padding of the stem conv (padding=3) and the layout change. */
int osd_pack_image(const float* src_nchw, void* dst, int n, int h, int w, int hp, int wp, int pad_t, int pad_l,
                   int dtype, void* stream);
/* NHWC `dtype` [n][h][w][c] (pixel stride `stride`, first channel c0, c channels) -> NCHW fp32 (the reference's
 * layout contract) */
int osd_nhwc_to_nchw_f32(const void* src, float* dst, int n, int h, int w, int c, int stride, int c0, int dtype,
                         void* stream);
/* NCHW fp32 -> dense NHWC `dtype` */
int osd_nchw_f32_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int dtype, void* stream);

/* F.max_pool2d(kernel 3, stride 2, pad 1) of the stem, modeling/backbone/resnet.py:336.  NHWC. */
int osd_maxpool3x3s2_fwd(const void* x, void* y, int n, int h, int w, int c, int ho, int wo, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GroupNorm(32, C) + ReLU of the FCOS towers (nn.GroupNorm, modeling/rpn/fcos/fcos.py:37-38,46-47).
 * stats: per (image, group) partial sum and sum of squares over an [n][hw][c] tensor ->
 * ws[n][OSD_GN_SPLITS][groups][2] fp32 (deterministic two-stage reduction, no atomics); finalize: a[n][c] = gamma*rstd, b[n][c] = beta - mean*gamma*rstd; apply: y = relu(x*a + b).
 * ---------------------------------------------------------------------------------------------------------------- */
#define OSD_GN_SPLITS 64
int osd_groupnorm_stats(const void* x, float* ws, int n, int hw, int c, int groups, int dtype, void* stream);
int osd_groupnorm_finalize(const float* ws, const float* gamma, const float* beta, float* a, float* b, int n, int hw,
                           int c, int groups, float eps, void* stream);
int osd_groupnorm_relu_apply(const void* x, const float* a, const float* b, void* y, int n, int hw, int c, int dtype,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * ROIAlign forward.  Replaces _C.roi_align_forward (csrc/ROIAlign.h:11-25; csrc/cuda/ROIAlign_cuda.cu:65-122,257-290).
 * x: NHWC [b][h][w][c]; rois: [r][5] fp32 (batch_idx, x1, y1, x2, y2); y: [r][ph][pw][c] fp32.
 * ---------------------------------------------------------------------------------------------------------------- */
int osd_roialign_fwd(const void* x, const float* rois, float* y, int b, int h, int w, int c, int num_rois,
                     float spatial_scale, int ph, int pw, int sampling_ratio, int dtype, void* stream);
/* batch_pooling: mean over the `shots` queries of each target image (generalized_rcnn.py:100-104).
 * x [b*shots][c] fp32 -> y [b][c] fp32 */
int osd_shot_mean(const float* x, float* y, int b, int shots, int c, void* stream);
/* The two above for ALL FPN levels of the query branch in one launch: SuppAlignLayer's 1 x 1 ROIAlign of every query's whole-image
 * box (generalized_rcnn.py:20-52, 257; csrc/cuda/ROIAlign_cuda.cu:65-122) followed by batch_pooling (:100-104).  xs[l]: NHWC
 * [batch * shots][hs[l]][ws[l]][c] `dtype`; rois [batch * shots][5] fp32, ROI b * shots + k = shot k of target image b; ys[l] [batch][c]
 * fp32.  xs / ys / hs / ws / scales are HOST arrays of n_levels <= 8 entries.  The arithmetic of osd_roialign_fwd (ph = pw = 1) +
 * osd_shot_mean per level (equal to 1e-5; the forward goldens' pooled vectors pin it). */
int osd_query_pool_levels(int n_levels, const void* const* xs, const int32_t* hs, const int32_t* ws, const float* scales,
                          const float* rois, int batch, int shots, int c, int sampling_ratio, float* const* ys, int dtype,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Query<->target correlation: y[n,h,w,c] = x[n,h,w,c] * q[n,c] (depthwise cross-correlation with a 1x1 query kernel),
 * generalized_rcnn.py:307-311.  x, y NHWC `dtype`; q [n][c] fp32.
 * ---------------------------------------------------------------------------------------------------------------- */
int osd_correlate_fwd(const void* x, const float* q, void* y, int n, int hw, int c, int dtype, void* stream);
/* The same for all FPN levels in ONE launch (the reference's list comprehension over levels, generalized_rcnn.py:307-311;
 * also the backward d_feat_l = g_l * q_l): xs / qs / ys HOST arrays of n_levels <= 6 device pointers ([n][hw_l][c], [n][c],
 * [n][hw_l][c]), hws HOST array. */
int osd_correlate_levels(int n_levels, const void* const* xs, const float* const* qs, void* const* ys, const int32_t* hws,
                         int n, int c, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Proposal pipeline (FCOSPostProcessor, modeling/rpn/fcos/inference.py:46-137,251-323) and NMS (_C.nms,
 * csrc/nms.h:10-28, csrc/cuda/nms.cu:23-131), batched over images, no host sync.
 * ---------------------------------------------------------------------------------------------------------------- */
/* score = sigmoid(cls)*sigmoid(ctr), decode + clip for one FPN level.  cls_ctr: [n][hw][cc_stride] (logit at +0,
 * centerness at +1), reg: [n][hw][reg_stride] (l,t,r,b).  Writes scores[n][total_locs] and boxes[n][total_locs][4]
 * at location offset `loc_offset` (fp32).  inference.py:53-79,104-117; locations fcos.py:220-234. */
int osd_fcos_score_decode(const void* cls_ctr, const void* reg, float* scores, float* boxes, int n, int h, int w,
                          int cc_stride, int reg_stride, int stride, int loc_offset, int total_locs, float img_h,
                          float img_w, int dtype, void* stream);
/* The same for a batch padded by to_image_list (structures/image_list.py:52-70; BatchCollator, data/collate_batch.py:15-20):
 * img_hw [n][2] fp32 on the device = every image's true (height, width), which is what clip_to_image uses
 * (fcos/inference.py:111-113); NULL = img_h / img_w for all images. */
int osd_fcos_score_decode_sizes(const void* cls_ctr, const void* reg, float* scores, float* boxes, int n, int h, int w,
                                int cc_stride, int reg_stride, int stride, int loc_offset, int total_locs, float img_h,
                                float img_w, const float* img_hw, int dtype, void* stream);
/* Per-level top-k of inference.py:97-102 (exact, by rank): within keys[img][lo .. lo+cnt) keep the `topn` largest
 * (ties: lower index first), write key -1 for the rest.  keys_in/keys_out: [n][total] fp32 (may alias). */
int osd_level_topk(const float* keys_in, float* keys_out, int n, int total, int lo, int cnt, int topn, void* stream);
/* Per-level top-k (inference.py:97-102) fused with the descending sort that NMS needs (csrc/cuda/nms.cu:73-75): the
 * candidates of each image are split into `n_levels` consecutive segments (level_lo/level_cnt, HOST arrays); within a
 * segment only the `topn` best (score desc, index asc) survive; survivors of all segments are ordered by (score desc,
 * index asc) and their boxes gathered: boxes_sorted [n][max_count][4], scores_sorted [n][max_count], idx_sorted
 * [n][max_count] (original index), counts [n] (written by the call).  n_levels = 0: one segment, no cut. */
int osd_rank_sort_gather(const float* keys, const float* boxes, int n, int total, int max_count,
                         const int32_t* level_lo, const int32_t* level_cnt, int n_levels, int topn,
                         float* boxes_sorted, float* scores_sorted, int32_t* idx_sorted, int32_t* counts, void* stream);
/* Greedy NMS over boxes ALREADY SORTED by descending score, per image, stopping after max_keep survivors
 * (= boxlist_nms + the post-NMS top-n of inference.py:316-321).  mask_ws: osd_nms_workspace_bytes(n, max_count)
 * bytes.  out_boxes [n][max_keep][4], out_scores [n][max_keep], out_pos [n][max_keep] (position in the sorted
 * list), out_count [n].  Suppression rule: IoU > thresh when cuda_semantics != 0 (csrc/cuda/nms.cu:60) else
 * IoU >= thresh (csrc/cpu/nms_cpu.cpp:60); "+1" areas as in both. */
int osd_nms_sorted(const float* boxes_sorted, const float* scores_sorted, const int32_t* counts, int n, int max_count,
                   float thresh, int cuda_semantics, int max_keep, uint64_t* mask_ws, float* out_boxes,
                   float* out_scores, int32_t* out_pos, int32_t* out_count, void* stream);
int64_t osd_nms_workspace_bytes(int n, int max_count);
/* _C.nms as one entry point (csrc/vision.cpp:8, csrc/nms.h:10-28; layers/nms.py:5): dets [num_boxes][4] fp32 xyxy, scores
 * [num_boxes] fp32 (any finite values, ties broken by lower index) -> keep_out [num_boxes] int64 = the ORIGINAL indices of
 * the kept boxes in ascending order (csrc/cuda/nms.cu:127-130, csrc/cpu/nms_cpu.cpp:64), the first *count_out of them
 * valid; count_out is a DEVICE int32 (the reference returns a tensor sized on the host after a blocking copy,
 * nms.cu:94-100; here the caller reads the count when it needs it).  Rule as osd_nms_sorted.  num_boxes == 0 -> count 0.
 * workspace: osd_nms_single_workspace_bytes(num_boxes) bytes. */
int osd_nms(const float* dets, const float* scores, int num_boxes, float thresh, int cuda_semantics, void* workspace,
            int64_t* keep_out, int32_t* count_out, void* stream);
int64_t osd_nms_single_workspace_bytes(int num_boxes);
/* osd_rank_sort_gather + osd_nms_sorted in one call that sorts only the HEAD of the order: greedy NMS stopping at max_keep
 * survivors almost never reads past the first 1.25 * max_keep candidates, while ranking all `total` candidates against
 * each other is O(total^2).  A per-image score threshold (two-level histogram) selects at least 1.25 * max_keep + 320
 * candidates, only those are ranked (exactly: whatever precedes a selected candidate is selected too), NMS runs on them,
 * and images that could not fill max_keep from the head are redone on the full order (launched unconditionally, exits at
 * once otherwise: no host round trip).  Same outputs as the two separate calls: out_boxes [n][max_keep][4], out_scores
 * [n][max_keep] (descending), out_count [n].  workspace: osd_proposals_workspace_bytes(n, total, max_count, max_keep). */
int osd_proposals_sort_nms(const float* keys, const float* boxes, int n, int total, int max_count, const int32_t* level_lo,
                           const int32_t* level_cnt, int n_levels, int topn, float thresh, int cuda_semantics, int max_keep,
                           void* workspace, float* out_boxes, float* out_scores, int32_t* out_count, void* stream);
/* The same with the size of the exactly sorted head under the caller's control: head_hint (0 = the default above) is the
 * number of candidates phase 1 may read; depth_out [n] (nullable, device) receives how deep the greedy scan actually read
 * (position of the last survivor + 1).  A trained head puts neighbouring locations on the same object, NMS suppresses
 * more, and the scan needs 1.3-1.4 x max_keep candidates instead of 1.0: callers feed depth_out of an earlier step back
 * (without synchronising: a lagged, non-blocking read) so that phase 2 stays the exception.  Exact for every hint. */
int osd_proposals_sort_nms_hint(const float* keys, const float* boxes, int n, int total, int max_count,
                                const int32_t* level_lo, const int32_t* level_cnt, int n_levels, int topn, float thresh,
                                int cuda_semantics, int max_keep, int head_hint, void* workspace, float* out_boxes,
                                float* out_scores, int32_t* out_count, int32_t* depth_out, void* stream);
int64_t osd_proposals_workspace_bytes(int n, int total, int max_count, int max_keep);

/* ------------------------------------------------------------------------------------------------------------------
 * Sigmoid focal loss.  Replaces _C.sigmoid_focalloss_forward/backward (csrc/SigmoidFocalLoss.h;
 * csrc/cuda/SigmoidFocalLoss_cuda.cu:21-58,62-101).  logits [m][classes] fp32, targets [m] int32.
 * ---------------------------------------------------------------------------------------------------------------- */
int osd_sigmoid_focal_fwd(const float* logits, const int32_t* targets, float* losses, int m, int classes, float gamma,
                          float alpha, void* stream);
int osd_sigmoid_focal_bwd(const float* logits, const int32_t* targets, const float* d_losses, float* d_logits, int m,
                          int classes, float gamma, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward pass (BASELINE.json configs[2..4]: forward + backward).  In the reference these are autograd's ATen
 * kernels (conv dgrad/wgrad, group_norm backward, ...), _C.roi_align_backward (csrc/ROIAlign.h:27-45,
 * csrc/cuda/ROIAlign_cuda.cu:178-254) and the FCOS loss of modeling/rpn/fcos/loss.py:213-276.
 * Data gradient of a stride-1 conv = osd_conv2d_fwd on dY with osd_pack_conv_weight_dgrad weights (pad' = r-1-pad),
 * its `mask` epilogue applying the producer's ReLU backward; stride-2 1x1 = that on the small grid + osd_scatter2x.
 * ---------------------------------------------------------------------------------------------------------------- */
/* OIHW fp32 (+ FrozenBN scale) -> [rows >= cin][r][s][cout_pad] with taps flipped: Wd[ci][r'][s'][co] = w[co][ci][R-1-r'][S-1-s'] */
int osd_pack_conv_weight_dgrad(const float* w, const float* scale, void* dst, int cout, int cin, int r, int s, int rows,
                               int cout_pad, int src_orsi, int dtype, void* stream);
/* Ordered mode of the weight-gradient launches (osd_conv2d_wgrad and its _grouped / _batched / _mixed / _multi forms):
 * with osd_conv_desc.ordered_ws set, every workgroup stores its partial tile into that caller-owned scratch buffer and a
 * second launch sums the tiles in a fixed order — dW (and db) come out bit-identical from run to run, and the partial tiles
 * travel as plain stores instead of fp32 atomics performed at the memory side.  With NULL (the default) the partial tiles
 * are added with atomics (ATen's conv backward makes no ordering promise either: engine/trainer.py:92).  A launch that
 * needs more than ordered_ws_bytes (workgroups x (tile + tile rows) x 4; 1 GiB covers every launch of the training step)
 * fails with OSD_ERR_WORKSPACE.  The buffer must not be shared by launches that run concurrently. */
/* dW[cout][r][s][cin] (fp32, ACCUMULATED with atomics: zero it first) += sum over pixels dy[m][co] * x[m@tap][ci].
 * d describes the FORWARD conv (x geometry, strides, pads, cout, out_stride = pixel stride of dy); scale (nullable) is a
 * per-Cout factor applied to the contribution (the folded FrozenBN scale); db (nullable, fp32 [cout], accumulated) also
 * receives the bias gradient sum_m dy[m][co] from the same pass over dy.
 * d->algo: 0 = default, else 1 + variant + 8 * split_target_code.  Variants 0..3: 128 x 128 channel tile, pixels per
 * stage x ring depth 32x3 / 64x2 / 32x4 / 64x3; variant 4: 256 x 256 channel tile on 8 waves (bf16).  Split targets
 * 512, 256, 128, 64, 1024, 768, 1536, 2048 workgroups (x 8 for fp32). */
int osd_conv2d_wgrad(const osd_conv_desc* d, const void* x, const void* dy, const float* scale, float* dw, float* db,
                     void* stream);
/* the same over n_seg <= 24 (x, dy) pairs that share the weights (the FPN levels of the FCOS towers): d gives the conv
 * geometry, ns/hs/ws (HOST arrays) the batch and input size of each pair; one launch, the atomic traffic into dW is paid once */
int osd_conv2d_wgrad_grouped(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                             const int32_t* ns, const int32_t* hs, const int32_t* ws, const float* scale, float* dw,
                             float* db, void* stream);
/* The FCOS prediction convs (cls_logits + centerness fused, bbox_pred: fcos.py:50-61; 3x3 / stride 1 / pad 1, cout <= 4,
 * cin a multiple of 256) over the FPN levels: a read-once kernel (one pass over x, the nine dy vectors of every pixel) with
 * per-workgroup partials folded by a second launch (16 adders per address instead of 500), instead of a 128-channel MFMA
 * tile spent on 2-4 channels.  dw [cout][3][3][cin] and db [cout] (nullable) are ACCUMULATED.  workspace:
 * osd_conv2d_wgrad_pred_workspace_bytes(n_seg, ns, hs, ws, cin) bytes. */
int osd_conv2d_wgrad_pred(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys, const int32_t* ns,
                          const int32_t* hs, const int32_t* ws, float* dw, float* db, void* workspace, void* stream);
int64_t osd_conv2d_wgrad_pred_workspace_bytes(int n_seg, const int32_t* ns, const int32_t* hs, const int32_t* ws, int cin);
/* The same convs' DATA gradient as a GEMM (round 5; autograd of fcos.py:50-61,91-97 in the reference): with the nine shifted dy vectors
 * of every input pixel side by side,
 *   G[q][tap * 4 + co] = dy[q - (tap / 3 - 1, tap % 3 - 1)][co]      (64 columns per pixel, 36 used; zero outside the map),
 * d x[q][ci] = sum over (tap, co) of G[q][tap * 4 + co] * w[co][tap][ci]: a 1x1 conv with K = 64 over G.
 *   osd_pred_dy_gather: d = the prediction conv's descriptor (dtype, cout <= 4, out_stride = channels per dy pixel); g receives
 *     [sum of n * h * w over the levels][64] elements of d->dtype, the levels one after the other;
 *   osd_pred_dgrad_pack: wd[cin][64] (dtype) <- the fp32 master w[cout][3][3][cin]: the packed weights of that 1x1 conv
 *     (osd_conv2d_fwd with cin = 64, cout = the prediction conv's cin, w_rows = cin, a zero bias);
 *   osd_conv2d_wgrad_pred_gathered: osd_conv2d_wgrad_pred from the caller's G (the gather is shared with the data gradient);
 *     workspace: 256 + 64 * cin * 4 bytes. */
int osd_pred_dy_gather(const osd_conv_desc* d, int n_seg, const void* const* dys, const int32_t* ns, const int32_t* hs,
                       const int32_t* ws, void* g, void* stream);
int osd_pred_dgrad_pack(int dtype, const float* w, int cout, int cin, void* wd, void* stream);
int osd_conv2d_wgrad_pred_gathered(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* g, const int32_t* ns,
                                   const int32_t* hs, const int32_t* ws, float* dw, float* db, void* workspace, void* stream);
/* n_seg <= 24 convs of IDENTICAL geometry (d: the shared forward descriptor incl. n, h, w) but different tensors AND
 * different weights — the repeated bottleneck blocks of a ResNet stage (resnet.py:295-315) — in one launch; xs / dys /
 * scales / dws / dbs: HOST arrays of per-conv device pointers (scales, dbs nullable as a whole or per entry).  The output
 * tiles of all convs share the workgroup budget: each needs 1/n_seg of the pixel splits, and of the atomic traffic, that a
 * launch of its own would. */
int osd_conv2d_wgrad_batched(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                             const float* const* scales, float* const* dws, float* const* dbs, void* stream);
/* The general form: n_seg <= 24 (x, dy) pairs, each with its own batch / height / width (ns, hs, ws) AND its own
 * dW / scale / db (pairs that share a weight repeat its pointers): e.g. the four convs of an FCOS tower over the five
 * FPN levels (20 pairs, 4 distinct dW) in one launch. */
int osd_conv2d_wgrad_multi(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                           const int32_t* ns, const int32_t* hs, const int32_t* ws, const float* const* scales,
                           float* const* dws, float* const* dbs, void* stream);
/* The most general form: every pair has its own conv descriptor (n, h, w, cin, cout, r, s, strides, pads, out_stride):
 * all weight gradients of a ResNet stage (1x1 and 3x3, stride 1 and 2, FPN laterals) in one launch; dtype and algo are
 * taken from descs[0]; descs is a HOST array of n_seg descriptors. */
int osd_conv2d_wgrad_mixed(int n_seg, const osd_conv_desc* descs, const void* const* xs, const void* const* dys,
                           const float* const* scales, float* const* dws, float* const* dbs, void* stream);
/* packed fp32 dW [cout][r][s][cin] -> OIHW fp32 gradient, multiplied by the folded FrozenBN scale (nullable);
 * accumulate != 0 adds to grad_oihw (weights shared over FPN levels) */
int osd_unpack_wgrad(const float* dw_packed, const float* scale, float* grad_oihw, int cout, int cin, int r, int s,
                     int accumulate, void* stream);
/* db[c] (fp32, accumulated with atomics) += sum over m of dy[m][c] */
int osd_bias_grad(const void* dy, float* db, int m, int c, int stride, int dtype, void* stream);
/* data gradient of strided convs without the implicit-GEMM path (the two 3x3/2 convs P6/P7): forward-packed weights,
 * optional addend (same shape as dx) and ReLU mask */
int osd_conv2d_dgrad_naive(const osd_conv_desc* d, const void* dy, const void* w_fwd_packed, const void* mask,
                           const void* addend, void* dx, void* stream);
/* dst = mask>0 ? (zero-inserted src + addend) : 0, where zero-inserted src[n,2ho,2wo,:] = src[n,ho,wo,:] (dst, mask,
 * addend [n][h][w][c]; src [n][ho][wo][c]; mask, addend nullable) */
int osd_scatter2x(const void* src, const void* mask, const void* addend, void* dst, int n, int h, int w, int ho, int wo,
                  int c, int dtype, void* stream);
/* out = (a + b) * (mask > 0); b and mask nullable */
int osd_add_mask(const void* a, const void* b, const void* mask, void* out, int64_t numel, int dtype, void* stream);
/* top[n,y,x,:] = prev[n,y,x,:] + sum_{2x2} inner[n,2y+i,2x+j,:]  (backward of the FPN nearest-2x top-down add; prev nullable) */
int osd_upsample2x_bwd(const void* inner, const void* prev, void* top, int n, int h, int w, int c, int dtype, void* stream);
/* dq[n][c] = sum_p g[n,p,c] * feat[n,p,c] (correlation backward w.r.t. the pooled query; d_feat = osd_correlate_fwd(g, q)) */
int osd_correlate_bwd_query(const void* g, const void* feat, float* dq, int n, int hw, int c, int dtype, void* stream);
/* all FPN levels in one launch: dqs[l] [n][c] fp32 (zeroed by the call) */
int osd_correlate_bwd_query_levels(int n_levels, const void* const* gs, const void* const* feats, float* const* dqs,
                                   const int32_t* hws, int n, int c, int dtype, void* stream);
/* _C.roi_align_backward on NHWC fp32: gx [b][h][w][c] (zeroed by the call) += taps of gy [r][ph][pw][c] */
int osd_roialign_bwd(const float* gy, const float* rois, float* gx, int b, int h, int w, int c, int num_rois,
                     float spatial_scale, int ph, int pw, int sampling_ratio, void* stream);
int osd_shot_mean_bwd(const float* gy, float* gx, int b, int shots, int c, void* stream);
/* backward of osd_query_pool_levels: outs[l] [batch * shots][hs[l]][ws[l]][c] `dtype` = the gradient of the query feature maps given
 * dqs[l] [batch][c] fp32 (osd_shot_mean_bwd, osd_roialign_bwd with ph = pw = 1 and osd_cast_f32 per level, in three launches for all
 * levels: the same arithmetic).  gx32: caller-owned fp32 scratch of sum_l batch * shots * hs[l] * ws[l] * c floats (zeroed by the call). */
int osd_query_pool_levels_bwd(int n_levels, const float* const* dqs, const int32_t* hs, const int32_t* ws, const float* scales,
                              const float* rois, int batch, int shots, int c, int sampling_ratio, float* gx32, void* const* outs,
                              int dtype, void* stream);
int osd_cast_f32(const float* src, void* dst, int64_t numel, int dtype, void* stream);
/* Gradient bucket on its way to (to_wire = 1: fp32 src -> bf16 dst) or back from (to_wire = 0: bf16 src -> fp32 dst) a
 * half-width all-reduce (DDP's bf16 compression hook is the reference-side equivalent; tools/train_net.py:83-88 itself
 * exchanges fp32).  numel % 8 == 0 (the buckets of the flat buffer are cut at multiples of 64). */
int osd_grad_wire_cast(const void* src, void* dst, int64_t numel, int to_wire, void* stream);
/* SGD with momentum over the flat fp32 master / gradient / momentum buffers in ONE launch (the reference uses
 * torch.optim.SGD with per-parameter groups, solver/build.py:8-26; same update rule: g += wd*p; buf = momentum*buf + g
 * (buf = g on the first step); p -= lr*lr_mult*buf).  table: device array of
 * struct { int64 off, numel; float lr_mult, wd; int32 first_block, n_blocks; } (32 bytes); block_entry[b] = entry of
 * workgroup b. */
int osd_sgd_momentum_multi(const void* table, const int32_t* block_entry, int n_blocks, float* params,
                           const float* grads, float* momentum_buf, float lr, float momentum, int first_step,
                           void* stream);
/* The same update, which ALSO writes the forward-form packed weights (osd_pack_multi with dgrad = 0: `dtype`, FrozenBN scale
 * folded, [rows][r*s][kpad] rows) of every conv weight it updates — the thread that holds four updated fp32 values stores them
 * again as `dtype`; the packed buffer's padding (rows >= cout, k >= cin) must already be zero and is not written.  Replaces
 * osd_sgd_momentum_multi + the forward half of the per-step repack (solver/build.py:8-26 has no counterpart for the latter: the
 * reference's convs read the fp32 parameters directly).  table: device array of struct { int64 off, numel; float lr_mult, wd;
 * int32 first_block, n_blocks; int64 dst_off (elements into `packed`; -1: update only), scale_off (floats into `scales`; -1:
 * none); int32 cin, rs, kpad, pad; } (64 bytes).  Tensors that start on a 16-byte boundary (and, packed, have cin % 4 == 0) take 16 bytes
 * per lane, the others one value per lane.  zero_grads != 0: the gradients are consumed — every element read is overwritten with
 * zero (optimizer.zero_grad() of engine/trainer.py:89 folded into the update: the weight-gradient kernels accumulate). */
int osd_sgd_momentum_pack_multi(const void* table, const int32_t* block_entry, int n_blocks, float* params,
                                float* grads, float* momentum_buf, const float* scales, void* packed, int dtype,
                                float lr, float momentum, int first_step, int zero_grads, void* stream);
/* GroupNorm + ReLU of one tower layer over ALL FPN levels (separate tensors sharing gamma/beta) in two launches, and
 * its backward in two launches: statistics per (level, image, slab), finalised inside the apply kernels.
 * xs/ys/us/dts/dus: HOST arrays of n_levels device pointers to [n][hw_l][c] tensors; hws: HOST array;
 * ab [n_levels][4][n][c] fp32 (written by fwd, read by bwd: planes a, b with y = relu(a x + b), then xa, xb with
 * xhat = xa x + xb, so that the backward pass never divides by gamma); ws: fwd n_levels*n*OSD_GN_SPLITS*groups*2 floats,
 * bwd n_levels*n*OSD_GN_SPLITS*(groups*2 + 2*c) floats (group sums + per-slab d gamma / d beta partials). */
int osd_groupnorm_relu_fwd_levels(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                  const float* gamma, const float* beta, float* ab, float* ws, int n, int c, int groups,
                                  float eps, int dtype, void* stream);
/* osd_groupnorm_relu_fwd_levels when osd_conv2d_fwd_multi_gn (forward statistics) has already accumulated the slab sums of the
 * levels whose bit is set in fused_mask (ws + l*n*OSD_GN_SPLITS*groups*2, zeroed by the caller before the conv) */
int osd_groupnorm_relu_fwd_levels_fused(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                        const float* gamma, const float* beta, float* ab, float* ws, int n, int c, int groups,
                                        float eps, int dtype, uint32_t fused_mask, void* stream);
/* The same two operations with ONE pass over HBM each (bf16 only; groupnorm_onepass.hip): a workgroup keeps its pixels in registers
 * between the statistics and the apply step and the workgroups of a (level, image) exchange their partial sums through `ws` behind
 * an arrival counter in `sync` (deterministic: partials are summed in a fixed order; only d gamma / d beta are atomic adds, one per
 * channel and (level, image), as in the two-launch form).  Same arguments and results as the two-launch entries (statistics summed in
 * another order: outputs agree to fp32 rounding of the sums), plus:
 *   ws:   osd_groupnorm_onepass_workspace_bytes(n_levels, hws, n, c, groups, backward) bytes, no initialisation needed;
 *   sync: osd_groupnorm_onepass_sync_bytes(n_levels, n) bytes, ZERO before the first launch that uses it; every launch leaves it zero
 *         again (no memset per launch).  Launches that may overlap in time (different streams) need sync buffers of their own.
 *         sync[2] != 0 after a launch: a workgroup gave up waiting for its job (~1 s); since ABI 4 such a workgroup also writes NaN
 *         into every output element it owns (and into the saved statistics), so the failure is visible in the data as well.
 *   Residency: the forward-progress argument needs two whole jobs resident at once; the launcher checks the occupancy API's answer
 *   for the device and returns OSD_ERR_UNSUPPORTED when it does not hold (the caller then uses the two-launch entries).
 * Replaces maskrcnn_benchmark/modeling/rpn/fcos/fcos.py:29-37 (GroupNorm + ReLU of a tower layer; torch.nn.GroupNorm forward and
 * autograd backward in the reference). */
int64_t osd_groupnorm_onepass_workspace_bytes(int n_levels, const int32_t* hws, int n, int c, int groups, int backward);
int64_t osd_groupnorm_onepass_sync_bytes(int n_levels, int n);
int osd_groupnorm_relu_fwd_levels_onepass(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                          const float* gamma, const float* beta, float* ab, float* ws, int32_t* sync, int n, int c,
                                          int groups, float eps, int dtype, void* stream);
int osd_groupnorm_relu_bwd_levels_onepass(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                          const int32_t* hws, const float* ab, const float* gamma, const float* beta, float* ws,
                                          int32_t* sync, float* dgamma, float* dbeta, int n, int c, int groups, int dtype, void* stream);
/* DIAGNOSTIC: osd_groupnorm_relu_fwd_levels_onepass with its last workgroup never started and `spin_limit` polls per wait, so that
 * the timeout path runs (error word set, NaN outputs for the incomplete job).  Leaves `sync` dirty: give it a buffer of its own.
 * No reference counterpart (torch.nn.GroupNorm, fcos.py:37, cannot fail this way); exists so that the failure path is tested. */
int osd_groupnorm_onepass_selftest_timeout(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                           const float* gamma, const float* beta, float* ab, float* ws, int32_t* sync, int n, int c,
                                           int groups, float eps, int dtype, int spin_limit, void* stream);
int osd_groupnorm_relu_bwd_levels(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                  const int32_t* hws, const float* ab, const float* gamma, const float* beta, float* ws,
                                  float* dgamma, float* dbeta, int n, int c, int groups, int dtype, void* stream);
/* the same when osd_conv2d_fwd_multi_gn has already accumulated the sums of some levels: bit l of fused_mask set = level l's
 * parts of ws (group sums at ws + l*n*OSD_GN_SPLITS*groups*2, d gamma / d beta partials at
 * ws + n_levels*n*OSD_GN_SPLITS*groups*2 + l*n*OSD_GN_SPLITS*2*c) are complete; the statistics pass skips those levels */
int osd_groupnorm_relu_bwd_levels_fused(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                        const int32_t* hws, const float* ab, const float* gamma, const float* beta, float* ws,
                                        float* dgamma, float* dbeta, int n, int c, int groups, int dtype, uint32_t fused_mask,
                                        void* stream);
/* osd_groupnorm_relu_bwd_levels which ALSO accumulates the bias gradient of the conv that produced us (fcos.py:29-37:
 * Conv2d(bias=True) -> GroupNorm -> ReLU): conv_dbias[c] += sum over levels, images and pixels of du.  No pass over du: per channel
 * sum_px du = rstd (gamma sum_px dz - N c1 - c2 sum_px xhat), and the statistics pass gathers sum_px xhat beside its other sums (one more
 * plane of slab partials: ws holds n_levels*n*OSD_GN_SPLITS*(groups*2 + 3*c) floats here).  One atomic add per channel and
 * (level, image).  The tower's weight-gradient launch then runs without its fused d-bias column sums (11 % of that launch). */
int osd_groupnorm_relu_bwd_levels_convbias(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                           const int32_t* hws, const float* ab, const float* gamma, const float* beta, float* ws,
                                           float* dgamma, float* dbeta, float* conv_dbias, int n, int c, int groups, int dtype,
                                           void* stream);
/* FCOS loss (modeling/rpn/fcos/loss.py:101-276; focal term = csrc/cuda/SigmoidFocalLoss_cuda.cu) for one FPN level.
 * phase 0 accumulates sums[5] = {num_pos, sum_w, sum_focal, sum_w*(1-giou), sum_bce} (zero them before the first
 * level); phase 1 writes d_cls_ctr [n][hw][grad_stride] (d logit, d centerness at +0/+1; the caller zero-fills the
 * rest, the layout the data-gradient conv consumes), d_reg [n][hw][grad_stride] (4 values) = gradient w.r.t. the
 * bbox_pred conv output (already through exp and the Scale), and accumulates d_scale_raw (divide by scale).
 * gt_boxes [n][max_gt][4] fp32 xyxy, gt_count [n]. */
int osd_fcos_loss_level(int phase, const void* cls_ctr, const void* reg, const float* gt_boxes, const int32_t* gt_count,
                        int max_gt, int n, int h, int w, int stride, float size_lo, float size_hi, float radius,
                        float gamma, float alpha, const float* scale_dev, float* sums, void* d_cls_ctr, void* d_reg,
                        int grad_stride, float* d_scale_raw, int dtype, void* stream);
/* The same for ALL FPN levels in one launch per phase (the per-level launches sit on the critical path between forward
 * and backward).  cls_ctrs / regs / scale_devs / d_cls_ctrs / d_regs / d_scale_raws: HOST arrays of n_levels (<= 6) device
 * pointers; hs / ws / strides / size_lo / size_hi: HOST arrays. */
int osd_fcos_loss_levels(int phase, int n_levels, const void* const* cls_ctrs, const void* const* regs,
                         const float* gt_boxes, const int32_t* gt_count, int max_gt, int n, const int32_t* hs,
                         const int32_t* ws, const int32_t* strides, const float* size_lo, const float* size_hi, float radius,
                         float gamma, float alpha, const float* const* scale_devs, float* sums, void* const* d_cls_ctrs,
                         void* const* d_regs, int grad_stride, float* const* d_scale_raws, int dtype, void* stream);
/* losses[4] = {loss_cls, loss_reg, loss_centerness, num_pos} */
int osd_fcos_loss_finalize(const float* sums, float* losses, int n, void* stream);
/* ... and, in the same launch, the gradient of the learnable per-level Scale (fcos.py:81, 95-97): d_scales[l] += d_scale_raw[l] /
 * scales[l] for l < n_levels (d_scale_raw: what osd_fcos_loss_levels(phase 1) accumulated, sum ds * log(reg)). */
int osd_fcos_loss_finalize_scales(const float* sums, float* losses, int n, const float* d_scale_raw, const float* scales,
                                  float* d_scales, int n_levels, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Second-stage few-shot ROI box head (modeling/roi_heads/box_head/box_head.py:81-259).  Its
 * convolutions and fully connected layers are osd_conv2d_fwd calls (a Linear is a 1x1 conv over [R][1][1][K]); the
 * three entry points below are the rest.
 * ---------------------------------------------------------------------------------------------------------------- */
#define OSD_MAX_ROI_LEVELS 8
/* Pooler.forward (modeling/poolers.py:93-124): every ROI is ROIAligned (pool x pool bins, `sampling_ratio` samples per
 * bin side; _C.roi_align_forward, csrc/cuda/ROIAlign_cuda.cu:11-122) from the FPN level LevelMapper picks
 * (poolers.py:11-42: floor(4 + log2(sqrt(area)/224 + 1e-6)) clamped to the levels, area with the "+1" of
 * structures/bounding_box.py:226-236).  xs / hs / ws / scales: HOST arrays of n_levels (device pointer to the NHWC
 * [n][h][w][c] map, its size, its spatial scale).  boxes [n][max_rois][4] fp32 xyxy, counts [n] (NULL: all max_rois are
 * valid; the reference needs equal counts, poolers.py:80 - here ROIs past counts[image] give zero rows).
 * y [n*max_rois][pool][pool][y_stride] `dtype` (channels 0..c-1 written).  level_out (optional) [n*max_rois] int32:
 * the level each ROI used, -1 past the count. */
int osd_roi_pool_levels(int n_levels, const void* const* xs, const int32_t* hs, const int32_t* ws, const float* scales,
                        const float* boxes, const int32_t* counts, void* y, int n, int c, int max_rois, int pool,
                        int sampling_ratio, int y_stride, int32_t* level_out, int dtype, void* stream);
/* nn.GroupNorm(groups, c) + nn.LeakyReLU(slope) over [n_samples][hw][c] ROI maps, hw = 49 (box_head.py:43-66; slope 0
 * = ReLU).  addend (optional, `dtype`, [..][hw][c]): added to x before the statistics; sample i reads addend map
 * (i / rois_per_add) * add_stride + add_offset.  This is how the `concat` comparison runs: conv1x1(cat(x, q)) =
 * conv1x1_x(x) + conv1x1_q(q) + bias, and the q half is one map per (image, shot) instead of one per ROI
 * (box_head.py:126,146-148) - rois_per_add = max_rois, add_stride = shots, add_offset = shot. */
int osd_groupnorm_act_rois(const void* x, const void* addend, const float* gamma, const float* beta, void* y,
                           int n_samples, int hw, int c, int groups, float eps, float slope, int rois_per_add,
                           int add_stride, int add_offset, int dtype, void* stream);
/* Per-class arg-max over shots (box_head.py:239-252), softmax (box_head/inference.py:66), BoxCoder.decode of the class-1
 * deltas (modeling/box_coder.py:50-95, weights reg_weights[4] on the HOST), clip_to_image (bounding_box.py:214-219).
 * pred [shots][n*max_rois][pred_stride] `dtype`: columns 0..1 = cls_score, 2..9 = bbox_pred (roi_box_predictors.py:88-99).
 * scores [n][max_rois] = class-1 probability, or -1 (dropped by osd_rank_sort_gather) for ROIs past counts[image] or
 * probability <= score_thresh (inference.py:136); boxes [n][max_rois][4].  logits_out [n*max_rois][2] / reg_out
 * [n*max_rois][8] (optional, fp32): the selected logits / deltas.  Follow with osd_rank_sort_gather + osd_nms_sorted
 * (thresh ROI_HEADS.NMS, filter_results inference.py:120-166). */
int osd_box_decode(const void* pred, const float* rois, const int32_t* counts, float* scores, float* boxes,
                   float* logits_out, float* reg_out, int n, int max_rois, int shots, int pred_stride,
                   const float* reg_weights, float img_h, float img_w, const float* img_hw, float score_thresh, int dtype,
                   void* stream);   /* img_hw: optional [n][2] device array of true (height, width) per image, as above */
/* add_gt_proposals of the TRAINING proposal path (modeling/rpn/fcos/inference.py:139-160,279): per image the kept
 * proposals (boxes [n][cap][4], scores [n][cap], counts [n]) followed by its ground-truth boxes (gt_boxes [n][max_gt][4],
 * gt_count [n]) with score 1 -> out_boxes [n][cap+max_gt][4], out_scores [n][cap+max_gt], out_counts [n]; rows past the
 * new count are zero. */
int osd_append_gt_boxes(const float* boxes, const float* scores, const int32_t* counts, const float* gt_boxes,
                        const int32_t* gt_count, float* out_boxes, float* out_scores, int32_t* out_counts, int n, int cap,
                        int max_gt, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Second stage, TRAINING path: FastRCNNLossComputation (modeling/roi_heads/box_head/loss.py).
 * ---------------------------------------------------------------------------------------------------------------- */
/* subsample (loss.py:234-301): per image match the proposals (boxes [n][max_props][4], counts [n]; the ground truth already
 * appended, fcos/inference.py:139-160) to the ground truth (gt_boxes [n][max_gt][4], gt_count [n], gt_labels [n][max_gt] or
 * NULL = all 1) by IoU with "+1" areas (boxlist_ops.py:221-256; Matcher with high = low = iou_thresh, matcher.py:52-83):
 * label 0 below the threshold, else the matched box's label; then BalancedPositiveNegativeSampler
 * (balanced_positive_negative_sampler.py:19-62): at most int(batch_per_image * positive_fraction) positives and
 * batch_per_image rows in all.  The reference draws torch.randperm(n)[:k]; here the caller supplies keys [n][max_props]
 * (uniform randoms) and the k smallest keys of each class win (ties: lower index) = the reference with randperm(n) :=
 * argsort(keys of that class's members), which is how the fixtures pin it.  Outputs, rows in ascending proposal order
 * (loss.py:292), rows past s_count[image] zero / label -1: s_boxes [n][batch][4], s_labels [n][batch], s_targets
 * [n][batch][4] = BoxCoder.encode(matched box, proposal) with reg_weights[4] (HOST; box_coder.py:21-50; background rows
 * against box 0 like matched_idxs.clamp(min=0), loss.py:70), s_index [n][batch] (proposal index), s_count [n]; optional
 * all_labels / all_matched [n][max_props] (label / matched box or -1 of every proposal). */
int osd_box_match_sample(const float* boxes, const int32_t* counts, const float* gt_boxes, const int32_t* gt_count,
                         const int32_t* gt_labels, const float* keys, int n, int max_props, int max_gt, int batch_per_image,
                         float positive_fraction, float iou_thresh, const float* reg_weights, float* s_boxes,
                         int32_t* s_labels, float* s_targets, int32_t* s_index, int32_t* s_count, int32_t* all_labels,
                         int32_t* all_matched, void* stream);
/* __call__ (loss.py:306-381, 'ce_loss', class-specific regression) with the weights of box_head.py:193-194: pred
 * [n*rois_per_image][pred_stride] `dtype` (columns 0..1 class logits, 2..9 box deltas, as osd_box_decode reads them) ->
 * losses[3] = {w_cls * cross_entropy (mean over the valid rows), w_box * smooth_l1(beta 1, summed over the positives' class
 * deltas) / valid rows, valid rows}; d_pred (nullable) [..][grad_stride] `dtype` = the gradient w.r.t. pred (zero rows past
 * s_count[image]).  Two classes (ROI_BOX_HEAD.NUM_CLASSES of the config of record): labels are 0 or 1; a label > 1 has no
 * columns in a 10-wide row, so nothing is read or written for such a row and losses[0..1] come back NaN. */
int osd_box_loss(const void* pred, const int32_t* labels, const float* targets, const int32_t* s_count, int n,
                 int rois_per_image, int pred_stride, float w_cls, float w_box, float* losses, void* d_pred, int grad_stride,
                 int dtype, void* stream);
/* backward of osd_groupnorm_act_rois: dx from dy (same arguments; statistics recomputed from x [+ addend]); dgamma / dbeta
 * [c] fp32 are ACCUMULATED; part_ws: n_samples * 2 * c floats.  The gradient w.r.t. the addend map of an image is the sum
 * of dx over that image's ROIs: osd_rois_sum. */
int osd_groupnorm_act_rois_bwd(const void* x, const void* addend, const float* gamma, const float* beta, const void* dy,
                               void* dx, float* part_ws, float* dgamma, float* dbeta, int n_samples, int hw, int c,
                               int groups, float eps, float slope, int rois_per_add, int add_stride, int add_offset,
                               int dtype, void* stream);
/* out[img][e] = sum_r x[img * rois_per_image + r][e], e < elems */
int osd_rois_sum(const void* x, void* out, int n, int rois_per_image, int64_t elems, int dtype, void* stream);
/* backward of osd_roi_pool_levels (Pooler + _C.roi_align_backward, csrc/cuda/ROIAlign_cuda.cu:178-254): dy
 * [n*max_rois][pool][pool][dy_stride] `dtype` is scattered into the fp32 level maps gxs[l] [n][h_l][w_l][c] (HOST array of
 * device pointers; ACCUMULATED with atomics: zero them first). */
int osd_roi_pool_levels_bwd(int n_levels, float* const* gxs, const int32_t* hs, const int32_t* ws, const float* scales,
                            const float* boxes, const int32_t* counts, const void* dy, int n, int c, int max_rois, int pool,
                            int sampling_ratio, int dy_stride, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Input transforms (data/transforms/transforms.py:27-92 in the Compose order of
 * data/transforms/build.py:39-46) + the zero padding of to_image_list (structures/image_list.py:52-70), one image per call:
 *   Resize = PIL.Image.resize((out_w, out_h), BILINEAR) on 8-bit RGB (Pillow's fixed-point ImagingResample: horizontal pass,
 *   then vertical; a pass whose size does not change is skipped), hflip, ToTensor (/255), Normalize (to_bgr255: channels
 *   reversed and * 255; then (x - mean[c]) / std[c]) — bit-exact against the reference's pipeline.
 * src_rgb_hwc: device uint8 [in_h][in_w][3].  (out_h, out_w) = Resize.get_size of the caller.  mean3 / std3: HOST floats.
 * The result lands in slot `batch_index` of dst, whose every image is dst_h x dst_w with this image at (pad_t, pad_l) and
 * zeros elsewhere: layout 0 = fp32 NCHW [n][3][dst_h][dst_w] (the reference's batch tensor; dtype ignored), layout 1 =
 * `dtype` NHWC4 [n][dst_h][dst_w][4] (the stem conv's padded input, as osd_pack_image writes it).
 * workspace: osd_image_transform_workspace_bytes(in_h, in_w, out_h, out_w) bytes. */
int osd_image_transform(const uint8_t* src_rgb_hwc, int in_h, int in_w, int out_h, int out_w, int flip, int to_bgr255,
                        const float* mean3, const float* std3, void* dst, int layout, int dtype, int batch_index, int dst_h,
                        int dst_w, int pad_t, int pad_l, void* workspace, void* stream);
int64_t osd_image_transform_workspace_bytes(int in_h, int in_w, int out_h, int out_w);
/* The same for a whole batch in THREE launches (tap tables of both axes, horizontal pass, vertical pass + normalise + pad
 * write): what BatchCollator does to the images the dataset returns (data/collate_batch.py:15-20 over
 * data/transforms/build.py:39-46).  srcs / in_hs / in_ws / out_hs / out_ws / flips (nullable = no flip) are HOST arrays
 * of n_images entries, srcs holding device pointers; image i goes to batch slot first_batch_index + i of dst.
 * workspace: the SUM of osd_image_transform_workspace_bytes over the images. */
int osd_image_transform_batch(int n_images, const uint8_t* const* srcs_rgb_hwc, const int32_t* in_hs, const int32_t* in_ws,
                              const int32_t* out_hs, const int32_t* out_ws, const int32_t* flips, int to_bgr255,
                              const float* mean3, const float* std3, void* dst, int layout, int dtype,
                              int first_batch_index, int dst_h, int dst_w, int pad_t, int pad_l, void* workspace,
                              void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Evaluation: the per-image matching of the PASCAL VOC detection metric, calc_detection_voc_prec_rec
 * (data/datasets/evaluation/voc/voc_eval.py:84-137, boxlist_iou structures/boxlist_ops.py:221-256).
 * det_boxes [n][max_det][4] xyxy, det_scores [n][max_det], det_labels [n][max_det], det_count [n]; gt_boxes [n][max_gt][4],
 * gt_labels, gt_difficult (0 / 1) [n][max_gt], gt_count [n] (max_gt <= 512).  Per detection, in the INPUT order:
 * matched_gt = index of the ground-truth box of its class with the largest IoU ('+1' on x2, y2 of both boxes, then the '+1'
 * areas of boxlist_iou; first maximum; -1 when that IoU < iou_thresh or the image has no box of the class) and
 * match = 1 (the highest-scoring detection matched to a non-difficult box; equal scores: the higher index), 0 (unmatched, or
 * a box already claimed) or -1 (matched to a difficult box: ignored), 0 past det_count. */
int osd_voc_match(const float* det_boxes, const float* det_scores, const int32_t* det_labels, const int32_t* det_count,
                  const float* gt_boxes, const int32_t* gt_labels, const uint8_t* gt_difficult, const int32_t* gt_count,
                  int n, int max_det, int max_gt, float iou_thresh, int8_t* match, int32_t* matched_gt, void* stream);

/* Precision / recall curves of every class in one launch (voc_eval.py:139-158).  flags_sorted: the match flags of the whole
 * dataset ordered by (class id ascending, score descending); class c owns [class_begin[c], class_begin[c + 1]) (n_classes + 1
 * entries); n_pos [n_classes] = the class's non-difficult ground-truth boxes.  prec / rec (float64, same indexing):
 * prec[i] = tp / (tp + fp) over the class's first i + 1 detections (NaN while only ignored ones were seen), rec[i] = tp / n_pos
 * (NaN for n_pos == 0).  Counts are integers and the divisions IEEE double: the reference's numpy values, bit for bit. */
int osd_voc_curves(const int8_t* flags_sorted, const int32_t* class_begin, const int32_t* n_pos, int n_classes, double* prec,
                   double* rec, void* stream);

/* Average precision of every class from its curves (calc_detection_voc_ap, voc_eval.py:161-216): use_07_metric != 0 -> the
 * 11-point PASCAL VOC 2007 metric (bit-identical to the reference's loop), else the area under the monotone precision envelope
 * (the same terms as the reference's, summed in a fixed tree order: equal to ~1e-16 relative).  has_prec[c] == 0 (class id never
 * seen) or has_rec[c] == 0 (no countable ground truth) -> NaN, like the reference.  ap [n_classes] float64. */
int osd_voc_ap(const double* prec, const double* rec, const int32_t* class_begin, const uint8_t* has_prec, const uint8_t* has_rec,
               int n_classes, int use_07_metric, double* ap, void* stream);

/* COCO-style matching, bbox (the reference's COCO configs: data/datasets/evaluation/coco/coco_eval.py:385-408 hands the
 * detections to pycocotools' COCOeval — cocodataset/cocoapi, unpinned, INSTALL.md:34-38, absent from this image; the algorithm
 * restated is its evaluateImg + maskApi.c bbIou: oracle/coco_eval_ref.py, PARITY UNPINNED).  One (image, category) pair per
 * workgroup: det_boxes_xywh [n_pairs][max_det][4] float64, sorted by descending score and cut at the largest maxDets by the caller,
 * det_count [n_pairs]; gt_boxes_xywh [n_pairs][max_gt][4], gt_area (the annotation's `area`), gt_crowd (0 / 1), gt_count
 * (max_gt <= 512).  iou_thrs [n_thrs <= 10] and area_ranges [n_areas <= 4][2] are HOST arrays.  Per (pair, area range, threshold,
 * detection): dt_match = 1 + the INPUT index of the ground-truth box it takes (0: none) — the still-free box (crowd boxes are
 * never used up) with the highest IoU >= threshold, non-ignored boxes preferred —, dt_ignore = matched to an ignored box, or
 * unmatched with w * h outside the range; gt_ignore [n_pairs][n_areas][max_gt] = crowd or `area` outside the range, in the
 * ignored-last order the accumulation concatenates.  IoU = intersection / union in float64, no '+1'; against a crowd box the union
 * is the detection's area.  Precision / recall tables and the 12 summary numbers: oneshotdet_amd/evaluation.py. */
int osd_coco_match(const double* det_boxes_xywh, const int32_t* det_count, const double* gt_boxes_xywh, const double* gt_area,
                   const uint8_t* gt_crowd, const int32_t* gt_count, int n_pairs, int max_det, int max_gt,
                   const double* iou_thrs, int n_thrs, const double* area_ranges, int n_areas, int32_t* dt_match,
                   uint8_t* dt_ignore, uint8_t* gt_ignore, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ONESHOTDET_HIP_H */
