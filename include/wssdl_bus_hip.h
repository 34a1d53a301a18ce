/*
 * wssdl_bus_hip.h -- C ABI of libwssdl_bus_hip.so, the MI355X (gfx950) implementation
 * of the wssdl_bus Faster-R-CNN detection hot path.
 *
 * Every entry point replaces one native/NumPy routine of the reference
 * (paths relative to code/lib of syshin1014/wssdl_bus, cited per function) and is
 * what the reference-side FFI for that routine would bind (INTEGRATION.md shows
 * the ctypes / tf.load_op_library-side stubs).
 *
 * Conventions
 *   - plain C: pointers and sizes only; no torch / HIP types in signatures.
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all data pointers are DEVICE pointers unless the name ends in `_host`.
 *   - outputs and workspaces are caller-allocated (the PyTorch caching allocator
 *     in the shipped host layer); `*_workspace_bytes` functions size them.
 *   - nothing here allocates, frees, copies to host or synchronises: every call
 *     only enqueues kernels on `stream`, so calls are hipGraph-capturable.
 *   - return value: 0 = WSSDL_OK, otherwise a wssdl_status code; never exit(),
 *     never prints (the reference's CUDA launcher prints and exit(-1)s on a
 *     launch error, roi_pooling_op_gpu.cu.cc:102-107 -- deliberately not kept).
 *   - layouts are the reference's: feature maps NHWC f32, rois [R,5] f32
 *     (batch_idx,x1,y1,x2,y2), gt_boxes [N,MAX_GT,5] f32 (x1,y1,x2,y2,cls).
 */
#ifndef WSSDL_BUS_HIP_H
#define WSSDL_BUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *wssdl_stream_t;

#if defined(__GNUC__)
#define WSSDL_API __attribute__((visibility("default")))
#else
#define WSSDL_API
#endif

enum wssdl_status {
    WSSDL_OK = 0,
    WSSDL_ERR_INVALID_ARGUMENT = 1,   /* bad shape / null pointer / unsupported size */
    WSSDL_ERR_WORKSPACE = 2,          /* workspace too small */
    WSSDL_ERR_LAUNCH = 3              /* hipGetLastError() != hipSuccess after a launch */
};

/* bin-boundary rounding of RoI pooling */
enum wssdl_roi_rounding {
    WSSDL_ROI_ROUND_CUDA = 0,  /* floor(ph*bin), ceil((ph+1)*bin): roi_pooling_op_gpu.cu.cc:51-58 (canonical) */
    WSSDL_ROI_ROUND_CPU = 1    /* (int)(ph*bin), (int)((ph+1)*bin): roi_pooling_op.cc:167-170 */
};

/* wssdl_roi_pool_forward's N when the caller does not know the batch size (the reference's ROIPoolForwardLaucher,
 * roi_pooling_op_gpu.h:17-21): no range check of rois[:, 0] above zero.  N == 0 with RoIs to pool is an error. */
#define WSSDL_ROI_BATCH_UNKNOWN (-1)

/* dataset switch of the anchor-target layer, anchor_target_layer_tf_bus.py:120-199 */
enum wssdl_dataset {
    WSSDL_DATASET_SNUBH = 0,     /* gt holds positive boxes first, then background boxes */
    WSSDL_DATASET_SNUBH_FG = 1,  /* same layout, background boxes ignored */
    WSSDL_DATASET_FG_ONLY = 2    /* e.g. 'UDIAT': every gt box is a positive */
};

#define WSSDL_MAX_ANCHORS 32     /* base anchors per cell (reference uses 9 or 12) */
/* roi_counts[i] / num_keep of an image whose NMS sweep gave up waiting for the mask blocks of the fused
 * launch (50 ms without progress, tuning key "nms_wait_us": the GPU is held by another process or kernel).  The image's rows are
 * incomplete; every consumer must treat a negative count as an error, never as "no proposals". */
#define WSSDL_NMS_TIMED_OUT (-1)
#define WSSDL_MAX_GT 64          /* gt boxes per image (reference: MAX_GT_PER_IMAGE = 20) */

/* library / build identification: returns a static string */
WSSDL_API const char *wssdl_version(void);
/* name of the last HIP error seen by this library on the calling thread ("" if none) */
WSSDL_API const char *wssdl_last_error(void);
/* Tuning knobs.  The library never reads the environment: a knob is an int the caller sets (process
 * wide, read at call time).  Keys: "roi_bwd_plan" (plan id of the list-driven RoI-pool backward, -1 =
 * by launch size), "roi_fwd_variant",
 * "roi_bwdc_variant", "roi_bwd_cg" (shapes of the compact forward / the fallback backwards, 0 =
 * automatic), "nms_one_pass" (1: the proposal layer skips the probe pass), "nms_fused" (0: mask and sweep of a one-pass
 * NMS as two launches instead of the fused one), "topk_sort" (order of the proposal candidates: 1 sorted runs +
 * cross ranks, the default; 0 the select + sample sort), "nms_wait_us" (how long a sweep of the fused launch waits for
 * the mask blocks before it reports WSSDL_NMS_TIMED_OUT; default 50000), "roi_fwd_blocks" / "roi_fwd_blocks_sort" /
 * "roi_fwd_blocks_parts" (block-table forward, see there), "roi_fwd_one_bin" (forward launches of fewer than 32768 bin rows x
 * 256-channel slices: waves per bin row, 7 = one wave per bin, the default; 0 = the sliced kernel).  Results do not depend on
 * any of them.  One more key is a fault injector for tests, not a knob: "nms_fused_fault" (> 0: the fused
 * NMS launch withholds image 0's progress counts and that image's sweep gives up after this many microseconds,
 * reporting WSSDL_NMS_TIMED_OUT).  Unknown key -> WSSDL_ERR_INVALID_ARGUMENT. */
WSSDL_API int wssdl_set_tuning(const char *key, int value);
WSSDL_API int wssdl_get_tuning(const char *key, int *value_host);

/* ------------------------------------------------------------------ a1, a2 ---
 * generate_anchors: rpn_msr/generate_anchors.py:37-97.  Host-side (9 boxes);
 * writes n_ratios*n_scales rows of 4 doubles to anchors_host, ratio-major.
 * Returns the number of anchors, or a negative wssdl_status. */
WSSDL_API int wssdl_generate_anchors_host(int base_size, const double *ratios_host, int n_ratios,
                                const double *scales_host, int n_scales, double *anchors_host);

/* shifted anchor grid: anchor_target_layer_tf_bus.py:59-73 == proposal_layer_tf_bus.py:55-71.
 * out [H*W*A, 4] f64, row (h*W+w)*A+a = base[a] + stride*(w,h,w,h). */
WSSDL_API int wssdl_shifted_anchors(const double *base_anchors_host, int A, int H, int W, int feat_stride,
                          double *out, wssdl_stream_t stream);

/* ------------------------------------------------------------------ a3, a4 ---
 * bbox_overlaps: utils/bbox.pyx:15-55.  boxes [N, box_stride>=4] f64, query
 * [K, query_stride>=4] f64 (only columns 0..3 are read), out [N,K] f64.  Bit-exact. */
WSSDL_API int wssdl_bbox_overlaps(const double *boxes, int64_t N, int box_stride, const double *query,
                        int64_t K, int query_stride, double *out, wssdl_stream_t stream);
/* bbox_overlaps_ui: utils/bbox_ui.pyx:12-47 (intersection / area of boxes[n]). */
WSSDL_API int wssdl_bbox_overlaps_ui(const double *boxes, int64_t N, int box_stride, const double *query,
                           int64_t K, int query_stride, double *out, wssdl_stream_t stream);

/* ---------------------------------------------------------------------- a8 ---
 * nms: fast_rcnn/nms_wrapper.py:13-21 -> nms/cpu_nms.pyx:17-68.
 * dets [n,5] f32 (x1,y1,x2,y2,score), any order (sorted internally by score
 * descending; equal scores: higher input index first).  f32 box arithmetic,
 * suppression when (double)iou >= thresh -- the vendored build's rule.
 * keep [max_keep] i32 receives kept indices into dets in score order (the
 * reference's caller truncates keep[:post_nms_topN]; pass max_keep = n for
 * all); *num_keep (device i32) receives the count written.  n == 0 -> count 0. */
WSSDL_API size_t wssdl_nms_workspace_bytes(int n);
WSSDL_API int wssdl_nms(const float *dets, int n, double thresh, int max_keep, int32_t *keep,
              int32_t *num_keep, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
/* The test path's NMS is a second file with the same rule: utils/nms.pyx:17-68 `nms` (imported as
 * utils.cython_nms at fast_rcnn/test_bus.py:10, called at :293,366,375) == wssdl_nms.
 * nms_new: utils/nms.pyx:70-123 (imported at test_bus.py:10, never called by the reference): box j is also
 * suppressed when it lies almost inside the kept box i or the other way round,
 *   ovr >= thresh  or  (double)(inter / area_i) > 0.95  or  (double)(inter / area_j) > 0.95
 * (:118-121; ovr1 / ovr2 are untyped there, i.e. the f32 quotients widened to Python floats).
 * Same arguments, workspace and outputs as wssdl_nms. */
WSSDL_API int wssdl_nms_new(const float *dets, int n, double thresh, int max_keep, int32_t *keep,
              int32_t *num_keep, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);

/* ------------------------------------------------------------ a6, a7, a8, a9 ---
 * proposal_layer: rpn_msr/proposal_layer_tf_bus.py:19-148 for all N images in
 * one call.  rpn_cls_prob [N,H,W,2A] f32 (fg prob = channels A..2A-1),
 * rpn_bbox_pred [N,H,W,4A] f32, im_info [N,im_info_stride] f32 (h,w,scale,...).
 * Decode (bbox_transform_inv, f32) -> clip -> min-size filter -> sort by score
 * -> top pre_nms_topN -> NMS -> top post_nms_topN.
 * Outputs: rois_padded [N, post_nms_topN, 5] f32 (batch_idx,x1,y1,x2,y2; rows
 * beyond the image's count are zero), roi_counts [N] i32.  Optional (may be
 * NULL) debug outputs used by the parity tests: decoded [N,H*W*A,4] f32 (after
 * clip), sorted_index [N, pre_nms_topN] i32 (anchor index of each pre-NMS
 * candidate in score order, -1 padded), sorted_count [N] i32. */
WSSDL_API size_t wssdl_proposal_workspace_bytes(int N, int H, int W, int A, int pre_nms_topN);
WSSDL_API int wssdl_proposal_layer(const float *rpn_cls_prob, const float *rpn_bbox_pred,
                         const float *im_info, int im_info_stride, int N, int H, int W,
                         const double *base_anchors_host, int A, int feat_stride,
                         int pre_nms_topN, int post_nms_topN, double nms_thresh, float min_size,
                         float *rois_padded, int32_t *roi_counts,
                         float *decoded, int32_t *sorted_index, int32_t *sorted_count,
                         void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
/* f2: the same layer fed with the raw RPN class scores rpn_cls_score [N,H,W,2A] (logits):
 * the reshape -> softmax -> reshape chain in front of the layer (networks/network.py:283-291,
 * 398-404, Resnet_train_bus.py:76-81) is fused into the decode kernel: the fg probability of
 * anchor a is softmax(score[a], score[A+a])[1]. */
WSSDL_API int wssdl_proposal_layer_from_logits(const float *rpn_cls_score, const float *rpn_bbox_pred,
                         const float *im_info, int im_info_stride, int N, int H, int W,
                         const double *base_anchors_host, int A, int feat_stride,
                         int pre_nms_topN, int post_nms_topN, double nms_thresh, float min_size,
                         float *rois_padded, int32_t *roi_counts,
                         float *decoded, int32_t *sorted_index, int32_t *sorted_count,
                         void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
/* gathers the per-image lists into the reference's contiguous blob [sum counts, 5]
 * (proposal_layer_tf_bus.py:144-146).  total_host must equal sum(roi_counts). */
WSSDL_API int wssdl_proposal_compact(const float *rois_padded, const int32_t *roi_counts, int N,
                           int post_nms_topN, float *rois_out, int total_host,
                           wssdl_stream_t stream);

/* ---------------------------------------------------------------------- a5 ---
 * anchor-target layer, rpn_msr/anchor_target_layer_tf_bus.py:19-303 / :328-628,
 * split at the two random sub-samplings (:202-217) so that the host layer can
 * either draw them from numpy's legacy RandomState (bit-identical to the
 * reference) or let the device do it.
 *
 * Stage 1  wssdl_anchor_labels: for each of the first n_images images: inside
 *   filter (:100-105), f64 IoU vs positive gt / intersection ratio vs background
 *   gt, labels before sub-sampling (:115-199).
 *   labels_pre [n_images, H*W*A] i8 in anchor order (h,w,a): -1/0/1 (outside = -1)
 *   argmax_gt  [n_images, H*W*A] i32: row of gt_boxes the targets regress to (-1 outside)
 *   counts     [n_images, 4] i32: (#inside, #fg, #bg, 0)
 * Stage 2a wssdl_anchor_subsample_device: random sub-sampling on the device
 *   (counter-based hash of (seed, image, anchor); same distribution as
 *   npr.choice(replace=False), not the same stream).  In place on labels_pre.  Given stage 1's
 *   counts the fg and the bg draw of an image run side by side (the bg quota only needs #fg).
 * Stage 2b (reference RNG) happens on the host: see
 *   wssdl_bus_amd/rpn_msr/anchor_target_layer_tf_bus.py.
 * Stage 3  wssdl_anchor_targets: final labels -> the four output blobs
 *   (:219-299): rpn_labels [n_out,1,A*H,W] f32, bbox_targets / inside / outside
 *   weights [n_out,4A,H,W] f32.  Images n_images..n_out-1 are the all-ignore
 *   weak images of the joint layer (:613-626) / anchor_target_layer_ws (:306-325). */
WSSDL_API size_t wssdl_anchor_workspace_bytes(int n_images);
WSSDL_API int wssdl_anchor_labels(const float *gt_boxes, int max_gt, const int32_t *num_gt_boxes,
                        const float *im_info, int im_info_stride, int n_images, int H, int W,
                        const double *base_anchors_host, int A, int feat_stride, int dataset,
                        double positive_overlap, double negative_overlap, int clobber_positives,
                        int8_t *labels_pre, int32_t *argmax_gt, int32_t *counts,
                        void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_anchor_subsample_device(int8_t *labels, int n_images, int total_anchors,
                                  int rpn_batchsize, double fg_fraction, uint64_t seed,
                                  const int32_t *counts /* of stage 1, or NULL: fg then bg in sequence */,
                                  wssdl_stream_t stream);
WSSDL_API int wssdl_anchor_targets(const int8_t *labels, const int32_t *argmax_gt, const float *gt_boxes,
                         int max_gt, int n_images, int n_out, int H, int W,
                         const double *base_anchors_host, int A, int feat_stride,
                         const float *inside_weights_host /* [4] */, double positive_weight,
                         float *rpn_labels, float *bbox_targets, float *inside_w, float *outside_w,
                         wssdl_stream_t stream);

/* --------------------------------------------------------------------- a10 ---
 * proposal-target layer, rpn_msr/proposal_target_layer_tf_bus.py:15-295.
 * Stage 1  wssdl_roi_gt_assign: rois [R,5] f32 x positive gt of each roi's image:
 *   f64 IoU (utils/bbox.pyx), max_overlap [R] f64, assignment [R] i32 (row of
 *   gt_boxes of that image; first maximum, numpy argmax).  (:233-238)
 * Stage 2  sampling (:241-262): on the host with the reference's numpy stream, or
 *   wssdl_roi_sample_device: per image images[s] draw min(fg_rois_per_image, #fg)
 *   rows with max_overlap >= fg_thresh and min(rois_per_image - n_fg, #bg) rows with
 *   bg_thresh_lo <= max_overlap < bg_thresh_hi, uniformly without replacement
 *   (counter-based hash of (seed, image, row); same distribution as npr.choice,
 *   not the same stream).  keep / is_fg [n_sample_images, rois_per_image]: rows of
 *   cand in candidate order, fg first, padded with -1; counts [n_sample_images, 2]
 *   = (n_fg, n_bg).  Rows of other images (batch index != images[s]) are ignored.
 * Stage 3  wssdl_roi_targets: for the kept rows: labels (bg clamped to 0, :265),
 *   bbox_transform in f32 (:220), expansion to 4*num_classes with inside/outside
 *   weights (:187-210, :89).  keep [n_keep] i32 indexes rois; is_fg [n_keep] u8.  A negative keep
 *   entry is a padding slot of a fixed-shape list (wssdl_roi_sample_device pads with -1 when an
 *   image runs short of candidates): its output row is (-1,0,0,0,0), label -1, zero targets and
 *   weights, so that consumers can run on the full shape without reading the counts back.
 *   normalize_host != NULL = cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED (:221-224): 8 host doubles, BBOX_NORMALIZE_MEANS
 *   then BBOX_NORMALIZE_STDS; targets = (f32 target - mean) / std evaluated in f64 and rounded to f32 once, as NumPy does
 *   when the f32 targets meet np.array(MEANS) / np.array(STDS). */
WSSDL_API int wssdl_roi_gt_assign(const float *rois, int R, const float *gt_boxes, int max_gt,
                        const int32_t *num_pos_boxes, int n_images, double *max_overlap,
                        int32_t *assignment, wssdl_stream_t stream);
/* Stage 0 (device path): candidate rows = rois [R,5] followed, when append_gt, by the max_gt gt
 *   slots of each image images[s] (:44-50); a slot's batch index is the image for its positive
 *   boxes (the first num_pos rows) and -1 otherwise.  cand [R + n_sample_images*max_gt, 5];
 *   num_pos_boxes [n_images] = boxes with class != 0 among the num_gt_boxes valid rows (:40-42). */
WSSDL_API int wssdl_roi_candidates(const float *rois, int R, const float *gt_boxes, int max_gt,
                         const int32_t *num_gt_boxes, int n_images, const int32_t *images,
                         int n_sample_images, int append_gt, float *cand, int32_t *num_pos_boxes,
                         wssdl_stream_t stream);
/* The device-sampled layer as one call: stages 0-3 (candidates, assignment, fg / bg draw, rows +
 *   targets) launched back to back, the intermediates carved from `workspace`
 *   (wssdl_proposal_target_device_workspace_bytes).  Outputs have the fixed shape
 *   n_sample_images * rois_per_image rows (rois_out [.,5], labels [.] f32, the three [., 4*num_classes]);
 *   an image that runs short of candidates leaves padding rows as described above. */
WSSDL_API size_t wssdl_proposal_target_device_workspace_bytes(int R, int n_images, int max_gt,
                                                    int n_sample_images, int rois_per_image, int append_gt);
WSSDL_API int wssdl_proposal_target_device(
    const float *rois, int R, const float *gt_boxes, int max_gt, const int32_t *num_gt_boxes, int n_images,
    const int32_t *images, int n_sample_images, int append_gt, int rois_per_image, int fg_rois_per_image,
    double fg_thresh, double bg_thresh_hi, double bg_thresh_lo, uint64_t seed, int num_classes,
    const float *inside_weights_host, const double *normalize_host, float *rois_out, float *labels, float *bbox_targets,
    float *inside_w, float *outside_w, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_roi_sample_device(const float *cand, const double *max_overlap, int Rc,
                            const int32_t *images, int n_sample_images, int rois_per_image,
                            int fg_rois_per_image, double fg_thresh, double bg_thresh_hi,
                            double bg_thresh_lo, uint64_t seed, int32_t *keep, uint8_t *is_fg,
                            int32_t *counts, wssdl_stream_t stream);
WSSDL_API int wssdl_roi_targets(const float *rois, const int32_t *keep, const uint8_t *is_fg, int n_keep,
                      const int32_t *assignment, const float *gt_boxes, int max_gt, int num_classes,
                      const float *inside_weights_host /* [4] */,
                      const double *normalize_host /* NULL, or means[4] then stds[4] */, float *rois_out,
                      float *labels, float *bbox_targets, float *inside_w, float *outside_w,
                      wssdl_stream_t stream);

/* ---------------------------------------------------------------- a11, a12 ---
 * RoiPool forward: roi_pooling_op.cc:31-52 (op), kernels roi_pooling_op_gpu.cu.cc:20-85
 * (rounding CUDA) / roi_pooling_op.cc:137-196 (rounding CPU).  N = WSSDL_ROI_BATCH_UNKNOWN (-1) = "batch size unknown": the
 * reference's ROIPoolForwardLaucher is not told it (roi_pooling_op_gpu.h:17-21) and never range-checks
 * rois[:, 0]; then only a negative batch index makes a RoI empty and the caller vouches for the rest (with N > 0 an
 * index >= N makes an empty RoI too; N == 0 with R > 0 returns WSSDL_ERR_INVALID_ARGUMENT).
 * bottom [N,H,W,C] f32, rois [R,5] f32, top [R,PH,PW,C] f32, argmax [R,PH,PW,C] i32
 * (flat NHWC index within the roi's image, -1 for an empty bin). */
WSSDL_API int wssdl_roi_pool_forward(const float *bottom, int N, int H, int W, int C, const float *rois,
                           int R, int pooled_h, int pooled_w, float spatial_scale, int rounding,
                           float *top, int32_t *argmax, wssdl_stream_t stream);
/* RoiPoolGrad: roi_pooling_op.cc:54-63 (op), roi_pooling_op_gpu.cu.cc:114-190 ==
 * roi_pooling_op.cc:383-458.  bottom_diff [N,H,W,C] f32 is fully written (no
 * pre-zeroing needed).  Deterministic: per element the f32 sum runs in the
 * reference's order roi^, ph^, pw^, so results are bit-identical to it. */
WSSDL_API int wssdl_roi_pool_backward(const float *top_diff, const int32_t *argmax, const float *rois,
                            int R, int N, int H, int W, int C, int pooled_h, int pooled_w,
                            float spatial_scale, float *bottom_diff, wssdl_stream_t stream);
/* The same op with a caller-owned workspace (wssdl_roi_pool_backward_workspace_bytes(R, N, H, W, pooled_h,
 * pooled_w); a TF launcher takes it from OpKernelContext::allocate_temp): the list-driven kernels of the training
 * path -- per (image, tile) the candidate bins in the reference's order, one wave per (image, tile, 128 channels)
 * -- reading the i32 arg-max.  Same bits as wssdl_roi_pool_backward; 1.4x faster on a train-sized RoI list (1.15 -> 0.80 ms).
 * Shapes the lists do not cover (C not a power of two, pooled size > 8, workspace NULL or short) run the kernel
 * of wssdl_roi_pool_backward. */
WSSDL_API int wssdl_roi_pool_backward_ws(const float *top_diff, const int32_t *argmax, const float *rois,
                            int R, int N, int H, int W, int C, int pooled_h, int pooled_w,
                            float spatial_scale, float *bottom_diff, void *workspace, size_t workspace_bytes,
                            wssdl_stream_t stream);

/* ------------------------------------------------- a11, a12: training path ---
 * The same pair with a ONE-BYTE arg-max.  In the reference the arg-max tensor never leaves the
 * op pair: networks/network.py:206-210 keeps only top_data and the registered gradient
 * (roi_pooling_op_grad.py:24-44) hands argmax straight to RoiPoolGrad, so its encoding is an
 * internal contract.  Both kernels are HBM-bound on top + argmax, and 4 + 1 bytes per element
 * instead of 4 + 4 is a 37 % cut of that traffic (re-reads of the backward included).
 *   code = (h - hstart) << 4 | (w - wstart)   within the bin's clipped window
 *          (roi_pooling_op_gpu.cu.cc:51-64 / roi_pooling_op.cc:167-176); 0xff = empty bin (-1).
 * Supported (wssdl_roi_pool_compact_supported == 1) when C % 32 == 0 and every window of a RoI
 * inside the map fits 15 x 16 cells: ceil((H+1)/pooled_h)+1 <= 15, ceil((W+1)/pooled_w)+1 <= 16
 * (7x7 bins: maps up to 97 x 104 cells).  A RoI reaching far outside the map can exceed that:
 * the forward then sets *overflow (optional device int32, OR-ed, never cleared) and that RoI's
 * codes are invalid -- callers feeding unclipped RoIs use the i32 pair above.
 * top / bottom_diff are bit-identical to the i32 pair; wssdl_roi_argmax_expand rebuilds the
 * reference's i32 argmax from the codes. */
WSSDL_API int wssdl_roi_pool_compact_supported(int H, int W, int C, int pooled_h, int pooled_w);
WSSDL_API int wssdl_roi_pool_forward_compact(const float *bottom, int N, int H, int W, int C,
                           const float *rois, int R, int pooled_h, int pooled_w, float spatial_scale,
                           int rounding, float *top, uint8_t *argmax8, int32_t *overflow,
                           wssdl_stream_t stream);
/* The same forward with the RoI geometry taken from a table: wssdl_roi_pool_forward_windows writes one
 * 32-byte entry per (roi, bin row) -- batch index, the clipped window rows, the window columns of the 7
 * bins (roi_pooling_op_gpu.cu.cc:36-64) -- and raises *overflow like the forward would;
 * wssdl_roi_pool_forward_compact_windows reads it with scalar loads instead of recomputing it in every
 * wave (~1/5 of the kernel's vector instructions).  7 x 7 bins, C % 256 == 0;
 * wssdl_roi_pool_forward_windows_bytes returns 0 for shapes it does not take (use the call above). */
WSSDL_API size_t wssdl_roi_pool_forward_windows_bytes(int R, int H, int W, int C, int pooled_h, int pooled_w);
WSSDL_API int wssdl_roi_pool_forward_windows(const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                                   int pooled_w, float spatial_scale, int rounding, void *table,
                                   size_t table_bytes, int32_t *overflow, wssdl_stream_t stream);
WSSDL_API int wssdl_roi_pool_forward_compact_windows(const float *bottom, int N, int H, int W, int C,
                                           const float *rois, int R, int pooled_h, int pooled_w,
                                           float spatial_scale, int rounding, const void *table,
                                           float *top, uint8_t *argmax8, wssdl_stream_t stream);
/* The same forward for MANY proposals per image (large, heavily overlapping bin windows: the weak images' 2000
 * proposals each), where scanning every cell of every window (sum of window areas x C) is the cost.  Per call the
 * feature map is reduced once to block-maximum tables -- the first maximum, by the reference's own scan
 * (roi_pooling_op_gpu.cu.cc:66-79: (h, w) order, strict >), of the k x k block at every cell, k = 2, 3, 4 -- and a
 * bin whose window is covered by the four k x k blocks at its corners reads four table entries instead of its
 * cells; "maximum value, then smallest (h, w)" is associative and idempotent, so top and arg-max are the same bits.
 * Other bins (and every bin when the map holds a -0.0, which equals +0.0 in the reference's compare) are scanned
 * cell by cell.  Order of calls, same stream: wssdl_roi_pool_forward_windows_blocks (the window table -- it also
 * carries the block choice per bin and the sort key of the bin row -- and the counters of `blocks` cleared),
 * wssdl_roi_pool_forward_blocks_prepare (tables + the bin rows sorted by (image, first window row) so that an XCD's
 * L2 holds the band of the tables its waves read), wssdl_roi_pool_forward_compact_blocks.
 * `blocks` = wssdl_roi_pool_forward_blocks_bytes() bytes, 256-byte aligned (0: shape not supported -- 7 x 7 bins,
 * C % 256 == 0, H, W in 4..255).  wssdl_roi_pool_forward_blocks_auto = 1 where the library suggests this form for
 * the launch shape ("roi_fwd_blocks": -1 that rule, 0 never, 1 wherever supported; "roi_fwd_blocks_sort" = 0 keeps
 * the bin rows in RoI order; "roi_fwd_blocks_parts" = 1, 2 (default), 4 or 7 waves per bin row).  The order array is
 * only as good as the call sequence: built on counters that wssdl_roi_pool_forward_windows_blocks did not clear it is
 * not a permutation, and rows of `top` stay unwritten (its entries are clamped, so nothing is read out of range). */
WSSDL_API size_t wssdl_roi_pool_forward_blocks_bytes(int R, int N, int H, int W, int C, int pooled_h, int pooled_w);
WSSDL_API int wssdl_roi_pool_forward_blocks_auto(int R, int N, int H, int W, int C, int pooled_h, int pooled_w);
WSSDL_API int wssdl_roi_pool_forward_windows_blocks(const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                                          int pooled_w, float spatial_scale, int rounding, void *table,
                                          size_t table_bytes, int32_t *overflow, void *blocks, size_t blocks_bytes,
                                          wssdl_stream_t stream);
WSSDL_API int wssdl_roi_pool_forward_blocks_prepare(const float *bottom, int N, int H, int W, int C, int R,
                                          int pooled_h, int pooled_w, const void *table, void *blocks,
                                          size_t blocks_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_roi_pool_forward_compact_blocks(const float *bottom, int N, int H, int W, int C, int R,
                                          int pooled_h, int pooled_w, const void *table, const void *blocks,
                                          size_t blocks_bytes, float *top, uint8_t *argmax8,
                                          wssdl_stream_t stream);
/* Backward in two calls.  The lists that drive it depend on the RoIs and the shapes only, so
 * wssdl_roi_pool_backward_prepare can run as soon as the RoIs exist (e.g. right behind the
 * forward): per (image, tile) it builds the stream of candidate bins in the reference's
 * summation order (roi, ph, pw) with the reference's in_roi / candidate-bin tests
 * (roi_pooling_op_gpu.cu.cc:141-151,169-177) evaluated once, and sorts the tiles by work.
 * It writes the plan it chose to *plan_host (a HOST int; -1 = not supported or no workspace:
 * the backward then runs a kernel that filters the RoIs itself).  wssdl_roi_pool_backward_compact
 * with that plan and the same workspace is then ONE kernel: one wave per (image, tile, 128
 * channels), heaviest tiles first.  workspace_bytes = 0 when the pooled size is not supported.
 * bottom_diff is bit-identical on every path. */
WSSDL_API size_t wssdl_roi_pool_backward_workspace_bytes(int R, int N, int H, int W, int pooled_h,
                            int pooled_w);
/* number of plans wssdl_roi_pool_backward_prepare can choose from ("roi_bwd_plan" = 0 .. count-1) */
WSSDL_API int wssdl_roi_pool_backward_plan_count(void);
/* byte offset, inside the workspace, of the int32[4] status block the prepare step fills:
 * [0] = 64-byte records in use, [1] = error flags (non-zero: the lists do not fit the workspace and
 * bottom_diff would be short -- cannot happen with wssdl_roi_pool_backward_workspace_bytes; the host
 * layer checks it with its other deferred flags). */
WSSDL_API size_t wssdl_roi_pool_backward_status_offset(int R, int N, int H, int W, int pooled_h, int pooled_w);
WSSDL_API int wssdl_roi_pool_backward_prepare(const float *rois, int R, int N, int H, int W, int C,
                            int pooled_h, int pooled_w, float spatial_scale, int rounding,
                            void *workspace, size_t workspace_bytes, int32_t *plan_host,
                            wssdl_stream_t stream);
WSSDL_API int wssdl_roi_pool_backward_compact(const float *top_diff, const uint8_t *argmax8,
                            const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                            int pooled_w, float spatial_scale, int rounding, float *bottom_diff,
                            void *workspace, size_t workspace_bytes, int plan, wssdl_stream_t stream);
/* Split form of the list-driven backward: DETERMINISTIC BUT NOT BIT-ORDERED.  A launch with few images is bound
 * by the longest slot chain one wave walks alone (every RoI of a 2000-RoI weak image reaches the tiles at the
 * image's centre), not by bandwidth: each tile's stream is cut into `segments` pieces walked by separate waves
 * (segment 0 into bottom_diff, the others into scratch) and the pieces are added in segment order.  The f32 sum
 * per element is then associated differently from roi_pooling_op_gpu.cu.cc:132-186 (roi^, ph^, pw^): the same
 * result on every run, every element within 1e-6 of ..._backward_compact relative to its own sum of |terms|
 * (measured ~2e-8) and the tensor within 1e-5 of its scale (north_star's tolerance for RoI pooling), not
 * bit-identical to it.  segments = 1 is the exact walk.
 * wssdl_roi_pool_backward_split_segments suggests a count by launch shape (8; 4 for two images with 2048 and for
 * three images with up to 3072 (image, channel) pairs, where the gain is the larger tiles' fewer re-read bytes; or
 * 1 = keep the exact walk: more than 4 images, fewer than 1000 RoIs per image, or more (image, channel) pairs than
 * that -- enough waves that the chains no longer bind); wssdl_roi_pool_backward_split_plan = the plan to prepare the lists with for the split form
 * (set "roi_bwd_plan" to it around ..._backward_prepare: large tiles, since the chains no longer matter); same
 * workspace as ..._backward_compact. */
WSSDL_API int wssdl_roi_pool_backward_split_segments(int R, int N, int H, int W, int C);
WSSDL_API int wssdl_roi_pool_backward_split_plan(void);
WSSDL_API size_t wssdl_roi_pool_backward_split_scratch_bytes(int N, int H, int W, int C, int segments);
WSSDL_API int wssdl_roi_pool_backward_compact_split(const float *top_diff, const uint8_t *argmax8,
                            const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                            int pooled_w, float spatial_scale, int rounding, float *bottom_diff,
                            void *workspace, size_t workspace_bytes, int plan, int segments,
                            void *scratch, size_t scratch_bytes, wssdl_stream_t stream);
/* Bin-owner form of the list-driven backward (round 5): DETERMINISTIC BUT NOT BIT-ORDERED, like the split form.
 * The exact walk lists a bin in every tile its window touches, so the launch reads top_diff / the codes ~1.5 x.  Here a
 * wave accumulates into a region that reaches past its tile (a halo of 1-3 cells) and a bin is listed once, by the tile
 * of its window's first cell (windows larger than the region continue in the next tile); the halos go to `scratch` and a
 * second kernel adds them to their owners in a fixed order (own, left, upper, upper-left).  Same result on every run;
 * the f32 sum per element is associated differently from roi_pooling_op_gpu.cu.cc:132-186, within the split form's
 * bounds.  Call ..._owner_prepare (instead of ..._backward_prepare; same workspace size) with the owner plan, then
 * ..._compact_owner with the same plan.  ..._owner_plan suggests a plan by launch shape, -1 = keep the exact walk. */
WSSDL_API int wssdl_roi_pool_backward_owner_plan_count(void);
WSSDL_API int wssdl_roi_pool_backward_owner_plan(int R, int N, int H, int W, int C);
/* the same rule for a given pooled size: -1 as well when the owner form does not take the launch (pooled_h or
 * pooled_w > 8, R * pooled_h * pooled_w * C >= 2^30 elements, N * H * W * C >= 2^31) -- what a caller that pools
 * other sizes than 7 x 7 must ask (wssdl_roi_pool_backward_owner_plan assumes 7 x 7) */
WSSDL_API int wssdl_roi_pool_backward_owner_plan_for(int R, int N, int H, int W, int C, int pooled_h, int pooled_w);
WSSDL_API size_t wssdl_roi_pool_backward_owner_scratch_bytes(int N, int H, int W, int C, int owner_plan);
WSSDL_API int wssdl_roi_pool_backward_owner_prepare(const float *rois, int R, int N, int H, int W, int C,
                            int pooled_h, int pooled_w, float spatial_scale, int rounding,
                            void *workspace, size_t workspace_bytes, int owner_plan, wssdl_stream_t stream);
WSSDL_API int wssdl_roi_pool_backward_compact_owner(const float *top_diff, const uint8_t *argmax8,
                            const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                            int pooled_w, float spatial_scale, int rounding, float *bottom_diff,
                            void *workspace, size_t workspace_bytes, int owner_plan,
                            void *scratch, size_t scratch_bytes, wssdl_stream_t stream);
/* The owner form with SEGMENTS (round 6), for launches with few (image, channel) pairs -- the alternating weak step's two
 * images, the reference's default 1 + 2 batch -- where one wave per (image, tile, 128 channels) leaves most of the chip
 * waiting for the longest streams: a tile's stream is cut into `segments` pieces walked by a wave each, every piece into
 * a region buffer of its own ([segments][N * tiles][region cells][C] f32 = ..._owner_split_scratch_bytes), and a merge
 * pass writes bottom_diff = the sum over segments of own cells + neighbours' halos, in a fixed order.  Same lists
 * (wssdl_roi_pool_backward_owner_prepare), deterministic, the owner form's tolerance (every (bin, cell) pair applied
 * exactly once).  wssdl_roi_pool_backward_owner_segments suggests the count by launch shape (1 = the plain owner form);
 * "roi_bwd_owner_segments" overrides it. */
WSSDL_API int wssdl_roi_pool_backward_owner_segments(int R, int N, int H, int W, int C);
WSSDL_API size_t wssdl_roi_pool_backward_owner_split_scratch_bytes(int N, int H, int W, int C, int owner_plan, int segments);
WSSDL_API int wssdl_roi_pool_backward_compact_owner_split(const float *top_diff, const uint8_t *argmax8,
                            const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                            int pooled_w, float spatial_scale, int rounding, float *bottom_diff,
                            void *workspace, size_t workspace_bytes, int owner_plan, int segments,
                            void *scratch, size_t scratch_bytes, wssdl_stream_t stream);
/* The bin-owner form reading the reference op's OWN arg-max layout (i32 flat index, roi_pooling_op_gpu.cu.cc:71-79): list
 * building + walk + halo merge in one call, owner plans 0 and 1.  Opt-in: RoiPoolGrad's declared contract
 * (wssdl_roi_pool_backward / _ws) stays the exact walk, bit for bit; this one is deterministic within the owner form's
 * bounds.  workspace = wssdl_roi_pool_backward_workspace_bytes, scratch = ..._owner_scratch_bytes. */
WSSDL_API int wssdl_roi_pool_backward_owner_i32(const float *top_diff, const int32_t *argmax, const float *rois, int R, int N,
                            int H, int W, int C, int pooled_h, int pooled_w, float spatial_scale, float *bottom_diff,
                            void *workspace, size_t workspace_bytes, int owner_plan, void *scratch, size_t scratch_bytes,
                            wssdl_stream_t stream);
WSSDL_API int wssdl_roi_argmax_expand(const uint8_t *argmax8, const float *rois, int R, int H, int W,
                            int C, int pooled_h, int pooled_w, float spatial_scale, int rounding,
                            int32_t *argmax, wssdl_stream_t stream);

/* --------------------------------------------------------------------- a13 ---
 * Multi-task loss of the supervised images and its gradients: fast_rcnn/train_bus.py:186-192 /
 * :605-610 (rpn_cross_entropy), :203-210 / :613-620 (rpn_loss_box, threshold |d| < 1 with the
 * sigma = 3 pieces as written), :218 / :623-630 (cross_entropy), :231-235 / :641-647 (loss_box).
 * Layouts are the layers' own (network.py:196-291): rpn_cls_score [n_images,H,W,2A] raw scores
 * (channel c*A + a: the reshape to [n, A*H, W, 2] is an index map), rpn_labels [n_images,1,A*H,W] i32
 * in {-1,0,1}, rpn_bbox_pred [n_images,H,W,4A], rpn targets / weights [n_images,4A,H,W]; the box term
 * covers the first n_box_images images (combined mode: IMS_PER_BATCH; the others get zero gradient).
 * cls_score [>= n_rows, num_classes], labels [n_rows] i32 (-1 = padding row: not counted),
 * bbox_pred [>= n_rows, 4*num_classes], bbox targets / weights [n_rows, 4*num_classes].
 * forward : losses[4] = rpn_cross_entropy, rpn_loss_box, cross_entropy, loss_box (f32; sums in f64,
 *           combined in a fixed order); the workspace keeps the two counts for the backward.
 * backward: grad_losses [4] f32 (device) = upstream gradient of each term; writes the gradients of
 *           the four prediction tensors in full (rows_total >= n_rows rows of the two head outputs:
 *           rows past n_rows and padding rows get zeros). */
WSSDL_API size_t wssdl_multi_task_loss_workspace_bytes(int n_images, int H, int W, int A);
WSSDL_API int wssdl_multi_task_loss_forward(
    const float *rpn_cls_score, const int32_t *rpn_labels, const float *rpn_bbox_pred,
    const float *rpn_bbox_targets, const float *rpn_inside_w, const float *rpn_outside_w, int n_images,
    int n_box_images, int H, int W, int A, const float *cls_score, const int32_t *labels,
    const float *bbox_pred, const float *bbox_targets, const float *bbox_inside_w,
    const float *bbox_outside_w, int n_rows, int num_classes, float *losses, void *workspace,
    size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_multi_task_loss_backward(
    const float *rpn_cls_score, const int32_t *rpn_labels, const float *rpn_bbox_pred,
    const float *rpn_bbox_targets, const float *rpn_inside_w, const float *rpn_outside_w, int n_images,
    int n_box_images, int H, int W, int A, const float *cls_score, const int32_t *labels,
    const float *bbox_pred, const float *bbox_targets, const float *bbox_inside_w,
    const float *bbox_outside_w, int n_rows, int rows_total, int num_classes, const float *grad_losses,
    const void *workspace, float *grad_rpn_cls_score, float *grad_rpn_bbox_pred, float *grad_cls_score,
    float *grad_bbox_pred, wssdl_stream_t stream);

/* ---------------------------------------------------------------------- f1 ---
 * MIL bag-instance selection: mil/core.py:11-46 (get_bag_logit) with the selectors
 * get_mal_max_logit :60-69, get_ben_max_logit :49-57, get_mass_max_logit :88-96.
 * instance_logits [R,num_classes] f32; the bag of row r is (int)(bag_of_row[r*bag_stride] -
 * bag_offset) (e.g. the batch-index column of the rois blob minus IMS_PER_BATCH,
 * fast_rcnn/train_bus.py:653); bag_labels [n_bags] i32.  Bags labelled 1 use selector_label1,
 * the others selector_other (train_bus.py:241,655).  row_out [n_bags] i32 receives the row to
 * gather (first extremum; -1 for an empty bag), count_out (optional) the instances per bag. */
enum wssdl_mil_selector { WSSDL_MIL_MAL_MAX = 0, WSSDL_MIL_BEN_MAX = 1, WSSDL_MIL_MASS_MAX = 2 };
WSSDL_API int wssdl_mil_select(const float *instance_logits, int R, int num_classes,
                     const float *bag_of_row, int bag_stride, float bag_offset,
                     const int32_t *bag_labels, int n_bags, int selector_label1,
                     int selector_other, int32_t *row_out, int32_t *count_out,
                     wssdl_stream_t stream);

/* The MIL term as one op: selection as above, then fast_rcnn/train_bus.py:246-260 / :657-671 --
 * softmax CE of the selected instance against the bag label, weighted by class_weights_host[label]
 * ([0, WS_MAL_PCT, 1 - WS_MAL_PCT], :252,:664; host array [num_classes]) and by `scale`
 * (1 - 0.99 * 0.9^floor(step/2000) or the constant, :248,:659), mean over the n_bags bags.  An empty
 * bag contributes 0 and still counts in the mean.  loss [1] f32; row_out [n_bags] i32 and bag_loss
 * [n_bags] f32 are outputs the backward (row_out) and the caller may read.  The backward writes the
 * gradient of the whole [R,num_classes] logits block (zeros except the selected rows);
 * grad_loss [1] f32 on the device. */
WSSDL_API int wssdl_mil_loss_forward(const float *instance_logits, int R, int num_classes,
                           const float *bag_of_row, int bag_stride, float bag_offset,
                           const int32_t *bag_labels, int n_bags, int selector_label1,
                           int selector_other, const float *class_weights_host, float scale,
                           float *loss, int32_t *row_out, float *bag_loss, wssdl_stream_t stream);
WSSDL_API int wssdl_mil_loss_backward(const float *instance_logits, int R, int num_classes,
                            const float *bag_of_row, int bag_stride, float bag_offset,
                            const int32_t *bag_labels, int n_bags, const int32_t *rows,
                            const float *class_weights_host, float scale, const float *grad_loss,
                            float *grad_logits, wssdl_stream_t stream);

/* ---------------------------------------------------------------------- f4 ---
 * The pinnable half of the host image path, on the device.  utils/blob.py:34-79
 * (prep_im_for_blob), :19-32 (im_list_to_blob), roi_data_layer/minibatch_bus.py:269-272 (grey plane
 * stacked three times, horizontal flip), datasets/imdb.py:106-121 (boxes of flipped images).
 * skimage.transform.resize (blob.py:74-77) sits in the middle: the steps around it are pinned by
 * fixtures from the reference's own blob.py, the resize follows the published algorithm of the
 * release the reference's README pins (scikit-image 0.14.2; the library is absent: parity unpinned):
 *   wssdl_image_prep     gray [h, row_stride] u8 -> out [h,w,3] f32 = what the reference hands to
 *                        the resize: flip, /255, optional brightness (+delta, clip to [0,1]),
 *                        optional contrast ((x - mean(x)) * factor + mean(x), clip), - pixel_mean/255.
 *                        delta / factor are the values the caller drew (the reference:
 *                        np.random.uniform, blob.py:50,55).  workspace: wssdl_image_prep_workspace_bytes.
 *   wssdl_image_to_blob  im [h,w,3] f64 (im_is_f64) or f32 = the resize's output -> blob
 *                        [n_images,Hmax,Wmax,3] f32, image `index`: x / scale (divide != 0: ResNet,
 *                        scale = pixel_std/255) or x * scale (VGG: 255), zero outside [h,w].
 *   wssdl_flip_boxes     in place on boxes [n, stride >= 4] f32: x1' = width - x2 - 1, x2' = width - x1 - 1.
 *   wssdl_image_resize   skimage.transform.resize(im, [rows, cols]) as blob.py:74-77 / fast_rcnn/test_bus.py:54-56 call
 *                        it (0.14 defaults: order 1, mode 'constant', cval 0, clip, no anti-aliasing):
 *                        im [h,w,channels] f32 or f64 (im_is_f64) -> out [rows,cols,channels] f64.
 *                        out(row, col) interpolates the input bilinearly at
 *                        (h/rows * (row + 0.5) - 0.5, w/cols * (col + 0.5) - 0.5); samples outside the
 *                        image read 0; the result is clipped to the input's [min, max].
 *   wssdl_image_warp     the general form = skimage.transform.warp(im, matrix, output_shape=(rows, cols),
 *                        order=1, mode, cval, clip): `matrix` = 9 doubles in HOST memory, row-major 3x3
 *                        inverse map (output (col,row,1) -> input (col,row,w)); serves rotate()
 *                        (blob.py:39-41) and a caller that brings skimage's own estimated matrix.
 *                        workspace: wssdl_image_warp_workspace_bytes (only read when clip != 0).
 *   wssdl_image_adjust_f64  blob.py:49-60 on the float64 image rotate() returns (weak images with
 *                        cfg.TRAIN.USE_ROTATION, :39-41): im = strided view [h,w,channels] f64 (row_stride in
 *                        elements; the crop :43-47 is a slice), the same brightness / contrast / mean steps as
 *                        wssdl_image_prep in f64 -> out [h,w,channels] f64 contiguous (the resize's input).
 *                        workspace: wssdl_image_prep_workspace_bytes. */
#define WSSDL_WARP_CONSTANT 0    /* mode='constant': cval outside the image */
#define WSSDL_WARP_EDGE 1        /* mode='edge': nearest edge pixel */
WSSDL_API size_t wssdl_image_warp_workspace_bytes(void);
WSSDL_API int wssdl_image_warp(const void *im, int im_is_f64, int h, int w, int channels, const double *matrix,
                     int rows, int cols, int mode, double cval, int clip, double *out, void *workspace,
                     size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_image_resize(const void *im, int im_is_f64, int h, int w, int channels, int rows, int cols,
                     double *out, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_image_adjust_f64(const double *im, int h, int w, int channels, int64_t row_stride,
                     int use_brightness, double brightness_delta, int use_contrast, double contrast_factor,
                     double pixel_mean, double *out, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API size_t wssdl_image_prep_workspace_bytes(void);
WSSDL_API int wssdl_image_prep(const uint8_t *gray, int h, int w, int row_stride, int flipped,
                     int use_brightness, float brightness_delta, int use_contrast,
                     float contrast_factor, double pixel_mean, float *out, void *workspace,
                     size_t workspace_bytes, wssdl_stream_t stream);
WSSDL_API int wssdl_image_to_blob(const void *im, int im_is_f64, int h, int w, double scale, int divide,
                     float *blob, int index, int n_images, int Hmax, int Wmax, wssdl_stream_t stream);
WSSDL_API int wssdl_flip_boxes(float *boxes, int n, int stride, float width, wssdl_stream_t stream);

/* ---------------------------------------------------------------------- f3 ---
 * The post-detection step of the test path, fast_rcnn/test_bus.py:360-401, batched over the classes:
 * for every class j = 1 .. num_classes-1 the rows with scores[r, j] > score_thresh, greedy NMS at
 * nms_thresh with the cpu_nms rule (utils/cython_nms) on (boxes[r, 4j:4j+4], scores[r, j]) in descending
 * score order, then the cap: if more than max_per_image (> 0) detections are left over all classes, only
 * those with a score >= the max_per_image-th largest stay (ties stay, like the reference's `>=`).
 *   scores [R, num_classes] f32, boxes [R, 4 * num_classes] f32 (class-wise decoded boxes);
 *   dets   [num_classes-1, R, 5] f32: row p of class j-1 = (x1, y1, x2, y2, score) of its p-th kept
 *          detection in descending score order; counts [num_classes-1] i32 = rows that survive the cap.
 * One set of launches for all classes, no host read-back.  num_classes <= 65. */
WSSDL_API size_t wssdl_post_detections_workspace_bytes(int R, int num_classes);
WSSDL_API int wssdl_post_detections(const float *scores, const float *boxes, int R, int num_classes,
                     float score_thresh, double nms_thresh, int max_per_image, float *dets,
                     int32_t *counts, void *workspace, size_t workspace_bytes, wssdl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WSSDL_BUS_HIP_H */
