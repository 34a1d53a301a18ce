// RPN proposal layer for gfx950 (MI355X): decode + clip + min-size filter ->
// score ranking / top-K -> greedy NMS -> top post_nms_topN, for all images of a
// step with one launch per stage.
//
// Reference: code/lib/rpn_msr/proposal_layer_tf_bus.py:19-148,
// fast_rcnn/bbox_transform.py:30-77 (bbox_transform_inv, clip_boxes),
// proposal_layer_tf_bus.py:151-156 (_filter_boxes).
//
// decode kernel: one lane per anchor, anchors enumerated (h, w, a) like the
// reference, so lane i reads the 16 bytes rpn_bbox_pred[... + 4*i] -- a perfectly
// coalesced dwordx4 stream -- and the shifted anchor is rebuilt in registers
// from the 9 base anchors (kernel argument) instead of being read from memory.
// All arithmetic is f32 in NumPy's operation order (separate multiply and add);
// exp is evaluated in f64 and rounded once to f32 (NumPy's own f32 exp is only
// accurate to ~2.5 ulp, so decoded boxes are tolerance-level by nature).
#include "order_sort.hip.h"

#include <stdlib.h>

namespace wssdl {

typedef float float4v __attribute__((ext_vector_type(4)));

struct DecodeArgs {
    const float *prob, *pred, *im_info;
    int info_stride, N, H, W, A, stride, from_logits;
    float min_size;
    BaseAnchors base;
    float *boxes;                      // [N, M, 4]
    unsigned long long *keys;          // [N, M] (decode kernel) or the sorted runs (decode + runs kernel)
    int *sorted_index;  long long n_sorted_index;
    int *n_sorted, *cand_fill;
    float *rois_padded;  long long n_rois_floats;
};

// the layer's buffers that later kernels count on being initialised (four memset launches folded into the
// first kernel): sorted_index = -1, the per-image counters = 0, rois_padded = 0
__device__ __forceinline__ void proposal_init_buffers(const DecodeArgs &a) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < a.n_sorted_index; i += nthreads) a.sorted_index[i] = -1;
    for (long long i = tid; i < a.n_rois_floats; i += nthreads) a.rois_padded[i] = 0.0f;
    if (tid < a.N) { a.n_sorted[tid] = 0;  a.cand_fill[tid] = 0; }
}

// anchor i of image n: its decoded, clipped box (stored) and its sort key (0 when the size filter drops it)
__device__ __forceinline__ unsigned long long decode_anchor(const DecodeArgs &g, int n, int i) {
    const int A = g.A, W = g.W, stride = g.stride;
    const int cell = i / A, a = i - cell * A;
    const int h = cell / W, w = cell - h * W;
    const float im_h = g.im_info[n * g.info_stride + 0];
    const float im_w = g.im_info[n * g.info_stride + 1];
    const float im_scale = g.im_info[n * g.info_stride + 2];
    const size_t cbase = (size_t)n * g.H * W + cell;
    // fg probability: channels A..2A-1 (proposal_layer_tf_bus.py:86)
    float score = g.prob[cbase * (2 * A) + A + a];
    if (g.from_logits) {
        // fused reshape -> softmax -> reshape (network.py:283-291,398-404): the pair of anchor
        // a is (score[a], score[A+a]); softmax as exp(x - max) / sum like TF's kernel
        const float bg = g.prob[cbase * (2 * A) + a];
        const float mx = fmaxf(bg, score);
        const float e0 = (float)exp((double)(bg - mx)), e1 = (float)exp((double)(score - mx));
        score = e1 / (e0 + e1);
    }
    const float4v d = *reinterpret_cast<const float4v *>(g.pred + (cbase * A + a) * 4);
    // anchors are integer-valued doubles cast to f32 (bbox_transform.py:34)
    const float ax1 = (float)(g.base.v[a][0] + (double)(stride * w));
    const float ay1 = (float)(g.base.v[a][1] + (double)(stride * h));
    const float ax2 = (float)(g.base.v[a][2] + (double)(stride * w));
    const float ay2 = (float)(g.base.v[a][3] + (double)(stride * h));
    float aw = ax2 - ax1;  aw = aw + 1.0f;
    float ah = ay2 - ay1;  ah = ah + 1.0f;
    float hx = 0.5f * aw, hy = 0.5f * ah;
    const float cx = ax1 + hx, cy = ay1 + hy;
    float pcx = d.x * aw;  pcx = pcx + cx;
    float pcy = d.y * ah;  pcy = pcy + cy;
    const float pw = (float)exp((double)d.z) * aw;
    const float ph = (float)exp((double)d.w) * ah;
    const float hpw = 0.5f * pw, hph = 0.5f * ph;
    float x1 = pcx - hpw, y1 = pcy - hph, x2 = pcx + hpw, y2 = pcy + hph;
    // clip_boxes (bbox_transform.py:63-77): max(min(v, im-1), 0)
    const float xm = im_w - 1.0f, ym = im_h - 1.0f;
    x1 = fmaxf(fminf(x1, xm), 0.0f);
    y1 = fmaxf(fminf(y1, ym), 0.0f);
    x2 = fmaxf(fminf(x2, xm), 0.0f);
    y2 = fmaxf(fminf(y2, ym), 0.0f);
    // _filter_boxes (proposal_layer_tf_bus.py:123,151-156)
    const float ms = g.min_size * im_scale;
    float bw = x2 - x1;  bw = bw + 1.0f;
    float bh = y2 - y1;  bh = bh + 1.0f;
    const bool valid = (bw >= ms) && (bh >= ms);
    float4v o;  o.x = x1;  o.y = y1;  o.z = x2;  o.w = y2;
    const int M = g.H * W * A;
    *reinterpret_cast<float4v *>(g.boxes + ((size_t)n * M + i) * 4) = o;
    return valid ? score_key(score, (unsigned)i) : 0ull;
}

__global__ __launch_bounds__(256) void proposal_decode_kernel(DecodeArgs g) {
    const int M = g.H * g.W * g.A;
    const long long total = (long long)g.N * M;
    proposal_init_buffers(g);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(e / M);
        const int i = (int)(e - (long long)n * M);
        g.keys[e] = decode_anchor(g, n, i);
    }
}

// decode + the first launch of the ordering (order_sort.hip) in one: workgroup (image, run) decodes the 2048
// anchors of its run, 8 consecutive ones per thread, and sorts their keys; g.keys receives the sorted runs.
__global__ __launch_bounds__(SORT_THREADS) void proposal_decode_runs_kernel(DecodeArgs g, int runs) {
    __shared__ typename RunSort::storage_type storage;
    const int M = g.H * g.W * g.A;
    proposal_init_buffers(g);
    const int n = blockIdx.x / runs, r = blockIdx.x - n * runs;
    const int first = r * RUN + threadIdx.x * SORT_ITEMS;
    unsigned long long k[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) k[i] = (first + i < M) ? decode_anchor(g, n, first + i) : 0ull;
    sort_and_store_run(k, storage, g.keys + (size_t)blockIdx.x * RUN);
}

__global__ __launch_bounds__(256) void proposal_gather_kernel(
    const float *__restrict__ boxes, const int *__restrict__ sorted_index,
    const int *__restrict__ n_sorted, int M, int topn, float *__restrict__ sorted_boxes) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= topn || p >= n_sorted[img]) return;
    const int i = sorted_index[(size_t)img * topn + p];
    const float4v b = *reinterpret_cast<const float4v *>(boxes + ((size_t)img * M + i) * 4);
    *reinterpret_cast<float4v *>(sorted_boxes + ((size_t)img * topn + p) * 4) = b;
}

__global__ __launch_bounds__(256) void proposal_compact_kernel(
    const float *__restrict__ rois_padded, const int *__restrict__ counts, int N, int post_topn,
    float *__restrict__ out, int total) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= counts[img]) return;
    int off = 0;
    for (int k = 0; k < img; ++k) off += max(counts[k], 0);      // (a count of -1 = WSSDL_NMS_TIMED_OUT: no rows)
    if (off + p >= total) return;
    const float *s = rois_padded + ((size_t)img * post_topn + p) * 5;
    float *d = out + (size_t)(off + p) * 5;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; d[4] = s[4];
}

struct ProposalWs {
    unsigned long long *keys, *cand, *thresh, *mask, *summ;
    float *boxes, *sorted_boxes;
    int *sorted_index, *n_sorted, *cand_fill, *kept, *done;
};

static size_t carve_proposal(void *ws, int N, int M, int topn, ProposalWs *out) {
    Carver c(ws);
    ProposalWs w;
    const int ncb = nms_mask_pitch(topn);
    w.keys = c.take<unsigned long long>((size_t)N * M);
    w.boxes = c.take<float>((size_t)N * M * 4);
    w.sorted_index = c.take<int>((size_t)N * topn);
    w.n_sorted = c.take<int>((size_t)N + 64);
    w.cand_fill = c.take<int>((size_t)N + 64);
    w.done = c.take<int>((size_t)N + 64);
    w.thresh = c.take<unsigned long long>((size_t)N + 32);
    w.cand = c.take<unsigned long long>((size_t)N * topn);
    w.kept = c.take<int>((size_t)N * ((size_t)topn + 64));
    w.sorted_boxes = c.take<float>((size_t)N * topn * 4);
    w.mask = c.take<unsigned long long>((size_t)N * topn * ncb);
    w.summ = c.take<unsigned long long>(nms_summary_alloc_words(N, topn));
    if (out) *out = w;
    return c.off;
}

static inline int effective_topn(int pre_nms_topN, int M) {
    // pre_nms_topN <= 0 means "no cap" (proposal_layer_tf_bus.py:130)
    return (pre_nms_topN > 0 && pre_nms_topN < M) ? pre_nms_topN : M;
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_proposal_workspace_bytes(int N, int H, int W, int A, int pre_nms_topN) {
    if (N < 1 || H < 1 || W < 1 || A < 1) return 256;
    int M = H * W * A;
    return carve_proposal(nullptr, N, M, effective_topn(pre_nms_topN, M), nullptr);
}

static int proposal_layer_impl(int from_logits, const float *rpn_cls_prob, const float *rpn_bbox_pred,
                                    const float *im_info, int im_info_stride, int N, int H, int W,
                                    const double *base_anchors_host, int A, int feat_stride,
                                    int pre_nms_topN, int post_nms_topN, double nms_thresh,
                                    float min_size, float *rois_padded, int32_t *roi_counts,
                                    float *decoded, int32_t *sorted_index, int32_t *sorted_count,
                                    void *workspace, size_t workspace_bytes,
                                    wssdl_stream_t stream) {
    if (N < 0 || H < 1 || W < 1 || im_info_stride < 3) return WSSDL_ERR_INVALID_ARGUMENT;
    BaseAnchors base;
    int rc = load_base_anchors(base_anchors_host, A, &base);
    if (rc) return rc;
    if (N == 0) return WSSDL_OK;
    if ((long long)H * W * A > (1LL << 24)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!rpn_cls_prob || !rpn_bbox_pred || !im_info || !rois_padded || !roi_counts || !workspace)
        return WSSDL_ERR_INVALID_ARGUMENT;
    const int M = H * W * A;
    const int topn = effective_topn(pre_nms_topN, M);
    // rois_padded is [N, pitch, 5]; post_nms_topN <= 0 means "no cap"
    // (proposal_layer_tf_bus.py:139), i.e. up to topn rows per image
    const int pitch = post_nms_topN > 0 ? post_nms_topN : topn;
    if (workspace_bytes < wssdl_proposal_workspace_bytes(N, H, W, A, pre_nms_topN))
        return WSSDL_ERR_WORKSPACE;
    ProposalWs w;
    carve_proposal(workspace, N, M, topn, &w);
    hipStream_t st = as_stream(stream);
    float *boxes = decoded ? decoded : w.boxes;
    int *sidx = w.sorted_index;
    int *nsorted = w.n_sorted;

    DecodeArgs g;
    g.prob = rpn_cls_prob;  g.pred = rpn_bbox_pred;  g.im_info = im_info;
    g.info_stride = im_info_stride;  g.N = N;  g.H = H;  g.W = W;  g.A = A;  g.stride = feat_stride;
    g.from_logits = from_logits;  g.min_size = min_size;  g.base = base;
    g.boxes = boxes;  g.keys = w.keys;
    g.sorted_index = sidx;  g.n_sorted_index = (long long)N * topn;
    g.n_sorted = nsorted;  g.cand_fill = w.cand_fill;
    g.rois_padded = rois_padded;  g.n_rois_floats = (long long)N * pitch * 5;
    // Order of the candidates.  Default (order_sort.hip): sorted runs + cross ranks, with the decode fused into the
    // run sort and the gather of the ranked boxes into the rank kernel: two launches in front of the NMS.  Scratch
    // (the sorted runs) = the suppression matrix, which is written later.  topk_sort = 2: decode, one device-wide
    // library sort, gather; topk_sort = 0: decode, the select + sample sort of nms.hip, gather.
    const size_t mask_bytes = sizeof(unsigned long long) * (size_t)N * topn * nms_mask_pitch(topn);
    const bool sortable = order_sort_supported(M, N) && order_sort_scratch_bytes(N, M) <= mask_bytes;
    if (tuning().topk_sort == 1 && sortable) {
        const int runs = order_runs_of(M);
        g.keys = w.mask;
        hipLaunchKernelGGL(proposal_decode_runs_kernel, dim3(N * runs), dim3(SORT_THREADS), 0, st, g, runs);
        if ((rc = check_launch())) return rc;
        if ((rc = launch_order_rank(w.mask, M, N, topn, sidx, nsorted, boxes, w.sorted_boxes, st))) return rc;
    } else {
        long long total = (long long)N * M;
        int blocks = cdiv(total, 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(proposal_decode_kernel, dim3(blocks), dim3(256), 0, st, g);
        if ((rc = check_launch())) return rc;
        if (tuning().topk_sort != 0 && sortable) {
            if ((rc = launch_order_sort(w.keys, M, N, topn, sidx, nsorted, w.cand_fill, w.mask, mask_bytes, st))) return rc;
        } else if ((rc = launch_rank_topk(w.keys, M, N, topn, w.cand, w.thresh, w.cand_fill, sidx, nsorted, w.mask,
                                          mask_bytes, st)))
            return rc;
        hipLaunchKernelGGL(proposal_gather_kernel, dim3(cdiv(topn, 256), N), dim3(256), 0, st, boxes,
                           sidx, nsorted, M, topn, w.sorted_boxes);
        if ((rc = check_launch())) return rc;
    }
    // w.cand is free once the ranking is done: it receives the transposed diagonal blocks
    // mask + sweep (two passes when a probe over the first candidates can settle an image, nms.hip);
    // the sweep writes (batch_idx, box) rows straight into rois_padded and stops after `pitch` kept boxes
    const bool one_pass = tuning().nms_one_pass != 0;       // tuning / A-B comparisons
    if ((rc = launch_nms_two_pass(w.sorted_boxes, topn * 4, nsorted, topn, N, nms_thresh, w.mask, w.cand, w.summ, pitch,
                                  nullptr, 0, nullptr, roi_counts, rois_padded, pitch <= topn ? w.kept : nullptr,
                                  one_pass ? nullptr : w.done, st)))
        return rc;
    if (sorted_index &&
        hipMemcpyAsync(sorted_index, sidx, sizeof(int) * (size_t)N * topn, hipMemcpyDeviceToDevice,
                       st) != hipSuccess)
        return WSSDL_ERR_LAUNCH;
    if (sorted_count &&
        hipMemcpyAsync(sorted_count, nsorted, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice,
                       st) != hipSuccess)
        return WSSDL_ERR_LAUNCH;
    return WSSDL_OK;
}

extern "C" int wssdl_proposal_layer(const float *rpn_cls_prob, const float *rpn_bbox_pred,
                                    const float *im_info, int im_info_stride, int N, int H, int W,
                                    const double *base_anchors_host, int A, int feat_stride,
                                    int pre_nms_topN, int post_nms_topN, double nms_thresh,
                                    float min_size, float *rois_padded, int32_t *roi_counts,
                                    float *decoded, int32_t *sorted_index, int32_t *sorted_count,
                                    void *workspace, size_t workspace_bytes,
                                    wssdl_stream_t stream) {
    return proposal_layer_impl(0, rpn_cls_prob, rpn_bbox_pred, im_info, im_info_stride, N, H, W,
                               base_anchors_host, A, feat_stride, pre_nms_topN, post_nms_topN,
                               nms_thresh, min_size, rois_padded, roi_counts, decoded, sorted_index,
                               sorted_count, workspace, workspace_bytes, stream);
}

extern "C" int wssdl_proposal_layer_from_logits(
    const float *rpn_cls_score, const float *rpn_bbox_pred, const float *im_info,
    int im_info_stride, int N, int H, int W, const double *base_anchors_host, int A,
    int feat_stride, int pre_nms_topN, int post_nms_topN, double nms_thresh, float min_size,
    float *rois_padded, int32_t *roi_counts, float *decoded, int32_t *sorted_index,
    int32_t *sorted_count, void *workspace, size_t workspace_bytes, wssdl_stream_t stream) {
    return proposal_layer_impl(1, rpn_cls_score, rpn_bbox_pred, im_info, im_info_stride, N, H, W,
                               base_anchors_host, A, feat_stride, pre_nms_topN, post_nms_topN,
                               nms_thresh, min_size, rois_padded, roi_counts, decoded, sorted_index,
                               sorted_count, workspace, workspace_bytes, stream);
}

extern "C" int wssdl_proposal_compact(const float *rois_padded, const int32_t *roi_counts, int N,
                                      int post_nms_topN, float *rois_out, int total_host,
                                      wssdl_stream_t stream) {
    if (N < 0 || post_nms_topN < 1 || total_host < 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0 || total_host == 0) return WSSDL_OK;
    if (!rois_padded || !roi_counts || !rois_out) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(proposal_compact_kernel, dim3(cdiv(post_nms_topN, 256), N), dim3(256), 0,
                       as_stream(stream), rois_padded, roi_counts, N, post_nms_topN, rois_out,
                       total_host);
    return check_launch();
}
