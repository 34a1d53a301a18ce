// f3: the post-detection step of the test path as ONE call, batched over the classes.
//
// Reference: code/lib/fast_rcnn/test_bus.py:360-401.  For every class j >= 1:
//   inds = where(scores[:, j] > thresh); cls_dets = hstack(boxes[inds, 4j:4j+4], scores[inds, j]) (f32);
//   keep = nms(cls_dets, cfg.TEST.NMS)  (utils/cython_nms == the cpu_nms rule: f32 arithmetic, f64 compare,
//   candidates in descending score order); cls_dets = cls_dets[keep].
// Then the cap (:389-396): if more than max_per_image detections are left over all classes,
//   image_thresh = sort(all scores)[-max_per_image]  and every class keeps its rows with score >= image_thresh.
//
// Here the classes play the role the images play in the proposal layer: one set of launches ranks,
// gathers and runs mask + sweep for all of them (nms.hip), the score filter is a zero key (no host
// read-back of `inds`), and the cap is a radix select over the kept scores (select.hip.h).  A class's
// kept rows are in descending score order, so "score >= image_thresh" keeps a PREFIX of them: the op
// writes dets[c] = the kept rows in that order and counts[c] = the length of the prefix.
#include "nms.hip.h"
#include "select.hip.h"

namespace wssdl {

struct PostWs {
    unsigned long long *keys, *cand, *thresh, *mask, *summ;
    float *boxes, *sorted_boxes;
    int *sorted_index, *n_sorted, *cand_fill, *kept, *keep, *num_keep;
};

static size_t carve_post(void *ws, int R, int nc, PostWs *out) {
    Carver c(ws);
    PostWs w;
    const int ncb = nms_mask_pitch(R);
    w.keys = c.take<unsigned long long>((size_t)nc * R);
    w.cand = c.take<unsigned long long>((size_t)nc * R);
    w.thresh = c.take<unsigned long long>((size_t)nc + 32);
    w.sorted_index = c.take<int>((size_t)nc * R);
    w.n_sorted = c.take<int>((size_t)nc + 64);
    w.cand_fill = c.take<int>((size_t)nc + 64);
    w.kept = c.take<int>((size_t)nc * ((size_t)R + 64));
    w.keep = c.take<int>((size_t)nc * R);
    w.num_keep = c.take<int>((size_t)nc + 64);
    w.boxes = c.take<float>((size_t)nc * R * 4);
    w.sorted_boxes = c.take<float>((size_t)nc * R * 4);
    w.mask = c.take<unsigned long long>((size_t)nc * R * ncb);
    w.summ = c.take<unsigned long long>(nms_summary_alloc_words(nc, R));
    if (out) *out = w;
    return c.off;
}

// keys of class c = j - 1: score_key(score, row) for rows above the score threshold, 0 otherwise;
// the class's box columns side by side; the initialisations the ranking expects
__global__ __launch_bounds__(256) void post_keys_kernel(const float *__restrict__ scores, const float *__restrict__ boxes,
                                                        int R, int K, float score_thresh,
                                                        unsigned long long *__restrict__ keys, float *__restrict__ cboxes,
                                                        int *__restrict__ sorted_index, int *__restrict__ n_sorted,
                                                        int *__restrict__ cand_fill) {
    const int nc = K - 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nc) { n_sorted[i] = 0;  cand_fill[i] = 0; }
    if (i >= (long long)nc * R) return;
    const int c = (int)(i / R), r = (int)(i - (long long)c * R);
    const float s = scores[(size_t)r * K + c + 1];
    keys[i] = (s > score_thresh) ? score_key(s, (unsigned)r) : 0ull;        // NaN fails the test like np.where
    sorted_index[i] = -1;
    const float *b = boxes + (size_t)r * 4 * K + 4 * (c + 1);
    float *o = cboxes + (size_t)i * 4;
    o[0] = b[0];  o[1] = b[1];  o[2] = b[2];  o[3] = b[3];
}

__global__ __launch_bounds__(256) void post_gather_kernel(const float *__restrict__ cboxes, const int *__restrict__ sorted_index,
                                                          const int *__restrict__ n_sorted, int R, int nc,
                                                          float *__restrict__ sorted_boxes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nc * R) return;
    const int c = (int)(i / R), p = (int)(i - (long long)c * R);
    if (p >= n_sorted[c]) return;
    const float *b = cboxes + ((size_t)c * R + sorted_index[i]) * 4;
    float *o = sorted_boxes + (size_t)i * 4;
    o[0] = b[0];  o[1] = b[1];  o[2] = b[2];  o[3] = b[3];
}

// One workgroup: the max_per_image-th largest kept score over all classes (test_bus.py:389-392), then the rows.
constexpr int CAP_BLOCK = 1024;
constexpr int CAP_LIST = 512;

__global__ __launch_bounds__(CAP_BLOCK) void post_cap_kernel(const float *__restrict__ scores, const float *__restrict__ cboxes,
                                                             const int *__restrict__ keep, const int *__restrict__ num_keep,
                                                             int R, int K, int max_per_image, float *__restrict__ dets,
                                                             int *__restrict__ counts) {
    __shared__ SelectScratch<CAP_LIST> sc;
    __shared__ int s_count[64];
    const int nc = K - 1, t = threadIdx.x;
    if (t < 64) s_count[t] = 0;
    // item i = (class c, kept position p); its key: score bits over a unique low word
    auto key_at = [&](int i, unsigned long long &v) -> bool {
        const int c = i / R, p = i - c * R;
        if (p >= num_keep[c]) return false;
        const int row = keep[(size_t)c * R + p];
        v = score_key(scores[(size_t)row * K + c + 1], (unsigned)i);
        return true;
    };
    int members = 0;
    const unsigned long long kth = block_radix_select<CAP_BLOCK, CAP_LIST, true>(
        key_at, nc * R, [max_per_image](int m) { return (max_per_image > 0 && m > max_per_image) ? max_per_image : 0; }, sc,
        &members);
    // image_thresh as order-preserving score bits (0: no cap -- every kept row passes)
    const unsigned cut = (unsigned)(kth >> 32);
    __syncthreads();
    for (int i = t; i < nc * R; i += CAP_BLOCK) {
        const int c = i / R, p = i - c * R;
        if (p >= num_keep[c]) continue;
        const int row = keep[(size_t)c * R + p];
        const float s = scores[(size_t)row * K + c + 1];
        const float *b = cboxes + ((size_t)c * R + row) * 4;
        float *o = dets + (size_t)i * 5;
        o[0] = b[0];  o[1] = b[1];  o[2] = b[2];  o[3] = b[3];  o[4] = s;
        if ((unsigned)(score_key(s, 0u) >> 32) >= cut && c < 64) atomicAdd(&s_count[c], 1);     // a prefix: rows are in descending order
    }
    __syncthreads();
    if (t < nc && t < 64) counts[t] = s_count[t];
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_post_detections_workspace_bytes(int R, int num_classes) {
    if (R < 1 || num_classes < 2) return 256;
    return carve_post(nullptr, R, num_classes - 1, nullptr);
}

extern "C" int wssdl_post_detections(const float *scores, const float *boxes, int R, int num_classes,
                                     float score_thresh, double nms_thresh, int max_per_image, float *dets,
                                     int32_t *counts, void *workspace, size_t workspace_bytes,
                                     wssdl_stream_t stream) {
    const int nc = num_classes - 1;
    if (R < 0 || num_classes < 2 || nc > 64 || !counts) return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    if (R == 0) {
        if (hipMemsetAsync(counts, 0, sizeof(int32_t) * nc, st) != hipSuccess) return WSSDL_ERR_LAUNCH;
        return WSSDL_OK;
    }
    if (!scores || !boxes || !dets || !workspace || (long long)nc * R > (1LL << 24)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < wssdl_post_detections_workspace_bytes(R, num_classes)) return WSSDL_ERR_WORKSPACE;
    PostWs w;
    carve_post(workspace, R, nc, &w);
    const int blocks = cdiv((long long)nc * R, 256);
    hipLaunchKernelGGL(post_keys_kernel, dim3(blocks), dim3(256), 0, st, scores, boxes, R, num_classes, score_thresh, w.keys,
                       w.boxes, w.sorted_index, w.n_sorted, w.cand_fill);
    int rc = check_launch();
    if (rc) return rc;
    if ((rc = launch_rank_topk(w.keys, R, nc, R, w.cand, w.thresh, w.cand_fill, w.sorted_index, w.n_sorted, w.mask,
                               sizeof(unsigned long long) * (size_t)nc * R * nms_mask_pitch(R), st)))
        return rc;
    hipLaunchKernelGGL(post_gather_kernel, dim3(blocks), dim3(256), 0, st, w.boxes, w.sorted_index, w.n_sorted, R, nc,
                       w.sorted_boxes);
    if ((rc = check_launch())) return rc;
    // w.cand is free once the ranking is done: it receives the transposed diagonal blocks of the mask
    if ((rc = launch_nms_two_pass(w.sorted_boxes, R * 4, w.n_sorted, R, nc, nms_thresh, w.mask, w.cand, w.summ, R,
                                  w.sorted_index, R, w.keep, w.num_keep, nullptr, w.kept, nullptr, st)))
        return rc;
    hipLaunchKernelGGL(post_cap_kernel, dim3(1), dim3(CAP_BLOCK), 0, st, scores, w.boxes, w.keep, w.num_keep, R, num_classes,
                       max_per_image, dets, counts);
    return check_launch();
}
