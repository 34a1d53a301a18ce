// RPN anchor-target layer for gfx950 (MI355X), all supervised images of a step
// per launch.
//
// Reference: code/lib/rpn_msr/anchor_target_layer_tf_bus.py:19-303 (alternating),
// :306-325 (weak), :328-628 (combined); utils/bbox.pyx:15-55, utils/bbox_ui.pyx:12-47
// (f64 overlaps); fast_rcnn/bbox_transform.py:10-28 (targets).
//
// The shifted anchor grid is never materialised: each lane rebuilds its anchor
// (h, w, a) from the <= 32 base anchors passed as a kernel argument; the <= 64 gt
// boxes of the image sit in LDS and are read as broadcasts.  f64 IoU in the
// Cython kernels' operation order (built with -ffp-contract=off), so the labels
// -- which depend on `overlap == column max` and `overlap >= 0.7` -- are
// bit-identical, including the zero-overlap quirk (:446-449).
//
//   anchor_gtmax   : per-gt column maximum over inside anchors (wave shuffle
//                    reduce -> LDS -> one u64 atomicMax per gt per workgroup;
//                    non-negative doubles order like their bit patterns).
//   anchor_label   : labels before sub-sampling, arg-max gt, counts.
//   anchor_subsample (optional, device RNG): exact-k random subset by 64-bit
//                    radix-select over counter-based hash keys.
//   anchor_targets : final labels -> rpn_labels [n,1,A*H,W] and the three
//                    [n,4A,H,W] blobs; lanes run along W so all 13 output planes
//                    are written with coalesced stores (13*A*K*4 B per image).
#include "common.hip.h"
#include "select.hip.h"

namespace wssdl {

struct GtShared {
    double x1[WSSDL_MAX_GT], y1[WSSDL_MAX_GT], x2[WSSDL_MAX_GT], y2[WSSDL_MAX_GT],
        area[WSSDL_MAX_GT];
    int num_gt, num_pos, n_ov, n_ui;   // n_ov boxes take IoU (rows 0..n_ov-1), n_ui rows follow
};

// gt rows of one image -> LDS, widened f32 -> f64 exactly as
// np.ascontiguousarray(gt, dtype=np.float) does (anchor_target_layer_tf_bus.py:131)
__device__ __forceinline__ void load_gt(GtShared &s, const float *__restrict__ gt_boxes, int max_gt,
                                        const int *__restrict__ num_gt_boxes, int img,
                                        int dataset) {
    const int t = threadIdx.x;
    int ng = num_gt_boxes[img];
    ng = min(max(ng, 0), min(max_gt, WSSDL_MAX_GT));
    const float *g = gt_boxes + (size_t)img * max_gt * 5;
    if (t < ng) {
        double x1 = g[t * 5 + 0], y1 = g[t * 5 + 1], x2 = g[t * 5 + 2], y2 = g[t * 5 + 3];
        s.x1[t] = x1; s.y1[t] = y1; s.x2[t] = x2; s.y2[t] = y2;
        s.area[t] = (x2 - x1 + 1) * (y2 - y1 + 1);
    }
    if (t == 0) {
        int np = 0;
        for (int k = 0; k < ng; ++k) np += (g[k * 5 + 4] != 0.0f) ? 1 : 0;   // :124-125
        s.num_gt = ng;
        s.num_pos = np;
        if (dataset == WSSDL_DATASET_SNUBH) { s.n_ov = np; s.n_ui = ng - np; }
        else if (dataset == WSSDL_DATASET_SNUBH_FG) { s.n_ov = np; s.n_ui = 0; }
        else { s.n_ov = ng; s.n_ui = 0; }
    }
    __syncthreads();
}

struct AnchorBox { double x1, y1, x2, y2; bool inside; };

__device__ __forceinline__ AnchorBox make_anchor(const BaseAnchors &base, int a, int h, int w,
                                                 int stride, float im_h, float im_w) {
    AnchorBox b;
    b.x1 = base.v[a][0] + (double)(stride * w);
    b.y1 = base.v[a][1] + (double)(stride * h);
    b.x2 = base.v[a][2] + (double)(stride * w);
    b.y2 = base.v[a][3] + (double)(stride * h);
    // :100-105 with _allowed_border = 0; im_info is f32, compared as double
    b.inside = (b.x1 >= 0.0) && (b.y1 >= 0.0) && (b.x2 < (double)im_w) && (b.y2 < (double)im_h);
    return b;
}

__device__ __forceinline__ double iou_f64(const AnchorBox &b, const GtShared &s, int k) {
    double iw = fmin(b.x2, s.x2[k]) - fmax(b.x1, s.x1[k]) + 1;
    if (iw > 0) {
        double ih = fmin(b.y2, s.y2[k]) - fmax(b.y1, s.y1[k]) + 1;
        if (ih > 0) {
            double ua = (b.x2 - b.x1 + 1) * (b.y2 - b.y1 + 1) + s.area[k] - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0;
}

__device__ __forceinline__ double ui_f64(const AnchorBox &b, double barea, const GtShared &s,
                                         int k) {
    double iw = fmin(b.x2, s.x2[k]) - fmax(b.x1, s.x1[k]) + 1;
    if (iw > 0) {
        double ih = fmin(b.y2, s.y2[k]) - fmax(b.y1, s.y1[k]) + 1;
        if (ih > 0) return iw * ih / barea;
    }
    return 0.0;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)v, off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

constexpr int AT_BLOCK = 256;

__global__ __launch_bounds__(AT_BLOCK) void anchor_gtmax_kernel(
    const float *__restrict__ gt_boxes, int max_gt, const int *__restrict__ num_gt_boxes,
    const float *__restrict__ im_info, int info_stride, int H, int W, BaseAnchors base, int A,
    int stride, int dataset, unsigned long long *__restrict__ gt_max /* [n_images, MAX_GT] */) {
    __shared__ GtShared s;
    __shared__ unsigned long long blk_max[WSSDL_MAX_GT];
    const int img = blockIdx.y;
    load_gt(s, gt_boxes, max_gt, num_gt_boxes, img, dataset);
    if (threadIdx.x < WSSDL_MAX_GT) blk_max[threadIdx.x] = 0ull;
    __syncthreads();
    const int total = H * W * A;
    const int i = blockIdx.x * AT_BLOCK + threadIdx.x;
    AnchorBox b;
    b.inside = false;
    if (i < total) {
        const int cell = i / A, a = i - cell * A;
        const int h = cell / W, w = cell - h * W;
        b = make_anchor(base, a, h, w, stride, im_info[img * info_stride + 0],
                        im_info[img * info_stride + 1]);
    }
    for (int k = 0; k < s.n_ov; ++k) {
        double ov = b.inside ? iou_f64(b, s, k) : 0.0;
        unsigned long long m = wave_max_u64((unsigned long long)__double_as_longlong(ov));
        if ((threadIdx.x & 63) == 0 && m != 0ull) atomicMax(&blk_max[k], m);
    }
    __syncthreads();
    if (threadIdx.x < s.n_ov && blk_max[threadIdx.x] != 0ull)
        atomicMax(&gt_max[(size_t)img * WSSDL_MAX_GT + threadIdx.x], blk_max[threadIdx.x]);
}

__global__ __launch_bounds__(AT_BLOCK) void anchor_label_kernel(
    const float *__restrict__ gt_boxes, int max_gt, const int *__restrict__ num_gt_boxes,
    const float *__restrict__ im_info, int info_stride, int H, int W, BaseAnchors base, int A,
    int stride, int dataset, double pos_thr, double neg_thr, int clobber,
    const unsigned long long *__restrict__ gt_max, signed char *__restrict__ labels_pre,
    int *__restrict__ argmax_gt, int *__restrict__ counts) {
    __shared__ GtShared s;
    __shared__ double gmax[WSSDL_MAX_GT];
    const int img = blockIdx.y;
    load_gt(s, gt_boxes, max_gt, num_gt_boxes, img, dataset);
    if (threadIdx.x < WSSDL_MAX_GT)
        gmax[threadIdx.x] =
            __longlong_as_double((long long)gt_max[(size_t)img * WSSDL_MAX_GT + threadIdx.x]);
    __syncthreads();
    __shared__ int blk_cnt[3];
    if (threadIdx.x < 3) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int total = H * W * A;
    const int i = blockIdx.x * AT_BLOCK + threadIdx.x;
    int label = -1, arg = -1;
    bool inside = false;
    if (i < total) {
        const int cell = i / A, a = i - cell * A;
        const int h = cell / W, w = cell - h * W;
        const AnchorBox b = make_anchor(base, a, h, w, stride, im_info[img * info_stride + 0],
                                        im_info[img * info_stride + 1]);
        inside = b.inside;
        if (b.inside) {
            double best = 0.0;
            bool hit = false;
            arg = (s.n_ov > 0) ? 0 : -1;      // no positive gt: the reference raises; we emit no target
            for (int k = 0; k < s.n_ov; ++k) {
                double ov = iou_f64(b, s, k);
                if (k == 0 || ov > best) { best = ov; arg = k; }   // numpy argmax: first maximum
                hit = hit || (ov == gmax[k]);                       // np.where(overlaps == gt_max), :139
            }
            if (dataset == WSSDL_DATASET_SNUBH) {
                if (s.n_ui > 0 && !clobber) {
                    const double barea = (b.x2 - b.x1 + 1) * (b.y2 - b.y1 + 1);
                    double mu = 0.0;
                    for (int k = 0; k < s.n_ui; ++k) {
                        double u = ui_f64(b, barea, s, s.n_ov + k);
                        if (k == 0 || u > mu) mu = u;
                    }
                    if (mu >= pos_thr) label = 0;                   // :149-151
                }
                if (hit) label = 1;                                 // :154
                if (best >= pos_thr) label = 1;                     // :157
            } else {
                if (!clobber && best < neg_thr) label = 0;          // :185-187
                if (hit) label = 1;
                if (best >= pos_thr) label = 1;
                if (clobber && best < neg_thr) label = 0;           // :195-197
            }
        }
        labels_pre[(size_t)img * total + i] = (signed char)label;
        argmax_gt[(size_t)img * total + i] = arg;
    }
    // counts: wave ballot -> LDS -> one global atomic per workgroup
    const unsigned long long m_in = __ballot(inside);
    const unsigned long long m_fg = __ballot(inside && label == 1);
    const unsigned long long m_bg = __ballot(inside && label == 0);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&blk_cnt[0], __popcll(m_in));
        atomicAdd(&blk_cnt[1], __popcll(m_fg));
        atomicAdd(&blk_cnt[2], __popcll(m_bg));
    }
    __syncthreads();
    if (threadIdx.x < 3 && blk_cnt[threadIdx.x] != 0)
        atomicAdd(&counts[img * 4 + threadIdx.x], blk_cnt[threadIdx.x]);
}

// ------------------------------------------------------- device sub-sampling ---
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned long long sample_key(unsigned long long seed, int img,
                                                         int phase, int i) {
    unsigned long long h = mix64(seed ^ mix64(((unsigned long long)img << 34) ^
                                             ((unsigned long long)phase << 32) ^ (unsigned)i));
    return (h & 0xFFFFFFFF00000000ull) | (unsigned)i;   // unique per anchor
}

constexpr int SS_BLOCK = 1024;
constexpr int SS_LIST = 512;

// keep exactly `quota` of the anchors with label == which (those with the smallest
// keys); the rest become -1.  No-op when there are <= quota of them.  Returns how many remain.
__device__ int subsample_one(signed char *lab, int total, int which, int quota,
                              unsigned long long seed, int img, int phase,
                              SelectScratch<SS_LIST> &sc) {
    const int t = threadIdx.x;
    int n = 0;
    // the quota-th smallest key (keys are unique) when there are more than quota members
    const unsigned long long cut = block_radix_select<SS_BLOCK, SS_LIST, false>(
        [=](int i, unsigned long long &v) {
            if (lab[i] != which) return false;
            v = sample_key(seed, img, phase, i);
            return true;
        }, total, [quota](int members) { return (members > quota && quota > 0) ? quota : 0; }, sc, &n);
    if (n <= quota) return n;
    if (quota <= 0) {
        for (int i = t; i < total; i += SS_BLOCK)
            if (lab[i] == which) lab[i] = -1;
        __syncthreads();
        return 0;
    }
    for (int i = t; i < total; i += SS_BLOCK)
        if (lab[i] == which && sample_key(seed, img, phase, i) > cut) lab[i] = -1;
    __syncthreads();
    return quota;
}

// With the fg / bg counts of the label stage at hand (gridDim.y == 2) the two draws of an image run in
// two workgroups side by side: the bg quota RPN_BATCHSIZE - #fg-after-sub-sampling (:212) only needs
// the NUMBER of fg anchors, min(#fg, num_fg), not which ones survive; each workgroup only rewrites
// anchors of its own class.  Same result as the sequential form (same keys, same cuts).
__global__ __launch_bounds__(SS_BLOCK) void anchor_subsample_kernel(signed char *labels, int total,
                                                                    int batchsize, int num_fg,
                                                                    unsigned long long seed,
                                                                    const int *__restrict__ counts) {
    __shared__ SelectScratch<SS_LIST> sc;
    const int img = blockIdx.x;
    signed char *lab = labels + (size_t)img * total;
    if (gridDim.y == 2) {
        if (blockIdx.y == 0) {
            subsample_one(lab, total, 1, num_fg, seed, img, 0, sc);               // :202-207
        } else {
            const int fg_left = min(counts[img * 4 + 1], max(num_fg, 0));
            subsample_one(lab, total, 0, batchsize - fg_left, seed, img, 1, sc);  // :212-217
        }
        return;
    }
    const int fg_left = subsample_one(lab, total, 1, num_fg, seed, img, 0, sc);   // :202-207
    // num_bg = RPN_BATCHSIZE - #fg after the first sub-sampling, :212
    subsample_one(lab, total, 0, batchsize - fg_left, seed, img, 1, sc);          // :213-217
}

// ---------------------------------------------------------------- targets ---
__global__ __launch_bounds__(AT_BLOCK) void anchor_targets_kernel(
    const signed char *__restrict__ labels, const int *__restrict__ argmax_gt,
    const float *__restrict__ gt_boxes, int max_gt, int n_images, int H, int W, BaseAnchors base,
    int A, int stride, float iw0, float iw1, float iw2, float iw3, double positive_weight,
    float *__restrict__ rpn_labels, float *__restrict__ bbox_targets, float *__restrict__ inside_w,
    float *__restrict__ outside_w) {
    __shared__ int s_cnt[3];
    const int img = blockIdx.y;
    const int total = H * W * A;
    const size_t plane = (size_t)H * W;
    float *lab_o = rpn_labels + (size_t)img * total;
    float *tg_o = bbox_targets + (size_t)img * total * 4;
    float *iw_o = inside_w + (size_t)img * total * 4;
    float *ow_o = outside_w + (size_t)img * total * 4;
    const int e = blockIdx.x * AT_BLOCK + threadIdx.x;       // element in (a, h, w) order
    if (img >= n_images) {
        // all-ignore weak images (:306-325, :613-626)
        if (e < total) {
            lab_o[e] = -1.0f;
            for (int j = 0; j < 4; ++j) {
                tg_o[(size_t)j * total + e] = 0.0f;
                iw_o[(size_t)j * total + e] = 0.0f;
                ow_o[(size_t)j * total + e] = 0.0f;
            }
        }
        return;
    }
    // every workgroup recounts the image's examples (21.5 KB of i8 from L2) so the
    // layer needs no extra launch or workspace
    const signed char *lab = labels + (size_t)img * total;
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int c_ex = 0, c_fg = 0;
    {
        // labels are -1 / 0 / 1 as bytes 0xff / 0x00 / 0x01: 16 at a time from the first 16-byte
        // aligned address (84 single-byte loads per thread made this kernel latency-bound)
        const int head = min((int)((16 - (reinterpret_cast<uintptr_t>(lab) & 15)) & 15), total);
        const int n16 = (total - head) >> 4;
        const uint4 *body = reinterpret_cast<const uint4 *>(lab + head);
        for (int i = threadIdx.x; i < n16; i += AT_BLOCK) {
            const uint4 v = body[i];
            const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c_ex += 4 - __popc(w4[q] & 0x80808080u);
                c_fg += __popc(w4[q] & ~(w4[q] >> 7) & 0x01010101u);
            }
        }
        for (int i = threadIdx.x; i < head; i += AT_BLOCK) {
            const int l = lab[i];
            c_ex += (l >= 0);  c_fg += (l == 1);
        }
        for (int i = head + (n16 << 4) + threadIdx.x; i < total; i += AT_BLOCK) {
            const int l = lab[i];
            c_ex += (l >= 0);  c_fg += (l == 1);
        }
    }
    atomicAdd(&s_cnt[0], c_ex);  atomicAdd(&s_cnt[1], c_fg);  atomicAdd(&s_cnt[2], c_ex - c_fg);
    __syncthreads();
    if (e >= total) return;
    float pos_w, neg_w;
    if (positive_weight < 0) {                      // :231-235: uniform 1/num_examples
        pos_w = neg_w = (float)(1.0 / (double)s_cnt[0]);
    } else {                                        // :237-242
        pos_w = (float)(positive_weight / (double)s_cnt[1]);
        neg_w = (float)((1.0 - positive_weight) / (double)s_cnt[2]);
    }
    const int a = e / (int)plane;
    const int hw = e - a * (int)plane;
    const int h = hw / W, w = hw - h * W;
    const int i = hw * A + a;                       // anchor index in (h, w, a) order
    const int l = lab[i];
    const int k = argmax_gt[(size_t)img * total + i];
    lab_o[e] = (float)l;                            // [1, A*H, W] at (a*H + h, w), :278-279
    float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k >= 0) {                                   // inside anchors only (_unmap fill 0, :259-262)
        const double ax1 = base.v[a][0] + (double)(stride * w);
        const double ay1 = base.v[a][1] + (double)(stride * h);
        const double ax2 = base.v[a][2] + (double)(stride * w);
        const double ay2 = base.v[a][3] + (double)(stride * h);
        const double ew = ax2 - ax1 + 1.0, eh = ay2 - ay1 + 1.0;
        const double ecx = ax1 + 0.5 * ew, ecy = ay1 + 0.5 * eh;
        // gt side stays f32 until it meets the f64 anchors (bbox_transform.py:16-19)
        const float *g = gt_boxes + ((size_t)img * max_gt + k) * 5;
        float gw = g[2] - g[0];  gw = gw + 1.0f;
        float gh = g[3] - g[1];  gh = gh + 1.0f;
        float hgw = 0.5f * gw, hgh = 0.5f * gh;
        const float gcx = g[0] + hgw, gcy = g[1] + hgh;
        t[0] = (float)(((double)gcx - ecx) / ew);
        t[1] = (float)(((double)gcy - ecy) / eh);
        t[2] = (float)log((double)gw / ew);
        t[3] = (float)log((double)gh / eh);
    }
    const float iw[4] = {iw0, iw1, iw2, iw3};
    const float ow = (l == 1) ? pos_w : ((l == 0) ? neg_w : 0.0f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {                   // channel a*4 + j, :283-297
        const size_t o = ((size_t)(a * 4 + j)) * plane + hw;
        tg_o[o] = t[j];
        iw_o[o] = (l == 1) ? iw[j] : 0.0f;
        ow_o[o] = ow;
    }
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_anchor_workspace_bytes(int n_images) {
    return ((size_t)(n_images > 0 ? n_images : 1) * WSSDL_MAX_GT * sizeof(unsigned long long) + 255) &
           ~size_t(255);
}

extern "C" int wssdl_anchor_labels(const float *gt_boxes, int max_gt, const int32_t *num_gt_boxes,
                                   const float *im_info, int im_info_stride, int n_images, int H,
                                   int W, const double *base_anchors_host, int A, int feat_stride,
                                   int dataset, double positive_overlap, double negative_overlap,
                                   int clobber_positives, int8_t *labels_pre, int32_t *argmax_gt,
                                   int32_t *counts, void *workspace, size_t workspace_bytes,
                                   wssdl_stream_t stream) {
    if (n_images < 0 || H < 1 || W < 1 || max_gt < 1 || im_info_stride < 2)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (dataset < WSSDL_DATASET_SNUBH || dataset > WSSDL_DATASET_FG_ONLY)
        return WSSDL_ERR_INVALID_ARGUMENT;
    BaseAnchors base;
    int rc = load_base_anchors(base_anchors_host, A, &base);
    if (rc) return rc;
    if (n_images == 0) return WSSDL_OK;
    if ((long long)H * W * A > (1LL << 24)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!gt_boxes || !num_gt_boxes || !im_info || !labels_pre || !argmax_gt || !counts ||
        !workspace)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < wssdl_anchor_workspace_bytes(n_images)) return WSSDL_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    unsigned long long *gt_max = static_cast<unsigned long long *>(workspace);
    if (hipMemsetAsync(gt_max, 0, sizeof(unsigned long long) * (size_t)n_images * WSSDL_MAX_GT,
                       st) != hipSuccess ||
        hipMemsetAsync(counts, 0, sizeof(int) * (size_t)n_images * 4, st) != hipSuccess)
        return WSSDL_ERR_LAUNCH;
    const int total = H * W * A;
    dim3 grid(cdiv(total, AT_BLOCK), n_images);
    hipLaunchKernelGGL(anchor_gtmax_kernel, grid, dim3(AT_BLOCK), 0, st, gt_boxes, max_gt,
                       num_gt_boxes, im_info, im_info_stride, H, W, base, A, feat_stride, dataset,
                       gt_max);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(anchor_label_kernel, grid, dim3(AT_BLOCK), 0, st, gt_boxes, max_gt,
                       num_gt_boxes, im_info, im_info_stride, H, W, base, A, feat_stride, dataset,
                       positive_overlap, negative_overlap, clobber_positives, gt_max,
                       reinterpret_cast<signed char *>(labels_pre), argmax_gt, counts);
    return check_launch();
}

extern "C" int wssdl_anchor_subsample_device(int8_t *labels, int n_images, int total_anchors,
                                             int rpn_batchsize, double fg_fraction, uint64_t seed,
                                             const int32_t *counts, wssdl_stream_t stream) {
    if (n_images < 0 || total_anchors < 1 || rpn_batchsize < 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_images == 0) return WSSDL_OK;
    if (!labels) return WSSDL_ERR_INVALID_ARGUMENT;
    const int num_fg = (int)(fg_fraction * (double)rpn_batchsize);      // int(), :202
    hipLaunchKernelGGL(anchor_subsample_kernel, dim3(n_images, counts ? 2 : 1), dim3(SS_BLOCK), 0,
                       as_stream(stream), reinterpret_cast<signed char *>(labels), total_anchors,
                       rpn_batchsize, num_fg, (unsigned long long)seed, counts);
    return check_launch();
}

extern "C" int wssdl_anchor_targets(const int8_t *labels, const int32_t *argmax_gt,
                                    const float *gt_boxes, int max_gt, int n_images, int n_out,
                                    int H, int W, const double *base_anchors_host, int A,
                                    int feat_stride, const float *inside_weights_host,
                                    double positive_weight, float *rpn_labels, float *bbox_targets,
                                    float *inside_w, float *outside_w, wssdl_stream_t stream) {
    if (n_images < 0 || n_out < n_images || H < 1 || W < 1 || max_gt < 1 || !inside_weights_host)
        return WSSDL_ERR_INVALID_ARGUMENT;
    BaseAnchors base;
    int rc = load_base_anchors(base_anchors_host, A, &base);
    if (rc) return rc;
    if (n_out == 0) return WSSDL_OK;
    if ((long long)H * W * A > (1LL << 24)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!rpn_labels || !bbox_targets || !inside_w || !outside_w) return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_images > 0 && (!labels || !argmax_gt || !gt_boxes)) return WSSDL_ERR_INVALID_ARGUMENT;
    const int total = H * W * A;
    hipLaunchKernelGGL(anchor_targets_kernel, dim3(cdiv(total, AT_BLOCK), n_out), dim3(AT_BLOCK), 0,
                       as_stream(stream), reinterpret_cast<const signed char *>(labels), argmax_gt,
                       gt_boxes, max_gt, n_images, H, W, base, A, feat_stride,
                       inside_weights_host[0], inside_weights_host[1], inside_weights_host[2],
                       inside_weights_host[3], positive_weight, rpn_labels, bbox_targets, inside_w,
                       outside_w);
    return check_launch();
}
