// Multi-task loss of the supervised images for gfx950: the four terms
// and their gradients in one forward and one backward launch.
//
// Reference: code/lib/fast_rcnn/train_bus.py
//   :186-192 / :605-610  rpn_cross_entropy  mean CE over the anchors whose label != -1
//   :203-210 / :613-620  rpn_loss_box       10 * mean over (image, channel) of
//                        sum_{h,w} out_w * [ 0.5 (3 in_w d)^2 s + (|d| - 0.5/9)(1 - s) ],  s = |d| < 1
//                        (the threshold 1 with the sigma = 3 pieces: the reference's formula as written);
//                        combined mode takes the first IMS_PER_BATCH images only (:613-616)
//   :218 / :623-630      cross_entropy      mean CE over the first len(label) rows of cls_score
//   :231-235 / :641-647  loss_box           mean over those rows of sum_j out_w * in_w * |pred - target|
// In the reference these are ~40 TF element-wise / reduction ops on tensors of at most 0.7 M
// elements (launch-bound); here every element is read once per direction.  Layouts are the layers'
// own: rpn_cls_score [N,H,W,2A] (channel c*A + a; the reshape of network.py:283-291 is an index map:
// row (a*H + h, w) of rpn_cls_score_reshape = channels (a, A + a) of cell (h, w)), rpn_labels
// [N,1,A*H,W] i32, rpn_bbox_pred [N,H,W,4A] against targets / weights [N,4A,H,W].
// Sums are accumulated in f64 per workgroup and combined in a fixed order (same result for the
// same launch shape); label -1 rows of the RoI list are padding (not rows of the reference's blob).
#include "common.hip.h"

#include <math.h>

namespace wssdl {

constexpr int MTL_BLOCK = 256;
constexpr int MTL_ITEMS = 8;                 // elements per thread and term
constexpr int MTL_MAX_CLASSES = 32;

struct MtlState {                            // head of the workspace, read by the backward
    double rpn_count, row_count;             // anchors with label != -1; rows with label != -1
};

__device__ __forceinline__ double block_sum(double v, double *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < MTL_BLOCK / 64; ++i) s += scratch[i];
    return s;                                // valid on thread 0
}

struct MtlArgs {
    const float *rpn_cls_score;
    const int *rpn_labels;
    const float *rpn_bbox_pred, *rpn_tg, *rpn_inw, *rpn_outw;
    int n_images, n_box_images, H, W, A;
    const float *cls_score;
    const int *labels;
    const float *bbox_pred, *tg, *inw, *outw;
    int n_rows, rows_total, K;
    int nb_cls, nb_box;                      // workgroups of the two RPN terms; one more for the R-CNN terms
};

// label / score addressing of anchor e = ((n*H + h)*W + w)*A + a
__device__ __forceinline__ void rpn_anchor(const MtlArgs &x, long long e, long long &score_off, long long &label_off) {
    const int a = (int)(e % x.A);
    const long long cell = e / x.A;                      // (n*H + h)*W + w
    const int w = (int)(cell % x.W);
    const long long nh = cell / x.W;
    const int h = (int)(nh % x.H);
    const long long n = nh / x.H;
    score_off = cell * (2 * x.A) + a;                    // bg; fg = + A
    label_off = ((n * x.A + a) * x.H + h) * x.W + w;     // [N,1,A*H,W]
}

// element e = ((n*H + h)*W + w)*4A + ch of rpn_bbox_pred  ->  offset in the [N,4A,H,W] tensors
__device__ __forceinline__ long long rpn_target_off(const MtlArgs &x, long long e) {
    const int C4 = 4 * x.A;
    const int ch = (int)(e % C4);
    const long long cell = e / C4;
    const int w = (int)(cell % x.W);
    const long long nh = cell / x.W;
    const int h = (int)(nh % x.H);
    const long long n = nh / x.H;
    return ((n * C4 + ch) * x.H + h) * x.W + w;
}

__device__ __forceinline__ float log_sum_exp2(float a, float b) {
    // (log1p of the smaller term: see lse_minus_max in common.hip.h)
    return fmaxf(a, b) + log1pf(expf(-fabsf(a - b)));
}

// partials: [nb_cls] CE sums, [nb_cls] counts, [nb_box] box sums, then 3 values of the R-CNN block
__global__ __launch_bounds__(MTL_BLOCK) void mtl_forward_kernel(MtlArgs x, double *__restrict__ partials) {
    __shared__ double scratch[MTL_BLOCK / 64];
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < x.nb_cls) {
        const long long total = (long long)x.n_images * x.H * x.W * x.A;
        double s = 0.0, c = 0.0;
        for (int i = 0; i < MTL_ITEMS; ++i) {
            const long long e = ((long long)b * MTL_ITEMS + i) * MTL_BLOCK + t;
            if (e >= total) break;
            long long so, lo;
            rpn_anchor(x, e, so, lo);
            const int l = x.rpn_labels[lo];
            if (l < 0) continue;
            const float bg = x.rpn_cls_score[so], fg = x.rpn_cls_score[so + x.A];
            // (max - own score) + log1p(...): adding the maximum first and subtracting it again would round the small term away
            s += (double)((fmaxf(bg, fg) - (l ? fg : bg)) + log1pf(expf(-fabsf(bg - fg))));
            c += 1.0;
        }
        const double ss = block_sum(s, scratch), cc = block_sum(c, scratch);
        if (t == 0) { partials[b] = ss;  partials[x.nb_cls + b] = cc; }
    } else if (b < x.nb_cls + x.nb_box) {
        const int bb = b - x.nb_cls;
        const long long total = (long long)x.n_box_images * x.H * x.W * 4 * x.A;
        double s = 0.0;
        for (int i = 0; i < MTL_ITEMS; ++i) {
            const long long e = ((long long)bb * MTL_ITEMS + i) * MTL_BLOCK + t;
            if (e >= total) break;
            const long long to = rpn_target_off(x, e);
            const float d = x.rpn_bbox_pred[e] - x.rpn_tg[to];
            const float iw = x.rpn_inw[to], ow = x.rpn_outw[to];
            const float ad = fabsf(d);
            const float q = iw * d * 3.0f;
            const float per = (ad < 1.0f) ? 0.5f * q * q : (ad - (float)(0.5 / 9.0));
            s += (double)(ow * per);
        }
        const double ss = block_sum(s, scratch);
        if (t == 0) partials[2 * x.nb_cls + bb] = ss;
    } else {
        // R-CNN terms: one workgroup, a thread per row
        double ce = 0.0, cnt = 0.0, box = 0.0;
        for (int r = t; r < x.n_rows; r += MTL_BLOCK) {
            const int l = x.labels[r];
            if (l >= 0 && l < x.K) {       // (a label outside [0, K) is ignored like padding, not read through)
                const float *sc = x.cls_score + (size_t)r * x.K;
                float m;
                const float lz = lse_minus_max(sc, x.K, &m);
                ce += (double)((m - sc[l]) + lz);
                cnt += 1.0;
            }
            const size_t o = (size_t)r * 4 * x.K;
            float row = 0.0f;
            for (int j = 0; j < 4 * x.K; ++j) row += x.outw[o + j] * (x.inw[o + j] * fabsf(x.bbox_pred[o + j] - x.tg[o + j]));
            box += (double)row;
        }
        const double a = block_sum(ce, scratch), c = block_sum(cnt, scratch), d = block_sum(box, scratch);
        if (t == 0) {
            double *p = partials + 2 * x.nb_cls + x.nb_box;
            p[0] = a;  p[1] = c;  p[2] = d;
        }
    }
}

__global__ __launch_bounds__(MTL_BLOCK) void mtl_finish_kernel(const double *__restrict__ partials, int nb_cls,
                                                               int nb_box, int n_box_images, int A,
                                                               MtlState *__restrict__ state, float *__restrict__ losses) {
    __shared__ double scratch[MTL_BLOCK / 64];
    const int t = threadIdx.x;
    double s = 0.0, c = 0.0, b = 0.0;
    for (int i = t; i < nb_cls; i += MTL_BLOCK) { s += partials[i];  c += partials[nb_cls + i]; }
    for (int i = t; i < nb_box; i += MTL_BLOCK) b += partials[2 * nb_cls + i];
    const double ss = block_sum(s, scratch), cc = block_sum(c, scratch), bb = block_sum(b, scratch);
    if (t == 0) {
        const double *p = partials + 2 * nb_cls + nb_box;
        state->rpn_count = cc;
        state->row_count = p[1];
        losses[0] = (float)(ss / cc);                                        // no labelled anchor: NaN, like the mean of nothing
        losses[1] = (float)(10.0 * bb / ((double)n_box_images * 4.0 * A));
        losses[2] = (float)(p[0] / p[1]);
        losses[3] = (float)(p[2] / (p[1] < 1.0 ? 1.0 : p[1]));
    }
}

__device__ __forceinline__ float sgnf(float d) { return (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f); }

__global__ __launch_bounds__(MTL_BLOCK) void mtl_backward_kernel(MtlArgs x, const MtlState *__restrict__ state,
                                                                 const float *__restrict__ gl, float *__restrict__ g_rpn_cls,
                                                                 float *__restrict__ g_rpn_box, float *__restrict__ g_cls,
                                                                 float *__restrict__ g_box) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < x.nb_cls) {
        const long long total = (long long)x.n_images * x.H * x.W * x.A;
        const float scale = (state->rpn_count > 0.0) ? (float)((double)gl[0] / state->rpn_count) : 0.0f;
        for (int i = 0; i < MTL_ITEMS; ++i) {
            const long long e = ((long long)b * MTL_ITEMS + i) * MTL_BLOCK + t;
            if (e >= total) break;
            long long so, lo;
            rpn_anchor(x, e, so, lo);
            const int l = x.rpn_labels[lo];
            float gb = 0.0f, gf = 0.0f;
            if (l >= 0) {
                const float bg = x.rpn_cls_score[so], fg = x.rpn_cls_score[so + x.A];
                const float lse = log_sum_exp2(bg, fg);
                // (two classes: the label's component is minus the other class's probability)
                const float pb = expf(bg - lse), pf = expf(fg - lse);
                gb = (l == 0 ? -pf : pb) * scale;
                gf = (l == 1 ? -pb : pf) * scale;
            }
            g_rpn_cls[so] = gb;
            g_rpn_cls[so + x.A] = gf;
        }
    } else if (b < x.nb_cls + x.nb_box) {
        // (covers ALL images: the ones beyond n_box_images get zeros)
        const int bb = b - x.nb_cls;
        const long long total = (long long)x.n_images * x.H * x.W * 4 * x.A;
        const long long live = (long long)x.n_box_images * x.H * x.W * 4 * x.A;
        const float scale = (float)(10.0 * (double)gl[1] / ((double)x.n_box_images * 4.0 * x.A));
        for (int i = 0; i < MTL_ITEMS; ++i) {
            const long long e = ((long long)bb * MTL_ITEMS + i) * MTL_BLOCK + t;
            if (e >= total) break;
            float g = 0.0f;
            if (e < live) {
                const long long to = rpn_target_off(x, e);
                const float d = x.rpn_bbox_pred[e] - x.rpn_tg[to];
                const float iw = x.rpn_inw[to], ow = x.rpn_outw[to];
                g = ow * ((fabsf(d) < 1.0f) ? 9.0f * iw * iw * d : sgnf(d)) * scale;
            }
            g_rpn_box[e] = g;
        }
    } else {
        const float sc_ce = (state->row_count > 0.0) ? (float)((double)gl[2] / state->row_count) : 0.0f;
        const float sc_box = (float)((double)gl[3] / (state->row_count < 1.0 ? 1.0 : state->row_count));
        // a thread per row, a workgroup per MTL_BLOCK rows (one workgroup for all 8512 rows of a combined step
        // was a 32 us chain)
        const int r = (b - x.nb_cls - x.nb_box) * MTL_BLOCK + t;
        if (r < x.rows_total) {
            float *gc = g_cls + (size_t)r * x.K;
            float *gb = g_box + (size_t)r * 4 * x.K;
            const int l = (r < x.n_rows) ? x.labels[r] : -1;
            if (l >= 0 && l < x.K) {
                const float *sc = x.cls_score + (size_t)r * x.K;
                float m;
                const float lz = lse_minus_max(sc, x.K, &m);
                const float lse = m + lz;
                // (the label's component as -(sum of the other probabilities): accurate when p_l -> 1)
                float others = 0.0f;
                for (int k = 0; k < x.K; ++k) {
                    const float pk = expf(sc[k] - lse);
                    gc[k] = pk * sc_ce;
                    others += (k == l) ? 0.0f : pk;
                }
                gc[l] = -others * sc_ce;
            } else {
                for (int k = 0; k < x.K; ++k) gc[k] = 0.0f;
            }
            if (r < x.n_rows) {
                const size_t o = (size_t)r * 4 * x.K;
                for (int j = 0; j < 4 * x.K; ++j)
                    gb[j] = x.outw[o + j] * x.inw[o + j] * sgnf(x.bbox_pred[o + j] - x.tg[o + j]) * sc_box;
            } else {
                for (int j = 0; j < 4 * x.K; ++j) gb[j] = 0.0f;
            }
        }
    }
}

static int fill_args(MtlArgs *x, const float *rpn_cls_score, const int32_t *rpn_labels, const float *rpn_bbox_pred,
                     const float *rpn_tg, const float *rpn_inw, const float *rpn_outw, int n_images, int n_box_images,
                     int H, int W, int A, const float *cls_score, const int32_t *labels, const float *bbox_pred,
                     const float *tg, const float *inw, const float *outw, int n_rows, int rows_total, int K,
                     bool backward) {
    if (n_images < 1 || n_box_images < 1 || n_box_images > n_images || H < 1 || W < 1 || A < 1 || A > WSSDL_MAX_ANCHORS)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_rows < 0 || rows_total < n_rows || K < 2 || K > MTL_MAX_CLASSES) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!rpn_cls_score || !rpn_labels || !rpn_bbox_pred || !rpn_tg || !rpn_inw || !rpn_outw) return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_rows > 0 && (!cls_score || !labels || !bbox_pred || !tg || !inw || !outw)) return WSSDL_ERR_INVALID_ARGUMENT;
    const long long anchors = (long long)n_images * H * W * A;
    if (anchors * 4 > 0x7fffffffLL * (long long)MTL_ITEMS) return WSSDL_ERR_INVALID_ARGUMENT;
    x->rpn_cls_score = rpn_cls_score;  x->rpn_labels = rpn_labels;  x->rpn_bbox_pred = rpn_bbox_pred;
    x->rpn_tg = rpn_tg;  x->rpn_inw = rpn_inw;  x->rpn_outw = rpn_outw;
    x->n_images = n_images;  x->n_box_images = n_box_images;  x->H = H;  x->W = W;  x->A = A;
    x->cls_score = cls_score;  x->labels = labels;  x->bbox_pred = bbox_pred;  x->tg = tg;  x->inw = inw;  x->outw = outw;
    x->n_rows = n_rows;  x->rows_total = rows_total;  x->K = K;
    const long long per = (long long)MTL_BLOCK * MTL_ITEMS;
    x->nb_cls = (int)((anchors + per - 1) / per);
    // the backward writes the box gradient of every image, the forward only reads the first n_box_images
    const long long box_elems = (long long)(backward ? n_images : n_box_images) * H * W * 4 * A;
    x->nb_box = (int)((box_elems + per - 1) / per);
    return WSSDL_OK;
}

static size_t mtl_workspace(int n_images, int H, int W, int A) {
    const long long per = (long long)MTL_BLOCK * MTL_ITEMS;
    const long long nb_cls = ((long long)n_images * H * W * A + per - 1) / per;
    const long long nb_box = ((long long)n_images * H * W * 4 * A + per - 1) / per;
    return 256 + sizeof(double) * (size_t)(2 * nb_cls + nb_box + 4);
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_multi_task_loss_workspace_bytes(int n_images, int H, int W, int A) {
    if (n_images < 1 || H < 1 || W < 1 || A < 1) return 0;
    return mtl_workspace(n_images, H, W, A);
}

extern "C" int wssdl_multi_task_loss_forward(
    const float *rpn_cls_score, const int32_t *rpn_labels, const float *rpn_bbox_pred, const float *rpn_bbox_targets,
    const float *rpn_inside_w, const float *rpn_outside_w, int n_images, int n_box_images, int H, int W, int A,
    const float *cls_score, const int32_t *labels, const float *bbox_pred, const float *bbox_targets,
    const float *bbox_inside_w, const float *bbox_outside_w, int n_rows, int num_classes, float *losses,
    void *workspace, size_t workspace_bytes, wssdl_stream_t stream) {
    MtlArgs x;
    int rc = fill_args(&x, rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_bbox_targets, rpn_inside_w, rpn_outside_w,
                       n_images, n_box_images, H, W, A, cls_score, labels, bbox_pred, bbox_targets, bbox_inside_w,
                       bbox_outside_w, n_rows, n_rows, num_classes, false);
    if (rc) return rc;
    if (!losses || !workspace) return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < mtl_workspace(n_images, H, W, A)) return WSSDL_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    MtlState *state = static_cast<MtlState *>(workspace);
    double *partials = reinterpret_cast<double *>(static_cast<char *>(workspace) + 256);
    hipLaunchKernelGGL(mtl_forward_kernel, dim3(x.nb_cls + x.nb_box + 1), dim3(MTL_BLOCK), 0, st, x, partials);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(mtl_finish_kernel, dim3(1), dim3(MTL_BLOCK), 0, st, partials, x.nb_cls, x.nb_box, n_box_images, A,
                       state, losses);
    return check_launch();
}

extern "C" int wssdl_multi_task_loss_backward(
    const float *rpn_cls_score, const int32_t *rpn_labels, const float *rpn_bbox_pred, const float *rpn_bbox_targets,
    const float *rpn_inside_w, const float *rpn_outside_w, int n_images, int n_box_images, int H, int W, int A,
    const float *cls_score, const int32_t *labels, const float *bbox_pred, const float *bbox_targets,
    const float *bbox_inside_w, const float *bbox_outside_w, int n_rows, int rows_total, int num_classes,
    const float *grad_losses, const void *workspace, float *grad_rpn_cls_score, float *grad_rpn_bbox_pred,
    float *grad_cls_score, float *grad_bbox_pred, wssdl_stream_t stream) {
    MtlArgs x;
    int rc = fill_args(&x, rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_bbox_targets, rpn_inside_w, rpn_outside_w,
                       n_images, n_box_images, H, W, A, cls_score, labels, bbox_pred, bbox_targets, bbox_inside_w,
                       bbox_outside_w, n_rows, rows_total, num_classes, true);
    if (rc) return rc;
    if (!grad_losses || !workspace || !grad_rpn_cls_score || !grad_rpn_bbox_pred) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rows_total > 0 && (!grad_cls_score || !grad_bbox_pred)) return WSSDL_ERR_INVALID_ARGUMENT;
    const int row_blocks = rows_total > 0 ? cdiv(rows_total, MTL_BLOCK) : 0;
    hipLaunchKernelGGL(mtl_backward_kernel, dim3(x.nb_cls + x.nb_box + row_blocks), dim3(MTL_BLOCK), 0, as_stream(stream), x,
                       static_cast<const MtlState *>(workspace), grad_losses, grad_rpn_cls_score, grad_rpn_bbox_pred,
                       grad_cls_score, grad_bbox_pred);
    return check_launch();
}
