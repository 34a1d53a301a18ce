// MIL bag-instance selection for gfx950.
//
// Reference: code/lib/mil/core.py:11-46 (get_bag_logit) with the selectors
// get_mal_max_logit :60-69 (arg-max of class-2 logit), get_ben_max_logit :49-57 (class 1),
// get_mass_max_logit :88-96 (arg-min of class-0 logit), as wired by
// fast_rcnn/train_bus.py:241 (alternating: mass-max for bag label 1, mal-max otherwise) and
// :655 (combined: mal-max for both).  ~20 TF slice/concat ops per weak image become one
// launch: one workgroup per bag scans the instances whose bag index matches, and returns the
// row to gather (first extremum, like tf.arg_max / tf.arg_min).  The gather itself stays a
// differentiable index_select in the host layer.
#include "common.hip.h"

namespace wssdl {

enum { MIL_SEL_MAL_MAX = 0, MIL_SEL_BEN_MAX = 1, MIL_SEL_MASS_MAX = 2 };

__global__ __launch_bounds__(256) void mil_select_kernel(
    const float *__restrict__ logits, int R, int K, const float *__restrict__ bag_of_row,
    int bag_stride, float bag_offset, const int *__restrict__ bag_labels, int sel_label1,
    int sel_other, int *__restrict__ row_out, int *__restrict__ count_out) {
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    __shared__ int s_cnt;
    const int bag = blockIdx.x, t = threadIdx.x;
    const int sel = (bag_labels[bag] == 1) ? sel_label1 : sel_other;
    const int col = (sel == MIL_SEL_MAL_MAX) ? 2 : ((sel == MIL_SEL_BEN_MAX) ? 1 : 0);
    const float sign = (sel == MIL_SEL_MASS_MAX) ? -1.0f : 1.0f;       // arg-min == arg-max of the negation
    if (t == 0) s_cnt = 0;
    __syncthreads();
    float best = 0.0f;
    int bi = -1, cnt = 0;
    for (int r = t; r < R; r += 256) {
        if ((int)(bag_of_row[(size_t)r * bag_stride] - bag_offset) != bag) continue;
        ++cnt;
        const float v = sign * logits[(size_t)r * K + col];
        if (bi < 0 || v > best) { best = v; bi = r; }                  // rows ascend: first extremum wins
    }
    s_val[t] = best;
    s_idx[t] = bi;
    atomicAdd(&s_cnt, cnt);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            const int oi = s_idx[t + s];
            const float ov = s_val[t + s];
            const int mi = s_idx[t];
            if (oi >= 0 && (mi < 0 || ov > s_val[t] || (ov == s_val[t] && oi < mi))) {
                s_val[t] = ov;
                s_idx[t] = oi;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        row_out[bag] = s_idx[0];
        if (count_out) count_out[bag] = s_cnt;
    }
}

// ---- the MIL loss as one op ("bag-logit selection + weighted CE") ----
// fast_rcnn/train_bus.py:239-260 / :650-671: per bag the selected instance's logits, softmax CE
// against the bag label, weighted by the class prior [0, WS_MAL_PCT, 1 - WS_MAL_PCT] (:252,:664)
// and the step-dependent scale (:248,:659), mean over the bags.  An empty bag contributes 0 (and
// still counts in the mean).  One workgroup per bag: selection as above, then the bag's term.
constexpr int MIL_MAX_CLASSES = 8;
struct MilWeights { float w[MIL_MAX_CLASSES]; };

__global__ __launch_bounds__(64) void mil_loss_finish_kernel(const float *__restrict__ bag_loss, int n_bags,
                                                            float scale, float *__restrict__ loss) {
    // bags in index order, f64: the same value for any launch shape
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < n_bags; ++b) s += (double)bag_loss[b];
        loss[0] = (float)((double)scale * s / (double)n_bags);
    }
}

__global__ __launch_bounds__(64) void mil_bag_term_kernel(const float *__restrict__ logits, int K,
                                                         const int *__restrict__ bag_labels,
                                                         const int *__restrict__ rows, MilWeights cw, int n_bags,
                                                         float *__restrict__ bag_loss) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_bags) return;
    const int r = rows[b], l = bag_labels[b];
    float v = 0.0f;
    if (r >= 0 && l >= 0 && l < K) {
        const float *s = logits + (size_t)r * K;
        float m;
        const float lz = lse_minus_max(s, K, &m);
        v = cw.w[l] * ((m - s[l]) + lz);
    }
    bag_loss[b] = v;
}

// gradient of the whole [R,K] logits block: zeros except the selected rows
__global__ __launch_bounds__(256) void mil_loss_backward_kernel(
    const float *__restrict__ logits, int R, int K, const float *__restrict__ bag_of_row, int bag_stride,
    float bag_offset, const int *__restrict__ bag_labels, const int *__restrict__ rows, MilWeights cw, int n_bags,
    float scale, const float *__restrict__ grad_loss, float *__restrict__ grad_logits) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float *g = grad_logits + (size_t)r * K;
    const int bag = (int)(bag_of_row[(size_t)r * bag_stride] - bag_offset);
    const bool mine = bag >= 0 && bag < n_bags && rows[bag] == r;
    const int l = mine ? bag_labels[bag] : -1;
    if (!mine || l < 0 || l >= K) {
        for (int k = 0; k < K; ++k) g[k] = 0.0f;
        return;
    }
    const float *s = logits + (size_t)r * K;
    float m;
    const float lz = lse_minus_max(s, K, &m);
    const float lse = m + lz;
    const float c = grad_loss[0] * scale * cw.w[l] / (float)n_bags;
    // the label's component is p_l - 1 = -(sum of the other probabilities): the sum keeps its accuracy when p_l -> 1
    float others = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float pk = expf(s[k] - lse);
        g[k] = c * pk;
        others += (k == l) ? 0.0f : pk;
    }
    g[l] = -c * others;
}

}  // namespace wssdl

extern "C" int wssdl_mil_select(const float *instance_logits, int R, int num_classes,
                                const float *bag_of_row, int bag_stride, float bag_offset,
                                const int32_t *bag_labels, int n_bags, int selector_label1,
                                int selector_other, int32_t *row_out, int32_t *count_out,
                                wssdl_stream_t stream) {
    if (R < 0 || num_classes < 3 || n_bags < 0 || bag_stride < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (selector_label1 < 0 || selector_label1 > 2 || selector_other < 0 || selector_other > 2)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_bags == 0) return WSSDL_OK;
    if (!bag_labels || !row_out || (R > 0 && (!instance_logits || !bag_of_row)))
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(wssdl::mil_select_kernel, dim3(n_bags), dim3(256), 0, wssdl::as_stream(stream),
                       instance_logits, R, num_classes, bag_of_row, bag_stride, bag_offset, bag_labels,
                       selector_label1, selector_other, row_out, count_out);
    return wssdl::check_launch();
}

static int mil_weights(const float *class_weights_host, int K, wssdl::MilWeights *out) {
    if (!class_weights_host || K < 3 || K > wssdl::MIL_MAX_CLASSES) return WSSDL_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < wssdl::MIL_MAX_CLASSES; ++k) out->w[k] = (k < K) ? class_weights_host[k] : 0.0f;
    return WSSDL_OK;
}

extern "C" int wssdl_mil_loss_forward(const float *instance_logits, int R, int num_classes,
                                      const float *bag_of_row, int bag_stride, float bag_offset,
                                      const int32_t *bag_labels, int n_bags, int selector_label1,
                                      int selector_other, const float *class_weights_host, float scale,
                                      float *loss, int32_t *row_out, float *bag_loss, wssdl_stream_t stream) {
    wssdl::MilWeights cw;
    int rc = mil_weights(class_weights_host, num_classes, &cw);
    if (rc) return rc;
    if (n_bags < 1 || !loss || !row_out || !bag_loss) return WSSDL_ERR_INVALID_ARGUMENT;
    rc = wssdl_mil_select(instance_logits, R, num_classes, bag_of_row, bag_stride, bag_offset, bag_labels, n_bags,
                          selector_label1, selector_other, row_out, nullptr, stream);
    if (rc) return rc;
    hipStream_t st = wssdl::as_stream(stream);
    hipLaunchKernelGGL(wssdl::mil_bag_term_kernel, dim3(wssdl::cdiv(n_bags, 64)), dim3(64), 0, st, instance_logits,
                       num_classes, bag_labels, row_out, cw, n_bags, bag_loss);
    if ((rc = wssdl::check_launch())) return rc;
    hipLaunchKernelGGL(wssdl::mil_loss_finish_kernel, dim3(1), dim3(64), 0, st, bag_loss, n_bags, scale, loss);
    return wssdl::check_launch();
}

extern "C" int wssdl_mil_loss_backward(const float *instance_logits, int R, int num_classes,
                                       const float *bag_of_row, int bag_stride, float bag_offset,
                                       const int32_t *bag_labels, int n_bags, const int32_t *rows,
                                       const float *class_weights_host, float scale, const float *grad_loss,
                                       float *grad_logits, wssdl_stream_t stream) {
    wssdl::MilWeights cw;
    int rc = mil_weights(class_weights_host, num_classes, &cw);
    if (rc) return rc;
    if (R < 0 || n_bags < 1 || bag_stride < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (R == 0) return WSSDL_OK;
    if (!instance_logits || !bag_of_row || !bag_labels || !rows || !grad_loss || !grad_logits)
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(wssdl::mil_loss_backward_kernel, dim3(wssdl::cdiv(R, 256)), dim3(256), 0,
                       wssdl::as_stream(stream), instance_logits, R, num_classes, bag_of_row, bag_stride, bag_offset,
                       bag_labels, rows, cw, n_bags, scale, grad_loss, grad_logits);
    return wssdl::check_launch();
}
