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
