// Library identification, error reporting and the host-side / trivial entry points.
#include "common.hip.h"

#include <math.h>
#include <string.h>

namespace wssdl {
static thread_local hipError_t g_last_error = hipSuccess;
void set_last_error(hipError_t e) { g_last_error = e; }
}  // namespace wssdl

namespace wssdl {
static Tuning g_tuning;
Tuning &tuning() { return g_tuning; }

static int *tuning_field(const char *key) {
    if (!key) return nullptr;
    Tuning &t = g_tuning;
    if (!strcmp(key, "roi_bwd_plan")) return &t.roi_bwd_plan;
    if (!strcmp(key, "roi_bwd_owner")) return &t.roi_bwd_owner;
    if (!strcmp(key, "roi_bwd_owner_segments")) return &t.roi_bwd_owner_segments;
    if (!strcmp(key, "roi_fwd_variant")) return &t.roi_fwd_variant;
    if (!strcmp(key, "roi_fwd_one_bin")) return &t.roi_fwd_one_bin;
    if (!strcmp(key, "roi_fwd_blocks")) return &t.roi_fwd_blocks;
    if (!strcmp(key, "roi_fwd_blocks_sort")) return &t.roi_fwd_blocks_sort;
    if (!strcmp(key, "roi_fwd_blocks_parts")) return &t.roi_fwd_blocks_parts;
    if (!strcmp(key, "roi_bwdc_variant")) return &t.roi_bwdc_variant;
    if (!strcmp(key, "roi_bwd_cg")) return &t.roi_bwd_cg;
    if (!strcmp(key, "nms_one_pass")) return &t.nms_one_pass;
    if (!strcmp(key, "nms_fused")) return &t.nms_fused;
    if (!strcmp(key, "nms_sparse")) return &t.nms_sparse;
    if (!strcmp(key, "nms_wait_us")) return &t.nms_wait_us;
    if (!strcmp(key, "nms_sweep_async")) return &t.nms_sweep_async;
    if (!strcmp(key, "nms_fused_fault")) return &t.nms_fused_fault;
    if (!strcmp(key, "topk_sort")) return &t.topk_sort;
    return nullptr;
}
}  // namespace wssdl

extern "C" int wssdl_set_tuning(const char *key, int value) {
    int *f = wssdl::tuning_field(key);
    if (!f) return WSSDL_ERR_INVALID_ARGUMENT;
    *f = value;
    return WSSDL_OK;
}

extern "C" int wssdl_get_tuning(const char *key, int *value_host) {
    int *f = wssdl::tuning_field(key);
    if (!f || !value_host) return WSSDL_ERR_INVALID_ARGUMENT;
    *value_host = *f;
    return WSSDL_OK;
}

extern "C" const char *wssdl_version(void) { return "wssdl_bus_hip 0.1 (gfx950)"; }

extern "C" const char *wssdl_last_error(void) {
    return wssdl::g_last_error == hipSuccess ? "" : hipGetErrorString(wssdl::g_last_error);
}

// rpn_msr/generate_anchors.py:37-97.  A (0,0,base-1,base-1) window is reshaped to
// each aspect ratio at constant area -- widths and heights rounded half-to-even
// like np.round (:82-83) -- then scaled about its centre (:93-96).
extern "C" int wssdl_generate_anchors_host(int base_size, const double *ratios, int n_ratios,
                                           const double *scales, int n_scales, double *out) {
    if (base_size < 1 || !ratios || !scales || !out || n_ratios < 1 || n_scales < 1)
        return -WSSDL_ERR_INVALID_ARGUMENT;
    if (n_ratios * n_scales > WSSDL_MAX_ANCHORS) return -WSSDL_ERR_INVALID_ARGUMENT;
    const double size = (double)base_size * (double)base_size;
    const double ctr = 0.5 * ((double)base_size - 1.0);
    int k = 0;
    for (int r = 0; r < n_ratios; ++r) {
        double w = nearbyint(sqrt(size / ratios[r]));     // default rounding mode = half-to-even
        double h = nearbyint(w * ratios[r]);
        for (int s = 0; s < n_scales; ++s, ++k) {
            double sw = w * scales[s], sh = h * scales[s];
            out[k * 4 + 0] = ctr - 0.5 * (sw - 1.0);
            out[k * 4 + 1] = ctr - 0.5 * (sh - 1.0);
            out[k * 4 + 2] = ctr + 0.5 * (sw - 1.0);
            out[k * 4 + 3] = ctr + 0.5 * (sh - 1.0);
        }
    }
    return k;
}

namespace wssdl {
// anchor_target_layer_tf_bus.py:59-73: row (h*W+w)*A+a = base[a] + stride*(w,h,w,h)
__global__ __launch_bounds__(256) void shifted_anchors_kernel(BaseAnchors base, int A, int H, int W,
                                                              int stride, double *__restrict__ out) {
    const int total = H * W * A * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int j = i & 3;
        int row = i >> 2;
        int a = row % A;
        int cell = row / A;
        int w = cell % W, h = cell / W;
        out[i] = base.v[a][j] + (double)(stride * ((j & 1) ? h : w));
    }
}
}  // namespace wssdl

extern "C" int wssdl_shifted_anchors(const double *base_host, int A, int H, int W, int feat_stride,
                                     double *out, wssdl_stream_t stream) {
    wssdl::BaseAnchors b;
    int rc = wssdl::load_base_anchors(base_host, A, &b);
    if (rc) return rc;
    if (H < 1 || W < 1 || !out || (long long)H * W * A * 4 > 0x7fffffffLL)
        return WSSDL_ERR_INVALID_ARGUMENT;
    int blocks = wssdl::cdiv((long long)H * W * A * 4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wssdl::shifted_anchors_kernel, dim3(blocks), dim3(256), 0,
                       wssdl::as_stream(stream), b, A, H, W, feat_stride, out);
    return wssdl::check_launch();
}
