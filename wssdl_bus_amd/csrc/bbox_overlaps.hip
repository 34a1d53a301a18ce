// f64 box overlaps for gfx950: bbox_overlaps (IoU) and bbox_overlaps_ui
// (intersection / area of the first box).
//
// Reference: code/lib/utils/bbox.pyx:15-55, code/lib/utils/bbox_ui.pyx:12-47.
// All-double arithmetic in the reference's operation order, compiled with
// -ffp-contract=off, IEEE division: results are bit-identical to the Cython
// kernels (anchor labels depend on `overlap == column max` and `>= 0.7`).
//
// One lane per box row n; the K query boxes (K <= ~20 on the hot path) are
// staged once per workgroup in LDS and read as broadcasts; each lane writes its
// K outputs as one contiguous row (K*8 bytes), rows of consecutive lanes are
// adjacent, so stores coalesce.  Launch-latency bound at hot-path sizes
// (8 151 x 3 doubles): see DESIGN.md.
#include "common.hip.h"

namespace wssdl {

constexpr int QTILE = 256;   // query boxes staged per pass

__device__ __forceinline__ double iou_pair(double bx1, double by1, double bx2, double by2,
                                           double qx1, double qy1, double qx2, double qy2,
                                           double qarea) {
    double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1;
    if (iw > 0) {
        double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1;
        if (ih > 0) {
            double ua = (bx2 - bx1 + 1) * (by2 - by1 + 1) + qarea - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0;
}

__device__ __forceinline__ double ui_pair(double bx1, double by1, double bx2, double by2,
                                          double barea, double qx1, double qy1, double qx2,
                                          double qy2) {
    double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1;
    if (iw > 0) {
        double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1;
        if (ih > 0) return iw * ih / barea;
    }
    return 0.0;
}

template <bool UI>
__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const double *__restrict__ boxes,
                                                            long long N, int bstride,
                                                            const double *__restrict__ query,
                                                            long long K, int qstride,
                                                            double *__restrict__ out) {
    __shared__ double q[QTILE][5];
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double bx1 = 0, by1 = 0, bx2 = 0, by2 = 0, barea = 0;
    if (n < N) {
        const double *b = boxes + n * bstride;
        bx1 = b[0]; by1 = b[1]; bx2 = b[2]; by2 = b[3];
        barea = (bx2 - bx1 + 1) * (by2 - by1 + 1);
    }
    for (long long k0 = 0; k0 < K; k0 += QTILE) {
        int kt = (int)((K - k0 < QTILE) ? (K - k0) : QTILE);
        __syncthreads();
        for (int i = threadIdx.x; i < kt; i += blockDim.x) {
            const double *qq = query + (k0 + i) * qstride;
            q[i][0] = qq[0]; q[i][1] = qq[1]; q[i][2] = qq[2]; q[i][3] = qq[3];
            q[i][4] = (qq[2] - qq[0] + 1) * (qq[3] - qq[1] + 1);
        }
        __syncthreads();
        if (n < N) {
            double *o = out + n * K + k0;
            for (int i = 0; i < kt; ++i)
                o[i] = UI ? ui_pair(bx1, by1, bx2, by2, barea, q[i][0], q[i][1], q[i][2], q[i][3])
                          : iou_pair(bx1, by1, bx2, by2, q[i][0], q[i][1], q[i][2], q[i][3], q[i][4]);
        }
    }
}

template <bool UI>
static int launch(const double *boxes, int64_t N, int bs, const double *query, int64_t K, int qs,
                  double *out, wssdl_stream_t stream) {
    if (N < 0 || K < 0 || bs < 4 || qs < 4) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0 || K == 0) return WSSDL_OK;
    if (!boxes || !query || !out) return WSSDL_ERR_INVALID_ARGUMENT;
    long long blocks = (N + 255) / 256;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(bbox_overlaps_kernel<UI>, dim3((unsigned)blocks), dim3(256), 0,
                       as_stream(stream), boxes, (long long)N, bs, query, (long long)K, qs, out);
    return check_launch();
}

}  // namespace wssdl

extern "C" int wssdl_bbox_overlaps(const double *boxes, int64_t N, int box_stride,
                                   const double *query, int64_t K, int query_stride, double *out,
                                   wssdl_stream_t stream) {
    return wssdl::launch<false>(boxes, N, box_stride, query, K, query_stride, out, stream);
}

extern "C" int wssdl_bbox_overlaps_ui(const double *boxes, int64_t N, int box_stride,
                                      const double *query, int64_t K, int query_stride,
                                      double *out, wssdl_stream_t stream) {
    return wssdl::launch<true>(boxes, N, box_stride, query, K, query_stride, out, stream);
}
