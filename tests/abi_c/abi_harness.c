/* A plain C caller of libwssdl_bus_hip.so: no Python, no torch, no C++ -- the drop-in boundary as a
 * maintainer's native code would use it (include/wssdl_bus_hip.h).  Device memory comes
 * from the HIP runtime's C API.  Reads a small problem from stdin, prints the results as text:
 *
 *   anchors                                   -> the 9 base anchors (wssdl_generate_anchors_host)
 *   iou  N K  <N*4 doubles> <K*4 doubles>     -> N x K overlaps        (wssdl_bbox_overlaps)
 *   nms  N thresh  <N*5 floats>               -> kept indices          (wssdl_nms, bitmask NMS with the cpu_nms rule)
 *   pool N H W C R  <N*H*W*C floats> <R*5 floats>  -> top [R,7,7,C] and argmax (wssdl_roi_pool_forward)
 *
 * tests/test_gpu_abi_c.py builds this with gcc and compares its output with the oracle. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wssdl_bus_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define CHECK_WS(x) do { int rc_ = (x); if (rc_ != WSSDL_OK) { fprintf(stderr, "wssdl status %d (%s) at %s:%d\n", rc_, wssdl_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

static void *dev_copy(const void *host, size_t bytes) {
    void *d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes ? bytes : 16));
    if (bytes) CHECK_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    return d;
}

int main(void) {
    char cmd[32];
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    while (scanf("%31s", cmd) == 1) {
        if (!strcmp(cmd, "anchors")) {
            const double ratios[3] = {0.5, 1.0, 2.0}, scales[3] = {8, 16, 32};
            double out[9 * 4];
            int n = wssdl_generate_anchors_host(16, ratios, 3, scales, 3, out);
            printf("anchors %d\n", n);
            for (int i = 0; i < n * 4; ++i) printf("%.17g%c", out[i], (i & 3) == 3 ? '\n' : ' ');
        } else if (!strcmp(cmd, "iou")) {
            long long N, K;
            if (scanf("%lld %lld", &N, &K) != 2) return 1;
            double *b = malloc(sizeof(double) * N * 4), *q = malloc(sizeof(double) * K * 4), *o = malloc(sizeof(double) * N * K);
            for (long long i = 0; i < N * 4; ++i) if (scanf("%lf", &b[i]) != 1) return 1;
            for (long long i = 0; i < K * 4; ++i) if (scanf("%lf", &q[i]) != 1) return 1;
            double *db = dev_copy(b, sizeof(double) * N * 4), *dq = dev_copy(q, sizeof(double) * K * 4), *dout = dev_copy(NULL, 0);
            CHECK_HIP(hipFree(dout));
            CHECK_HIP(hipMalloc((void **)&dout, sizeof(double) * N * K));
            CHECK_WS(wssdl_bbox_overlaps(db, N, 4, dq, K, 4, dout, st));
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(o, dout, sizeof(double) * N * K, hipMemcpyDeviceToHost));
            printf("iou %lld %lld\n", N, K);
            for (long long i = 0; i < N * K; ++i) printf("%.17g%c", o[i], ((i + 1) % K) ? ' ' : '\n');
            hipFree(db); hipFree(dq); hipFree(dout); free(b); free(q); free(o);
        } else if (!strcmp(cmd, "nms")) {
            int N;
            double thresh;
            if (scanf("%d %lf", &N, &thresh) != 2) return 1;
            float *d = malloc(sizeof(float) * N * 5);
            for (int i = 0; i < N * 5; ++i) if (scanf("%f", &d[i]) != 1) return 1;
            float *dd = dev_copy(d, sizeof(float) * N * 5);
            size_t wsb = wssdl_nms_workspace_bytes(N);
            void *ws = NULL;
            int32_t *keep = NULL, *nk = NULL;
            CHECK_HIP(hipMalloc(&ws, wsb));
            CHECK_HIP(hipMalloc((void **)&keep, sizeof(int32_t) * N));
            CHECK_HIP(hipMalloc((void **)&nk, sizeof(int32_t)));
            CHECK_WS(wssdl_nms(dd, N, thresh, N, keep, nk, ws, wsb, st));
            CHECK_HIP(hipStreamSynchronize(st));
            int32_t n_keep = 0, *hk = malloc(sizeof(int32_t) * N);
            CHECK_HIP(hipMemcpy(&n_keep, nk, sizeof(int32_t), hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(hk, keep, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
            printf("nms %d\n", n_keep);
            for (int i = 0; i < n_keep; ++i) printf("%d%c", hk[i], i + 1 == n_keep ? '\n' : ' ');
            if (n_keep == 0) printf("\n");
            hipFree(dd); hipFree(ws); hipFree(keep); hipFree(nk); free(d); free(hk);
        } else if (!strcmp(cmd, "pool")) {
            int N, H, W, C, R;
            if (scanf("%d %d %d %d %d", &N, &H, &W, &C, &R) != 5) return 1;
            size_t nf = (size_t)N * H * W * C, nt = (size_t)R * 49 * C;
            float *f = malloc(sizeof(float) * nf), *r = malloc(sizeof(float) * R * 5), *top = malloc(sizeof(float) * nt);
            int32_t *arg = malloc(sizeof(int32_t) * nt);
            for (size_t i = 0; i < nf; ++i) if (scanf("%f", &f[i]) != 1) return 1;
            for (int i = 0; i < R * 5; ++i) if (scanf("%f", &r[i]) != 1) return 1;
            float *df = dev_copy(f, sizeof(float) * nf), *dr = dev_copy(r, sizeof(float) * R * 5), *dt = NULL;
            int32_t *da = NULL;
            CHECK_HIP(hipMalloc((void **)&dt, sizeof(float) * nt));
            CHECK_HIP(hipMalloc((void **)&da, sizeof(int32_t) * nt));
            /* the checked form first (the real N: a batch index >= N would be an empty RoI), then the way the reference's
             * launcher body would call it: ROIPoolForwardLaucher is not told the batch size (roi_pooling_op_gpu.h:17-21),
             * N = WSSDL_ROI_BATCH_UNKNOWN.  Every index here is in range, so both give the
             * same tensors; the second call's are the ones printed.  N = 0 with RoIs is refused. */
            CHECK_WS(wssdl_roi_pool_forward(df, N, H, W, C, dr, R, 7, 7, 1.0f / 16.0f, WSSDL_ROI_ROUND_CUDA, dt, da, st));
            if (wssdl_roi_pool_forward(df, 0, H, W, C, dr, R, 7, 7, 1.0f / 16.0f, WSSDL_ROI_ROUND_CUDA, dt, da, st) !=
                WSSDL_ERR_INVALID_ARGUMENT) {
                fprintf(stderr, "N = 0 with RoIs must be WSSDL_ERR_INVALID_ARGUMENT\n");
                return 3;
            }
            CHECK_WS(wssdl_roi_pool_forward(df, WSSDL_ROI_BATCH_UNKNOWN, H, W, C, dr, R, 7, 7, 1.0f / 16.0f, WSSDL_ROI_ROUND_CUDA,
                                            dt, da, st));
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(top, dt, sizeof(float) * nt, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(arg, da, sizeof(int32_t) * nt, hipMemcpyDeviceToHost));
            printf("pool %zu\n", nt);
            for (size_t i = 0; i < nt; ++i) printf("%.9g %d\n", top[i], arg[i]);
            hipFree(df); hipFree(dr); hipFree(dt); hipFree(da); free(f); free(r); free(top); free(arg);
        } else {
            fprintf(stderr, "unknown command %s\n", cmd);
            return 1;
        }
    }
    CHECK_HIP(hipStreamDestroy(st));
    printf("done %s\n", wssdl_version());
    return 0;
}
