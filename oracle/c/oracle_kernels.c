/*
 * oracle_kernels.c -- CPU restatement of the native kernels on the wssdl_bus
 * detection hot path.  TEST INFRASTRUCTURE ONLY: linked/loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker; the
 * product path (wssdl_bus_amd/) never loads it.
 *
 * Pinning status (DESIGN.md "Oracle"):
 *   orc_bbox_overlaps / orc_bbox_overlaps_ui / orc_cpu_nms are checked
 *   bit-for-bit against the reference's own Cython kernels (oracle/_ref, built
 *   from code/lib/utils/bbox.pyx, bbox_ui.pyx, code/lib/nms/cpu_nms.pyx) and
 *   against golden vectors produced by the imported reference.
 *   orc_roi_pool_* : PARITY UNPINNED by the reference -- its TF op cannot be
 *   built here (needs TensorFlow headers) and its only test asserts nothing
 *   (code/lib/roi_pooling_layer/roi_pooling_op_test.py).  Pinned instead by
 *   hand-computed known-answer cases in tests/test_oracle_roi_pool.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is compiled without FMA contraction
 * and anchor labels depend on exact f64 equality of IoU values.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ IoU --- */

/* follows code/lib/utils/bbox.pyx:15-55.  boxes [N,4], query [K,4], out [N,K]
 * (row-major f64).  +1 pixel convention; 0 unless iw>0 and ih>0. */
void orc_bbox_overlaps(const double *boxes, int64_t N, const double *query,
                       int64_t K, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)N * (size_t)K);
    for (int64_t k = 0; k < K; ++k) {
        const double *q = query + 4 * k;
        double qarea = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
        for (int64_t n = 0; n < N; ++n) {
            const double *b = boxes + 4 * n;
            double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
            if (iw > 0) {
                double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
                if (ih > 0) {
                    double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + qarea - iw * ih;
                    out[n * K + k] = iw * ih / ua;
                }
            }
        }
    }
}

/* follows code/lib/utils/bbox_ui.pyx:12-47: intersection / area(boxes[n]). */
void orc_bbox_overlaps_ui(const double *boxes, int64_t N, const double *query,
                          int64_t K, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)N * (size_t)K);
    for (int64_t n = 0; n < N; ++n) {
        const double *b = boxes + 4 * n;
        double barea = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
        for (int64_t k = 0; k < K; ++k) {
            const double *q = query + 4 * k;
            double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
            if (iw > 0) {
                double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
                if (ih > 0)
                    out[n * K + k] = iw * ih / barea;
            }
        }
    }
}

/* ------------------------------------------------------------------ NMS --- */

static inline float f32max(float a, float b) { return a >= b ? a : b; }   /* cpu_nms.pyx:11 */
static inline float f32min(float a, float b) { return a <= b ? a : b; }   /* cpu_nms.pyx:14 */

/* follows code/lib/nms/cpu_nms.pyx:17-68.  dets [n,5] f32 (x1,y1,x2,y2,score).
 * `order` is the visiting order (the caller computes scores.argsort()[::-1] so
 * that the tie rule is NumPy's, cpu_nms.pyx:25).  All box arithmetic is f32;
 * the threshold test is (double)ovr >= thresh (vendored cpu_nms.c:2495 compares
 * PyFloat objects).  Returns the number of kept indices written to keep[]. */
static int64_t nms_greedy(const float *dets, int64_t n, const int64_t *order,
                          double thresh, int64_t *keep, int contain);

int64_t orc_cpu_nms(const float *dets, int64_t n, const int64_t *order,
                    double thresh, int64_t *keep)
{
    return nms_greedy(dets, n, order, thresh, keep, 0);
}

/* follows code/lib/utils/nms.pyx:70-123 (`nms_new`; `nms` :17-68 of that file is line for line the
 * rule of cpu_nms.pyx, i.e. orc_cpu_nms).  ovr1 / ovr2 are not cdef'd there: the f32 quotients
 * inter / iarea and inter / areas[j] become Python floats and are compared with 0.95 in f64
 * (:118-120); a box is suppressed when `ovr >= thresh or ovr1 > 0.95 or ovr2 > 0.95`. */
int64_t orc_nms_new(const float *dets, int64_t n, const int64_t *order,
                    double thresh, int64_t *keep)
{
    return nms_greedy(dets, n, order, thresh, keep, 1);
}

static int64_t nms_greedy(const float *dets, int64_t n, const int64_t *order,
                          double thresh, int64_t *keep, int contain)
{
    float *areas = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    unsigned char *sup = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t nk = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *d = dets + 5 * i;
        /* numpy f32 vector ops: (x2 - x1 + 1) * (y2 - y1 + 1), cpu_nms.pyx:24 */
        float w = d[2] - d[0];
        w = w + 1.0f;
        float h = d[3] - d[1];
        h = h + 1.0f;
        areas[i] = w * h;
    }
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i])
            continue;
        keep[nk++] = i;
        const float *di = dets + 5 * i;
        float ix1 = di[0], iy1 = di[1], ix2 = di[2], iy2 = di[3], iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j])
                continue;
            const float *dj = dets + 5 * j;
            float xx1 = f32max(ix1, dj[0]);
            float yy1 = f32max(iy1, dj[1]);
            float xx2 = f32min(ix2, dj[2]);
            float yy2 = f32min(iy2, dj[3]);
            float w = xx2 - xx1;
            w = f32max(0.0f, w + 1.0f);
            float h = yy2 - yy1;
            h = f32max(0.0f, h + 1.0f);
            float inter = w * h;
            float den = iarea + areas[j];
            den = den - inter;
            float ovr = inter / den;
            if ((double)ovr >= thresh)
                sup[j] = 1;
            else if (contain) {                               /* nms.pyx:118-120 */
                float ovr1 = inter / iarea;
                float ovr2 = inter / areas[j];
                if ((double)ovr1 > 0.95 || (double)ovr2 > 0.95)
                    sup[j] = 1;
            }
        }
    }
    free(areas);
    free(sup);
    return nk;
}

/* -------------------------------------------------------------- RoI pool --- */

enum { ORC_ROUND_CUDA = 0, ORC_ROUND_CPU = 1 };

typedef struct {
    int batch, sw, sh, ew, eh, rw, rh;
    float bin_h, bin_w;
} roi_geom;

/* roi_pooling_op.cc:152-165 == roi_pooling_op_gpu.cu.cc:36-49 */
static roi_geom roi_geometry(const float *r, float scale, int PH, int PW)
{
    roi_geom g;
    g.batch = (int)r[0];
    g.sw = (int)roundf(r[1] * scale);
    g.sh = (int)roundf(r[2] * scale);
    g.ew = (int)roundf(r[3] * scale);
    g.eh = (int)roundf(r[4] * scale);
    g.rw = g.ew - g.sw + 1; if (g.rw < 1) g.rw = 1;
    g.rh = g.eh - g.sh + 1; if (g.rh < 1) g.rh = 1;
    g.bin_h = (float)g.rh / (float)PH;
    g.bin_w = (float)g.rw / (float)PW;
    return g;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Forward.  bottom [N,H,W,C] f32 NHWC, rois [R,5] f32 (batch,x1,y1,x2,y2),
 * top [R,PH,PW,C] f32, argmax [R,PH,PW,C] i32 (flat NHWC index inside the image,
 * -1 when the bin is empty).
 *   mode ORC_ROUND_CUDA: roi_pooling_op_gpu.cu.cc:51-58  floor(ph*bin), ceil((ph+1)*bin)
 *   mode ORC_ROUND_CPU : roi_pooling_op.cc:167-170        (int)(ph*bin),  (int)((ph+1)*bin)
 * Strict '>' against -FLT_MAX: first maximum in (h,w) scan order wins
 * (roi_pooling_op.cc:184-192). */
void orc_roi_pool_forward(const float *bottom, int N, int H, int W, int C,
                          const float *rois, int R, int PH, int PW, float scale,
                          int mode, float *top, int32_t *argmax)
{
    (void)N;
    for (int r = 0; r < R; ++r) {
        roi_geom g = roi_geometry(rois + 5 * r, scale, PH, PW);
        const float *img = bottom + (size_t)g.batch * H * W * C;
        for (int ph = 0; ph < PH; ++ph) {
            for (int pw = 0; pw < PW; ++pw) {
                int hstart, hend, wstart, wend;
                if (mode == ORC_ROUND_CPU) {
                    hstart = (int)((float)ph * g.bin_h);
                    wstart = (int)((float)pw * g.bin_w);
                    hend = (int)((float)(ph + 1) * g.bin_h);
                    wend = (int)((float)(pw + 1) * g.bin_w);
                } else {
                    hstart = (int)floorf((float)ph * g.bin_h);
                    wstart = (int)floorf((float)pw * g.bin_w);
                    hend = (int)ceilf((float)(ph + 1) * g.bin_h);
                    wend = (int)ceilf((float)(pw + 1) * g.bin_w);
                }
                hstart = clampi(hstart + g.sh, 0, H);
                hend = clampi(hend + g.sh, 0, H);
                wstart = clampi(wstart + g.sw, 0, W);
                wend = clampi(wend + g.sw, 0, W);
                int empty = (hend <= hstart) || (wend <= wstart);
                size_t o = (((size_t)r * PH + ph) * PW + pw) * C;
                for (int c = 0; c < C; ++c) {
                    float maxval = empty ? 0.0f : -FLT_MAX;
                    int maxidx = -1;
                    for (int h = hstart; h < hend; ++h)
                        for (int w = wstart; w < wend; ++w) {
                            int idx = (h * W + w) * C + c;
                            if (img[idx] > maxval) {
                                maxval = img[idx];
                                maxidx = idx;
                            }
                        }
                    top[o + c] = maxval;
                    argmax[o + c] = maxidx;
                }
            }
        }
    }
}

/* Backward, literally the reference's gather: roi_pooling_op.cc:387-457 ==
 * roi_pooling_op_gpu.cu.cc:121-189.  For every bottom element, over all RoIs of
 * its image whose rounded box contains (h,w), over the candidate bins, add
 * top_diff where argmax points at this element; f32 sum in order roi^, ph^, pw^.
 * O(N*H*W*C*R): use on small cases only. */
void orc_roi_pool_backward(const float *top_diff, const int32_t *argmax,
                           const float *rois, int R, int N, int H, int W, int C,
                           int PH, int PW, float scale, float *bottom_diff)
{
    roi_geom *gs = (roi_geom *)malloc(sizeof(roi_geom) * (size_t)(R > 0 ? R : 1));
    for (int r = 0; r < R; ++r)
        gs[r] = roi_geometry(rois + 5 * r, scale, PH, PW);
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                float *bd = bottom_diff + (((size_t)n * H + h) * W + w) * C;
                for (int c = 0; c < C; ++c)
                    bd[c] = 0.0f;
                for (int r = 0; r < R; ++r) {
                    const roi_geom *g = gs + r;
                    if (g->batch != n)
                        continue;
                    if (!(w >= g->sw && w <= g->ew && h >= g->sh && h <= g->eh))
                        continue;
                    int phstart = (int)floorf((float)(h - g->sh) / g->bin_h);
                    int phend = (int)ceilf((float)(h - g->sh + 1) / g->bin_h);
                    int pwstart = (int)floorf((float)(w - g->sw) / g->bin_w);
                    int pwend = (int)ceilf((float)(w - g->sw + 1) / g->bin_w);
                    phstart = clampi(phstart, 0, PH);
                    phend = clampi(phend, 0, PH);
                    pwstart = clampi(pwstart, 0, PW);
                    pwend = clampi(pwend, 0, PW);
                    size_t off = (size_t)r * PH * PW * C;
                    for (int ph = phstart; ph < phend; ++ph)
                        for (int pw = pwstart; pw < pwend; ++pw) {
                            size_t o = off + ((size_t)ph * PW + pw) * C;
                            for (int c = 0; c < C; ++c)
                                if (argmax[o + c] == (h * W + w) * C + c)
                                    bd[c] += top_diff[o + c];
                        }
                }
            }
    free(gs);
}

/* Same result as orc_roi_pool_backward, restated as an ordered scatter so that
 * full-size cases finish in seconds: walk (roi^, ph^, pw^) and add each
 * top_diff into the element its argmax names, after applying the reference's
 * in_roi and candidate-bin tests to that element.  Per destination the f32
 * additions happen in the same order as in the gather.  Verified equal
 * (bitwise) to orc_roi_pool_backward in tests/test_oracle_roi_pool.py. */
void orc_roi_pool_backward_scatter(const float *top_diff, const int32_t *argmax,
                                   const float *rois, int R, int N, int H, int W,
                                   int C, int PH, int PW, float scale,
                                   float *bottom_diff)
{
    memset(bottom_diff, 0, sizeof(float) * (size_t)N * H * W * C);
    for (int r = 0; r < R; ++r) {
        roi_geom g = roi_geometry(rois + 5 * r, scale, PH, PW);
        if (g.batch < 0 || g.batch >= N)
            continue;
        float *img = bottom_diff + (size_t)g.batch * H * W * C;
        for (int ph = 0; ph < PH; ++ph)
            for (int pw = 0; pw < PW; ++pw) {
                size_t o = (((size_t)r * PH + ph) * PW + pw) * C;
                for (int c = 0; c < C; ++c) {
                    int idx = argmax[o + c];
                    if (idx < 0)
                        continue;
                    if (idx % C != c)
                        continue;
                    int cell = idx / C;
                    int h = cell / W, w = cell % W;
                    if (h >= H)
                        continue;
                    if (!(w >= g.sw && w <= g.ew && h >= g.sh && h <= g.eh))
                        continue;
                    int phstart = clampi((int)floorf((float)(h - g.sh) / g.bin_h), 0, PH);
                    int phend = clampi((int)ceilf((float)(h - g.sh + 1) / g.bin_h), 0, PH);
                    int pwstart = clampi((int)floorf((float)(w - g.sw) / g.bin_w), 0, PW);
                    int pwend = clampi((int)ceilf((float)(w - g.sw + 1) / g.bin_w), 0, PW);
                    if (ph >= phstart && ph < phend && pw >= pwstart && pw < pwend)
                        img[idx] += top_diff[o + c];
                }
            }
    }
}

/* The ordered scatter restricted to channels [c0, c1) and WITHOUT the zero fill, for the
 * cpu_baseline leg: channels are independent, so the caller zeroes bottom_diff once and runs
 * one call per host thread on disjoint channel ranges; per destination element the f32
 * additions still happen in the order roi^, ph^, pw^ (results identical to the function above). */
void orc_roi_pool_backward_scatter_channels(const float *top_diff, const int32_t *argmax,
                                            const float *rois, int R, int N, int H, int W,
                                            int C, int PH, int PW, float scale, int c0, int c1,
                                            float *bottom_diff)
{
    for (int r = 0; r < R; ++r) {
        roi_geom g = roi_geometry(rois + 5 * r, scale, PH, PW);
        if (g.batch < 0 || g.batch >= N)
            continue;
        float *img = bottom_diff + (size_t)g.batch * H * W * C;
        for (int ph = 0; ph < PH; ++ph)
            for (int pw = 0; pw < PW; ++pw) {
                size_t o = (((size_t)r * PH + ph) * PW + pw) * C;
                for (int c = c0; c < c1; ++c) {
                    int idx = argmax[o + c];
                    if (idx < 0 || idx % C != c)
                        continue;
                    int cell = idx / C;
                    int h = cell / W, w = cell % W;
                    if (h >= H)
                        continue;
                    if (!(w >= g.sw && w <= g.ew && h >= g.sh && h <= g.eh))
                        continue;
                    int phstart = clampi((int)floorf((float)(h - g.sh) / g.bin_h), 0, PH);
                    int phend = clampi((int)ceilf((float)(h - g.sh + 1) / g.bin_h), 0, PH);
                    int pwstart = clampi((int)floorf((float)(w - g.sw) / g.bin_w), 0, PW);
                    int pwend = clampi((int)ceilf((float)(w - g.sw + 1) / g.bin_w), 0, PW);
                    if (ph >= phstart && ph < phend && pw >= pwstart && pw < pwend)
                        img[idx] += top_diff[o + c];
                }
            }
    }
}

/* Threaded forward for the cpu_baseline leg: the reference shards the flat
 * output range over TF's intra-op pool (roi_pooling_op.cc:198-203); here the
 * RoI range is split over `nthreads` OpenMP-free pthreads-free workers by the
 * caller (python threads release the GIL inside ctypes), so this entry point
 * just processes RoIs [r0, r1). */
void orc_roi_pool_forward_range(const float *bottom, int N, int H, int W, int C,
                                const float *rois, int r0, int r1, int PH, int PW,
                                float scale, int mode, float *top, int32_t *argmax)
{
    orc_roi_pool_forward(bottom, N, H, W, C, rois + 5 * (size_t)r0, r1 - r0, PH, PW,
                         scale, mode, top + (size_t)r0 * PH * PW * C,
                         argmax + (size_t)r0 * PH * PW * C);
}
