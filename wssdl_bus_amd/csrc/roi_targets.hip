// Proposal-target layer kernels for gfx950 (MI355X).
//
// Reference: code/lib/rpn_msr/proposal_target_layer_tf_bus.py:228-280 (_sample_rois),
// :187-226 (_get_bbox_regression_labels, _compute_targets), utils/bbox.pyx:15-55,
// fast_rcnn/bbox_transform.py:10-28.
//
// roi_gt_assign : one lane per candidate RoI: f64 IoU against the positive gt boxes
//                 of the RoI's image (bit-identical to the Cython kernel), maximum
//                 and first arg-max.  The random fg/bg sampling between the two
//                 kernels belongs to the host layer (it consumes numpy's legacy
//                 RandomState stream in the reference).
// roi_sample    : the device alternative to that host step (cfg.SAMPLING_RNG =
//                 'device'): one workgroup per supervised image draws exactly
//                 min(quota, #) fg and bg rows by 64-bit radix select on
//                 counter-based hash keys; rows come out in candidate order
//                 (fg first), so a run is reproducible from the seed alone.
// roi_targets   : one lane per sampled RoI: label (background clamped to 0, :265),
//                 all-f32 bbox_transform (:220), expansion into the 4*num_classes
//                 layout with inside / outside weights (:199-209, :89).
#include "common.hip.h"
#include "select.hip.h"

namespace wssdl {

// IoU arg-max of one candidate row (batch index, box) over the positive boxes of its image (:236-240)
__device__ __forceinline__ void roi_gt_assign_row(const float *b, const float *__restrict__ gt_boxes, int max_gt, int n_pos,
                                                  int n_images, double &best, int &arg) {
    const int img = (int)b[0];
    best = 0.0;
    arg = -1;
    if (img >= 0 && img < n_images) {
        const double bx1 = b[1], by1 = b[2], bx2 = b[3], by2 = b[4];
        const int np = min(n_pos, max_gt);
        const float *g = gt_boxes + (size_t)img * max_gt * 5;
        for (int k = 0; k < np; ++k) {
            const double qx1 = g[k * 5 + 0], qy1 = g[k * 5 + 1], qx2 = g[k * 5 + 2], qy2 = g[k * 5 + 3];
            const double qarea = (qx2 - qx1 + 1) * (qy2 - qy1 + 1);
            double ov = 0.0;
            double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1;
            if (iw > 0) {
                double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1;
                if (ih > 0) {
                    double ua = (bx2 - bx1 + 1) * (by2 - by1 + 1) + qarea - iw * ih;
                    ov = iw * ih / ua;
                }
            }
            if (k == 0 || ov > best) { best = ov; arg = k; }   // numpy argmax: first maximum
        }
    }
}

__global__ __launch_bounds__(256) void roi_gt_assign_kernel(
    const float *__restrict__ rois, int R, const float *__restrict__ gt_boxes, int max_gt,
    const int *__restrict__ num_pos, int n_images, double *__restrict__ max_overlap,
    int *__restrict__ assignment) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *b = rois + (size_t)r * 5;
    const int img = (int)b[0];
    double best;
    int arg;
    roi_gt_assign_row(b, gt_boxes, max_gt, (img >= 0 && img < n_images) ? num_pos[img] : 0, n_images, best, arg);
    max_overlap[r] = best;
    assignment[r] = arg;
}

// cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED (proposal_target_layer_tf_bus.py:221-224):
// targets = (targets - np.array(MEANS)) / np.array(STDS) -- the f32 targets meet f64 arrays, so the subtraction and the
// division happen in f64 and the result is rounded to f32 once (the hstack + astype(np.float32) of :225-226)
struct RoiTargetNorm {
    int on;
    double mean[4], std[4];
};

__global__ __launch_bounds__(256) void roi_targets_kernel(
    const float *__restrict__ rois, const int *__restrict__ keep,
    const unsigned char *__restrict__ is_fg, int n_keep, const int *__restrict__ assignment,
    const float *__restrict__ gt_boxes, int max_gt, int num_classes, float iw0, float iw1,
    float iw2, float iw3, RoiTargetNorm norm, float *__restrict__ rois_out, float *__restrict__ labels,
    float *__restrict__ bbox_targets, float *__restrict__ inside_w, float *__restrict__ outside_w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_keep) return;
    const int r = keep[p];
    float *ro = rois_out + (size_t)p * 5;
    const int width = 4 * num_classes;
    float *to = bbox_targets + (size_t)p * width;
    float *io = inside_w + (size_t)p * width;
    float *oo = outside_w + (size_t)p * width;
    for (int j = 0; j < width; ++j) { to[j] = 0.0f; io[j] = 0.0f; oo[j] = 0.0f; }
    if (r < 0) {
        // padding slot of a fixed-shape keep list (an image ran short of candidates): a row no
        // consumer acts on -- batch index -1 (RoI pooling treats it as empty), label -1 (ignored)
        ro[0] = -1.0f; ro[1] = 0.0f; ro[2] = 0.0f; ro[3] = 0.0f; ro[4] = 0.0f;
        labels[p] = -1.0f;
        return;
    }
    const float *b = rois + (size_t)r * 5;
    ro[0] = b[0]; ro[1] = b[1]; ro[2] = b[2]; ro[3] = b[3]; ro[4] = b[4];
    const int img = (int)b[0];
    const int k = assignment[r];
    float label = 0.0f;
    if (k >= 0 && is_fg[p]) {
        const float *g = gt_boxes + ((size_t)img * max_gt + k) * 5;
        label = g[4];
        const int cls = (int)label;
        if (label > 0.0f && cls < num_classes) {
            // bbox_transform with f32 operands on both sides (:220)
            float ew = b[3] - b[1];  ew = ew + 1.0f;
            float eh = b[4] - b[2];  eh = eh + 1.0f;
            float hew = 0.5f * ew, heh = 0.5f * eh;
            const float ecx = b[1] + hew, ecy = b[2] + heh;
            float gw = g[2] - g[0];  gw = gw + 1.0f;
            float gh = g[3] - g[1];  gh = gh + 1.0f;
            float hgw = 0.5f * gw, hgh = 0.5f * gh;
            const float gcx = g[0] + hgw, gcy = g[1] + hgh;
            float t0 = (gcx - ecx) / ew;
            float t1 = (gcy - ecy) / eh;
            float t2 = (float)log((double)(gw / ew));
            float t3 = (float)log((double)(gh / eh));
            if (norm.on) {
                t0 = (float)(((double)t0 - norm.mean[0]) / norm.std[0]);
                t1 = (float)(((double)t1 - norm.mean[1]) / norm.std[1]);
                t2 = (float)(((double)t2 - norm.mean[2]) / norm.std[2]);
                t3 = (float)(((double)t3 - norm.mean[3]) / norm.std[3]);
            }
            const int s = 4 * cls;
            to[s + 0] = t0; to[s + 1] = t1; to[s + 2] = t2; to[s + 3] = t3;
            io[s + 0] = iw0; io[s + 1] = iw1; io[s + 2] = iw2; io[s + 3] = iw3;
            oo[s + 0] = iw0 > 0.0f ? 1.0f : 0.0f;
            oo[s + 1] = iw1 > 0.0f ? 1.0f : 0.0f;
            oo[s + 2] = iw2 > 0.0f ? 1.0f : 0.0f;
            oo[s + 3] = iw3 > 0.0f ? 1.0f : 0.0f;
        }
    }
    labels[p] = label;
}

// ------------------------------------------------------- candidate rows ---
// :40-50 on the device: cand = rpn_rois followed, per supervised image, by all max_gt slots of
// its gt array; a slot carries the image's batch index when it is one of the image's positive
// boxes (the first num_pos rows: class != 0 among the num_gt valid rows) and -1 otherwise, so
// that it can never be drawn.  Also emits num_pos per image.
__global__ __launch_bounds__(256) void roi_candidates_kernel(
    const float *__restrict__ rois, int R, const float *__restrict__ gt_boxes, int max_gt,
    const int *__restrict__ num_gt, int n_images, const int *__restrict__ images, int S,
    int append_gt, float *__restrict__ cand, int *__restrict__ num_pos, double *__restrict__ max_overlap /* or NULL */,
    int *__restrict__ assignment) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    auto count_pos = [&](int img) {
        const int ng = min(max(num_gt[img], 0), max_gt);
        int c = 0;
        for (int k = 0; k < ng; ++k) c += gt_boxes[((size_t)img * max_gt + k) * 5 + 4] != 0.0f ? 1 : 0;
        return c;
    };
    if (row < n_images) num_pos[row] = count_pos(row);
    const int total = R + (append_gt ? S * max_gt : 0);
    if (row >= total) return;
    float *o = cand + (size_t)row * 5;
    float c[5];
    if (row < R) {
        const float *b = rois + (size_t)row * 5;
        c[0] = b[0]; c[1] = b[1]; c[2] = b[2]; c[3] = b[3]; c[4] = b[4];
    } else {
        const int s = (row - R) / max_gt, k = (row - R) % max_gt;
        const int img = images[s];
        const float *g = gt_boxes + ((size_t)img * max_gt + k) * 5;
        c[0] = (k < count_pos(img)) ? (float)img : -1.0f;
        c[1] = g[0]; c[2] = g[1]; c[3] = g[2]; c[4] = g[3];
    }
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3]; o[4] = c[4];
    if (max_overlap) {
        // the assignment of wssdl_roi_gt_assign in the same launch (the row's own count of positives: the
        // num_pos array is written by other threads of this launch)
        const int img = (int)c[0];
        double best;
        int arg;
        roi_gt_assign_row(c, gt_boxes, max_gt, (img >= 0 && img < n_images) ? count_pos(img) : 0, n_images, best, arg);
        max_overlap[row] = best;
        assignment[row] = arg;
    }
}

// ------------------------------------------------------- device sampling ---
constexpr int RS_BLOCK = 1024;

__device__ __forceinline__ unsigned long long rs_mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// unique per candidate: hash in the high word, candidate index in the low word
__device__ __forceinline__ unsigned long long rs_key(unsigned long long seed, int img, int i) {
    unsigned long long h = rs_mix64(seed ^ rs_mix64(((unsigned long long)(unsigned)img << 34) ^ (unsigned)i));
    return (h & 0xFFFFFFFF00000000ull) | (unsigned)i;
}

struct RoiClassifier {          // _sample_rois :241, :253-254
    const float *cand;
    const double *ov;
    int img;
    double fg_thresh, bg_hi, bg_lo;
    // 1 = fg, 0 = bg, -1 = neither / other image
    __device__ __forceinline__ int operator()(int i) const {
        if ((int)cand[(size_t)i * 5] != img) return -1;
        const double o = ov[i];
        if (o >= fg_thresh) return 1;
        return (o < bg_hi && o >= bg_lo) ? 0 : -1;
    }
};

constexpr int RS_LIST = 512;

// key of the quota-th smallest among the candidates of class `which` (all of them when there
// are no more than quota: returns ~0)
__device__ unsigned long long rs_select(const RoiClassifier &cls, int lo, int hi, int which, int n_have,
                                        int quota, unsigned long long seed,
                                        SelectScratch<RS_LIST> &sc) {
    if (n_have <= quota || quota <= 0) return ~0ull;
    return block_radix_select<RS_BLOCK, RS_LIST, false>(
        [=](int j, unsigned long long &v) {
            const int i = lo + j;
            if (cls(i) != which) return false;
            v = rs_key(seed, cls.img, i);
            return true;
        }, hi - lo, [quota](int) { return quota; }, sc);
}

__device__ int rs_block_exclusive_scan(int v, int *s_wave, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RS_BLOCK / 64; ++w) {
        const int x = s_wave[w];
        base += (w < wave) ? x : 0;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(RS_BLOCK) void roi_sample_kernel(
    const float *__restrict__ cand, const double *__restrict__ max_overlap, int Rc,
    const int *__restrict__ images, int rois_per_image, int fg_rois_per_image, double fg_thresh,
    double bg_hi, double bg_lo, unsigned long long seed, int *__restrict__ keep,
    unsigned char *__restrict__ is_fg, int *__restrict__ counts) {
    __shared__ SelectScratch<RS_LIST> sc;
    __shared__ int s_wave[RS_BLOCK / 64];
    const int s = blockIdx.x, t = threadIdx.x;
    RoiClassifier cls;
    cls.cand = cand;  cls.ov = max_overlap;  cls.img = images[s];
    cls.fg_thresh = fg_thresh;  cls.bg_hi = bg_hi;  cls.bg_lo = bg_lo;
    // The rows of image `img` usually are one stretch of the candidate list (the proposal blob is ordered by
    // image, the gt rows follow): [lo, hi) = the span that holds all of them, found by one coalesced pass over
    // the image column; everything after it only walks the span (8512 candidates of 8 images: every pass
    // was 9 dependent load pairs per thread, 50 us for the kernel).  Any order of the rows stays correct.
    __shared__ int s_lo, s_hi;
    if (t == 0) { s_lo = Rc;  s_hi = 0; }
    __syncthreads();
    {
        int my_lo = Rc, my_hi = 0;
#pragma unroll 4
        for (int i = t; i < Rc; i += RS_BLOCK)
            if ((int)cand[(size_t)i * 5] == cls.img) { my_lo = min(my_lo, i);  my_hi = i + 1; }
        // wave minimum / maximum first: ~1000 lanes on the two LDS words cost more than the passes this saves
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            my_lo = min(my_lo, __shfl_xor(my_lo, off, 64));
            my_hi = max(my_hi, __shfl_xor(my_hi, off, 64));
        }
        if ((t & 63) == 0 && my_hi > 0) { atomicMin(&s_lo, my_lo);  atomicMax(&s_hi, my_hi); }
    }
    __syncthreads();
    const int lo = min(s_lo, s_hi), hi = s_hi;
    // each thread owns a contiguous slice of the span so that the output keeps the candidate order; the class
    // of its rows is kept in two bit masks (slices of up to 32 rows; longer ones re-read)
    const int per = (hi - lo + RS_BLOCK - 1) / RS_BLOCK;
    const int i0 = min(lo + t * per, hi), i1 = min(i0 + per, hi);
    const bool masked = per <= 32;
    unsigned m_fg = 0u, m_bg = 0u;
    int cf = 0, cb = 0;
    for (int i = i0; i < i1; ++i) {
        const int c = cls(i);
        cf += c == 1;
        cb += c == 0;
        if (masked) { m_fg |= (unsigned)(c == 1) << (i - i0);  m_bg |= (unsigned)(c == 0) << (i - i0); }
    }
    auto class_of = [&](int i) { return masked ? (int)((m_fg >> (i - i0)) & 1u) - (int)(1u & ~((m_fg | m_bg) >> (i - i0))) : cls(i); };
    int have_fg, have_bg;
    rs_block_exclusive_scan(cf, s_wave, have_fg);
    rs_block_exclusive_scan(cb, s_wave, have_bg);
    const int n_fg = min(fg_rois_per_image, have_fg);                 // :243
    const int n_bg = min(rois_per_image - n_fg, have_bg);             // :256-258
    // the two draws only share these counts: with gridDim.y == 2 workgroup (s, 0) draws and emits
    // the fg rows, workgroup (s, 1) the bg rows (and the padding) -- half the latency of the chain
    const bool do_fg = gridDim.y == 1 || blockIdx.y == 0, do_bg = gridDim.y == 1 || blockIdx.y == 1;
    unsigned long long t_fg = 0ull, t_bg = 0ull;
    if (do_fg) t_fg = rs_select(cls, lo, hi, 1, have_fg, n_fg, seed, sc);
    if (do_bg) t_bg = rs_select(cls, lo, hi, 0, have_bg, n_bg, seed ^ 0x5bd1e995ull, sc);
    const bool emit_fg = do_fg && n_fg > 0, emit_bg = do_bg && n_bg > 0;
    cf = cb = 0;
    if (emit_fg || emit_bg)
        for (int i = i0; i < i1; ++i) {
            const int c = class_of(i);
            if (c == 1) cf += (emit_fg && rs_key(seed, cls.img, i) <= t_fg);
            else if (c == 0) cb += (emit_bg && rs_key(seed ^ 0x5bd1e995ull, cls.img, i) <= t_bg);
        }
    int tot;
    int pf = rs_block_exclusive_scan(cf, s_wave, tot);
    int pb = n_fg + rs_block_exclusive_scan(cb, s_wave, tot);
    int *kp = keep + (size_t)s * rois_per_image;
    unsigned char *fp = is_fg + (size_t)s * rois_per_image;
    if (emit_fg || emit_bg)
        for (int i = i0; i < i1; ++i) {
            const int c = class_of(i);
            if (c == 1 && emit_fg && rs_key(seed, cls.img, i) <= t_fg) { kp[pf] = i; fp[pf] = 1; ++pf; }
            else if (c == 0 && emit_bg && rs_key(seed ^ 0x5bd1e995ull, cls.img, i) <= t_bg) { kp[pb] = i; fp[pb] = 0; ++pb; }
        }
    if (do_bg) {
        for (int p = n_fg + n_bg + t; p < rois_per_image; p += RS_BLOCK) { kp[p] = -1; fp[p] = 0; }
        if (t == 0) { counts[2 * s] = n_fg; counts[2 * s + 1] = n_bg; }
    }
}

}  // namespace wssdl

using namespace wssdl;

extern "C" int wssdl_roi_gt_assign(const float *rois, int R, const float *gt_boxes, int max_gt,
                                   const int32_t *num_pos_boxes, int n_images, double *max_overlap,
                                   int32_t *assignment, wssdl_stream_t stream) {
    if (R < 0 || max_gt < 1 || n_images < 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (R == 0) return WSSDL_OK;
    if (!rois || !gt_boxes || !num_pos_boxes || !max_overlap || !assignment)
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(roi_gt_assign_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream),
                       rois, R, gt_boxes, max_gt, num_pos_boxes, n_images, max_overlap, assignment);
    return check_launch();
}

extern "C" int wssdl_roi_targets(const float *rois, const int32_t *keep, const uint8_t *is_fg,
                                 int n_keep, const int32_t *assignment, const float *gt_boxes,
                                 int max_gt, int num_classes, const float *inside_weights_host,
                                 const double *normalize_host, float *rois_out, float *labels, float *bbox_targets,
                                 float *inside_w, float *outside_w, wssdl_stream_t stream) {
    if (n_keep < 0 || max_gt < 1 || num_classes < 1 || !inside_weights_host)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_keep == 0) return WSSDL_OK;
    if (!rois || !keep || !is_fg || !assignment || !gt_boxes || !rois_out || !labels ||
        !bbox_targets || !inside_w || !outside_w)
        return WSSDL_ERR_INVALID_ARGUMENT;
    RoiTargetNorm norm;
    norm.on = normalize_host ? 1 : 0;
    for (int j = 0; j < 4; ++j) {
        norm.mean[j] = normalize_host ? normalize_host[j] : 0.0;
        norm.std[j] = normalize_host ? normalize_host[4 + j] : 1.0;
    }
    hipLaunchKernelGGL(roi_targets_kernel, dim3(cdiv(n_keep, 256)), dim3(256), 0, as_stream(stream),
                       rois, keep, is_fg, n_keep, assignment, gt_boxes, max_gt, num_classes,
                       inside_weights_host[0], inside_weights_host[1], inside_weights_host[2],
                       inside_weights_host[3], norm, rois_out, labels, bbox_targets, inside_w, outside_w);
    return check_launch();
}

extern "C" int wssdl_roi_sample_device(const float *cand, const double *max_overlap, int Rc,
                                       const int32_t *images, int n_sample_images,
                                       int rois_per_image, int fg_rois_per_image, double fg_thresh,
                                       double bg_thresh_hi, double bg_thresh_lo, uint64_t seed,
                                       int32_t *keep, uint8_t *is_fg, int32_t *counts,
                                       wssdl_stream_t stream) {
    if (Rc < 0 || n_sample_images < 0 || rois_per_image < 1 || fg_rois_per_image < 0 ||
        fg_rois_per_image > rois_per_image)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_sample_images == 0) return WSSDL_OK;
    if (!images || !keep || !is_fg || !counts || (Rc > 0 && (!cand || !max_overlap)))
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(roi_sample_kernel, dim3(n_sample_images, 2), dim3(RS_BLOCK), 0, as_stream(stream),
                       cand, max_overlap, Rc, images, rois_per_image, fg_rois_per_image, fg_thresh,
                       bg_thresh_hi, bg_thresh_lo, (unsigned long long)seed, keep, is_fg, counts);
    return check_launch();
}

extern "C" int wssdl_roi_candidates(const float *rois, int R, const float *gt_boxes, int max_gt,
                                    const int32_t *num_gt_boxes, int n_images, const int32_t *images,
                                    int n_sample_images, int append_gt, float *cand,
                                    int32_t *num_pos_boxes, wssdl_stream_t stream) {
    if (R < 0 || max_gt < 1 || n_images < 1 || n_sample_images < 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!gt_boxes || !num_gt_boxes || !num_pos_boxes || !cand || (R > 0 && !rois) ||
        (n_sample_images > 0 && !images))
        return WSSDL_ERR_INVALID_ARGUMENT;
    const int total = R + (append_gt ? n_sample_images * max_gt : 0);
    const int threads = total > n_images ? total : n_images;
    hipLaunchKernelGGL(roi_candidates_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, as_stream(stream),
                       rois, R, gt_boxes, max_gt, num_gt_boxes, n_images, images, n_sample_images,
                       append_gt, cand, num_pos_boxes, static_cast<double *>(nullptr), static_cast<int *>(nullptr));
    return check_launch();
}

// ------------------------------------------------ the device-sampled layer as one call ---
// proposal_target_layer_tf_bus.py:15-97 / :99-160 for the supervised images with the device sampler:
// candidates (rois + every gt slot, :44-50) -> IoU arg-max assignment (:236-240) -> fg / bg draw
// (:243-261) -> rows, labels, targets, weights (:263-280, :187-226) as four launches behind ONE entry
// point, the intermediates carved from the caller's workspace (four separate calls from Python cost
// more host time than the kernels run).  The output has the fixed shape
// n_sample_images * rois_per_image rows; an image that runs short leaves rows (-1,0,0,0,0) with
// label -1 and zero weights.
namespace wssdl {
struct PtWs {
    float *cand;
    double *max_ov;
    int *assign, *num_pos, *keep, *counts;
    unsigned char *is_fg;
};
static size_t carve_pt(void *ws, int Rc, int n_images, int n_keep, int S, PtWs *out) {
    Carver c(ws);
    PtWs w;
    w.cand = c.take<float>((size_t)Rc * 5);
    w.max_ov = c.take<double>((size_t)Rc);
    w.assign = c.take<int>((size_t)Rc);
    w.num_pos = c.take<int>((size_t)n_images);
    w.keep = c.take<int>((size_t)n_keep);
    w.counts = c.take<int>((size_t)S * 2);
    w.is_fg = c.take<unsigned char>((size_t)n_keep);
    if (out) *out = w;
    return c.off;
}
}  // namespace wssdl

extern "C" size_t wssdl_proposal_target_device_workspace_bytes(int R, int n_images, int max_gt, int n_sample_images,
                                                               int rois_per_image, int append_gt) {
    if (R < 0 || n_images < 1 || max_gt < 1 || n_sample_images < 0 || rois_per_image < 1) return 0;
    const int Rc = R + (append_gt ? n_sample_images * max_gt : 0);
    return wssdl::carve_pt(nullptr, Rc, n_images, n_sample_images * rois_per_image, n_sample_images, nullptr) + 256;
}

extern "C" int wssdl_proposal_target_device(
    const float *rois, int R, const float *gt_boxes, int max_gt, const int32_t *num_gt_boxes, int n_images,
    const int32_t *images, int n_sample_images, int append_gt, int rois_per_image, int fg_rois_per_image,
    double fg_thresh, double bg_thresh_hi, double bg_thresh_lo, uint64_t seed, int num_classes,
    const float *inside_weights_host, const double *normalize_host, float *rois_out, float *labels, float *bbox_targets,
    float *inside_w, float *outside_w, void *workspace, size_t workspace_bytes, wssdl_stream_t stream) {
    if (R < 0 || n_images < 1 || max_gt < 1 || n_sample_images < 0 || rois_per_image < 1 || num_classes < 1)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (n_sample_images == 0) return WSSDL_OK;
    if (!workspace) return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < wssdl_proposal_target_device_workspace_bytes(R, n_images, max_gt, n_sample_images,
                                                                       rois_per_image, append_gt))
        return WSSDL_ERR_WORKSPACE;
    const int Rc = R + (append_gt ? n_sample_images * max_gt : 0);
    const int n_keep = n_sample_images * rois_per_image;
    wssdl::PtWs w;
    wssdl::carve_pt(workspace, Rc, n_images, n_keep, n_sample_images, &w);
    if (!gt_boxes || !num_gt_boxes || (R > 0 && !rois) || !images) return WSSDL_ERR_INVALID_ARGUMENT;
    // candidates + assignment in one launch (three launches for the layer: each costs more host time than it runs)
    hipLaunchKernelGGL(wssdl::roi_candidates_kernel, dim3(cdiv(Rc > n_images ? Rc : n_images, 256)), dim3(256), 0,
                       as_stream(stream), rois, R, gt_boxes, max_gt, num_gt_boxes, n_images, images, n_sample_images,
                       append_gt, w.cand, w.num_pos, w.max_ov, w.assign);
    int rc = check_launch();
    if (rc) return rc;
    if ((rc = wssdl_roi_sample_device(w.cand, w.max_ov, Rc, images, n_sample_images, rois_per_image, fg_rois_per_image,
                                      fg_thresh, bg_thresh_hi, bg_thresh_lo, seed, w.keep, w.is_fg, w.counts, stream)))
        return rc;
    return wssdl_roi_targets(w.cand, w.keep, w.is_fg, n_keep, w.assign, gt_boxes, max_gt, num_classes,
                             inside_weights_host, normalize_host, rois_out, labels, bbox_targets, inside_w, outside_w, stream);
}
