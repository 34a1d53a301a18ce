// Training-mode batch norm over the rows of an [M, C] f32 matrix, fused with ReLU, for the
// per-RoI ResNet head (networks/roi_head.py).  NOT part of the drop-in C ABI of the detection
// hot path (include/wssdl_bus_hip.h): this is plumbing around it, in its own library
// (libwssdl_plumbing_hip.so).  The head's activations are [R*h*w, C] with R*h*w up to ~4e5
// rows: with stock elementwise ops a BN+ReLU layer costs ~19 passes over the tensor per
// training step (forward 5, backward 14); here
//   forward : column sums (1 read)            -> y = relu(x*scale + shift)        (1 read, 1 write)
//   backward: column sums of g and g*x, g = dy masked by the recomputed activation (2 reads)
//                                              -> dx = a*g - k0 - k1*x             (2 reads, 1 write)
// Column sums are accumulated in f64 per thread, reduced in a fixed order (partials per
// workgroup, then one thread per column): results do not depend on scheduling.
//
// LIVE-ROW MASK (the *_masked entry points).  The rows come in groups of `per` (the h*w positions of
// one RoI); `mask[roi]` = 0 marks a dead RoI -- a padding row of the fixed-shape RoI blob, or of a
// supervised image that ran short of candidates.  Dead rows are skipped by the column sums (not even
// loaded), the statistics are taken over the live rows (n = per * sum(mask), computed on the device:
// no host read-back), and the layer writes zeros for dead rows in both directions, so the live rows
// come out exactly as if the blob had been compacted first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PLUMB_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_PARTIAL_BLOCKS = 1024;

typedef float float4v __attribute__((ext_vector_type(4)));

// Partial column sums of one row slab.  Thread t owns float4 column (t % L) + cc * L and the
// rows r0 + t / L + k * RS; L = min(C/4, 256), RS = 256 / L.
// MODE 0: s = sum x,  q = sum x*x
// MODE 1: s = sum g,  q = sum g*x   with g = dy, masked by (x*scale + shift > 0) when RELU
template <int MODE, bool RELU, bool MASKED>
__global__ __launch_bounds__(BLOCK) void rowbn_partial_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ scale,
    const float *__restrict__ shift, long long M, int C, long long rows_per_block,
    double *__restrict__ partial, const float *__restrict__ mask, int per) {
    __shared__ double red[BLOCK][8];
    const int C4 = C >> 2;
    const int L = C4 < BLOCK ? C4 : BLOCK;
    const int RS = BLOCK / L;
    const int t = threadIdx.x;
    const int lc = t % L, lr = t / L;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    double *out = partial + (size_t)blockIdx.x * 2 * C;
    for (int cc = 0; cc * L < C4; ++cc) {
        const int c4 = cc * L + lc;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        float4v sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
        if (MODE == 1 && RELU) {
            sc = reinterpret_cast<const float4v *>(scale)[c4];
            sh = reinterpret_cast<const float4v *>(shift)[c4];
        }
        if (lr < RS) {
            long long r = r0 + lr;
            // two rows in flight per step
            for (; r + RS < r1; r += 2 * RS) {
                bool live0 = true, live1 = true;
                if (MASKED) {
                    live0 = mask[(unsigned)r / (unsigned)per] != 0.0f;
                    live1 = mask[(unsigned)(r + RS) / (unsigned)per] != 0.0f;
                    if (!live0 && !live1) continue;
                }
                const float4v zero4 = {0, 0, 0, 0};
                const float4v a0 = live0 ? reinterpret_cast<const float4v *>(x + (size_t)r * C)[c4] : zero4;
                const float4v a1 = live1 ? reinterpret_cast<const float4v *>(x + (size_t)(r + RS) * C)[c4] : zero4;
                float4v g0, g1;
                if (MODE == 1) {
                    g0 = live0 ? reinterpret_cast<const float4v *>(dy + (size_t)r * C)[c4] : zero4;
                    g1 = live1 ? reinterpret_cast<const float4v *>(dy + (size_t)(r + RS) * C)[c4] : zero4;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) {
                        s[j] += (double)a0[j] + (double)a1[j];
                        q[j] += (double)a0[j] * (double)a0[j] + (double)a1[j] * (double)a1[j];
                    } else {
                        float u0 = g0[j], u1 = g1[j];
                        if (RELU) {
                            if (!(a0[j] * sc[j] + sh[j] > 0.0f)) u0 = 0.0f;
                            if (!(a1[j] * sc[j] + sh[j] > 0.0f)) u1 = 0.0f;
                        }
                        if (MASKED) {               // a dead row adds nothing (x = 0 would still pass the ReLU test)
                            if (!live0) u0 = 0.0f;
                            if (!live1) u1 = 0.0f;
                        }
                        s[j] += (double)u0 + (double)u1;
                        q[j] += (double)u0 * (double)a0[j] + (double)u1 * (double)a1[j];
                    }
                }
            }
            for (; r < r1; r += RS) {
                if (MASKED && mask[(unsigned)r / (unsigned)per] == 0.0f) continue;
                const float4v a0 = reinterpret_cast<const float4v *>(x + (size_t)r * C)[c4];
                float4v g0;
                if (MODE == 1) g0 = reinterpret_cast<const float4v *>(dy + (size_t)r * C)[c4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) {
                        s[j] += (double)a0[j];
                        q[j] += (double)a0[j] * (double)a0[j];
                    } else {
                        float u0 = g0[j];
                        if (RELU && !(a0[j] * sc[j] + sh[j] > 0.0f)) u0 = 0.0f;
                        s[j] += (double)u0;
                        q[j] += (double)u0 * (double)a0[j];
                    }
                }
            }
        }
        // fixed-order reduction over the RS row phases of a column
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[t][j] = s[j]; red[t][4 + j] = q[j]; }
        __syncthreads();
        if (lr == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double ss = 0.0, qq = 0.0;
                for (int k = 0; k < RS; ++k) { ss += red[k * L + lc][j]; qq += red[k * L + lc][4 + j]; }
                out[c4 * 4 + j] = ss;
                out[C + c4 * 4 + j] = qq;
            }
        }
    }
}

// Sum of the per-workgroup partials of 16 columns, in a fixed order: 64 row groups of 16 lanes
// each add every 64th partial (in order), then lane-wise the 64 group sums are added in order.
constexpr int FIN_COLS = 16, FIN_GROUPS = 64;

__device__ __forceinline__ void finish_reduce(const double *__restrict__ partial, int nblocks, int C,
                                              int c, int grp, double (*red)[FIN_COLS][2], double &s,
                                              double &q) {
    double ss = 0.0, qq = 0.0;
    if (c < C)
        for (int b = grp; b < nblocks; b += FIN_GROUPS) {
            ss += partial[(size_t)b * 2 * C + c];
            qq += partial[(size_t)b * 2 * C + C + c];
        }
    red[grp][threadIdx.x % FIN_COLS][0] = ss;
    red[grp][threadIdx.x % FIN_COLS][1] = qq;
    __syncthreads();
    s = 0.0;
    q = 0.0;
    if (grp == 0)
        for (int g = 0; g < FIN_GROUPS; ++g) {
            s += red[g][threadIdx.x][0];
            q += red[g][threadIdx.x][1];
        }
}

// rows the statistics are taken over: M, or per * (number of live RoIs) with a mask (at least 1).
// Every thread of the workgroup returns the same value; the sum runs in a fixed order.
__device__ __forceinline__ double live_rows(const float *__restrict__ mask, int n_rois, int per, long long M,
                                            double *scratch /* [FIN_COLS * FIN_GROUPS] LDS */) {
    if (!mask) return (double)M;
    double c = 0.0;
    for (int i = threadIdx.x; i < n_rois; i += FIN_COLS * FIN_GROUPS) c += mask[i] != 0.0f ? 1.0 : 0.0;
    __syncthreads();
    scratch[threadIdx.x] = c;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < FIN_COLS * FIN_GROUPS; ++i) t += scratch[i];
    __syncthreads();
    t *= (double)per;
    return t < 1.0 ? 1.0 : t;
}

// forward finish: mean / biased var / scale / shift per column
__global__ __launch_bounds__(FIN_COLS * FIN_GROUPS) void rowbn_fwd_finish_kernel(
    const double *__restrict__ partial, int nblocks, int C, long long M,
    const float *__restrict__ weight, const float *__restrict__ bias, float eps,
    float *__restrict__ mean, float *__restrict__ var, float *__restrict__ rstd,
    float *__restrict__ scale, float *__restrict__ shift, const float *__restrict__ mask, int n_rois, int per,
    float *__restrict__ count) {
    __shared__ double red[FIN_GROUPS][FIN_COLS][2];
    const int grp = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    const double Mn = live_rows(mask, n_rois, per, M, &red[0][0][0]);
    if (count && blockIdx.x == 0 && threadIdx.x == 0) count[0] = (float)Mn;
    double s, q;
    finish_reduce(partial, nblocks, C, c, grp, red, s, q);
    if (grp != 0 || c >= C) return;
    const double mu = s / Mn;
    double v = q / Mn - mu * mu;
    if (v < 0.0) v = 0.0;
    const float rs = (float)(1.0 / sqrt(v + (double)eps));
    const float scl = rs * weight[c];
    mean[c] = (float)mu;
    var[c] = (float)v;
    rstd[c] = rs;
    scale[c] = scl;
    shift[c] = bias[c] - (float)mu * scl;
}

// backward finish: dweight, dbias and the three coefficients of dx = a*g - k0 - k1*x
__global__ __launch_bounds__(FIN_COLS * FIN_GROUPS) void rowbn_bwd_finish_kernel(
    const double *__restrict__ partial, int nblocks, int C, long long M,
    const float *__restrict__ weight, const float *__restrict__ mean,
    const float *__restrict__ rstd, float *__restrict__ dweight, float *__restrict__ dbias,
    float *__restrict__ coef, const float *__restrict__ mask, int n_rois, int per) {
    __shared__ double red[FIN_GROUPS][FIN_COLS][2];
    const int grp = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    const double Mn = live_rows(mask, n_rois, per, M, &red[0][0][0]);
    double sg, sgx;
    finish_reduce(partial, nblocks, C, c, grp, red, sg, sgx);
    if (grp != 0 || c >= C) return;
    const double mu = mean[c], rs = rstd[c], w = weight[c];
    const double sum_g_xhat = (sgx - mu * sg) * rs;
    const double a = w * rs;
    const double k1 = a * rs * sum_g_xhat / Mn;
    const double k0 = a * sg / Mn - k1 * mu;
    dweight[c] = (float)sum_g_xhat;
    dbias[c] = (float)sg;
    coef[c] = (float)a;
    coef[C + c] = (float)k0;
    coef[2 * C + c] = (float)k1;
}

template <bool RELU, bool MASKED>
__global__ __launch_bounds__(BLOCK) void rowbn_apply_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
    long long total4, int C4, float *__restrict__ y, const float *__restrict__ mask, int per) {
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total4;
         i += (long long)gridDim.x * BLOCK) {
        const int c4 = (int)(i % C4);
        if (MASKED && mask[(unsigned)(i / C4) / (unsigned)per] == 0.0f) {
            const float4v zero4 = {0, 0, 0, 0};
            reinterpret_cast<float4v *>(y)[i] = zero4;
            continue;
        }
        const float4v a = reinterpret_cast<const float4v *>(x)[i];
        const float4v sc = reinterpret_cast<const float4v *>(scale)[c4];
        const float4v sh = reinterpret_cast<const float4v *>(shift)[c4];
        float4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = a[j] * sc[j] + sh[j];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            o[j] = v;
        }
        reinterpret_cast<float4v *>(y)[i] = o;
    }
}

template <bool RELU, bool MASKED>
__global__ __launch_bounds__(BLOCK) void rowbn_apply_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ coef, long long total4, int C4,
    float *__restrict__ dx, const float *__restrict__ mask, int per) {
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total4;
         i += (long long)gridDim.x * BLOCK) {
        const int c4 = (int)(i % C4);
        if (MASKED && mask[(unsigned)(i / C4) / (unsigned)per] == 0.0f) {
            const float4v zero4 = {0, 0, 0, 0};
            reinterpret_cast<float4v *>(dx)[i] = zero4;
            continue;
        }
        const float4v a = reinterpret_cast<const float4v *>(x)[i];
        const float4v g = reinterpret_cast<const float4v *>(dy)[i];
        const float4v ka = reinterpret_cast<const float4v *>(coef)[c4];
        const float4v k0 = reinterpret_cast<const float4v *>(coef + C)[c4];
        const float4v k1 = reinterpret_cast<const float4v *>(coef + 2 * C)[c4];
        float4v sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
        if (RELU) {
            sc = reinterpret_cast<const float4v *>(scale)[c4];
            sh = reinterpret_cast<const float4v *>(shift)[c4];
        }
        float4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = g[j];
            if (RELU && !(a[j] * sc[j] + sh[j] > 0.0f)) u = 0.0f;
            o[j] = ka[j] * u - k0[j] - k1[j] * a[j];
        }
        reinterpret_cast<float4v *>(dx)[i] = o;
    }
}

inline int partial_blocks(long long M, int C) {
    const int C4 = C / 4;
    const int L = C4 < BLOCK ? C4 : BLOCK;
    const int RS = BLOCK / L;
    long long want = (M + (long long)RS * 16 - 1) / ((long long)RS * 16);   // >= 16 row steps each
    if (want < 1) want = 1;
    return (int)(want < MAX_PARTIAL_BLOCKS ? want : MAX_PARTIAL_BLOCKS);
}

inline bool shape_ok(long long M, int C) {
    if (M < 1 || C < 4 || (C & 3)) return false;
    const int C4 = C / 4;
    return C4 <= BLOCK ? (BLOCK % C4 == 0) : (C4 % BLOCK == 0);
}

inline int apply_grid(long long total4) {
    long long b = (total4 + BLOCK - 1) / BLOCK;
    return (int)(b < 65536 ? b : 65536);
}

}  // namespace

// bytes of scratch for the partial sums (f64) of one call
PLUMB_API size_t wsplumb_rowbn_workspace_bytes(long long M, int C) {
    if (!shape_ok(M, C)) return 0;
    return (size_t)partial_blocks(M, C) * 2 * (size_t)C * sizeof(double);
}

// 1 when the kernels support the shape (C % 4 == 0 and C/4 divides or is a multiple of 256)
PLUMB_API int wsplumb_rowbn_supported(long long M, int C) { return shape_ok(M, C) ? 1 : 0; }

static int forward_impl(const float *x, long long M, int C, const float *weight, const float *bias, float eps,
                        int relu, float *y, float *mean, float *var, float *rstd, float *scale, float *shift,
                        const float *mask, int n_rois, int per, float *count, void *workspace,
                        size_t workspace_bytes, void *stream) {
    if (!shape_ok(M, C) || workspace_bytes < wsplumb_rowbn_workspace_bytes(M, C)) return 1;
    if (mask && (per < 1 || n_rois < 1 || (long long)n_rois * per != M || M > 0x7fffffffLL)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = partial_blocks(M, C);
    const long long rpb = (M + nb - 1) / nb;
    double *partial = static_cast<double *>(workspace);
    if (mask)
        hipLaunchKernelGGL((rowbn_partial_kernel<0, false, true>), dim3(nb), dim3(BLOCK), 0, st, x, nullptr,
                           nullptr, nullptr, M, C, rpb, partial, mask, per);
    else
        hipLaunchKernelGGL((rowbn_partial_kernel<0, false, false>), dim3(nb), dim3(BLOCK), 0, st, x, nullptr,
                           nullptr, nullptr, M, C, rpb, partial, nullptr, 1);
    hipLaunchKernelGGL(rowbn_fwd_finish_kernel, dim3((C + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS * FIN_GROUPS), 0, st, partial, nb, C,
                       M, weight, bias, eps, mean, var, rstd, scale, shift, mask, n_rois, per, count);
    const long long total4 = M * (C / 4);
#define WSPLUMB_APPLY(RELU, MASKED) \
    hipLaunchKernelGGL((rowbn_apply_fwd_kernel<RELU, MASKED>), dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x, scale, \
                       shift, total4, C / 4, y, mask, per)
    if (relu) { if (mask) WSPLUMB_APPLY(true, true); else WSPLUMB_APPLY(true, false); }
    else { if (mask) WSPLUMB_APPLY(false, true); else WSPLUMB_APPLY(false, false); }
#undef WSPLUMB_APPLY
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// y = act(batch_norm(x)); writes mean, var (biased), rstd, scale = rstd*weight,
// shift = bias - mean*scale (all [C]).  Returns 0 on success.
PLUMB_API int wsplumb_rowbn_forward(const float *x, long long M, int C, const float *weight,
                                    const float *bias, float eps, int relu, float *y, float *mean,
                                    float *var, float *rstd, float *scale, float *shift,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    return forward_impl(x, M, C, weight, bias, eps, relu, y, mean, var, rstd, scale, shift, nullptr, 0, 1, nullptr,
                        workspace, workspace_bytes, stream);
}

// the same over the live rows only: mask [n_rois] f32 (0 = dead), rows r*per .. r*per+per-1 belong to
// RoI r (M = n_rois * per); dead rows of y are written as zeros; count[0] receives the number of live
// rows (>= 1) as a float, for the caller's running-variance correction
PLUMB_API int wsplumb_rowbn_forward_masked(const float *x, long long M, int C, const float *weight,
                                           const float *bias, float eps, int relu, const float *mask,
                                           int n_rois, int per, float *y, float *mean, float *var,
                                           float *rstd, float *scale, float *shift, float *count,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    if (!mask || !count) return 1;
    return forward_impl(x, M, C, weight, bias, eps, relu, y, mean, var, rstd, scale, shift, mask, n_rois, per, count,
                        workspace, workspace_bytes, stream);
}

// y = act(x*scale + shift) with given per-column scale / shift (inference statistics)
PLUMB_API int wsplumb_rowbn_apply(const float *x, long long M, int C, const float *scale,
                                  const float *shift, int relu, float *y, void *stream) {
    if (M < 1 || C < 4 || (C & 3)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total4 = M * (C / 4);
    if (relu)
        hipLaunchKernelGGL((rowbn_apply_fwd_kernel<true, false>), dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y, nullptr, 1);
    else
        hipLaunchKernelGGL((rowbn_apply_fwd_kernel<false, false>), dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y, nullptr, 1);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

static int backward_impl(const float *x, const float *dy, long long M, int C, const float *weight,
                         const float *mean, const float *rstd, const float *scale, const float *shift, int relu,
                         float *dx, float *dweight, float *dbias, float *coef, const float *mask, int n_rois,
                         int per, void *workspace, size_t workspace_bytes, void *stream) {
    if (!shape_ok(M, C) || workspace_bytes < wsplumb_rowbn_workspace_bytes(M, C)) return 1;
    if (mask && (per < 1 || n_rois < 1 || (long long)n_rois * per != M || M > 0x7fffffffLL)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = partial_blocks(M, C);
    const long long rpb = (M + nb - 1) / nb;
    double *partial = static_cast<double *>(workspace);
#define WSPLUMB_PARTIAL(RELU, MASKED) \
    hipLaunchKernelGGL((rowbn_partial_kernel<1, RELU, MASKED>), dim3(nb), dim3(BLOCK), 0, st, x, dy, scale, shift, M, C, \
                       rpb, partial, mask, per)
    if (relu) { if (mask) WSPLUMB_PARTIAL(true, true); else WSPLUMB_PARTIAL(true, false); }
    else { if (mask) WSPLUMB_PARTIAL(false, true); else WSPLUMB_PARTIAL(false, false); }
#undef WSPLUMB_PARTIAL
    hipLaunchKernelGGL(rowbn_bwd_finish_kernel, dim3((C + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS * FIN_GROUPS), 0, st, partial, nb, C,
                       M, weight, mean, rstd, dweight, dbias, coef, mask, n_rois, per);
    const long long total4 = M * (C / 4);
#define WSPLUMB_APPLY(RELU, MASKED) \
    hipLaunchKernelGGL((rowbn_apply_bwd_kernel<RELU, MASKED>), dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x, dy, scale, \
                       shift, coef, total4, C / 4, dx, mask, per)
    if (relu) { if (mask) WSPLUMB_APPLY(true, true); else WSPLUMB_APPLY(true, false); }
    else { if (mask) WSPLUMB_APPLY(false, true); else WSPLUMB_APPLY(false, false); }
#undef WSPLUMB_APPLY
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// gradients of wsplumb_rowbn_forward: dx [M,C], dweight [C], dbias [C]; coef is [3*C] scratch
PLUMB_API int wsplumb_rowbn_backward(const float *x, const float *dy, long long M, int C,
                                     const float *weight, const float *mean, const float *rstd,
                                     const float *scale, const float *shift, int relu, float *dx,
                                     float *dweight, float *dbias, float *coef, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    return backward_impl(x, dy, M, C, weight, mean, rstd, scale, shift, relu, dx, dweight, dbias, coef, nullptr, 0, 1,
                         workspace, workspace_bytes, stream);
}

// gradients of wsplumb_rowbn_forward_masked (dead rows: dy ignored, dx = 0)
PLUMB_API int wsplumb_rowbn_backward_masked(const float *x, const float *dy, long long M, int C,
                                            const float *weight, const float *mean, const float *rstd,
                                            const float *scale, const float *shift, int relu,
                                            const float *mask, int n_rois, int per, float *dx,
                                            float *dweight, float *dbias, float *coef, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    if (!mask) return 1;
    return backward_impl(x, dy, M, C, weight, mean, rstd, scale, shift, relu, dx, dweight, dbias, coef, mask, n_rois, per,
                         workspace, workspace_bytes, stream);
}
