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
template <int MODE, bool RELU>
__global__ __launch_bounds__(BLOCK) void rowbn_partial_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ scale,
    const float *__restrict__ shift, long long M, int C, long long rows_per_block,
    double *__restrict__ partial) {
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
                const float4v a0 = reinterpret_cast<const float4v *>(x + (size_t)r * C)[c4];
                const float4v a1 = reinterpret_cast<const float4v *>(x + (size_t)(r + RS) * C)[c4];
                float4v g0, g1;
                if (MODE == 1) {
                    g0 = reinterpret_cast<const float4v *>(dy + (size_t)r * C)[c4];
                    g1 = reinterpret_cast<const float4v *>(dy + (size_t)(r + RS) * C)[c4];
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
                        s[j] += (double)u0 + (double)u1;
                        q[j] += (double)u0 * (double)a0[j] + (double)u1 * (double)a1[j];
                    }
                }
            }
            for (; r < r1; r += RS) {
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

// forward finish: mean / biased var / scale / shift per column
__global__ __launch_bounds__(FIN_COLS * FIN_GROUPS) void rowbn_fwd_finish_kernel(
    const double *__restrict__ partial, int nblocks, int C, long long M,
    const float *__restrict__ weight, const float *__restrict__ bias, float eps,
    float *__restrict__ mean, float *__restrict__ var, float *__restrict__ rstd,
    float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ double red[FIN_GROUPS][FIN_COLS][2];
    const int grp = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    double s, q;
    finish_reduce(partial, nblocks, C, c, grp, red, s, q);
    if (grp != 0 || c >= C) return;
    const double mu = s / (double)M;
    double v = q / (double)M - mu * mu;
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
    float *__restrict__ coef) {
    __shared__ double red[FIN_GROUPS][FIN_COLS][2];
    const int grp = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    double sg, sgx;
    finish_reduce(partial, nblocks, C, c, grp, red, sg, sgx);
    if (grp != 0 || c >= C) return;
    const double mu = mean[c], rs = rstd[c], w = weight[c];
    const double sum_g_xhat = (sgx - mu * sg) * rs;
    const double a = w * rs;
    const double k1 = a * rs * sum_g_xhat / (double)M;
    const double k0 = a * sg / (double)M - k1 * mu;
    dweight[c] = (float)sum_g_xhat;
    dbias[c] = (float)sg;
    coef[c] = (float)a;
    coef[C + c] = (float)k0;
    coef[2 * C + c] = (float)k1;
}

template <bool RELU>
__global__ __launch_bounds__(BLOCK) void rowbn_apply_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
    long long total4, int C4, float *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total4;
         i += (long long)gridDim.x * BLOCK) {
        const int c4 = (int)(i % C4);
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

template <bool RELU>
__global__ __launch_bounds__(BLOCK) void rowbn_apply_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ coef, long long total4, int C4,
    float *__restrict__ dx) {
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total4;
         i += (long long)gridDim.x * BLOCK) {
        const int c4 = (int)(i % C4);
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

// y = act(batch_norm(x)); writes mean, var (biased), rstd, scale = rstd*weight,
// shift = bias - mean*scale (all [C]).  Returns 0 on success.
PLUMB_API int wsplumb_rowbn_forward(const float *x, long long M, int C, const float *weight,
                                    const float *bias, float eps, int relu, float *y, float *mean,
                                    float *var, float *rstd, float *scale, float *shift,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!shape_ok(M, C) || workspace_bytes < wsplumb_rowbn_workspace_bytes(M, C)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = partial_blocks(M, C);
    const long long rpb = (M + nb - 1) / nb;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL((rowbn_partial_kernel<0, false>), dim3(nb), dim3(BLOCK), 0, st, x, nullptr,
                       nullptr, nullptr, M, C, rpb, partial);
    hipLaunchKernelGGL(rowbn_fwd_finish_kernel, dim3((C + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS * FIN_GROUPS), 0, st, partial, nb, C,
                       M, weight, bias, eps, mean, var, rstd, scale, shift);
    const long long total4 = M * (C / 4);
    if (relu)
        hipLaunchKernelGGL(rowbn_apply_fwd_kernel<true>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y);
    else
        hipLaunchKernelGGL(rowbn_apply_fwd_kernel<false>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// y = act(x*scale + shift) with given per-column scale / shift (inference statistics)
PLUMB_API int wsplumb_rowbn_apply(const float *x, long long M, int C, const float *scale,
                                  const float *shift, int relu, float *y, void *stream) {
    if (M < 1 || C < 4 || (C & 3)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total4 = M * (C / 4);
    if (relu)
        hipLaunchKernelGGL(rowbn_apply_fwd_kernel<true>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y);
    else
        hipLaunchKernelGGL(rowbn_apply_fwd_kernel<false>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x,
                           scale, shift, total4, C / 4, y);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// gradients of wsplumb_rowbn_forward: dx [M,C], dweight [C], dbias [C]; coef is [3*C] scratch
PLUMB_API int wsplumb_rowbn_backward(const float *x, const float *dy, long long M, int C,
                                     const float *weight, const float *mean, const float *rstd,
                                     const float *scale, const float *shift, int relu, float *dx,
                                     float *dweight, float *dbias, float *coef, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!shape_ok(M, C) || workspace_bytes < wsplumb_rowbn_workspace_bytes(M, C)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = partial_blocks(M, C);
    const long long rpb = (M + nb - 1) / nb;
    double *partial = static_cast<double *>(workspace);
    if (relu)
        hipLaunchKernelGGL((rowbn_partial_kernel<1, true>), dim3(nb), dim3(BLOCK), 0, st, x, dy, scale,
                           shift, M, C, rpb, partial);
    else
        hipLaunchKernelGGL((rowbn_partial_kernel<1, false>), dim3(nb), dim3(BLOCK), 0, st, x, dy, scale,
                           shift, M, C, rpb, partial);
    hipLaunchKernelGGL(rowbn_bwd_finish_kernel, dim3((C + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS * FIN_GROUPS), 0, st, partial, nb, C,
                       M, weight, mean, rstd, dweight, dbias, coef);
    const long long total4 = M * (C / 4);
    if (relu)
        hipLaunchKernelGGL(rowbn_apply_bwd_kernel<true>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x, dy,
                           scale, shift, coef, total4, C / 4, dx);
    else
        hipLaunchKernelGGL(rowbn_apply_bwd_kernel<false>, dim3(apply_grid(total4)), dim3(BLOCK), 0, st, x, dy,
                           scale, shift, coef, total4, C / 4, dx);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
