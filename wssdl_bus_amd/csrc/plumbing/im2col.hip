// 3x3 patch gather (im2col) and its adjoint on NHWC f32 tensors, for the 3x3 convolutions of
// the per-RoI head (networks/roi_head.py): the head runs its convolutions as GEMMs, and the
// stock route (F.pad -> unfold -> permute -> reshape, and unfold's backward) cost ~11 ms per
// step in copy kernels.  Plumbing library (libwssdl_plumbing_hip.so), not the drop-in C ABI.
//   cols[(r*OH + oy)*OW + ox][(ky*3 + kx)*C + c] = x[r][oy*S + ky - PT][ox*S + kx - PL][c]  (0 outside)
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PLUMB_API extern "C" __attribute__((visibility("default")))

namespace {

typedef float float4v __attribute__((ext_vector_type(4)));

// one thread = one float4 of cols; consecutive threads walk c, then the tap, then the pixel:
// fully coalesced writes, reads coalesced per tap
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float *__restrict__ x, int H, int W,
                                                       int C4, int OH, int OW, int S, int PT, int PL,
                                                       long long total4, float *__restrict__ cols) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4;
         i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int tap = (int)(t % 9);
        t /= 9;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const long long r = t / OH;
        const int y = oy * S + tap / 3 - PT, xx = ox * S + tap % 3 - PL;
        float4v v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && xx >= 0 && xx < W)
            v = reinterpret_cast<const float4v *>(x)[((r * H + y) * W + xx) * C4 + c4];
        reinterpret_cast<float4v *>(cols)[i] = v;
    }
}

// adjoint: one thread = one float4 of dx, sums the (at most 9) patch entries that copied it,
// in a fixed (ky, kx) order
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float *__restrict__ dcols, int H, int W,
                                                       int C4, int OH, int OW, int S, int PT, int PL,
                                                       long long total4, float *__restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4;
         i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int xx = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const long long r = t / H;
        float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ny = y + PT - ky;
            if (ny < 0 || ny % S != 0) continue;
            const int oy = ny / S;
            if (oy >= OH) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nx = xx + PL - kx;
                if (nx < 0 || nx % S != 0) continue;
                const int ox = nx / S;
                if (ox >= OW) continue;
                const float4v v = reinterpret_cast<const float4v *>(
                    dcols)[(((r * OH + oy) * OW + ox) * 9 + (ky * 3 + kx)) * C4 + c4];
                acc += v;
            }
        }
        reinterpret_cast<float4v *>(dx)[i] = acc;
    }
}

inline int grid_for(long long total4) {
    long long b = (total4 + 255) / 256;
    return (int)(b < 262144 ? b : 262144);
}

}  // namespace

PLUMB_API int wsplumb_im2col3x3(const float *x, long long R, int H, int W, int C, int OH, int OW,
                                int S, int PT, int PL, float *cols, void *stream) {
    if (R < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || OH < 1 || OW < 1 || S < 1) return 1;
    const long long total4 = R * OH * OW * 9 * (C / 4);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, H, W, C / 4, OH, OW, S, PT, PL, total4, cols);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

PLUMB_API int wsplumb_col2im3x3(const float *dcols, long long R, int H, int W, int C, int OH, int OW,
                                int S, int PT, int PL, float *dx, void *stream) {
    if (R < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || OH < 1 || OW < 1 || S < 1) return 1;
    const long long total4 = R * H * W * (C / 4);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(total4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dcols, H, W, C / 4, OH, OW, S, PT, PL, total4, dx);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
