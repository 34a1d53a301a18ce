// RoI max-pooling forward / backward for gfx950 (MI355X).
//
// Reference: code/lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:20-85 (forward,
// canonical rounding), roi_pooling_op.cc:137-196 (forward, CPU rounding),
// roi_pooling_op_gpu.cu.cc:114-190 == roi_pooling_op.cc:383-458 (backward).
//
// Forward  : HBM-write bound.  One lane owns 4 consecutive channels of one output
//            bin: 16-byte coalesced loads of the NHWC feature map (which stays in
//            L2 / Infinity Cache: 38x63x1024 f32 = 9.8 MB per image) and 16-byte
//            non-temporal stores of top + argmax (written once, never re-read
//            by this kernel).
// Backward : the reference gathers per bottom element over ALL RoIs
//            (O(N*H*W*C*R)).  Here each workgroup owns an 8x8-cell x 256-channel
//            tile of bottom_diff in LDS (64 KiB), one lane per channel.  It
//            filters the RoI list down to the RoIs of its image that touch the
//            tile (order-preserving ballot compaction), then walks them in RoI
//            order, bins in (ph, pw) order, adding top_diff into the LDS cell
//            its argmax names.  A lane is the only writer of its channel, so
//            per element the f32 additions happen in exactly the reference's
//            order (roi^, ph^, pw^): the result is bit-identical and needs no
//            atomics and no pre-zeroing; every bottom_diff element is written
//            exactly once with a coalesced store.
#include "common.hip.h"

#include <float.h>

namespace wssdl {

struct RoiGeom {
    int batch, sw, sh, ew, eh;
    float bin_h, bin_w;
};

// roi_pooling_op_gpu.cu.cc:36-49 == roi_pooling_op.cc:152-165
__device__ __forceinline__ RoiGeom roi_geometry(const float *__restrict__ r, float scale, int PH,
                                                int PW) {
    RoiGeom g;
    g.batch = (int)r[0];
    g.sw = (int)roundf(r[1] * scale);
    g.sh = (int)roundf(r[2] * scale);
    g.ew = (int)roundf(r[3] * scale);
    g.eh = (int)roundf(r[4] * scale);
    int rw = max(g.ew - g.sw + 1, 1);
    int rh = max(g.eh - g.sh + 1, 1);
    g.bin_h = (float)rh / (float)PH;
    g.bin_w = (float)rw / (float)PW;
    return g;
}

__device__ __forceinline__ void bin_window(const RoiGeom &g, int ph, int pw, int H, int W,
                                           int rounding, int &hs, int &he, int &ws, int &we) {
    if (rounding == WSSDL_ROI_ROUND_CPU) {          // roi_pooling_op.cc:167-170
        hs = (int)((float)ph * g.bin_h);
        ws = (int)((float)pw * g.bin_w);
        he = (int)((float)(ph + 1) * g.bin_h);
        we = (int)((float)(pw + 1) * g.bin_w);
    } else {                                        // roi_pooling_op_gpu.cu.cc:51-58
        hs = (int)floorf((float)ph * g.bin_h);
        ws = (int)floorf((float)pw * g.bin_w);
        he = (int)ceilf((float)(ph + 1) * g.bin_h);
        we = (int)ceilf((float)(pw + 1) * g.bin_w);
    }
    hs = min(max(hs + g.sh, 0), H);
    he = min(max(he + g.sh, 0), H);
    ws = min(max(ws + g.sw, 0), W);
    we = min(max(we + g.sw, 0), W);
}

typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ forward ---
// VEC = 4: one lane = 4 channels (C % 4 == 0);  VEC = 1: scalar fallback.
template <int VEC>
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(
    const float *__restrict__ bottom, int N, int H, int W, int C, const float *__restrict__ rois,
    int R, int PH, int PW, float scale, int rounding, float *__restrict__ top,
    int *__restrict__ argmax) {
    const int CV = C / VEC;
    const long long total = (long long)R * PH * PW * CV;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int cv = (int)(idx % CV);
        long long bin = idx / CV;
        int pw = (int)(bin % PW);
        int ph = (int)((bin / PW) % PH);
        int r = (int)(bin / ((long long)PW * PH));
        RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
        int hs, he, ws, we;
        bin_window(g, ph, pw, H, W, rounding, hs, he, ws, we);
        bool empty = (he <= hs) || (we <= ws) || g.batch < 0 || g.batch >= N;
        const int c0 = cv * VEC;
        if (VEC == 4) {
            float4v mv = empty ? (float4v)(0.0f) : (float4v)(-FLT_MAX);
            int4v mi = (int4v)(-1);
            if (!empty) {
                const float *img = bottom + (size_t)g.batch * H * W * C;
                for (int h = hs; h < he; ++h) {
                    for (int w = ws; w < we; ++w) {
                        int base = (h * W + w) * C + c0;
                        float4v v = *reinterpret_cast<const float4v *>(img + base);
                        if (v.x > mv.x) { mv.x = v.x; mi.x = base; }
                        if (v.y > mv.y) { mv.y = v.y; mi.y = base + 1; }
                        if (v.z > mv.z) { mv.z = v.z; mi.z = base + 2; }
                        if (v.w > mv.w) { mv.w = v.w; mi.w = base + 3; }
                    }
                }
            }
            size_t o = (size_t)bin * C + c0;
            __builtin_nontemporal_store(mv, reinterpret_cast<float4v *>(top + o));
            __builtin_nontemporal_store(mi, reinterpret_cast<int4v *>(argmax + o));
        } else {
            float mv = empty ? 0.0f : -FLT_MAX;
            int mi = -1;
            if (!empty) {
                const float *img = bottom + (size_t)g.batch * H * W * C;
                for (int h = hs; h < he; ++h)
                    for (int w = ws; w < we; ++w) {
                        int base = (h * W + w) * C + c0;
                        float v = img[base];
                        if (v > mv) { mv = v; mi = base; }
                    }
            }
            size_t o = (size_t)bin * C + c0;
            top[o] = mv;
            argmax[o] = mi;
        }
    }
}

// ----------------------------------------------------------------- backward ---
template <int TH, int TW, int CG>
__global__ __launch_bounds__(CG) void roi_pool_bwd_kernel(
    const float *__restrict__ top_diff, const int *__restrict__ argmax,
    const float *__restrict__ rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
    float *__restrict__ bottom_diff, int tiles_h, int tiles_w, int cgroups) {
    constexpr int KPT = 4;                 // RoIs tested per thread per filter round
    constexpr int CHUNK = KPT * CG;
    constexpr int NW = CG / WSSDL_WAVE;
    __shared__ float acc[TH * TW * CG];
    __shared__ int list[CHUNK];
    __shared__ int wave_cnt[KPT][NW];

    int b = blockIdx.x;
    const int cg = b % cgroups;  b /= cgroups;
    const int tx = b % tiles_w;  b /= tiles_w;
    const int ty = b % tiles_h;
    const int n = b / tiles_h;
    const int tc = threadIdx.x;
    const int c = cg * CG + tc;
    const bool c_ok = c < C;
    const int h0 = ty * TH, w0 = tx * TW;
    const int h1 = min(h0 + TH, H) - 1, w1 = min(w0 + TW, W) - 1;   // inclusive
    const int lane = tc & (WSSDL_WAVE - 1), wave = tc / WSSDL_WAVE;

#pragma unroll
    for (int i = 0; i < TH * TW; ++i) acc[i * CG + tc] = 0.0f;

    for (int base = 0; base < R; base += CHUNK) {
        // ---- filter: RoIs of image n whose rounded box touches the tile, in RoI order
        bool hit[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            int r = base + k * CG + tc;
            hit[k] = false;
            if (r < R) {
                RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
                hit[k] = (g.batch == n) && g.sw <= w1 && g.ew >= w0 && g.sh <= h1 && g.eh >= h0;
            }
            unsigned long long m = __ballot(hit[k]);
            if (lane == 0) wave_cnt[k][wave] = __popcll(m);
        }
        __syncthreads();
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            unsigned long long m = __ballot(hit[k]);
            int before = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                int wc = wave_cnt[k][w];
                before += (w < wave) ? wc : 0;
            }
            if (hit[k]) {
                int pos = cnt + before + __popcll(m & ((1ull << lane) - 1ull));
                list[pos] = base + k * CG + tc;
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) cnt += wave_cnt[k][w];
        }
        __syncthreads();

        // ---- walk the touching RoIs in order; every lane = one channel
        for (int i = 0; i < cnt; ++i) {
            const int r = list[i];
            RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
            const int hlo = max(h0, g.sh), hhi = min(h1, g.eh);
            const int wlo = max(w0, g.sw), whi = min(w1, g.ew);
            // candidate bins of the tile's cells: phstart/phend are monotone in h
            // (roi_pooling_op_gpu.cu.cc:169-177), so the union over the tile rows
            // is [phstart(hlo), phend(hhi)).
            int ph0 = (int)floorf((float)(hlo - g.sh) / g.bin_h);
            int ph1 = (int)ceilf((float)(hhi - g.sh + 1) / g.bin_h);
            int pw0 = (int)floorf((float)(wlo - g.sw) / g.bin_w);
            int pw1 = (int)ceilf((float)(whi - g.sw + 1) / g.bin_w);
            ph0 = min(max(ph0, 0), PH);  ph1 = min(max(ph1, 0), PH);
            pw0 = min(max(pw0, 0), PW);  pw1 = min(max(pw1, 0), PW);
            for (int ph = ph0; ph < ph1; ++ph) {
                for (int pw = pw0; pw < pw1; ++pw) {
                    if (!c_ok) continue;
                    size_t o = (((size_t)r * PH + ph) * PW + pw) * C + c;
                    int idx = argmax[o];
                    if (idx < 0) continue;
                    int cell = idx / C;
                    if (idx - cell * C != c) continue;
                    int h = cell / W, w = cell - h * W;
                    if (h < hlo || h > hhi || w < wlo || w > whi) continue;   // tile & in_roi
                    int phs = (int)floorf((float)(h - g.sh) / g.bin_h);
                    int phe = (int)ceilf((float)(h - g.sh + 1) / g.bin_h);
                    int pws = (int)floorf((float)(w - g.sw) / g.bin_w);
                    int pwe = (int)ceilf((float)(w - g.sw + 1) / g.bin_w);
                    phs = min(max(phs, 0), PH);  phe = min(max(phe, 0), PH);
                    pws = min(max(pws, 0), PW);  pwe = min(max(pwe, 0), PW);
                    if (ph >= phs && ph < phe && pw >= pws && pw < pwe) {
                        float *a = &acc[((h - h0) * TW + (w - w0)) * CG + tc];
                        *a = *a + top_diff[o];
                    }
                }
            }
        }
        __syncthreads();
    }

    if (c_ok) {
        float *img = bottom_diff + (size_t)n * H * W * C;
        for (int i = 0; i < TH * TW; ++i) {
            int h = h0 + i / TW, w = w0 + i % TW;
            if (h < H && w < W) img[((size_t)h * W + w) * C + c] = acc[i * CG + tc];
        }
    }
}

template <int CG>
static int launch_bwd(const float *top_diff, const int *argmax, const float *rois, int R, int N,
                      int H, int W, int C, int PH, int PW, float scale, float *bottom_diff,
                      hipStream_t st) {
    constexpr int TH = 8, TW = 8;
    int tiles_h = cdiv(H, TH), tiles_w = cdiv(W, TW), cgroups = cdiv(C, CG);
    long long blocks = (long long)N * tiles_h * tiles_w * cgroups;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL((roi_pool_bwd_kernel<TH, TW, CG>), dim3((unsigned)blocks), dim3(CG), 0, st,
                       top_diff, argmax, rois, R, N, H, W, C, PH, PW, scale, bottom_diff, tiles_h,
                       tiles_w, cgroups);
    return check_launch();
}

}  // namespace wssdl

using namespace wssdl;

extern "C" int wssdl_roi_pool_forward(const float *bottom, int N, int H, int W, int C,
                                      const float *rois, int R, int pooled_h, int pooled_w,
                                      float spatial_scale, int rounding, float *top,
                                      int32_t *argmax, wssdl_stream_t stream) {
    if (N < 0 || H < 1 || W < 1 || C < 1 || R < 0 || pooled_h < 1 || pooled_w < 1)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if ((long long)H * W * C > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;  // argmax is i32
    if (R == 0) return WSSDL_OK;
    if (!bottom || !rois || !top || !argmax || N < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(bottom) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(top) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(argmax) & 15) == 0);
    long long total = (long long)R * pooled_h * pooled_w * (vec ? C / 4 : C);
    long long blocks = (total + 255) / 256;
    if (blocks > (1LL << 22)) blocks = 1LL << 22;     // grid-stride beyond 4M workgroups
    if (vec)
        hipLaunchKernelGGL(roi_pool_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, bottom,
                           N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top,
                           argmax);
    else
        hipLaunchKernelGGL(roi_pool_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, bottom,
                           N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top,
                           argmax);
    return check_launch();
}

extern "C" int wssdl_roi_pool_backward(const float *top_diff, const int32_t *argmax,
                                       const float *rois, int R, int N, int H, int W, int C,
                                       int pooled_h, int pooled_w, float spatial_scale,
                                       float *bottom_diff, wssdl_stream_t stream) {
    if (N < 0 || H < 1 || W < 1 || C < 1 || R < 0 || pooled_h < 1 || pooled_w < 1)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if ((long long)H * W * C > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax || !rois)))
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    if (C > 128)
        return launch_bwd<256>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                               spatial_scale, bottom_diff, st);
    if (C > 64)
        return launch_bwd<128>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                               spatial_scale, bottom_diff, st);
    return launch_bwd<64>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                          spatial_scale, bottom_diff, st);
}
