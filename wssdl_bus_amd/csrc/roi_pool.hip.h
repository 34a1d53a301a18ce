// Geometry shared by the RoI-pool kernels (roi_pool.hip: reference argmax layout;
// roi_pool_compact.hip: training path with a 1-byte argmax).
//
// Reference: code/lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:36-58 (RoI rounding, bin
// windows, canonical rounding), roi_pooling_op.cc:152-170 (CPU rounding), :169-177 of the
// .cu.cc (candidate bins of a bottom cell in the backward pass).
#pragma once
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

// 1-byte arg-max of the training path (roi_pool_compact.hip): (h - hstart) << 4 | (w - wstart), 0xff = empty bin
constexpr unsigned ARG8_EMPTY = 0xffu;
constexpr int ARG8_MAX_WIN_H = 15;    // dh <= 14: the code 0xff = (15, 15) can never be produced
constexpr int ARG8_MAX_WIN_W = 16;
// window table of the forward: one 32-byte entry per (roi, bin row), written by roi_windows_kernel
constexpr int WIN_ENTRY_WORDS = 8;

typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

// candidate pooled-bin range of one bottom row / column, roi_pooling_op_gpu.cu.cc:169-177
__device__ __forceinline__ void cand_range(int d, float bin, int P, int &s, int &e) {
    s = (int)floorf((float)d / bin);
    e = (int)ceilf((float)(d + 1) / bin);
    s = min(max(s, 0), P);
    e = min(max(e, 0), P);
}

constexpr unsigned TOUCH_GENERIC = 0xfu;
constexpr long long BWD_MIN_WORKGROUPS = 1024;   // one per workgroup slot of the chip (256 CUs x 4)

template <int TN, int FW = 8>
__device__ __forceinline__ void touch_axis(int t0, int t1, int rs, int re, float bin, int P, int &p0,
                                           int &pn, unsigned long long &mask) {
    // tile cells t0..t1 (inclusive, <= TN of them), RoI cells rs..re; mask bit FW*k + j
    static_assert(TN <= FW && FW * 8 <= 64, "one FW-bit field per candidate bin");
    const int lo = max(t0, rs), hi = min(t1, re);
    int s[TN], e[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s[j] = e[j] = 0;
        if (t0 + j >= lo && t0 + j <= hi) cand_range(t0 + j - rs, bin, P, s[j], e[j]);
    }
    int a, z, t;
    cand_range(lo - rs, bin, P, a, t);
    cand_range(hi - rs, bin, P, t, z);
    p0 = a;
    pn = z - a;
    mask = 0ull;
    if (lo > hi) { pn = 0; return; }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (a + k >= s[j] && a + k < e[j]) mask |= 1ull << (FW * k + j);
}


// wave-uniform forward with the i32 arg-max (roi_pool_compact.hip); WSSDL_ROWS_I32_UNSUPPORTED = shape not taken
constexpr int WSSDL_ROWS_I32_UNSUPPORTED = -1000;
int launch_fwd_rows_i32(const float *bottom, int N, int H, int W, int C, const float *rois, int R, int pooled_h,
                        int pooled_w, float spatial_scale, int rounding, float *top, int32_t *argmax, hipStream_t st);

// block-table forward (roi_pool_blocks.hip)
bool blocks_supported(int R, int N, int H, int W, int C, int PH, int PW);
size_t blocks_workspace_bytes(int R, int N, int H, int W, int C);
void blocks_zero_region(void *ws, int R, int N, int H, int W, int C, unsigned **ptr, int *words);

// list-driven backward of the training path (roi_pool_walk.hip)
int walk_plan_count();
size_t walk_flags_offset(int R, int N, int H, int W, int PH, int PW);
bool walk_supported(int R, int N, int H, int W, int C, int PH, int PW);
size_t walk_workspace_bytes(int R, int N, int H, int W, int PH, int PW);
int walk_prepare(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale, int rounding,
                 void *workspace, size_t workspace_bytes, int *plan_out, hipStream_t st, int force_plan = -1);
int walk_split_plan();
int walk_plan_auto_id(int N, int H, int W, int C);
int launch_walk(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C, int PH,
                int PW, float *bottom_diff, void *workspace, size_t workspace_bytes, int plan, hipStream_t st,
                int nseg = 1, float *partial = nullptr, bool i32 = false /* arg8 points at the i32 arg-max */);
bool walk_i32_supported(int R, int N, int H, int W, int C, int PH, int PW);
bool walk_i32_plan_built(int id);
int walk_split_segments(int R, int N, int H, int W, int C);
// bin-owner form (round 5): every bin listed by one tile, halos merged afterwards
int owner_plan_count();
int owner_plan_auto(int R, int N, int H, int W, int C);
bool owner_supported(int R, int N, int H, int W, int C, int PH, int PW);
size_t owner_scratch_bytes(int N, int H, int W, int C, int plan, int nseg = 1);
int owner_split_segments(int R, int N, int H, int W, int C);
int owner_prepare(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale, int rounding,
                  void *workspace, size_t workspace_bytes, int plan, hipStream_t st);
int launch_owner(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C, int PH, int PW,
                 float *bottom_diff, void *workspace, size_t workspace_bytes, int plan, float *halo, hipStream_t st,
                 bool i32 = false /* arg8 points at the i32 arg-max (owner plans 0 and 1) */,
                 int nseg = 1 /* > 1: a tile's stream walked by nseg waves, halo = nseg region buffers (round 6) */);

}  // namespace wssdl
