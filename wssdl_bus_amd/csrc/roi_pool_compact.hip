// RoI max-pooling, TRAINING path with a 1-byte arg-max, for gfx950 (MI355X).
//
// In the reference the arg-max tensor is an internal hand-off: network.py:206-210 keeps only
// top_data ([0] of the op's outputs) and the registered gradient (roi_pooling_op_grad.py:24-44)
// feeds argmax straight back into RoiPoolGrad.  Its i32 layout (a flat NHWC index,
// roi_pooling_op_gpu.cu.cc:71-79) costs as many bytes as the activations, and both kernels are
// HBM-bound on exactly those bytes.  Here the pair (forward, backward) agrees on one byte per
// element instead:
//
//     code = (h - hstart) << 4 | (w - wstart)        0xff = empty bin (the reference's -1)
//
// where [hstart, hend) x [wstart, wend) is the bin's CLIPPED window (roi_pooling_op_gpu.cu.cc:
// 51-64 / roi_pooling_op.cc:167-176).  Windows are at most 15 rows x 16 columns for every RoI
// that lies inside a feature map of up to 97 x 104 cells with 7 x 7 bins (the host checks the
// map size; a window beyond that sets *overflow).  wssdl_roi_argmax_expand turns the codes
// back into the reference's i32 indices (tests compare those with the oracle bit for bit).
//
// Forward : roi_pool_fwd_rows_kernel -- one wave per (roi, ph) bin row x 256 channels, scalar window
//           loops, the row walked through its 7 bins with the shared columns kept in registers, RoI
//           geometry from the table of roi_windows_kernel; channel slice <-> XCD as in roi_pool.hip.
//           (roi_pool_fwd_compact_kernel, the round-1 sliced form with the 1-byte store, serves the
//           shapes the wave-uniform kernel does not take.)
// Backward: roi_pool_walk.hip (lists built by the prepare step, one wave per (image, tile, 128
//           channels)).  The kernel in THIS file is the fallback without a workspace: the tile-owner
//           kernel of roi_pool.hip (4x4-cell x 256-channel tiles of bottom_diff in LDS, RoIs filtered
//           per tile in RoI order, candidate bins walked in (ph, pw) order, one lane per channel =>
//           the reference's f32 summation order, bit-identical) reading 4 + 1 bytes per visited
//           element; its (RoI, tile) record carries the window start of each candidate bin row /
//           column relative to the tile, so decoding a code is two adds and two mask look-ups (no
//           division by W or C as with the flat index).
#include "roi_pool.hip.h"

namespace wssdl {

// (ARG8_EMPTY, ARG8_MAX_WIN_H / _W and the window-table layout: roi_pool.hip.h)

// clipped window start of bin p: roi_pooling_op_gpu.cu.cc:51-52,61-63 / roi_pooling_op.cc:167-168,173-175
__device__ __forceinline__ int win_start(int p, float bin, int rs, int limit, int rounding) {
    const float v = (float)p * bin;
    const int s = (rounding == WSSDL_ROI_ROUND_CPU) ? (int)v : (int)floorf(v);
    return min(max(s + rs, 0), limit);
}

__device__ __forceinline__ int win_end(int p, float bin, int rs, int limit, int rounding) {
    const float v = (float)(p + 1) * bin;
    const int e = (rounding == WSSDL_ROI_ROUND_CPU) ? (int)v : (int)ceilf(v);
    return min(max(e + rs, 0), limit);
}

// ------------------------------------------------------------------ forward ---
// Workgroup b serves channel slice b % 8 (one XCD's L2 then holds only its slice of the feature
// map); a lane owns 4 channels of one (roi, ph) bin row and walks its PW bins.  C % 32 == 0.
template <int BATCH>
__global__ __launch_bounds__(256) void roi_pool_fwd_compact_kernel(
    const float *__restrict__ bottom, int N, int H, int W, int C, const float *__restrict__ rois,
    int R, int PH, int PW, float scale, int rounding, float *__restrict__ top,
    unsigned *__restrict__ arg8 /* [R,PH,PW,C] bytes, written 4 at a time */,
    int *__restrict__ overflow, int lanes_per_row /* = C/32 */, int rows_per_block) {
    const int slice = blockIdx.x & 7;
    const long long rows = (long long)R * PH;
    const long long row = (long long)(blockIdx.x >> 3) * rows_per_block + threadIdx.x / lanes_per_row;
    if (row >= rows) return;
    const int lane_in_row = threadIdx.x % lanes_per_row;
    const int c0 = slice * (C >> 3) + lane_in_row * 4;
    const int r = (int)(row / PH), ph = (int)(row - (long long)r * PH);
    const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
    const bool bad = g.batch < 0 || g.batch >= N;
    const float *img = bottom + (size_t)(bad ? 0 : g.batch) * H * W * C;
    const int hs = win_start(ph, g.bin_h, g.sh, H, rounding);
    const int he = win_end(ph, g.bin_h, g.sh, H, rounding);
    size_t o = ((size_t)row * PW) * C + c0;
    for (int pw = 0; pw < PW; ++pw, o += C) {
        const int ws = win_start(pw, g.bin_w, g.sw, W, rounding);
        const int we = win_end(pw, g.bin_w, g.sw, W, rounding);
        const bool empty = (he <= hs) || (we <= ws) || bad;
        float4v mv = empty ? (float4v)(0.0f) : (float4v)(-FLT_MAX);
        unsigned m0 = ARG8_EMPTY, m1 = ARG8_EMPTY, m2 = ARG8_EMPTY, m3 = ARG8_EMPTY;
        if (!empty) {
            if ((he - hs > ARG8_MAX_WIN_H || we - ws > ARG8_MAX_WIN_W) && overflow) atomicOr(overflow, 1);
            // BATCH cells of a window row are fetched together; slots past the window's last
            // column re-read that column: an equal value never passes the strict >, so the scan
            // order and the first-maximum rule (roi_pooling_op_gpu.cu.cc:71-79) are untouched.
            for (int h = hs; h < he; ++h) {
                const int row_base = h * W * C + c0;
                const unsigned rcode = (unsigned)(h - hs) << 4;
                for (int w = ws; w < we; w += BATCH) {
                    float4v v[BATCH];
                    unsigned code[BATCH];
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) {
                        const int wj = min(w + j, we - 1);
                        code[j] = rcode | (unsigned)(wj - ws);
                        v[j] = *reinterpret_cast<const float4v *>(img + row_base + wj * C);
                    }
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) {
                        if (v[j].x > mv.x) { mv.x = v[j].x; m0 = code[j]; }
                        if (v[j].y > mv.y) { mv.y = v[j].y; m1 = code[j]; }
                        if (v[j].z > mv.z) { mv.z = v[j].z; m2 = code[j]; }
                        if (v[j].w > mv.w) { mv.w = v[j].w; m3 = code[j]; }
                    }
                }
            }
        }
        __builtin_nontemporal_store(mv, reinterpret_cast<float4v *>(top + o));
        __builtin_nontemporal_store(m0 | (m1 << 8) | (m2 << 16) | (m3 << 24), arg8 + (o >> 2));
    }
}

// Window table of the forward (7 x 7 bins): one 32-byte entry per (roi, ph) bin row,
//   word 0 = batch index, word 1 = hs | he << 8 (clipped window rows of the bin row),
//   bytes 8..14 = wstart of the 7 bin columns, bytes 16..22 = wend.
// Written by roi_windows_kernel (one lane per bin row), read by the forward with two scalar loads: the
// RoI geometry -- coordinates, two IEEE divisions, floor / ceil per bin -- then costs a forward wave no
// vector instruction at all instead of ~170 of its ~890.

__global__ __launch_bounds__(256) void roi_windows_kernel(const float *__restrict__ rois, int R, int N, int H, int W,
                                                          float scale, int rounding, unsigned *__restrict__ table,
                                                          int *__restrict__ overflow, unsigned *__restrict__ zero,
                                                          int zero_words) {
    // (block-table forward: the counters of its sort start from zero -- cleared here instead of by a memset launch)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_words; i += gridDim.x * 256) zero[i] = 0u;
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= (long long)R * 7) return;
    const int r = (int)(item / 7), ph = (int)(item - (long long)r * 7);
    const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, 7, 7);
    const int hs = win_start(ph, g.bin_h, g.sh, H, rounding), he = win_end(ph, g.bin_h, g.sh, H, rounding);
    unsigned wsb[2] = {0u, 0u}, web[2] = {0u, 0u}, blk = 0u;
    bool wide = false, any = false;
    const int a = he - hs;
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
        const int ws = win_start(pw, g.bin_w, g.sw, W, rounding), we = win_end(pw, g.bin_w, g.sw, W, rounding);
        wsb[pw >> 2] |= (unsigned)ws << (8 * (pw & 3));
        web[pw >> 2] |= (unsigned)we << (8 * (pw & 3));
        wide |= we - ws > ARG8_MAX_WIN_W;
        any |= we > ws;
        // block-table forward (roi_pool_blocks.hip): the window is the union of four k x k blocks anchored at its
        // corners when k <= min(a, b) and max(a, b) <= 2k; the largest such k in {2, 3, 4}, as k - 1 (0: none)
        const int b = we - ws, lo = min(a, b), hi = max(a, b);
        unsigned kc = 0u;
#pragma unroll
        for (int k = 2; k <= 4; ++k)
            if (k <= lo && hi <= 2 * k) kc = (unsigned)(k - 1);
        blk |= kc << (2 * pw);
    }
    const bool bad = g.batch < 0 || g.batch >= N;
    unsigned *e = table + (size_t)item * WIN_ENTRY_WORDS;
    e[0] = (unsigned)g.batch;
    e[1] = (unsigned)hs | ((unsigned)he << 8);
    e[2] = wsb[0];  e[3] = wsb[1];  e[4] = web[0];  e[5] = web[1];
    e[6] = blk;                                                            // 2 bits per bin column
    e[7] = bad ? 0u : (unsigned)(g.batch * H + min(hs, H - 1));           // sort key of the bin row: (image, first window row)
    if (overflow && !bad && he > hs && any && (wide || he - hs > ARG8_MAX_WIN_H)) atomicOr(overflow, 1);
}

// Forward, wave-uniform form: ONE wave = one (roi, ph) bin row x 64*CPL channels, so the window
// bounds, the loops over the window and the cell offsets are scalar (SALU); a cell costs one
// buffer load (scalar cell offset + constant lane offset) and 3 VALU per channel.  The sliced
// kernel above packs two bin rows with different windows into a wave: its loops are vector code
// under exec masks and every cell pays 64-bit address arithmetic (measured: 0.49 ms with the
// loads removed, 0.59 ms with the stores removed, 0.72 ms together at R = 8512 for 2.2 GB of
// HBM traffic -- instruction issue, not bandwidth; this form: 0.44 / 0.63 / 0.59 ms).  Channel
// slice <-> blockIdx % 8 as above (without the slicing: 1.13 ms).  Tried without gain: four
// cells in flight, 7 rows (one RoI) per workgroup, plain instead of non-temporal stores (+5-10 %);
// 512 channels per wave (coarser L2 slices): 0.80 ms; the stores alone take 0.32-0.37 ms.
template <int CPL>
struct LaneVec;
template <>
struct LaneVec<4> {
    typedef float4v vec;
    static __device__ __forceinline__ vec load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
};
typedef float float2w __attribute__((ext_vector_type(2)));
template <>
struct LaneVec<2> {
    typedef float2w vec;
    static __device__ __forceinline__ vec load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
};

// I32 (round 4): the same kernel with the reference op's own second output, `argmax` [R,PH,PW,C] i32 = the flat NHWC
// index inside the RoI's image, -1 for an empty bin (roi_pooling_op.cc:31-52, roi_pooling_op_gpu.cu.cc:71-79): the
// running arg-max is then the cell index h * W + w (no window-relative code, so any window size is fine and
// nothing can overflow), turned into (cell * C + channel) at the store.  N <= 0 means "batch size unknown": the
// reference's ROIPoolForwardLaucher is not told it (roi_pooling_op_gpu.h:17-21), so only a negative batch
// index makes a RoI empty.
template <int CPL, int RPW /* waves per workgroup */, int PWS /* PW when known at compile time, else 0 */,
          bool WHOLE_ROI /* a wave walks all PH bin rows of one RoI instead of one bin row */,
          bool TAB /* the windows come from the table of roi_windows_kernel (PH = PWS = 7, one bin row per wave) */,
          bool I32 = false /* i32 flat-index arg-max instead of the 1-byte codes */,
          int SPLIT = 0 /* > 0: SPLIT waves share a bin row, ceil(PW / SPLIT) bins each (small launches: more, shorter waves) */>
__global__ __launch_bounds__(64 * RPW) void roi_pool_fwd_rows_kernel(
    const float *__restrict__ bottom, int N, int H, int W, int C, const float *__restrict__ rois,
    int R, int PH, int PW, float scale, int rounding, float *__restrict__ top,
    unsigned char *__restrict__ arg8, int *__restrict__ overflow, int slices,
    const unsigned *__restrict__ table /* window table (7 x 7 bins, one bin row per wave) or NULL */) {
    constexpr unsigned EMPTY = I32 ? 0xffffffffu : ARG8_EMPTY;
    typedef typename LaneVec<CPL>::vec vec;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx -> (channel slice, group of RPW items): all groups of a slice share blockIdx % 8.
    // (divisions by run-time values are ~20 VALU instructions each: shifts when `slices` is a power
    // of two, a constant divisor when PH == PWS)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const bool pow2 = (slices & (slices - 1)) == 0;
    const int sh = 31 - __builtin_clz(slices);
    int slice, group;
    if (slices >= 8) {
        const int per = slices >> 3;
        if (pow2) { slice = xcd + 8 * (q & (per - 1));  group = q >> (sh - 3); }
        else { slice = xcd + 8 * (q % per);  group = q / per; }
    } else {       // 1, 2 or 4 slices
        slice = xcd & (slices - 1);
        group = (q << (3 - sh)) + (xcd >> sh);
    }
    // item = one RoI (WHOLE_ROI) or one (roi, ph) bin row
    static_assert(SPLIT == 0 || (PWS == 0 && !WHOLE_ROI && !TAB), "split bin rows: the generic bin loop, no table");
    const long long witem = (long long)group * RPW + wave;
    const int per_wave = SPLIT > 0 ? (PW + SPLIT - 1) / SPLIT : PW;
    const int pw_first = SPLIT > 0 ? (int)(witem % SPLIT) * per_wave : 0, pw_last = min(PW, pw_first + per_wave);
    const long long item = SPLIT > 0 ? witem / SPLIT : witem;
    const long long items = WHOLE_ROI ? (long long)R : (long long)R * PH;
    if (item >= items) return;
    const int r = WHOLE_ROI ? (int)item : ((PWS > 0 && PH == PWS) ? (int)(item / (PWS > 0 ? PWS : 1)) : (int)(item / PH));
    const int ph_first = WHOLE_ROI ? 0 : (int)(item - (long long)r * PH);
    const int ph_last = WHOLE_ROI ? PH : ph_first + 1;
    // The geometry of the RoI costs ~170 VALU instructions per wave (coordinates, two IEEE
    // divisions, the windows): a fifth of a one-row wave's VALU work, and the kernel's VALU is 75 %
    // busy (SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES).  Paying it once per RoI was measured both ways
    // and lost both times: a wave that walks the whole RoI (WHOLE_ROI: 0.80 ms, 34 k waves of very
    // unequal length) and 7-wave workgroups whose wave 0 shares it through LDS (+4 %: the barrier
    // and the coarser workgroup slots cost more than the instructions saved).
    static_assert(!TAB || (PWS == 7 && !WHOLE_ROI), "the window table describes 7 x 7 bins, one bin row per entry");
    int batch, t_hs = 0, t_he = 0;
    unsigned t_ws0 = 0u, t_ws1 = 0u, t_we0 = 0u, t_we1 = 0u;
    int my_ws = 0, my_we = 0, my_hs = 0, my_he = 0;
    constexpr bool use_table = TAB;
    if constexpr (use_table) {
        // (uniform address: two scalar loads)
        const unsigned *e = table + (size_t)item * WIN_ENTRY_WORDS;
        batch = (int)e[0];
        t_hs = (int)(e[1] & 0xffu);
        t_he = (int)((e[1] >> 8) & 0xffu);
        t_ws0 = e[2];  t_ws1 = e[3];  t_we0 = e[4];  t_we1 = e[5];
    } else {
        const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);        // wave-uniform values
        batch = __builtin_amdgcn_readfirstlane(g.batch);
        // windows of the bin columns and rows: lane p computes bin column p and bin row p once (a lane's
        // own copy per bin cost ~20 VALU instructions per bin), read back with v_readlane; PW, PH <= 64
        const int pme = (lane < PW) ? lane : 0;
        my_ws = win_start(pme, g.bin_w, g.sw, W, rounding);
        my_we = win_end(pme, g.bin_w, g.sw, W, rounding);
        const int phme = (lane < PH) ? lane : 0;
        my_hs = win_start(phme, g.bin_h, g.sh, H, rounding);
        my_he = win_end(phme, g.bin_h, g.sh, H, rounding);
        const unsigned long long wide = __ballot(lane < PW && my_we - my_ws > ARG8_MAX_WIN_W);    // (every lane votes)
        const unsigned long long tall = __ballot(lane < PH && lane >= ph_first && lane < ph_last && my_he - my_hs > ARG8_MAX_WIN_H);
        const unsigned long long cols = __ballot(lane < PW && my_we > my_ws);
        const unsigned long long rows_live = __ballot(lane < PH && lane >= ph_first && lane < ph_last && my_he > my_hs);
        const bool bad0 = batch < 0 || (N > 0 && batch >= N);
        if (!I32 && overflow && !bad0 && lane == 0 && ((wide != 0ull && rows_live != 0ull) || (tall != 0ull && cols != 0ull)))
            atomicOr(overflow, 1);
    }
    const bool bad = batch < 0 || (N > 0 && batch >= N);
    const int c0 = (slice * 64 + lane) * CPL;
    const bool lane_ok = c0 < C;
    const int voff = (lane_ok ? c0 : 0) * 4;
    const int cell_bytes = C * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(bottom + (size_t)(bad ? 0 : batch) * H * W * C), 0, H * W * cell_bytes, 0x00020000);

    for (int ph = ph_first; ph < ph_last; ++ph) {
    const int hs = use_table ? t_hs : __builtin_amdgcn_readlane(my_hs, ph);
    const int he = use_table ? t_he : __builtin_amdgcn_readlane(my_he, ph);
    const bool row_dead = (he <= hs) || bad;
    const size_t o_row = (((size_t)r * PH + ph) * PW) * C + c0;
    // one bin: the reference's scan (h ascending, w ascending, strict >: roi_pooling_op_gpu.cu.cc:66-79)
    auto pool_bin = [&](int ws, int we, vec &mv, unsigned (&mi)[CPL]) {
        const bool empty = row_dead || (we <= ws);
#pragma unroll
        for (int k = 0; k < CPL; ++k) { mv[k] = empty ? 0.0f : -FLT_MAX;  mi[k] = EMPTY; }
        if (empty) return;
        for (int h = hs; h < he; ++h) {
            const int so_row = h * W * cell_bytes;
            const unsigned rcode = I32 ? (unsigned)(h * W + ws) : (unsigned)(h - hs) << 4;      // code of (h, ws)
            int w = ws;
            if constexpr (SPLIT > 0) {
                for (; w + 3 < we; w += 4) {      // experiment: four cells in flight on the short waves of a small launch
                    const vec v0 = LaneVec<CPL>::load(rs, voff, so_row + w * cell_bytes);
                    const vec v1 = LaneVec<CPL>::load(rs, voff, so_row + (w + 1) * cell_bytes);
                    const vec v2 = LaneVec<CPL>::load(rs, voff, so_row + (w + 2) * cell_bytes);
                    const vec v3 = LaneVec<CPL>::load(rs, voff, so_row + (w + 3) * cell_bytes);
                    const unsigned code0 = rcode + (unsigned)(w - ws);
#pragma unroll
                    for (int k = 0; k < CPL; ++k) if (v0[k] > mv[k]) { mv[k] = v0[k];  mi[k] = code0; }
#pragma unroll
                    for (int k = 0; k < CPL; ++k) if (v1[k] > mv[k]) { mv[k] = v1[k];  mi[k] = code0 + 1u; }
#pragma unroll
                    for (int k = 0; k < CPL; ++k) if (v2[k] > mv[k]) { mv[k] = v2[k];  mi[k] = code0 + 2u; }
#pragma unroll
                    for (int k = 0; k < CPL; ++k) if (v3[k] > mv[k]) { mv[k] = v3[k];  mi[k] = code0 + 3u; }
                }
            }
            for (; w + 1 < we; w += 2) {          // two cells in flight
                const vec v0 = LaneVec<CPL>::load(rs, voff, so_row + w * cell_bytes);
                const vec v1 = LaneVec<CPL>::load(rs, voff, so_row + (w + 1) * cell_bytes);
                const unsigned code0 = rcode + (unsigned)(w - ws), code1 = code0 + 1u;
#pragma unroll
                for (int k = 0; k < CPL; ++k) if (v0[k] > mv[k]) { mv[k] = v0[k];  mi[k] = code0; }
#pragma unroll
                for (int k = 0; k < CPL; ++k) if (v1[k] > mv[k]) { mv[k] = v1[k];  mi[k] = code1; }
            }
            if (w < we) {
                const vec v0 = LaneVec<CPL>::load(rs, voff, so_row + w * cell_bytes);
                const unsigned code0 = rcode + (unsigned)(w - ws);
#pragma unroll
                for (int k = 0; k < CPL; ++k) if (v0[k] > mv[k]) { mv[k] = v0[k];  mi[k] = code0; }
            }
        }
    };
    auto store_bin = [&](int pw, const vec &mv, const unsigned (&m)[CPL]) {
        const size_t o = o_row + (size_t)pw * C;
        if (lane_ok)
        {
            __builtin_nontemporal_store(mv, reinterpret_cast<vec *>(top + o));
            if constexpr (I32) {
                // cell index -> flat NHWC index of the lane's channels (roi_pooling_op_gpu.cu.cc:71-79), -1 = empty
                typedef int ivec __attribute__((ext_vector_type(CPL)));
                ivec idx;
#pragma unroll
                for (int k = 0; k < CPL; ++k) idx[k] = m[k] == EMPTY ? -1 : (int)(m[k] * (unsigned)C) + c0 + k;
                __builtin_nontemporal_store(idx, reinterpret_cast<ivec *>(reinterpret_cast<int *>(arg8) + o));
            } else {
                const unsigned codes = (CPL == 4) ? (m[0] | (m[1] << 8) | (m[2 % CPL] << 16) | (m[3 % CPL] << 24))
                                                  : (m[0] | (m[1] << 8));
                if (CPL == 4) __builtin_nontemporal_store(codes, reinterpret_cast<unsigned *>(arg8 + o));
                else __builtin_nontemporal_store((unsigned short)codes, reinterpret_cast<unsigned short *>(arg8 + o));
            }
        }
    };
    if constexpr (PWS > 0) {
        // PW known at compile time: the bin row is walked ROW BY ROW through all its bins, with the
        // PWS running maxima in registers (8 per bin).
        // * Adjacent bins share a column whenever their boundary is not a whole cell
        //   (floor / ceil, roi_pooling_op_gpu.cu.cc:55-58): the last cell a bin loads in a window row is
        //   handed to the next bin in registers instead of being loaded again -- 26 % fewer loads on the
        //   proposals of a train step (SQ_INSTS_VMEM_RD); the loads all go to L2 (the reuse distance
        //   between two bins of a wave is ~190 KiB of other waves' traffic, the L1 holds 32), which runs at
        //   55 % of its peak here.  A bin's own scan order is untouched: (h ascending, w ascending), strict >.
        // * The stores of the row are issued together behind its last bin.  On gfx9-family hardware
        //   stores and loads share the in-order vmcnt counter: a wave that waits for a load also waits
        //   for every store it issued before it, so a store per bin puts a store acknowledgement on
        //   the wave's critical path per bin instead of per row.
        vec res[PWS];
        unsigned mi[PWS][CPL];
        int wss[PWS], wes[PWS];
#pragma unroll
        for (int pw = 0; pw < PWS; ++pw) {
            if (use_table) {
                wss[pw] = (int)(((pw < 4 ? t_ws0 : t_ws1) >> (8 * (pw & 3))) & 0xffu);
                wes[pw] = (int)(((pw < 4 ? t_we0 : t_we1) >> (8 * (pw & 3))) & 0xffu);
            } else {
                wss[pw] = __builtin_amdgcn_readlane(my_ws, pw);
                wes[pw] = __builtin_amdgcn_readlane(my_we, pw);
            }
            const bool empty = row_dead || (wes[pw] <= wss[pw]);
#pragma unroll
            for (int k = 0; k < CPL; ++k) { res[pw][k] = empty ? 0.0f : -FLT_MAX;  mi[pw][k] = EMPTY; }
        }
        auto upd = [](vec &mv, unsigned (&m)[CPL], const vec &v, unsigned code) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) if (v[k] > mv[k]) { mv[k] = v[k];  m[k] = code; }
        };
        if (!row_dead) {
            for (int h = hs; h < he; ++h) {
                const int so_row = h * W * cell_bytes;
                const unsigned rcode0 = I32 ? (unsigned)(h * W) : (unsigned)(h - hs) << 4;
                vec carry = (vec)(0.0f);          // cell (h, carry_w): the last one the previous bin loaded
                int carry_w = -1;
#pragma unroll
                for (int pw = 0; pw < PWS; ++pw) {
                    const int ws = wss[pw], we = wes[pw];
                    if (we <= ws) continue;
                    const unsigned rcode = I32 ? rcode0 + (unsigned)ws : rcode0;      // code of (h, ws)
                    int w = ws;
                    if (carry_w == ws) { upd(res[pw], mi[pw], carry, rcode);  w = ws + 1; }
                    for (; w + 1 < we; w += 2) {          // two cells in flight
                        const vec v0 = LaneVec<CPL>::load(rs, voff, so_row + w * cell_bytes);
                        const vec v1 = LaneVec<CPL>::load(rs, voff, so_row + (w + 1) * cell_bytes);
                        const unsigned code0 = rcode + (unsigned)(w - ws);
                        upd(res[pw], mi[pw], v0, code0);
                        upd(res[pw], mi[pw], v1, code0 + 1u);
                        carry = v1;
                    }
                    if (w < we) {
                        const vec v0 = LaneVec<CPL>::load(rs, voff, so_row + w * cell_bytes);
                        upd(res[pw], mi[pw], v0, rcode + (unsigned)(w - ws));
                        carry = v0;
                    }
                    carry_w = we - 1;
                }
            }
        }
#pragma unroll
        for (int pw = 0; pw < PWS; ++pw) store_bin(pw, res[pw], mi[pw]);
    } else {
        for (int pw = pw_first; pw < pw_last; ++pw) {
            vec mv;
            unsigned mi[CPL];
            pool_bin(__builtin_amdgcn_readlane(my_ws, pw), __builtin_amdgcn_readlane(my_we, pw), mv, mi);
            store_bin(pw, mv, mi);
        }
    }
    }
}

// codes -> the reference's flat NHWC index (roi_pooling_op_gpu.cu.cc:71-79); one lane = 4 channels
__global__ __launch_bounds__(256) void roi_argmax_expand_kernel(
    const unsigned *__restrict__ arg8, const float *__restrict__ rois, long long total4, int H, int W,
    int C, int PH, int PW, float scale, int rounding, int4v *__restrict__ out) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long bin = i / C4;
        const int c0 = (int)(i - bin * C4) * 4;
        const int pw = (int)(bin % PW), ph = (int)((bin / PW) % PH);
        const int r = (int)(bin / ((long long)PW * PH));
        const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
        const int hs = win_start(ph, g.bin_h, g.sh, H, rounding);
        const int ws = win_start(pw, g.bin_w, g.sw, W, rounding);
        const unsigned codes = arg8[i];
        int4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned a = (codes >> (8 * j)) & 0xffu;
            const int idx = ((hs + (int)(a >> 4)) * W + ws + (int)(a & 15u)) * C + c0 + j;
            o[j] = (a == ARG8_EMPTY) ? -1 : idx;
        }
        out[i] = o;
    }
}

// ----------------------------------------------------------------- backward ---
// One (RoI, tile) intersection (32 B in LDS).  geo / rowmask / colmask as in roi_pool.hip:
//   geo     = ph0 | pw0 << 8 | phn << 16 | pwn << 20 | (r - chunk base) << 24
//   rowmask : bit 4*k + j <=> tile row h0+j lies in the RoI and bin row ph0+k is one of its
//             candidate rows (roi_pooling_op_gpu.cu.cc:141-151,169-177); colmask: bit 8*k + j
//   hs      : byte k = (clipped window start of bin row ph0+k) - h0, clamped to [-16, 15];
//   ws      : byte k likewise for bin column pw0+k and w0.
struct TouchRecC {
    unsigned geo;
    unsigned rowmask;
    unsigned long long colmask;
    unsigned long long hs;
    unsigned long long ws;
};

struct WalkCtxC {
    float *acc;            // LDS tile [TH*TW][CG]
    int tc;
    bool lane_ok;          // channel < C
    int voff8, voff;       // lane's byte offset inside a bin of arg8 / top_diff
};

template <int PWN, int NROWS, int TW, int CG>
__device__ __forceinline__ void visit_rows_c(const WalkCtxC &x, __amdgpu_buffer_rsrc_t ra,
                                             __amdgpu_buffer_rsrc_t rt, int so8, int so, int C, int PW,
                                             unsigned rowmask, int row0, unsigned long long colmask,
                                             unsigned long long hs, unsigned long long ws) {
    unsigned a[NROWS][PWN];
    float td[NROWS][PWN];
#pragma unroll
    for (int q = 0; q < NROWS; ++q)
#pragma unroll
        for (int j = 0; j < PWN; ++j) {
            const int bin = q * PW + j;
            a[q][j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(ra, x.voff8, so8 + bin * C, 0);
            td[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, x.voff, so + bin * C * 4, 0));
        }
    // keep every loaded value live here: otherwise the compiler sinks the top_diff loads into
    // the (rare) hit branch and serialises them
#pragma unroll
    for (int q = 0; q < NROWS; ++q)
#pragma unroll
        for (int j = 0; j < PWN; ++j) asm volatile("" : "+v"(a[q][j]), "+v"(td[q][j]));
#pragma unroll
    for (int q = 0; q < NROWS; ++q) {
        const int sh4 = 4 * (row0 + q), sh8 = 8 * (row0 + q);
        const unsigned rm = (rowmask >> sh4) & 0xfu;
        const int hsq = (int)(signed char)(hs >> sh8);
#pragma unroll
        for (int j = 0; j < PWN; ++j) {
            const unsigned cmk = (unsigned)(colmask >> (8 * j)) & 0xffu;
            const int wsj = (int)(signed char)(ws >> (8 * j));
            const unsigned code = a[q][j];
            const int th = hsq + (int)(code >> 4), tw = wsj + (int)(code & 15u);
            // tile, in_roi and candidate-bin tests: the masks have no bits at positions a cell
            // outside the tile would index (th, tw in [-16, 30]; shifts use the low 5 bits)
            const unsigned bits = (rm >> (th & 31)) & (cmk >> (tw & 31)) & 1u;
            const bool ok = (bits != 0u) & (code != ARG8_EMPTY) & x.lane_ok;
            if (ok) {
                float *p = &x.acc[(th * TW + tw) * CG + x.tc];
                *p = *p + td[q][j];
            }
        }
    }
}

template <int PWN, int TW, int CG, int MAXB>
__device__ __forceinline__ void visit_roi_c(const WalkCtxC &x, __amdgpu_buffer_rsrc_t ra,
                                            __amdgpu_buffer_rsrc_t rt, int so8, int so, int C, int PW,
                                            unsigned rowmask, int phn, unsigned long long colmask,
                                            unsigned long long hs, unsigned long long ws) {
    constexpr int NR = (MAXB / PWN) >= 4 ? 4 : ((MAXB / PWN) >= 3 ? 3 : 2);
    int rb = 0;
    for (; rb + NR <= phn; rb += NR, so8 += NR * PW * C, so += NR * PW * C * 4)
        visit_rows_c<PWN, NR, TW, CG>(x, ra, rt, so8, so, C, PW, rowmask, rb, colmask, hs, ws);
    const int rem = phn - rb;
    if (NR > 3 && rem == 3)
        visit_rows_c<PWN, 3, TW, CG>(x, ra, rt, so8, so, C, PW, rowmask, rb, colmask, hs, ws);
    else if (NR > 2 && rem == 2)
        visit_rows_c<PWN, 2, TW, CG>(x, ra, rt, so8, so, C, PW, rowmask, rb, colmask, hs, ws);
    else if (rem == 1)
        visit_rows_c<PWN, 1, TW, CG>(x, ra, rt, so8, so, C, PW, rowmask, rb, colmask, hs, ws);
}

template <int TH, int TW, int CG, int CHUNK, int MAXB, int MINB>
__global__ __launch_bounds__(CG, MINB) void roi_pool_bwd_compact_kernel(
    const float *__restrict__ top_diff, const unsigned char *__restrict__ arg8,
    const float *__restrict__ rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
    int rounding, float *__restrict__ bottom_diff, int tiles_h, int tiles_w, int cgroups) {
    static_assert(TH <= 4 && TW <= 8, "4 mask bits per candidate bin row, 8 per column");
    static_assert(CHUNK <= 256, "8-bit RoI index inside a filter round");
    static_assert(CHUNK % CG == 0 || CHUNK < CG, "whole filter rounds");
    constexpr int KPT = CHUNK >= CG ? CHUNK / CG : 1;   // RoIs tested per thread per filter round
    constexpr int NW = CG / WSSDL_WAVE;
    __shared__ float acc[TH * TW * CG];
    __shared__ TouchRecC list[CHUNK];
    __shared__ int wave_cnt[KPT][NW];
    __shared__ int roi_span[2];

    // Workgroup -> (image, channel group, tile): all tiles of one (image, channel group) pair
    // share blockIdx % 8 (observed: one XCD), so border bins re-read by neighbouring tiles can
    // meet in one L2.  Placement affects speed only.
    const int pairs = N * cgroups, tiles = tiles_h * tiles_w;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = xcd + 8 * (slot / tiles);
    if (pair >= pairs) return;
    const int tile = slot % tiles;
    const int cg = pair % cgroups, n = pair / cgroups;
    const int tx = tile % tiles_w, ty = tile / tiles_w;
    const int tc = threadIdx.x;
    const int c = cg * CG + tc;
    const bool c_ok = c < C;
    const int h0 = ty * TH, w0 = tx * TW;
    const int h1 = min(h0 + TH, H) - 1, w1 = min(w0 + TW, W) - 1;   // inclusive
    const int lane = tc & (WSSDL_WAVE - 1), wave = tc / WSSDL_WAVE;
    const int cl = c_ok ? c : C - 1;       // lanes past C read channel C-1 and never accumulate
    const int roi_elems = PH * PW * C;
    WalkCtxC wx;
    wx.acc = acc;  wx.tc = tc;  wx.lane_ok = c_ok;  wx.voff8 = cl;  wx.voff = cl * 4;

#pragma unroll
    for (int i = 0; i < TH * TW; ++i) acc[i * CG + tc] = 0.0f;

    // ---- the span of RoI indices that belong to image n (RoIs normally arrive grouped by image;
    // any order stays correct)
    if (tc == 0) { roi_span[0] = R; roi_span[1] = -1; }
    __syncthreads();
    {
        int lo = R, hi = -1;
        for (int r = tc; r < R; r += CG)
            if ((int)rois[(size_t)r * 5] == n) { lo = min(lo, r); hi = r; }
        if (hi >= 0) { atomicMin(&roi_span[0], lo); atomicMax(&roi_span[1], hi); }
    }
    __syncthreads();
    const int r_begin = roi_span[0], r_end = roi_span[1] + 1;

    for (int base = r_begin; base < r_end; base += CHUNK) {
        // ---- filter: RoIs of image n whose rounded box touches the tile, in RoI order
        bool hit[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int r = base + k * CG + tc;
            hit[k] = false;
            if (r < r_end && k * CG + tc < CHUNK) {
                const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
                hit[k] = (g.batch == n) && g.sw <= w1 && g.ew >= w0 && g.sh <= h1 && g.eh >= h0;
            }
            const unsigned long long m = __ballot(hit[k]);
            if (lane == 0) wave_cnt[k][wave] = __popcll(m);
        }
        __syncthreads();
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const unsigned long long m = __ballot(hit[k]);
            int before = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int wc = wave_cnt[k][w];
                before += (w < wave) ? wc : 0;
            }
            if (hit[k])   // only the RoI's index for now; finished below
                list[cnt + before + __popcll(m & ((1ull << lane) - 1ull))].geo = (unsigned)(k * CG + tc);
#pragma unroll
            for (int w = 0; w < NW; ++w) cnt += wave_cnt[k][w];
        }
        __syncthreads();
        // finish the records with the hits packed into the first threads: candidate ranges of the
        // tile's rows / columns (the reference's backward tests) and the forward's window starts
        for (int t = tc; t < cnt; t += CG) {
            TouchRecC q;
            const unsigned rel = list[t].geo;
            const RoiGeom g = roi_geometry(rois + (size_t)(base + (int)rel) * 5, scale, PH, PW);
            int ph0, phn, pw0, pwn;
            unsigned long long rm;
            touch_axis<TH, 4>(h0, h1, g.sh, g.eh, g.bin_h, PH, ph0, phn, rm);
            touch_axis<TW, 8>(w0, w1, g.sw, g.ew, g.bin_w, PW, pw0, pwn, q.colmask);
            q.hs = q.ws = 0ull;
            if (phn <= 0 || pwn <= 0) phn = pwn = 0;                          // nothing to visit
            else if (phn > 8 || pwn > 8) phn = pwn = (int)TOUCH_GENERIC;
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int a = min(max(win_start(ph0 + k, g.bin_h, g.sh, H, rounding) - h0, -16), 15);
                    const int b = min(max(win_start(pw0 + k, g.bin_w, g.sw, W, rounding) - w0, -16), 15);
                    q.hs |= (unsigned long long)((unsigned)a & 0xffu) << (8 * k);
                    q.ws |= (unsigned long long)((unsigned)b & 0xffu) << (8 * k);
                }
            }
            q.rowmask = (unsigned)rm;
            q.geo = (unsigned)ph0 | ((unsigned)pw0 << 8) | ((unsigned)phn << 16) | ((unsigned)pwn << 20) |
                    (rel << 24);
            list[t] = q;
        }
        __syncthreads();

        // ---- walk the touching RoIs in order; every lane = one channel, so per element the f32
        // additions happen in the reference's order (roi^, ph^, pw^).  The record is wave-uniform.
        for (int i = 0; i < cnt; ++i) {
            const unsigned geo = (unsigned)__builtin_amdgcn_readfirstlane((int)list[i].geo);
            const int r = base + (int)(geo >> 24);
            const int ph0 = geo & 0xff, pw0 = (geo >> 8) & 0xff;
            const int phn = (geo >> 16) & 0xf, pwn = (geo >> 20) & 0xf;
            const size_t rbin0 = (size_t)r * PH * PW;
            if (phn == 0) continue;
            if (phn != (int)TOUCH_GENERIC) {
#define WSSDL_RFL64(v) (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((v) >> 32)) << 32) | \
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(v)))
                const unsigned rowmask = (unsigned)__builtin_amdgcn_readfirstlane((int)list[i].rowmask);
                const unsigned long long colmask = WSSDL_RFL64(list[i].colmask);
                const unsigned long long hs = WSSDL_RFL64(list[i].hs);
                const unsigned long long ws = WSSDL_RFL64(list[i].ws);
#undef WSSDL_RFL64
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<unsigned char *>(arg8 + rbin0 * C), 0, roi_elems, 0x00020000);
                const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(top_diff + rbin0 * C), 0, roi_elems * 4, 0x00020000);
                const int so8 = (ph0 * PW + pw0) * C;      // scalar byte offset of the first bin
#define WSSDL_VISIT(K) visit_roi_c<K, TW, CG, MAXB>(wx, ra, rt, so8, so8 * 4, C, PW, rowmask, phn, colmask, hs, ws)
                switch (pwn) {                                       // wave-uniform
                    case 1: WSSDL_VISIT(1); break;
                    case 2: WSSDL_VISIT(2); break;
                    case 3: WSSDL_VISIT(3); break;
                    case 4: WSSDL_VISIT(4); break;
                    case 5: WSSDL_VISIT(5); break;
                    case 6: WSSDL_VISIT(6); break;
                    case 7: WSSDL_VISIT(7); break;
                    default: WSSDL_VISIT(8); break;
                }
#undef WSSDL_VISIT
            } else {
                // pooled sizes with more than 8 candidate bin rows / columns per tile: one bin at a
                // time, the reference's tests evaluated per lane
                const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
                const int hlo = max(h0, g.sh), hhi = min(h1, g.eh);
                const int wlo = max(w0, g.sw), whi = min(w1, g.ew);
                int pa, pz, qa, qz, t;
                cand_range(hlo - g.sh, g.bin_h, PH, pa, t);
                cand_range(hhi - g.sh, g.bin_h, PH, t, pz);
                cand_range(wlo - g.sw, g.bin_w, PW, qa, t);
                cand_range(whi - g.sw, g.bin_w, PW, t, qz);
                for (int ph = pa; ph < pz; ++ph) {
                    const int bhs = win_start(ph, g.bin_h, g.sh, H, rounding);
                    for (int pw = qa; pw < qz; ++pw) {
                        const size_t bo = (rbin0 + (size_t)(ph * PW + pw)) * C;
                        const unsigned code = (arg8 + bo)[cl];
                        const float tv = (top_diff + bo)[cl];
                        if (code == ARG8_EMPTY || !c_ok) continue;
                        const int h = bhs + (int)(code >> 4);
                        const int w = win_start(pw, g.bin_w, g.sw, W, rounding) + (int)(code & 15u);
                        if (h < hlo || h > hhi || w < wlo || w > whi) continue;
                        int rs, re, cs, ce;
                        cand_range(h - g.sh, g.bin_h, PH, rs, re);
                        cand_range(w - g.sw, g.bin_w, PW, cs, ce);
                        if (ph >= rs && ph < re && pw >= cs && pw < ce) {
                            float *a = &acc[((h - h0) * TW + (w - w0)) * CG + tc];
                            *a = *a + tv;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    if (c_ok) {
        float *img = bottom_diff + (size_t)n * H * W * C;
#pragma unroll
        for (int i = 0; i < TH * TW; ++i) {
            const int h = h0 + i / TW, w = w0 + i % TW;
            if (h < H && w < W) img[((size_t)h * W + w) * C + c] = acc[i * CG + tc];
        }
    }
}

template <int TH, int TW, int CG, int CHUNK, int MAXB = 8, int MINB = 1>
static int launch_bwd_c(const float *top_diff, const unsigned char *arg8, const float *rois, int R, int N,
                        int H, int W, int C, int PH, int PW, float scale, int rounding,
                        float *bottom_diff, hipStream_t st) {
    int tiles_h = cdiv(H, TH), tiles_w = cdiv(W, TW), cgroups = cdiv(C, CG);
    // 8 interleaved queues (one per blockIdx % 8) of ceil(pairs / 8) * tiles workgroups each
    long long blocks = 8LL * cdiv((long long)N * cgroups, 8) * tiles_h * tiles_w;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL((roi_pool_bwd_compact_kernel<TH, TW, CG, CHUNK, MAXB, MINB>), dim3((unsigned)blocks),
                       dim3(CG), 0, st, top_diff, arg8, rois, R, N, H, W, C, PH, PW, scale, rounding,
                       bottom_diff, tiles_h, tiles_w, cgroups);
    return check_launch();
}

static bool compact_supported(int H, int W, int C, int PH, int PW) {
    if (H < 1 || W < 1 || C < 32 || (C % 32) != 0 || C / 32 > 256 || PH < 1 || PW < 1) return false;
    if (PH > 64 || PW > 64) return false;       // (the forward keeps the bin windows in the lanes of a wave)
    // a RoI inside the map spans at most H+1 (W+1) cells after rounding: windows of at most
    // ceil((H+1)/PH) + 1 rows, ceil((W+1)/PW) + 1 columns
    return cdiv(H + 1, PH) + 1 <= ARG8_MAX_WIN_H && cdiv(W + 1, PW) + 1 <= ARG8_MAX_WIN_W;
}

}  // namespace wssdl

using namespace wssdl;

extern "C" int wssdl_roi_pool_compact_supported(int H, int W, int C, int pooled_h, int pooled_w) {
    return compact_supported(H, W, C, pooled_h, pooled_w) ? 1 : 0;
}

// windows table: 7 x 7 bins and a shape the wave-uniform kernel takes with 256-channel waves
static bool window_table_supported(int H, int W, int C, int pooled_h, int pooled_w) {
    if (!compact_supported(H, W, C, pooled_h, pooled_w) || pooled_h != 7 || pooled_w != 7) return false;
    if (C % 256 != 0 || H > 255 || W > 255) return false;
    const int slices = C / 256;
    return ((slices >= 8 && slices % 8 == 0) || (slices < 8 && 8 % slices == 0)) && (long long)H * W * C * 4 < 0x7fffffffLL;
}

extern "C" size_t wssdl_roi_pool_forward_windows_bytes(int R, int H, int W, int C, int pooled_h, int pooled_w) {
    if (R <= 0 || !window_table_supported(H, W, C, pooled_h, pooled_w)) return 0;
    return (size_t)R * 7 * WIN_ENTRY_WORDS * sizeof(unsigned);
}

extern "C" int wssdl_roi_pool_forward_windows(const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                                              int pooled_w, float spatial_scale, int rounding, void *table,
                                              size_t table_bytes, int32_t *overflow, wssdl_stream_t stream) {
    if (R < 0 || N < 1 || !window_table_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (R == 0) return WSSDL_OK;
    if (!rois || !table || (reinterpret_cast<uintptr_t>(table) & 31)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (table_bytes < wssdl_roi_pool_forward_windows_bytes(R, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_WORKSPACE;
    hipLaunchKernelGGL(roi_windows_kernel, dim3(cdiv((long long)R * 7, 256)), dim3(256), 0, as_stream(stream), rois, R, N,
                       H, W, spatial_scale, rounding, static_cast<unsigned *>(table), overflow, nullptr, 0);
    return check_launch();
}

extern "C" int wssdl_roi_pool_forward_windows_blocks(const float *rois, int R, int N, int H, int W, int C, int pooled_h,
                                                     int pooled_w, float spatial_scale, int rounding, void *table,
                                                     size_t table_bytes, int32_t *overflow, void *blocks,
                                                     size_t blocks_bytes, wssdl_stream_t stream) {
    if (R < 1 || N < 1 || !window_table_supported(H, W, C, pooled_h, pooled_w) ||
        !blocks_supported(R, N, H, W, C, pooled_h, pooled_w))
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!rois || !table || (reinterpret_cast<uintptr_t>(table) & 31) || !blocks || (reinterpret_cast<uintptr_t>(blocks) & 255))
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (table_bytes < wssdl_roi_pool_forward_windows_bytes(R, H, W, C, pooled_h, pooled_w) ||
        blocks_bytes < blocks_workspace_bytes(R, N, H, W, C))
        return WSSDL_ERR_WORKSPACE;
    unsigned *zero = nullptr;
    int zero_words = 0;
    blocks_zero_region(blocks, R, N, H, W, C, &zero, &zero_words);
    hipLaunchKernelGGL(roi_windows_kernel, dim3(cdiv((long long)R * 7, 256)), dim3(256), 0, as_stream(stream), rois, R, N,
                       H, W, spatial_scale, rounding, static_cast<unsigned *>(table), overflow, zero, zero_words);
    return check_launch();
}

static int forward_compact(const float *bottom, int N, int H, int W, int C, const float *rois, int R, int pooled_h,
                           int pooled_w, float spatial_scale, int rounding, float *top, uint8_t *argmax8,
                           int32_t *overflow, const unsigned *table, wssdl_stream_t stream);

extern "C" int wssdl_roi_pool_forward_compact(const float *bottom, int N, int H, int W, int C,
                                              const float *rois, int R, int pooled_h, int pooled_w,
                                              float spatial_scale, int rounding, float *top,
                                              uint8_t *argmax8, int32_t *overflow,
                                              wssdl_stream_t stream) {
    return forward_compact(bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, argmax8,
                           overflow, nullptr, stream);
}

extern "C" int wssdl_roi_pool_forward_compact_windows(const float *bottom, int N, int H, int W, int C,
                                                      const float *rois, int R, int pooled_h, int pooled_w,
                                                      float spatial_scale, int rounding, const void *table,
                                                      float *top, uint8_t *argmax8, wssdl_stream_t stream) {
    if (!table || !window_table_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    return forward_compact(bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, argmax8,
                           nullptr, static_cast<const unsigned *>(table), stream);
}

static int forward_compact(const float *bottom, int N, int H, int W, int C, const float *rois, int R, int pooled_h,
                           int pooled_w, float spatial_scale, int rounding, float *top, uint8_t *argmax8,
                           int32_t *overflow, const unsigned *table, wssdl_stream_t stream) {
    if (N < 0 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (R == 0) return WSSDL_OK;
    if (!bottom || !rois || !top || !argmax8 || N < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(bottom) & 15) || (reinterpret_cast<uintptr_t>(top) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax8) & 3))
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    const int lanes_per_row = C / 32;
    const int rows_per_block = 256 / lanes_per_row;
    const long long row_blocks = ((long long)R * pooled_h + rows_per_block - 1) / rows_per_block;
    if (row_blocks * 8 > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    unsigned *a8 = reinterpret_cast<unsigned *>(argmax8);
    int variant = tuning().roi_fwd_variant;
    if (table) variant = 0;
    // wave-uniform kernel, 256 (or 128) channels per wave.  Variants: 0 = automatic, 1 = one bin row per
    // wave with a store per bin, 2 = 128-channel waves, 3 = 7 one-row waves per workgroup, 4 = one bin
    // row per wave, 5 = a whole RoI per wave, 9 = the sliced round-1 form
    // a test-sized RoI list (R = 300: 8400 one-row waves) is latency-bound: 0.087 ms on one-row waves, 0.067 on the sliced
    // kernel (which spreads a bin row over more lanes), 0.056-0.062 on one wave per bin (below) at 63 x 100 x 1024
    const bool small_launch = variant == 0 && !table && (long long)R * pooled_h * cdiv(C, 256) < 32768;
    // Small launches (round 6): one wave per BIN.  A launch of R * PH * C / 256 < 32768 one-row waves leaves SIMDs idle and each
    // wave walks a chain of 7 bins; split seven ways the waves are 7 x as many and a seventh as long.  Same scan per bin, same
    // bits.  R x C sweep against the sliced kernel (profiles/r06_small_forward_one_bin_sweep.log): 0.45-0.86 x the time
    // everywhere below the bound, 0.96-1.02 x at it, 1.3 x beyond (where the rows kernel takes over).  "roi_fwd_one_bin":
    // 7 (default) / 4 = waves per bin row, 0 = the sliced kernel, + 100 = also beyond the bound (experiments).
    const int split = tuning().roi_fwd_one_bin;
    if ((small_launch || (split >= 100 && variant == 0 && !table)) && split > 0 && C % 256 == 0 && (long long)H * W * C * 4 < 0x7fffffffLL) {
        // SPLIT waves per bin row: R * PH * SPLIT * C / 256 waves with ceil(PW / SPLIT) bins each
        const int slices = C / 256;
        if ((slices >= 8 && slices % 8 == 0) || (slices < 8 && 8 % slices == 0)) {
            const int sp = split % 100 >= 7 ? 7 : 4;
            const long long items = (long long)R * pooled_h * sp;
            const long long groups = (items + 3) / 4;
            const long long blocks = slices >= 8 ? groups * slices : 8 * ((groups + 8 / slices - 1) / (8 / slices));
            if (blocks <= 0x7fffffffLL) {
#define WSSDL_FWD_SPLIT(SP) \
    hipLaunchKernelGGL((roi_pool_fwd_rows_kernel<4, 4, 0, false, false, false, SP>), dim3((unsigned)blocks), dim3(256), 0, st, bottom, N, H, \
                       W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, argmax8, overflow, slices, nullptr)
                if (sp == 7) WSSDL_FWD_SPLIT(7); else WSSDL_FWD_SPLIT(4);
#undef WSSDL_FWD_SPLIT
                return check_launch();
            }
        }
    }
    if (variant != 9 && !small_launch) {
        const int cpl = (variant == 2 || C % 256 != 0) ? 2 : 4;
        const int slices = cdiv(C, 64 * cpl);
        const int rpw = (variant == 3 && cpl == 4) ? 7 : 4;
        if (((slices >= 8 && slices % 8 == 0) || (slices < 8 && 8 % slices == 0)) &&
            (long long)H * W * C * 4 < 0x7fffffffLL) {
            // a wave walks a whole RoI when that still gives every SIMD its 8 waves (R * slices >= 8192),
            // else one (roi, ph) bin row; pooled_w == 7 is known at compile time: the stores of a row are
            // issued together
            const bool many = (long long)R * slices >= 8192;
            const bool whole = pooled_w == 7 && variant == 5 && many;      // (measured slower: 0.80 against 0.57 ms)
            const long long items = whole ? (long long)R : (long long)R * pooled_h;
            const long long groups = (items + rpw - 1) / rpw;
            long long blocks = slices >= 8 ? groups * slices : 8 * ((groups + 8 / slices - 1) / (8 / slices));
            if (blocks <= 0x7fffffffLL) {
#define WSSDL_FWD_ROWS(CPL, RPW, PWS, WHOLE) \
    hipLaunchKernelGGL((roi_pool_fwd_rows_kernel<CPL, RPW, PWS, WHOLE, false>), dim3((unsigned)blocks), dim3(64 * RPW), 0, st, \
                       bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, argmax8, overflow, \
                       slices, nullptr)
#define WSSDL_FWD_TAB(CPL) \
    hipLaunchKernelGGL((roi_pool_fwd_rows_kernel<CPL, 4, 7, false, true>), dim3((unsigned)blocks), dim3(64 * 4), 0, st, \
                       bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, argmax8, overflow, \
                       slices, table)
                if (table && cpl == 4 && rpw == 4 && !whole) WSSDL_FWD_TAB(4);
                else if (table) return WSSDL_ERR_INVALID_ARGUMENT;
                else if (variant == 3 && cpl == 4) WSSDL_FWD_ROWS(4, 7, 0, false);
                else if (cpl == 2) { if (whole) WSSDL_FWD_ROWS(2, 4, 7, true); else WSSDL_FWD_ROWS(2, 4, 0, false); }
                else if (whole) WSSDL_FWD_ROWS(4, 4, 7, true);
                else if ((variant == 0 || variant == 4) && pooled_w == 7) WSSDL_FWD_ROWS(4, 4, 7, false);
                else WSSDL_FWD_ROWS(4, 4, 0, false);
#undef WSSDL_FWD_ROWS
#undef WSSDL_FWD_TAB
                return check_launch();
            }
        }
    }
    if (row_blocks * 8 <= 4096)      // less than ~4 workgroups per CU: latency-bound, fetch 2 cells at a time
        hipLaunchKernelGGL(roi_pool_fwd_compact_kernel<2>, dim3((unsigned)(row_blocks * 8)), dim3(256), 0, st,
                           bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, a8,
                           overflow, lanes_per_row, rows_per_block);
    else
        hipLaunchKernelGGL(roi_pool_fwd_compact_kernel<1>, dim3((unsigned)(row_blocks * 8)), dim3(256), 0, st,
                           bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, a8,
                           overflow, lanes_per_row, rows_per_block);
    return check_launch();
}

// The wave-uniform forward with the reference op's i32 arg-max (wssdl_roi_pool_forward).  Returns
// WSSDL_ROWS_I32_UNSUPPORTED when the shape is not one the kernel takes (the caller then runs the sliced kernel).
int wssdl::launch_fwd_rows_i32(const float *bottom, int N, int H, int W, int C, const float *rois, int R, int pooled_h,
                        int pooled_w, float spatial_scale, int rounding, float *top, int32_t *argmax, hipStream_t st) {
    if (C % 128 != 0 || pooled_h > 64 || pooled_w > 64 || (long long)H * W * C * 4 >= 0x7fffffffLL)
        return WSSDL_ROWS_I32_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(bottom) & 15) || (reinterpret_cast<uintptr_t>(top) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax) & 15))
        return WSSDL_ROWS_I32_UNSUPPORTED;
    // (a test-sized RoI list: one wave per bin, as on the 1-byte path; the sliced kernel where that form does not apply)
    if ((long long)R * pooled_h * cdiv(C, 256) < 32768) {
        const int sl = C / 256;
        if (tuning().roi_fwd_one_bin <= 0 || C % 256 != 0 || !((sl >= 8 && sl % 8 == 0) || (sl < 8 && 8 % sl == 0)))
            return WSSDL_ROWS_I32_UNSUPPORTED;
        const long long groups7 = ((long long)R * pooled_h * 7 + 3) / 4;
        const long long blocks7 = sl >= 8 ? groups7 * sl : 8 * ((groups7 + 8 / sl - 1) / (8 / sl));
        if (blocks7 > 0x7fffffffLL) return WSSDL_ROWS_I32_UNSUPPORTED;
        hipLaunchKernelGGL((roi_pool_fwd_rows_kernel<4, 4, 0, false, false, true, 7>), dim3((unsigned)blocks7), dim3(256), 0, st, bottom, N,
                           H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, reinterpret_cast<unsigned char *>(argmax),
                           nullptr, sl, nullptr);
        return check_launch();
    }
    const int cpl = (C % 256 != 0) ? 2 : 4;
    const int slices = cdiv(C, 64 * cpl);
    if (!((slices >= 8 && slices % 8 == 0) || (slices < 8 && 8 % slices == 0))) return WSSDL_ROWS_I32_UNSUPPORTED;
    const int rpw = 4;
    const long long items = (long long)R * pooled_h;
    const long long groups = (items + rpw - 1) / rpw;
    const long long blocks = slices >= 8 ? groups * slices : 8 * ((groups + 8 / slices - 1) / (8 / slices));
    if (blocks > 0x7fffffffLL) return WSSDL_ROWS_I32_UNSUPPORTED;
    unsigned char *a = reinterpret_cast<unsigned char *>(argmax);
#define WSSDL_FWD_I32(CPL, PWS) \
    hipLaunchKernelGGL((roi_pool_fwd_rows_kernel<CPL, 4, PWS, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, \
                       bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top, a, nullptr, slices, \
                       nullptr)
    if (cpl == 4) { if (pooled_w == 7) WSSDL_FWD_I32(4, 7); else WSSDL_FWD_I32(4, 0); }
    else { if (pooled_w == 7) WSSDL_FWD_I32(2, 7); else WSSDL_FWD_I32(2, 0); }
#undef WSSDL_FWD_I32
    return check_launch();
}

extern "C" int wssdl_roi_argmax_expand(const uint8_t *argmax8, const float *rois, int R, int H, int W,
                                       int C, int pooled_h, int pooled_w, float spatial_scale,
                                       int rounding, int32_t *argmax, wssdl_stream_t stream) {
    if (R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if ((long long)H * W * C > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    if (R == 0) return WSSDL_OK;
    if (!argmax8 || !rois || !argmax) return WSSDL_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(argmax8) & 3) || (reinterpret_cast<uintptr_t>(argmax) & 15))
        return WSSDL_ERR_INVALID_ARGUMENT;
    const long long total4 = (long long)R * pooled_h * pooled_w * (C / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > (1LL << 20)) blocks = 1LL << 20;
    hipLaunchKernelGGL(roi_argmax_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const unsigned *>(argmax8), rois, total4, H, W, C, pooled_h, pooled_w,
                       spatial_scale, rounding, reinterpret_cast<int4v *>(argmax));
    return check_launch();
}

extern "C" size_t wssdl_roi_pool_backward_workspace_bytes(int R, int N, int H, int W, int pooled_h,
                                                          int pooled_w) {
    if (R < 0 || N < 1 || H < 1 || W < 1 || pooled_h < 1 || pooled_w < 1 || pooled_h > 8 || pooled_w > 8) return 0;
    return walk_workspace_bytes(R, N, H, W, pooled_h, pooled_w);
}

extern "C" int wssdl_roi_pool_backward_plan_count(void) { return walk_plan_count(); }

extern "C" size_t wssdl_roi_pool_backward_status_offset(int R, int N, int H, int W, int pooled_h, int pooled_w) {
    return walk_flags_offset(R, N, H, W, pooled_h, pooled_w);
}

extern "C" int wssdl_roi_pool_backward_prepare(const float *rois, int R, int N, int H, int W, int C,
                                               int pooled_h, int pooled_w, float spatial_scale,
                                               int rounding, void *workspace, size_t workspace_bytes,
                                               int32_t *plan_host, wssdl_stream_t stream) {
    if (!plan_host) return WSSDL_ERR_INVALID_ARGUMENT;
    *plan_host = -1;
    if (N < 1 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (!workspace || !walk_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_OK;   // plan -1: fallback kernel
    if (R > 0 && !rois) return WSSDL_ERR_INVALID_ARGUMENT;
    int plan = -1;
    const int rc = walk_prepare(rois, R, N, H, W, C, pooled_h, pooled_w, spatial_scale, rounding, workspace,
                                workspace_bytes, &plan, as_stream(stream));
    if (rc == WSSDL_OK) *plan_host = plan;
    return rc;
}

extern "C" int wssdl_roi_pool_backward_split_segments(int R, int N, int H, int W, int C) {
    return walk_split_segments(R, N, H, W, C);
}

extern "C" int wssdl_roi_pool_backward_split_plan(void) { return walk_split_plan(); }

extern "C" size_t wssdl_roi_pool_backward_split_scratch_bytes(int N, int H, int W, int C, int segments) {
    if (segments <= 1 || N < 1 || H < 1 || W < 1 || C < 1) return 0;
    return (size_t)(segments - 1) * (size_t)N * H * W * C * sizeof(float);
}

extern "C" int wssdl_roi_pool_backward_compact_split(const float *top_diff, const uint8_t *argmax8, const float *rois,
                                                     int R, int N, int H, int W, int C, int pooled_h, int pooled_w,
                                                     float spatial_scale, int rounding, float *bottom_diff,
                                                     void *workspace, size_t workspace_bytes, int plan, int segments,
                                                     void *scratch, size_t scratch_bytes, wssdl_stream_t stream) {
    if (N < 0 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax8 || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    if (plan < 0 || !workspace || !walk_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (segments < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (segments > 1 && (!scratch || scratch_bytes < wssdl_roi_pool_backward_split_scratch_bytes(N, H, W, C, segments)))
        return WSSDL_ERR_WORKSPACE;
    return launch_walk(top_diff, argmax8, R, N, H, W, C, pooled_h, pooled_w, bottom_diff, workspace, workspace_bytes, plan,
                       as_stream(stream), segments, static_cast<float *>(scratch));
}

extern "C" int wssdl_roi_pool_backward_owner_plan_count(void) { return owner_plan_count(); }

extern "C" int wssdl_roi_pool_backward_owner_plan(int R, int N, int H, int W, int C) {
    return owner_plan_auto(R, N, H, W, C);
}

extern "C" int wssdl_roi_pool_backward_owner_plan_for(int R, int N, int H, int W, int C, int pooled_h, int pooled_w) {
    if (R < 0 || N < 1 || !compact_supported(H, W, C, pooled_h, pooled_w) || !owner_supported(R, N, H, W, C, pooled_h, pooled_w))
        return -1;
    return owner_plan_auto(R, N, H, W, C);
}

extern "C" size_t wssdl_roi_pool_backward_owner_scratch_bytes(int N, int H, int W, int C, int owner_plan) {
    return owner_scratch_bytes(N, H, W, C, owner_plan);
}

extern "C" int wssdl_roi_pool_backward_owner_prepare(const float *rois, int R, int N, int H, int W, int C,
                                                     int pooled_h, int pooled_w, float spatial_scale, int rounding,
                                                     void *workspace, size_t workspace_bytes, int owner_plan,
                                                     wssdl_stream_t stream) {
    if (N < 1 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!workspace || !owner_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (R > 0 && !rois) return WSSDL_ERR_INVALID_ARGUMENT;
    return owner_prepare(rois, R, N, H, W, C, pooled_h, pooled_w, spatial_scale, rounding, workspace, workspace_bytes,
                         owner_plan, as_stream(stream));
}

extern "C" int wssdl_roi_pool_backward_compact_owner(const float *top_diff, const uint8_t *argmax8, const float *rois,
                                                     int R, int N, int H, int W, int C, int pooled_h, int pooled_w,
                                                     float spatial_scale, int rounding, float *bottom_diff,
                                                     void *workspace, size_t workspace_bytes, int owner_plan,
                                                     void *scratch, size_t scratch_bytes, wssdl_stream_t stream) {
    if (N < 0 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax8 || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!workspace || !owner_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    const size_t need = owner_scratch_bytes(N, H, W, C, owner_plan);
    if (need == 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!scratch || scratch_bytes < need || (reinterpret_cast<uintptr_t>(scratch) & 15) ||
        (reinterpret_cast<uintptr_t>(bottom_diff) & 15))
        return WSSDL_ERR_WORKSPACE;
    return launch_owner(top_diff, argmax8, R, N, H, W, C, pooled_h, pooled_w, bottom_diff, workspace, workspace_bytes,
                        owner_plan, static_cast<float *>(scratch), as_stream(stream));
}

extern "C" int wssdl_roi_pool_backward_owner_segments(int R, int N, int H, int W, int C) {
    return owner_split_segments(R, N, H, W, C);
}

extern "C" size_t wssdl_roi_pool_backward_owner_split_scratch_bytes(int N, int H, int W, int C, int owner_plan, int segments) {
    return owner_scratch_bytes(N, H, W, C, owner_plan, segments);
}

extern "C" int wssdl_roi_pool_backward_compact_owner_split(const float *top_diff, const uint8_t *argmax8, const float *rois,
                                                           int R, int N, int H, int W, int C, int pooled_h, int pooled_w,
                                                           float spatial_scale, int rounding, float *bottom_diff,
                                                           void *workspace, size_t workspace_bytes, int owner_plan,
                                                           int segments, void *scratch, size_t scratch_bytes,
                                                           wssdl_stream_t stream) {
    if (N < 0 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax8 || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!workspace || !owner_supported(R, N, H, W, C, pooled_h, pooled_w) || segments < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    const size_t need = owner_scratch_bytes(N, H, W, C, owner_plan, segments);
    if (need == 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!scratch || scratch_bytes < need || (reinterpret_cast<uintptr_t>(scratch) & 15) ||
        (reinterpret_cast<uintptr_t>(bottom_diff) & 15))
        return WSSDL_ERR_WORKSPACE;
    return launch_owner(top_diff, argmax8, R, N, H, W, C, pooled_h, pooled_w, bottom_diff, workspace, workspace_bytes,
                        owner_plan, static_cast<float *>(scratch), as_stream(stream), false, segments);
}

extern "C" int wssdl_roi_pool_backward_owner_i32(const float *top_diff, const int32_t *argmax, const float *rois, int R, int N,
                                                 int H, int W, int C, int pooled_h, int pooled_w, float spatial_scale,
                                                 float *bottom_diff, void *workspace, size_t workspace_bytes, int owner_plan,
                                                 void *scratch, size_t scratch_bytes, wssdl_stream_t stream) {
    if (N < 0 || R < 0 || H < 1 || W < 1 || C < 1 || pooled_h < 1 || pooled_w < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!workspace || !owner_supported(R, N, H, W, C, pooled_h, pooled_w) || !walk_i32_supported(R, N, H, W, C, pooled_h, pooled_w))
        return WSSDL_ERR_INVALID_ARGUMENT;
    const size_t need = owner_scratch_bytes(N, H, W, C, owner_plan);
    if (need == 0) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!scratch || scratch_bytes < need || (reinterpret_cast<uintptr_t>(scratch) & 15) ||
        (reinterpret_cast<uintptr_t>(bottom_diff) & 15) || (reinterpret_cast<uintptr_t>(argmax) & 7) ||
        (reinterpret_cast<uintptr_t>(top_diff) & 7))
        return WSSDL_ERR_WORKSPACE;
    // (the lists carry window starts this path does not read: the rounding mode does not matter)
    const int rc = owner_prepare(rois, R, N, H, W, C, pooled_h, pooled_w, spatial_scale, WSSDL_ROI_ROUND_CUDA, workspace,
                                 workspace_bytes, owner_plan, as_stream(stream));
    if (rc != WSSDL_OK) return rc;
    return launch_owner(top_diff, reinterpret_cast<const unsigned char *>(argmax), R, N, H, W, C, pooled_h, pooled_w,
                        bottom_diff, workspace, workspace_bytes, owner_plan, static_cast<float *>(scratch), as_stream(stream),
                        true);
}

extern "C" int wssdl_roi_pool_backward_compact(const float *top_diff, const uint8_t *argmax8,
                                               const float *rois, int R, int N, int H, int W, int C,
                                               int pooled_h, int pooled_w, float spatial_scale,
                                               int rounding, float *bottom_diff, void *workspace,
                                               size_t workspace_bytes, int plan, wssdl_stream_t stream) {
    if (N < 0 || R < 0 || !compact_supported(H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax8 || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    if (plan >= 0) {          // lists prepared by wssdl_roi_pool_backward_prepare: the walk kernel alone
        if (!workspace || !walk_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
        return launch_walk(top_diff, argmax8, R, N, H, W, C, pooled_h, pooled_w, bottom_diff, workspace,
                           workspace_bytes, plan, st);
    }
    const int variant = tuning().roi_bwdc_variant;      // tuning: fallback kernel shapes
    // fallback: tile-owner kernel with the RoI filter inside (no workspace; any pooled size)
    // channels per workgroup: 256 when that still yields enough workgroups to fill the chip
    int cg = C > 128 ? 256 : (C > 64 ? 128 : 64);
    const long long tiles = (long long)cdiv(H, 4) * cdiv(W, 8);       // counted in 4x8 tiles
    while (cg > 64 && (long long)N * cdiv(C, cg) * tiles < BWD_MIN_WORKGROUPS) cg >>= 1;
    {       // tuning override
        const int v = tuning().roi_bwd_cg;
        if (v == 64 || v == 128 || v == 256) cg = v;
    }
#define WSSDL_BWDC(TH, TW, CGV, CHUNK, MAXB, MINB) \
    launch_bwd_c<TH, TW, CGV, CHUNK, MAXB, MINB>(top_diff, argmax8, rois, R, N, H, W, C, pooled_h, pooled_w, \
                                                 spatial_scale, rounding, bottom_diff, st)
    if (cg == 256) {
        switch (variant) {
            case 1: return WSSDL_BWDC(4, 4, 256, 218, 8, 7);     // 7 workgroups / CU, longer filter rounds
            case 2: return WSSDL_BWDC(4, 8, 256, 240, 8, 4);     // 4x8 tiles: fewer border re-reads, 4-5 wg / CU
            case 3: return WSSDL_BWDC(4, 8, 256, 240, 16, 4);
            case 4: return WSSDL_BWDC(4, 4, 256, 126, 12, 6);
            case 5: return WSSDL_BWDC(4, 4, 256, 126, 8, 8);     // 8 workgroups / CU (20 KiB of LDS each)
            default: return WSSDL_BWDC(4, 4, 256, 218, 8, 7);
        }
    }
    if (cg == 128) return WSSDL_BWDC(4, 8, 128, 128, 8, 1);
    return WSSDL_BWDC(4, 8, 64, 64, 8, 1);
#undef WSSDL_BWDC
}
