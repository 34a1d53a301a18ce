// RoI max-pooling forward / backward for gfx950 (MI355X).
//
// Reference: code/lib/roi_pooling_layer/roi_pooling_op_gpu.cu.cc:20-85 (forward,
// canonical rounding), roi_pooling_op.cc:137-196 (forward, CPU rounding),
// roi_pooling_op_gpu.cu.cc:114-190 == roi_pooling_op.cc:383-458 (backward).
//
// Forward  : HBM-write bound.  A lane owns 4 consecutive channels (16-byte loads of the
//            NHWC feature map, 16-byte non-temporal stores of top + argmax) of one
//            (roi, ph) bin row and walks its PW bins.  Workgroup b serves channel
//            slice b % 8, i.e. (observed round-robin placement) one XCD whose L2
//            then only holds its slice of the feature map.
// Backward : the reference gathers per bottom element over ALL RoIs
//            (O(N*H*W*C*R)).  Here each workgroup owns a 4x8-cell x 256-channel
//            tile of bottom_diff in LDS (32 KiB), one lane per channel.  Filter
//            phase: the RoIs of its image that touch the tile are compacted in
//            RoI order and the reference's candidate-bin formulas for the
//            tile's rows / columns are evaluated once per (RoI, tile) into two
//            64-bit masks.  Walk phase: per RoI (scalar registers) the
//            candidate bins are visited in (ph, pw) order, two bin rows at a
//            time, straight-line code specialised on the column count; loads
//            are buffer loads (scalar descriptor + scalar bin offset + lane
//            offset); a lane decodes its argmax, tests two mask bits and adds
//            top_diff into its LDS cell.  A lane is the only writer of its
//            channel, so per element the f32 additions happen in exactly the
//            reference's order (roi^, ph^, pw^): bit-identical, no atomics, no
//            pre-zeroing, every bottom_diff element written once, coalesced.
#include "roi_pool.hip.h"

#include <stdlib.h>

// cache policy of the backward's streaming loads: 0 = default.  nt (2) was measured 5 % slower:
// the bins re-read by neighbouring tiles profit from staying in L2.
#define WSSDL_LD_AUX 0


namespace wssdl {

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

// Forward, XCD-sliced: workgroup b serves channel slice b % 8 (C/8 channels) -- under the
// observed round-robin placement that is one XCD, whose 4 MiB L2 then only ever holds its
// slice of the feature map (1.2 MB per 38x63x1024 image instead of 9.8 MB).  A lane owns 4
// channels of one (roi, ph) bin row and walks its PW bins, so the RoI geometry is computed
// once per PW outputs.  Requires C % 32 == 0.  Placement affects speed only.
// BATCH cells of a window row fetched together: 2 helps small launches (latency-bound: 0.088 ->
// 0.072 ms at R = 300), 1 is best once the chip is full (the duplicated loads cost throughput).
template <int BATCH>
__global__ __launch_bounds__(256) void roi_pool_fwd_sliced_kernel(
    const float *__restrict__ bottom, int N, int H, int W, int C, const float *__restrict__ rois,
    int R, int PH, int PW, float scale, int rounding, float *__restrict__ top,
    int *__restrict__ argmax, int lanes_per_row /* = C/32 */, int rows_per_block) {
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
    int hs, he;
    if (rounding == WSSDL_ROI_ROUND_CPU) {
        hs = (int)((float)ph * g.bin_h);
        he = (int)((float)(ph + 1) * g.bin_h);
    } else {
        hs = (int)floorf((float)ph * g.bin_h);
        he = (int)ceilf((float)(ph + 1) * g.bin_h);
    }
    hs = min(max(hs + g.sh, 0), H);
    he = min(max(he + g.sh, 0), H);
    size_t o = ((size_t)row * PW) * C + c0;
    for (int pw = 0; pw < PW; ++pw, o += C) {
        int ws, we;
        if (rounding == WSSDL_ROI_ROUND_CPU) {
            ws = (int)((float)pw * g.bin_w);
            we = (int)((float)(pw + 1) * g.bin_w);
        } else {
            ws = (int)floorf((float)pw * g.bin_w);
            we = (int)ceilf((float)(pw + 1) * g.bin_w);
        }
        ws = min(max(ws + g.sw, 0), W);
        we = min(max(we + g.sw, 0), W);
        const bool empty = (he <= hs) || (we <= ws) || bad;
        float4v mv = empty ? (float4v)(0.0f) : (float4v)(-FLT_MAX);
        int4v mi = (int4v)(-1);
        if (!empty) {
            // BATCH cells of a window row are fetched together (independent loads in flight
            // instead of one dependent load per compare).  Slots past the window's last
            // column re-read that column: an equal value never passes the strict >, so the scan
            // order and the first-maximum rule are untouched.
            for (int h = hs; h < he; ++h) {
                const int row_base = h * W * C + c0;
                for (int w = ws; w < we; w += BATCH) {
                    float4v v[BATCH];
                    int base[BATCH];
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) {
                        base[j] = row_base + min(w + j, we - 1) * C;
                        v[j] = *reinterpret_cast<const float4v *>(img + base[j]);
                    }
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) {
                        if (v[j].x > mv.x) { mv.x = v[j].x; mi.x = base[j]; }
                        if (v[j].y > mv.y) { mv.y = v[j].y; mi.y = base[j] + 1; }
                        if (v[j].z > mv.z) { mv.z = v[j].z; mi.z = base[j] + 2; }
                        if (v[j].w > mv.w) { mv.w = v[j].w; mi.w = base[j] + 3; }
                    }
                }
            }
        }
        __builtin_nontemporal_store(mv, reinterpret_cast<float4v *>(top + o));
        __builtin_nontemporal_store(mi, reinterpret_cast<int4v *>(argmax + o));
    }
}

// ----------------------------------------------------------------- backward ---
struct FastDiv {          // n / d = (n * magic) >> shift with one full-rate v_mul_u32_u24
    unsigned magic;       // < 2^24
    int shift;
};

// One (RoI, tile) intersection, produced by the filter phase (16 B in LDS).
//   geo     = ph0 | pw0 << 8 | phn << 16 | pwn << 20 | (r - chunk base) << 24: the candidate bins of
//             the tile's cells are [ph0, ph0+phn) x [pw0, pw0+pwn)  (phstart/phend are monotone in
//             h, so the union over the tile's rows is one interval; same for columns)
//   rowmask : bit 4*k + j  <=>  tile row h0+j lies in the RoI (in_roi) and bin row ph0+k is one of
//             its candidate rows (tiles are at most 4 rows high); colmask: bit 8*k + j likewise
//             for columns.  phn or pwn > 8 (possible only for pooled sizes > 8): GENERIC
//             (phn = pwn = 15), the masks are unused.
struct TouchRec {
    unsigned geo;
    unsigned rowmask;
    unsigned long long colmask;
};

// Per-wave constants of the walk (kept in registers across RoIs)
struct WalkCtx {
    float *acc;            // LDS tile [TH*TW][CG]
    int tc, h0, w0, W, C, cm, cshift, cmask;
    FastDiv divw;
    int voff;              // lane's byte offset inside a bin
};

// Visit NROWS (1 or 2) candidate bin rows x PWN candidate bin columns of one RoI: straight-line
// code, 2*NROWS*PWN buffer loads issued back to back (scalar descriptor + scalar bin offset +
// lane offset), then the in-order accumulation.  PWN and NROWS are compile-time so that no
// slot needs a predicate; the caller switches on the (wave-uniform) column count.
template <int PWN, int NROWS, int TW, int CG, bool FAST, int FWR = 8>
__device__ __forceinline__ void visit_rows(const WalkCtx &x, __amdgpu_buffer_rsrc_t ra,
                                           __amdgpu_buffer_rsrc_t rt, int so_row, int bin_bytes,
                                           int row_bytes, unsigned long long rowmask, int row0,
                                           unsigned long long colmask) {
    int idx[NROWS][PWN];
    float td[NROWS][PWN];
#pragma unroll
    for (int q = 0; q < NROWS; ++q)
#pragma unroll
        for (int j = 0; j < PWN; ++j) {
            const int so = so_row + q * row_bytes + j * bin_bytes;
            idx[q][j] = (int)__builtin_amdgcn_raw_buffer_load_b32(ra, x.voff, so, WSSDL_LD_AUX);
            td[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, x.voff, so, WSSDL_LD_AUX));
        }
    // keep every loaded value live here: otherwise the compiler sinks the top_diff loads into
    // the (rare) hit branch and serialises them
#pragma unroll
    for (int q = 0; q < NROWS; ++q)
#pragma unroll
        for (int j = 0; j < PWN; ++j) asm volatile("" : "+v"(idx[q][j]), "+v"(td[q][j]));
#pragma unroll
    for (int q = 0; q < NROWS; ++q) {
        const unsigned rm = (unsigned)(rowmask >> (FWR * (row0 + q))) & ((1u << FWR) - 1u);
#pragma unroll
        for (int j = 0; j < PWN; ++j) {
            const unsigned cmk = (unsigned)(colmask >> (8 * j)) & 0xffu;
            const int id = idx[q][j];
            int cell, cc, h, w;
            if (FAST) {
                cell = id >> x.cshift;
                cc = id & x.cmask;
                h = (int)(__umul24((unsigned)cell, x.divw.magic) >> x.divw.shift);
                w = cell - (int)__umul24((unsigned)h, (unsigned)x.W);
            } else {
                cell = id / x.C;
                cc = id - cell * x.C;
                h = cell / x.W;
                w = cell - h * x.W;
            }
            const unsigned dh = (unsigned)(h - x.h0), dw = (unsigned)(w - x.w0);
            // tile, in_roi and candidate-bin tests: two mask look-ups, no branches
            const unsigned bits = (rm >> (dh & 7u)) & (cmk >> (dw & 7u)) & 1u;
            const bool ok = (bits != 0u) & ((dh | dw) < 8u) & (id >= 0) & (cc == x.cm);
            if (ok) {
                float *a = &x.acc[(dh * TW + dw) * CG + x.tc];
                *a = *a + td[q][j];
            }
        }
    }
}

template <int PWN, int TW, int CG, bool FAST, int MAXB, int FWR = 8>
__device__ __forceinline__ void visit_roi(const WalkCtx &x, __amdgpu_buffer_rsrc_t ra,
                                          __amdgpu_buffer_rsrc_t rt, int so_row, int bin_bytes,
                                          int row_bytes, unsigned long long rowmask, int phn,
                                          unsigned long long colmask) {
    // NR bin rows per batch: at most MAXB (argmax, top_diff) load pairs in flight per wave
    constexpr int NR = (MAXB / PWN) >= 4 ? 4 : ((MAXB / PWN) >= 3 ? 3 : 2);
    int rb = 0;
    for (; rb + NR <= phn; rb += NR, so_row += NR * row_bytes)
        visit_rows<PWN, NR, TW, CG, FAST, FWR>(x, ra, rt, so_row, bin_bytes, row_bytes, rowmask, rb, colmask);
    const int rem = phn - rb;
    if (NR > 3 && rem == 3)
        visit_rows<PWN, 3, TW, CG, FAST, FWR>(x, ra, rt, so_row, bin_bytes, row_bytes, rowmask, rb, colmask);
    else if (NR > 2 && rem == 2)
        visit_rows<PWN, 2, TW, CG, FAST, FWR>(x, ra, rt, so_row, bin_bytes, row_bytes, rowmask, rb, colmask);
    else if (rem == 1)
        visit_rows<PWN, 1, TW, CG, FAST, FWR>(x, ra, rt, so_row, bin_bytes, row_bytes, rowmask, rb, colmask);
}

// FAST: C is a power of two (idx -> cell by shift) and cell / W fits a 24-bit multiply.
template <int TH, int TW, int CG, int CHUNK, bool FAST, int MAXB, int MINB>
__global__ __launch_bounds__(CG, MINB) void roi_pool_bwd_kernel(
    const float *__restrict__ top_diff, const int *__restrict__ argmax,
    const float *__restrict__ rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
    float *__restrict__ bottom_diff, int tiles_h, int tiles_w, int cgroups, int cshift,
    FastDiv divw) {
    static_assert(TH <= 4 && TW <= 8, "4 mask bits per candidate bin row, 8 per column");
    static_assert(CHUNK <= 256, "8-bit RoI index inside a filter round");
    static_assert(CHUNK % CG == 0 || CHUNK < CG, "whole filter rounds");
    constexpr int KPT = CHUNK >= CG ? CHUNK / CG : 1;   // RoIs tested per thread per filter round
    constexpr int NW = CG / WSSDL_WAVE;
    __shared__ float acc[TH * TW * CG];
    __shared__ TouchRec list[CHUNK];
    __shared__ int wave_cnt[KPT][NW];
    __shared__ int roi_span[2];

    // Workgroup -> (image, channel group, tile).  Workgroups are dealt round-robin over the 8
    // XCDs (blockIdx % 8; observed, used for speed only): all tiles of one (image, channel
    // group) pair are given the same blockIdx % 8, so the bins that straddle tile borders --
    // read by 2-4 neighbouring tiles walking the same RoI list -- are served by one L2.
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
    const int cmask = (1 << cshift) - 1;
    const int cl = c_ok ? c : C - 1;       // lanes past C read channel C-1 ...
    const int cm = c_ok ? c : -1;          // ... and never match
    // Loads go through buffer descriptors (one per RoI, rebuilt from scalars): the lane only
    // contributes this byte offset, the bin offset travels in the scalar soffset operand, so
    // visiting a bin costs no vector address arithmetic.
    const int roi_bytes = PH * PW * C * 4;
    WalkCtx wx;
    wx.acc = acc;  wx.tc = tc;  wx.h0 = h0;  wx.w0 = w0;  wx.W = W;  wx.C = C;  wx.cm = cm;
    wx.cshift = cshift;  wx.cmask = cmask;  wx.divw = divw;  wx.voff = cl * 4;

#pragma unroll
    for (int i = 0; i < TH * TW; ++i) acc[i * CG + tc] = 0.0f;

    // ---- the span of RoI indices that belong to image n: the filter rounds below only scan
    // that (RoIs normally arrive grouped by image; any order stays correct)
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
        // ---- filter: RoIs of image n whose rounded box touches the tile, in RoI order.
        // The thread that tests a RoI also evaluates the reference's candidate-bin formulas
        // for the tile's rows and columns once, so the walk below only tests bit masks.
        bool hit[KPT];
        RoiGeom gk[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int r = base + k * CG + tc;
            hit[k] = false;
            if (r < r_end && k * CG + tc < CHUNK) {
                gk[k] = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
                const RoiGeom &g = gk[k];
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
        // finish the records with the hits packed into the first threads (no divergence
        // between hit and non-hit lanes): candidate ranges of the tile's rows / columns
        for (int t = tc; t < cnt; t += CG) {
            TouchRec q;
            const unsigned rel = list[t].geo;
            const RoiGeom g = roi_geometry(rois + (size_t)(base + (int)rel) * 5, scale, PH, PW);
            int ph0, phn, pw0, pwn;
            unsigned long long rm;
            touch_axis<TH, 4>(h0, h1, g.sh, g.eh, g.bin_h, PH, ph0, phn, rm);
            touch_axis<TW, 8>(w0, w1, g.sw, g.ew, g.bin_w, PW, pw0, pwn, q.colmask);
            if (phn <= 0 || pwn <= 0) phn = pwn = 0;                          // nothing to visit
            else if (phn > 8 || pwn > 8) phn = pwn = (int)TOUCH_GENERIC;
            q.rowmask = (unsigned)rm;
            q.geo = (unsigned)ph0 | ((unsigned)pw0 << 8) | ((unsigned)phn << 16) | ((unsigned)pwn << 20) |
                    (rel << 24);
            list[t] = q;
        }
        __syncthreads();

        // ---- walk the touching RoIs in order; every lane = one channel, so per element the
        // f32 additions happen in the reference's order (roi^, ph^, pw^).  The record is
        // wave-uniform (scalar registers); bins are visited BB at a time: 2*BB loads from a
        // scalar base + lane offset are issued back to back, then accumulated in order.
        for (int i = 0; i < cnt; ++i) {
            const unsigned geo = (unsigned)__builtin_amdgcn_readfirstlane((int)list[i].geo);
            const int r = base + (int)(geo >> 24);
            const int ph0 = geo & 0xff, pw0 = (geo >> 8) & 0xff;
            const int phn = (geo >> 16) & 0xf, pwn = (geo >> 20) & 0xf;
            const size_t rbin0 = (size_t)r * PH * PW;
            if (phn == 0) continue;
            if (phn != (int)TOUCH_GENERIC) {
                const unsigned long long rowmask =
                    (unsigned)__builtin_amdgcn_readfirstlane((int)list[i].rowmask);
                const unsigned long long colmask =
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(list[i].colmask >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)list[i].colmask);
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<int *>(argmax + rbin0 * C), 0, roi_bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(top_diff + rbin0 * C), 0, roi_bytes, 0x00020000);
                const int bin_bytes = C * 4, row_bytes = PW * C * 4;
                const int so_row = (ph0 * PW + pw0) * bin_bytes;     // scalar byte offset of the first bin
#define WSSDL_VISIT(K) visit_roi<K, TW, CG, FAST, MAXB, 4>(wx, ra, rt, so_row, bin_bytes, row_bytes, rowmask, phn, colmask)
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
                // pooled sizes with more than 8 candidate bin rows / columns per tile: one bin
                // at a time, the reference's tests evaluated per lane
                const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
                const int hlo = max(h0, g.sh), hhi = min(h1, g.eh);
                const int wlo = max(w0, g.sw), whi = min(w1, g.ew);
                int pa, pz, qa, qz, t;
                cand_range(hlo - g.sh, g.bin_h, PH, pa, t);
                cand_range(hhi - g.sh, g.bin_h, PH, t, pz);
                cand_range(wlo - g.sw, g.bin_w, PW, qa, t);
                cand_range(whi - g.sw, g.bin_w, PW, t, qz);
                for (int ph = pa; ph < pz; ++ph) {
                    for (int pw = qa; pw < qz; ++pw) {
                        const size_t bo = (rbin0 + (size_t)(ph * PW + pw)) * C;
                        const int id = (argmax + bo)[cl];
                        const float tv = (top_diff + bo)[cl];
                        if (id < 0) continue;
                        const int cell = id / C, cc = id - cell * C;
                        const int h = cell / W, w = cell - h * W;
                        if (cc != cm || h < hlo || h > hhi || w < wlo || w > whi) continue;
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
static int launch_bwd(const float *top_diff, const int *argmax, const float *rois, int R, int N,
                      int H, int W, int C, int PH, int PW, float scale, float *bottom_diff,
                      hipStream_t st) {
    int tiles_h = cdiv(H, TH), tiles_w = cdiv(W, TW), cgroups = cdiv(C, CG);
    // 8 interleaved queues (one per blockIdx % 8) of ceil(pairs / 8) * tiles workgroups each
    long long blocks = 8LL * cdiv((long long)N * cgroups, 8) * tiles_h * tiles_w;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    int cshift = 0;
    while ((1 << cshift) < C) ++cshift;
    const bool cpow2 = (1 << cshift) == C;
    // cell / W for every cell < H*W by one 24-bit multiply: (cell * magic) >> shift with
    // cell, magic < 2^24 and cell * magic < 2^32; verified exhaustively (H*W is small)
    FastDiv dw;
    dw.magic = 0;
    dw.shift = 0;
    bool fast = cpow2 && (long long)H * W < (1 << 16);
    if (fast) {
        const unsigned cells = (unsigned)H * (unsigned)W;
        bool found = false;
        for (int sft = 8; sft <= 24 && !found; ++sft) {
            unsigned long long mg = ((1ULL << sft) + (unsigned)W - 1) / (unsigned)W;
            if (mg >= (1ULL << 24) || mg * (cells ? cells - 1 : 0) >= (1ULL << 32)) continue;
            bool exact = true;
            for (unsigned n = 0; n < cells && exact; ++n)
                exact = (unsigned)((n * mg) >> sft) == n / (unsigned)W;
            if (exact) { dw.magic = (unsigned)mg; dw.shift = sft; found = true; }
        }
        fast = found;
    }
    if (fast)
        hipLaunchKernelGGL((roi_pool_bwd_kernel<TH, TW, CG, CHUNK, true, MAXB, MINB>), dim3((unsigned)blocks),
                           dim3(CG), 0, st, top_diff, argmax, rois, R, N, H, W, C, PH, PW, scale,
                           bottom_diff, tiles_h, tiles_w, cgroups, cshift, dw);
    else
        hipLaunchKernelGGL((roi_pool_bwd_kernel<TH, TW, CG, CHUNK, false, MAXB, MINB>), dim3((unsigned)blocks),
                           dim3(CG), 0, st, top_diff, argmax, rois, R, N, H, W, C, PH, PW, scale,
                           bottom_diff, tiles_h, tiles_w, cgroups, cshift, dw);
    return check_launch();
}

}  // namespace wssdl

using namespace wssdl;

extern "C" int wssdl_roi_pool_forward(const float *bottom, int N, int H, int W, int C,
                                      const float *rois, int R, int pooled_h, int pooled_w,
                                      float spatial_scale, int rounding, float *top,
                                      int32_t *argmax, wssdl_stream_t stream) {
    if (H < 1 || W < 1 || C < 1 || R < 0 || pooled_h < 1 || pooled_w < 1)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (rounding != WSSDL_ROI_ROUND_CUDA && rounding != WSSDL_ROI_ROUND_CPU)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if ((long long)H * W * C > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;  // argmax is i32
    if (R == 0) return WSSDL_OK;
    if (!bottom || !rois || !top || !argmax) return WSSDL_ERR_INVALID_ARGUMENT;
    // N = WSSDL_ROI_BATCH_UNKNOWN (-1): the batch size is unknown -- the reference's ROIPoolForwardLaucher is not told
    // it (roi_pooling_op_gpu.h:17-21) and never range-checks the batch index -- so only a negative index makes a RoI
    // empty here and an index beyond the caller's tensor is read, as in the reference; with N > 0 an index >= N makes
    // an empty RoI too.  N == 0 with RoIs to pool is a caller's mistake, not a request for the unchecked form.
    if (N == 0 || N < WSSDL_ROI_BATCH_UNKNOWN) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == WSSDL_ROI_BATCH_UNKNOWN) N = 0x7fffffff;
    hipStream_t st = as_stream(stream);
    {   // the round-3 kernel (one wave per bin row, scalar windows, shared columns) with an i32 store
        const int rc = launch_fwd_rows_i32(bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale, rounding, top,
                                           argmax, st);
        if (rc != WSSDL_ROWS_I32_UNSUPPORTED) return rc;
    }
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(bottom) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(top) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(argmax) & 15) == 0);
    if (vec && C % 32 == 0 && C / 32 <= 256) {
        const int lanes_per_row = C / 32;
        const int rows_per_block = 256 / lanes_per_row;
        const long long row_blocks = ((long long)R * pooled_h + rows_per_block - 1) / rows_per_block;
        if (row_blocks * 8 <= 0x7fffffffLL) {
            if (row_blocks * 8 <= 4096)      // less than ~4 workgroups per CU
                hipLaunchKernelGGL(roi_pool_fwd_sliced_kernel<2>, dim3((unsigned)(row_blocks * 8)), dim3(256),
                                   0, st, bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale,
                                   rounding, top, argmax, lanes_per_row, rows_per_block);
            else
                hipLaunchKernelGGL(roi_pool_fwd_sliced_kernel<1>, dim3((unsigned)(row_blocks * 8)), dim3(256),
                                   0, st, bottom, N, H, W, C, rois, R, pooled_h, pooled_w, spatial_scale,
                                   rounding, top, argmax, lanes_per_row, rows_per_block);
            return check_launch();
        }
    }
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

// RoiPoolGrad on the list-driven walk (roi_pool_walk.hip) with the reference's i32 arg-max: prepare + walk in one
// call, the lists in a caller-owned workspace (wssdl_roi_pool_backward_workspace_bytes).  Shapes the walk does not
// take (C not a power of two, pooled size > 8, ...) run the tile-owner kernel of wssdl_roi_pool_backward.
extern "C" int wssdl_roi_pool_backward_ws(const float *top_diff, const int32_t *argmax, const float *rois, int R, int N,
                                          int H, int W, int C, int pooled_h, int pooled_w, float spatial_scale,
                                          float *bottom_diff, void *workspace, size_t workspace_bytes,
                                          wssdl_stream_t stream) {
    if (N < 0 || H < 1 || W < 1 || C < 1 || R < 0 || pooled_h < 1 || pooled_w < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax || !rois))) return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace && (reinterpret_cast<uintptr_t>(top_diff) & 7) == 0 && (reinterpret_cast<uintptr_t>(argmax) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(bottom_diff) & 7) == 0 && walk_i32_supported(R, N, H, W, C, pooled_h, pooled_w) &&
        workspace_bytes >= walk_workspace_bytes(R, N, H, W, pooled_h, pooled_w)) {
        hipStream_t st = as_stream(stream);
        int plan = -1;
        // 8 bytes per element instead of 5: the train-sized launch is even more bandwidth-bound than on the 1-byte
        // path and wants the larger 6x8 tiles (fewer border re-reads): 0.83 against 0.87 ms at R = 8512 x 1024
        // channels (tools/bwd_fixed_sweep.py --i32); "roi_bwd_plan" still overrides
        // a tuned plan the i32 form is not built for (the knob must not change results, let alone fail the call): the
        // automatic choice instead
        const int tuned = tuning().roi_bwd_plan;
        const int auto_plan = walk_plan_auto_id(N, H, W, C) == 11 ? 9 : walk_plan_auto_id(N, H, W, C);
        const int force = (tuned >= 0 && walk_i32_plan_built(tuned)) ? tuned : auto_plan;
        // (the window starts the lists also carry are not read on this path: the rounding mode does not matter)
        int rc = walk_prepare(rois, R, N, H, W, C, pooled_h, pooled_w, spatial_scale, WSSDL_ROI_ROUND_CUDA, workspace,
                              workspace_bytes, &plan, st, force);
        if (rc != WSSDL_OK) return rc;
        return launch_walk(top_diff, reinterpret_cast<const unsigned char *>(argmax), R, N, H, W, C, pooled_h, pooled_w,
                           bottom_diff, workspace, workspace_bytes, plan, st, 1, nullptr, true);
    }
    return wssdl_roi_pool_backward(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w, spatial_scale, bottom_diff,
                                   stream);
}

extern "C" int wssdl_roi_pool_backward(const float *top_diff, const int32_t *argmax,
                                       const float *rois, int R, int N, int H, int W, int C,
                                       int pooled_h, int pooled_w, float spatial_scale,
                                       float *bottom_diff, wssdl_stream_t stream) {
    if (N < 0 || H < 1 || W < 1 || C < 1 || R < 0 || pooled_h < 1 || pooled_w < 1)
        return WSSDL_ERR_INVALID_ARGUMENT;
    if (pooled_h > 255 || pooled_w > 255) return WSSDL_ERR_INVALID_ARGUMENT;   // 8-bit bin tables
    if ((long long)H * W * C > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    if (N == 0) return WSSDL_OK;
    if (!bottom_diff || (R > 0 && (!top_diff || !argmax || !rois)))
        return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    // Tile size: the kernel is bound by the serial latency of a wave's walk (load batch -> wait ->
    // dependent LDS read-modify-writes), hidden only by waves per CU, not by HBM bandwidth:
    // 4x4-cell tiles (16 KiB of LDS per 256-channel workgroup, 63 VGPRs -> 28 waves per CU)
    // re-read MORE bytes than 4x8 tiles (32 KiB, 16 waves per CU) and are 5-9 % faster; 8x8
    // tiles (8 waves per CU) re-read 18 % less and are 50-70 % slower whatever the depth of
    // their load batches.
    // Channels per workgroup: 256 when that still yields enough workgroups to fill the chip,
    // fewer otherwise (small batches / narrow feature maps).
    int cg = C > 128 ? 256 : (C > 64 ? 128 : 64);
    const long long tiles = (long long)cdiv(H, 4) * cdiv(W, 8);       // counted in 4x8 tiles
    while (cg > 64 && (long long)N * cdiv(C, cg) * tiles < BWD_MIN_WORKGROUPS) cg >>= 1;
    {       // tuning override
        const int v = tuning().roi_bwd_cg;
        if (v == 64 || v == 128 || v == 256) cg = v;
    }
    if (cg == 256)
        return launch_bwd<4, 4, 256, 254, 8, 8>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                                                spatial_scale, bottom_diff, st);
    if (cg == 128)
        return launch_bwd<4, 8, 128, 256>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                                          spatial_scale, bottom_diff, st);
    return launch_bwd<4, 8, 64, 128>(top_diff, argmax, rois, R, N, H, W, C, pooled_h, pooled_w,
                                     spatial_scale, bottom_diff, st);
}
