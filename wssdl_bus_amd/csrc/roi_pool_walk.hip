// RoI-pool backward, training path (1-byte arg-max), list-driven: gfx950 (MI355X).
//
// Reference semantics: roi_pooling_op_gpu.cu.cc:114-190 == roi_pooling_op.cc:383-458 (per bottom
// element, over the RoIs of its image that contain it and over its candidate bins, add top_diff
// where the arg-max matches -- f32, in the order roi, ph, pw).  Results are bit-identical to that.
//
// What measurement of the tile-owner kernel (roi_pool_compact.hip, kept as the fallback) showed:
//   * it is not bandwidth-bound: the walk without any load takes 0.5 ms, the loads without the
//     walk 0.6 ms, the RoI filter every workgroup repeats 0.15 ms (R = 8512, C = 1024);
//   * the busiest tiles (image centre of the 2000-RoI weak images) are serial chains of ~320
//     (RoI, tile) records at ~3 us each -- load, wait for HBM, accumulate -- and take 80 % of the
//     kernel's span, while half of the chip idles behind them.
// So the structure here is:
//   1. span     first / last RoI index of every image (one atomic pair per wave and image)
//   2. axes     one lane per (RoI, tile row) / (RoI, tile column): the reference's in_roi and
//               candidate-bin tests and the forward's window starts, as 16-byte table entries
//      count    one workgroup per (image, tile): its candidate bins
//      order    one workgroup: offsets (prefix sum) and the tiles sorted by work, heaviest first
//               (a single cursor claimed with one atomic per tile serialised: 65 us for 1280 tiles)
//   3. fill     one workgroup per (image, tile): the tile's SLOT STREAM in the reference's order
//               (roi^, ph^, pw^): 8 bytes per candidate bin = element offset of the bin in
//               top_diff / arg8 + the masks and window offsets that decode a code for this tile
//               (the reference's in_roi / candidate-bin tests, evaluated once here)
//   4. walk     one WAVE per (image, tile, 128 channels), launched heaviest tiles first: a
//               record = 8 slots (64 B) is fetched by 16 lanes and broadcast to scalar registers;
//               the data of the next DEPTH-1 records is in flight while a record is accumulated
//               (straight-line code: 8 slots, no per-slot branches, out-of-range offsets for the
//               padding slots of a tile's last record); a lane owns TWO channels (one 2-byte +
//               one 8-byte load per slot and lane); accumulators live in LDS (18.5 KiB per wave for
//               the default 6x6-cell tiles), branch-free (misses add 0.0 to a spare cell).  A lane is the
//               only writer of its channels: the f32 additions run in exactly the reference's order.
// The lists are shared by all channel groups, the walk needs no barrier at all, and no workgroup
// repeats the RoI filter.
//
// Round 3, measured and NOT kept ("gangs"): walking GH x GW neighbouring tiles in one
// workgroup so that the border bins they share are fetched once.  Co-scheduling alone left the L2 hit
// rate where it was (13 %); pacing the waves of a gang by RoI segment (barrier or a sliding window)
// brought the fetched bytes from 3.08 to 2.65-2.75 GB but cost more time than it saved (0.57-1.2 ms
// against 0.51): lock-step waves cannot keep enough loads in flight, and 8-wave workgroups quantise
// the launch.  The free-running walk below stays.
#include "roi_pool.hip.h"

namespace wssdl {

constexpr unsigned ARG8_EMPTY_W = 0xffu;
constexpr int WALK_SLOTS = 8;            // slots per record (64 B)
constexpr int WALK_MAX_SEGMENTS = 16;    // split form: segments a tile's stream may be cut into

__device__ __forceinline__ int win_start_w(int p, float bin, int rs, int limit, int rounding) {
    const float v = (float)p * bin;
    const int s = (rounding == WSSDL_ROI_ROUND_CPU) ? (int)v : (int)floorf(v);
    return min(max(s + rs, 0), limit);
}

struct WalkWs {
    int *tile_slots;    // [items] candidate bins of each (image, tile)
    int *tile_off;      // [items] first record of the tile's stream
    int *order;         // [items] items sorted by tile_slots, heaviest first
    int *img_span;      // [N][2]  (R - first RoI index, one-past-last RoI index) of each image (zeroed per call)
    int *total;         // [4]     records in use, error flags                                  (zeroed per call)
    unsigned long long *rowtab;  // [R][tiles_h][2]  AxisEntry of every (RoI, tile row)
    unsigned long long *coltab;  // [R][tiles_w][2]  ... (RoI, tile column)
    unsigned long long *slots;   // [cap_records * 8]
};

static size_t carve_walk(void *ws, int R, int N, int tiles_h, int tiles_w, long long cap_records, WalkWs *out) {
    Carver c(ws);
    WalkWs w;
    const int tiles = tiles_h * tiles_w;
    const size_t items = (size_t)N * tiles;
    w.img_span = c.take<int>((size_t)N * 2);
    w.total = c.take<int>(4);
    w.tile_slots = c.take<int>(items);
    w.tile_off = c.take<int>(items);
    w.order = c.take<int>(items);
    w.rowtab = c.take<unsigned long long>((size_t)R * tiles_h * 2);
    w.coltab = c.take<unsigned long long>((size_t)R * tiles_w * 2);
    w.slots = c.take<unsigned long long>((size_t)cap_records * WALK_SLOTS);
    if (out) *out = w;
    return c.off;
}

// upper bound of the records a call can need (every bin counted in every tile its window can reach)
static long long walk_record_bound(int R, int N, int H, int W, int PH, int PW, int TH, int TW) {
    const int win_h = cdiv(H + 1, PH) + 1, win_w = cdiv(W + 1, PW) + 1;
    const long long per_bin = (long long)cdiv(win_h + TH - 1, TH) * cdiv(win_w + TW - 1, TW);
    const long long slots = (long long)R * PH * PW * per_bin;
    const long long tiles = (long long)N * cdiv(H, TH) * cdiv(W, TW);
    return slots / WALK_SLOTS + tiles + 8;        // + one partly filled record per tile
}

constexpr int FILL_BLOCK = 1024;

// ---- 1. span: first / last RoI of every image ---------------------------------------------------
__global__ __launch_bounds__(256) void walk_span_kernel(const float *__restrict__ rois, int R, int N,
                                                        int *__restrict__ img_span) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int n = -1;
    if (r < R) {
        n = (int)rois[(size_t)r * 5];
        if (n < 0 || n >= N) n = -1;
    }
    // RoIs arrive grouped by image almost always: one atomic pair per wave and image
    unsigned long long todo = __ballot(n >= 0);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int n0 = __builtin_amdgcn_readlane(n, leader);
        const unsigned long long same = __ballot(n == n0);
        if (lane == leader) {
            const int r0 = r - lane;
            atomicMax(&img_span[n0 * 2], R - (r0 + __ffsll((long long)same) - 1));
            atomicMax(&img_span[n0 * 2 + 1], r0 + 64 - __clzll((long long)same));
        }
        todo &= ~same;
    }
}

// block-wide exclusive scan of one int per thread (BLOCK threads); returns the block total via *total
template <int BLOCK>
__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_sums /* [BLOCK/64] LDS */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();                       // wave_sums may still be read from the previous call
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const int s = wave_sums[w];
        before += (w < wave) ? s : 0;
        all += s;
    }
    *total = all;
    return before + incl - v;
}

// ---- 2. fill: the slot stream of one (image, tile), in (roi, ph, pw) order ---------------------
//   slot = w0 | w1 << 32:  w0 = element offset of the bin (r * PH*PW*C + bin * C),
//   w1 = rm | cm << 8 | (hs & 31) << 16 | (ws & 31) << 21 where, for this tile,
//     rm : 8 bits, bit j <=> tile row h0+j lies in the RoI and the bin's row is one of its candidate
//          rows (roi_pooling_op_gpu.cu.cc:141-151,169-177); cm: 8 bits likewise for columns;
//     hs : (clipped window start of the bin's row) - h0, clamped to [-16, 15]; ws likewise.
//   Padding slots of the last record: w0 = total elements (out of range: loads return 0), w1 = 0.
// One (RoI, tile row) or (RoI, tile column): 16 bytes.
//   mask  : bit 8*k + j <=> tile line t0+j lies in the RoI and bin p0+k is one of its candidates
//           (the reference's in_roi and candidate tests, roi_pooling_op_gpu.cu.cc:141-151,169-177)
//   info  : bits 5k..5k+4 = (clipped window start of bin p0+k) - t0, clamped to [-16, 15], k < 8;
//           bits 40..47 = p0, bits 48..51 = pn (0: the RoI does not touch this line of tiles)
struct AxisEntry {
    unsigned long long mask, info;
};

template <int TN>
__device__ __forceinline__ AxisEntry axis_entry(int t0, int limit, int rs, int re, float bin, int P, int rounding) {
    AxisEntry e;
    int p0, pn;
    const int t1 = min(t0 + TN, limit) - 1;
    touch_axis<TN, 8>(t0, t1, rs, re, bin, P, p0, pn, e.mask);
    e.info = 0ull;
    if (pn <= 0 || re < rs) { e.mask = 0ull;  return e; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = min(max(win_start_w(p0 + k, bin, rs, limit, rounding) - t0, -16), 15);
        e.info |= (unsigned long long)((unsigned)a & 31u) << (5 * k);
    }
    e.info |= ((unsigned long long)(unsigned)p0 << 40) | ((unsigned long long)(unsigned)pn << 48);
    return e;
}

// one lane per (RoI, tile row) and per (RoI, tile column): the reference's tests are evaluated
// tiles_h + tiles_w times per RoI instead of once per (RoI, tile) and axis
template <int TH, int TW>
__global__ __launch_bounds__(256) void walk_axes_kernel(
    const float *__restrict__ rois, int R, int N, int H, int W, int PH, int PW, float scale, int rounding,
    int tiles_h, int tiles_w, unsigned long long *__restrict__ rowtab, unsigned long long *__restrict__ coltab) {
    const int per = tiles_h + tiles_w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * per) return;
    const int r = (int)(i / per), a = (int)(i - (long long)r * per);
    const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
    const bool bad = g.batch < 0 || g.batch >= N;
    AxisEntry e;
    if (a < tiles_h) {
        e = axis_entry<TH>(a * TH, H, g.sh, g.eh, g.bin_h, PH, rounding);
        if (bad) e.mask = e.info = 0ull;
        rowtab[((size_t)r * tiles_h + a) * 2] = e.mask;
        rowtab[((size_t)r * tiles_h + a) * 2 + 1] = e.info;
    } else {
        const int tx = a - tiles_h;
        e = axis_entry<TW>(tx * TW, W, g.sw, g.ew, g.bin_w, PW, rounding);
        if (bad) e.mask = e.info = 0ull;
        coltab[((size_t)r * tiles_w + tx) * 2] = e.mask;
        coltab[((size_t)r * tiles_w + tx) * 2 + 1] = e.info;
    }
}

// ---- bin-owner form (round 5) ---------------------------------------------------------------------
// The exact walk lists a bin in EVERY tile its window touches (1.58 x the bins for 6x6 tiles on the default
// workload: the bytes the launch re-reads).  Here a wave accumulates into a REGION of RN lines that starts at its
// tile's origin and reaches RN - SN lines into the next tile (tiles repeat every SN lines, SN < RN), and a bin
// is listed by the FIRST tile whose region can hold it: the tile of its window's first line.  A window that
// does not fit continues in the tile of its first uncovered line, and so on (a chain; rare: windows are 2-3
// cells), so every line of a window belongs to exactly one (tile, mask) listing per axis and every
// (bin, cell) pair is applied once.  What lands outside the tile's own SN x SN cells goes to a halo buffer that
// walk_merge_kernel adds to the owners in a fixed order: deterministic, NOT the reference's association
// (roi_pooling_op_gpu.cu.cc:132-186 sums roi^, ph^, pw^ per cell) -- a separate entry point, like the split form.
__device__ __forceinline__ int win_end_w(int p, float bin, int rs, int limit, int rounding) {
    const float v = (float)(p + 1) * bin;
    const int e = (rounding == WSSDL_ROI_ROUND_CPU) ? (int)v : (int)ceilf(v);        // roi_pooling_op.cc:169-170 / _gpu.cu.cc:53-58
    return min(max(e + rs, 0), limit);
}

template <int RN, int SN>
__device__ __forceinline__ AxisEntry axis_entry_own(int T, int limit, int rs, int re, float bin, int P, int rounding) {
    static_assert(SN >= 1 && SN <= RN && RN <= 8 && RN - SN <= SN, "halo no larger than the tile; 8-bit masks");
    AxisEntry e;
    const int t0 = T * SN;
    int p0, pn;
    const int t1 = min(t0 + RN, limit) - 1;
    unsigned long long m;
    touch_axis<RN, 8>(t0, t1, rs, re, bin, P, p0, pn, m);
    e.mask = e.info = 0ull;
    if (pn <= 0 || re < rs) return e;
    int first = -1, last = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned bits = 0u;
        if (k < pn) {
            const int s = win_start_w(p0 + k, bin, rs, limit, rounding), z = win_end_w(p0 + k, bin, rs, limit, rounding);
            // the chain of listings of this window: u = first line not covered yet
            int u = s;
            for (int it = 0; it < 64 && u < z; ++it) {
                const int t = u / SN;
                if (t == T) {
                    const int lo = u - t0, hi = min(z, t0 + RN) - t0;        // lines [lo, hi) of the region
                    bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                    break;
                }
                if (t > T) break;
                u = t * SN + RN;
            }
            bits &= (unsigned)(m >> (8 * k)) & 0xffu;
        }
        if (bits) { if (first < 0) first = k;  last = k; }
        m = (m & ~(0xffull << (8 * k))) | ((unsigned long long)bits << (8 * k));
    }
    if (first < 0) return e;
    p0 += first;
    pn = last - first + 1;
    e.mask = m >> (8 * first);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = min(max(win_start_w(p0 + k, bin, rs, limit, rounding) - t0, -16), 15);
        e.info |= (unsigned long long)((unsigned)a & 31u) << (5 * k);
    }
    e.info |= ((unsigned long long)(unsigned)p0 << 40) | ((unsigned long long)(unsigned)pn << 48);
    return e;
}

template <int RH, int RW, int SH, int SW>
__global__ __launch_bounds__(256) void walk_axes_own_kernel(
    const float *__restrict__ rois, int R, int N, int H, int W, int PH, int PW, float scale, int rounding,
    int tiles_h, int tiles_w, unsigned long long *__restrict__ rowtab, unsigned long long *__restrict__ coltab) {
    const int per = tiles_h + tiles_w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * per) return;
    const int r = (int)(i / per), a = (int)(i - (long long)r * per);
    const RoiGeom g = roi_geometry(rois + (size_t)r * 5, scale, PH, PW);
    const bool bad = g.batch < 0 || g.batch >= N;
    AxisEntry e;
    if (a < tiles_h) {
        e = axis_entry_own<RH, SH>(a, H, g.sh, g.eh, g.bin_h, PH, rounding);
        if (bad) e.mask = e.info = 0ull;
        rowtab[((size_t)r * tiles_h + a) * 2] = e.mask;
        rowtab[((size_t)r * tiles_h + a) * 2 + 1] = e.info;
    } else {
        const int tx = a - tiles_h;
        e = axis_entry_own<RW, SW>(tx, W, g.sw, g.ew, g.bin_w, PW, rounding);
        if (bad) e.mask = e.info = 0ull;
        coltab[((size_t)r * tiles_w + tx) * 2] = e.mask;
        coltab[((size_t)r * tiles_w + tx) * 2 + 1] = e.info;
    }
}

__device__ __forceinline__ int axis_pn(unsigned long long info) { return (int)(info >> 48) & 15; }
__device__ __forceinline__ int axis_p0(unsigned long long info) { return (int)(info >> 40) & 255; }

// pass 1: the number of candidate bins of one (image, tile)
__global__ __launch_bounds__(FILL_BLOCK) void walk_count_kernel(
    const float *__restrict__ rois, int R, int tiles_h, int tiles_w, const int *__restrict__ img_span,
    const unsigned long long *__restrict__ rowtab, const unsigned long long *__restrict__ coltab,
    int *__restrict__ tile_slots) {
    __shared__ int wave_sums[FILL_BLOCK / 64];
    const int tiles = tiles_h * tiles_w;
    const int item = blockIdx.x;
    const int n = item / tiles, tile = item - n * tiles;
    const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
    const int lo = R - img_span[n * 2], hi = img_span[n * 2 + 1];       // (R, 0) when the image has no RoI
    int mine = 0;
    for (int r = lo + (int)threadIdx.x; r < hi; r += FILL_BLOCK) {
        if ((int)rois[(size_t)r * 5] != n) continue;
        mine += axis_pn(rowtab[((size_t)r * tiles_h + ty) * 2 + 1]) * axis_pn(coltab[((size_t)r * tiles_w + tx) * 2 + 1]);
    }
    int nslots;
    block_exclusive_scan<FILL_BLOCK>(mine, wave_sums, &nslots);
    if (threadIdx.x == 0) tile_slots[item] = nslots;
}

// Lean decode (round 5, bin-owner plans): a slot's row mask is an INTERVAL of region lines -- in_roi is one, the
// candidate test phstart(h) <= ph < phend(h) (roi_pooling_op_gpu.cu.cc:169-177) intersects a prefix and a suffix
// because floor((h - rs) / bin) and ceil((h - rs + 1) / bin) are monotone in h, the owner form's cover range is one --
// and so is the column mask.  With th = hs + ch (ch = code >> 4) the test "bit th of rm" becomes
// (ch - lo_h) <u n_h with lo_h = first line of the interval - hs, and the cell th * RW + tw = ch * RW + cw + base with
// base = hs * RW + ws: two subtract-and-compare pairs and one multiply-add per channel instead of two variable
// shifts, four ANDs and the range bookkeeping.  The empty code 0xff (ch = 15) can never pass: lo_h + n_h > 15 would
// need a window of 16 rows.  w1 = lo_h (signed 6) | n_h (4) << 6 | lo_w (signed 6) << 10 | n_w (4) << 16 | base (signed 10) << 20.
// (walk_fill_kernel builds the word from per-axis halves: the column halves once per RoI, the row half once per bin row;
// an empty mask on either axis gives n_h = 0, which fails every code; a clamped window start keeps lo out of the codes' range.)

// pass 2 (after the offsets are known): the slot stream of one (image, tile), in (roi, ph, pw) order
//   slot = w0 | w1 << 32:  w0 = element offset of the bin (r * PH*PW*C + bin * C),
//   w1 = rm | cm << 8 | (hs & 31) << 16 | (ws & 31) << 21: the 8 mask bits of the bin's row /
//   column for this tile's rows / columns and the window starts relative to the tile.
//   Padding slots of the last record: w0 = total elements (out of range: loads return 0), w1 = 0.
__global__ __launch_bounds__(FILL_BLOCK) void walk_fill_kernel(
    const float *__restrict__ rois, int C, int PW, int PHPW, int R, int tiles_h, int tiles_w,
    const int *__restrict__ img_span, const unsigned long long *__restrict__ rowtab,
    const unsigned long long *__restrict__ coltab, const int *__restrict__ tile_off,
    const int *__restrict__ tile_slots, unsigned long long *__restrict__ slots, long long cap_records,
    unsigned total_elems, int *__restrict__ total, int lean_rw) {
    // lean_rw > 0 (owner plans with the lean decode, round 5): w1 carries the masks as INTERVALS in code space and the
    // region cell of code (0, 0) -- "Lean decode" above -- for a region lean_rw cells wide
    __shared__ int wave_sums[FILL_BLOCK / 64];
    const int tiles = tiles_h * tiles_w;
    const int item = blockIdx.x;
    const int n = item / tiles, tile = item - n * tiles;
    const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
    const int lo = R - img_span[n * 2], hi = img_span[n * 2 + 1];
    const int nslots = tile_slots[item];
    const int nrec = (nslots + WALK_SLOTS - 1) / WALK_SLOTS;
    const long long first = (long long)tile_off[item] * WALK_SLOTS;
    const long long cap = cap_records * WALK_SLOTS;
    int run = 0;
    for (int base = lo; base < hi; base += FILL_BLOCK) {
        const int r = base + (int)threadIdx.x;
        unsigned long long rmask = 0, rinfo = 0, cmask = 0, cinfo = 0;
        int nb = 0;
        if (r < hi && (int)rois[(size_t)r * 5] == n) {
            rinfo = rowtab[((size_t)r * tiles_h + ty) * 2 + 1];
            cinfo = coltab[((size_t)r * tiles_w + tx) * 2 + 1];
            nb = axis_pn(rinfo) * axis_pn(cinfo);
            if (nb > 0) {
                rmask = rowtab[((size_t)r * tiles_h + ty) * 2];
                cmask = coltab[((size_t)r * tiles_w + tx) * 2];
            }
        }
        int tot;
        const int ex = block_exclusive_scan<FILL_BLOCK>(nb, wave_sums, &tot);
        if (nb > 0) {
            long long pos = first + run + ex;
            const int phn = axis_pn(rinfo), pwn = axis_pn(cinfo);
            const unsigned e0 = (unsigned)r * (unsigned)(PHPW * C) + (unsigned)((axis_p0(rinfo) * PW + axis_p0(cinfo)) * C);
            if (lean_rw > 0) {
                // the column halves of the lean words once per RoI (at most 8 bins), the row half once per bin row
                unsigned cpart[8];
                int cws[8];
                bool good = true;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned cm = (unsigned)(cmask >> (8 * j)) & 0xffu;
                    const int ws = ((int)(((unsigned)(cinfo >> (5 * j)) & 31u) << 27)) >> 27;
                    const int w0 = cm ? __ffs((int)cm) - 1 : 0, wn = __popc(cm);
                    good = good && (cm == (((1u << wn) - 1u) << w0));
                    cpart[j] = (((unsigned)min(max(w0 - ws, -32), 31) & 63u) << 10) | ((unsigned)wn << 16);
                    cws[j] = ws;
                }
                for (int q = 0; q < phn; ++q) {
                    const unsigned rm = (unsigned)(rmask >> (8 * q)) & 0xffu;
                    const int hs = ((int)(((unsigned)(rinfo >> (5 * q)) & 31u) << 27)) >> 27;
                    const int h0 = rm ? __ffs((int)rm) - 1 : 0, hn = __popc(rm);
                    good = good && (rm == (((1u << hn) - 1u) << h0));
                    const unsigned rpart = ((unsigned)min(max(h0 - hs, -32), 31) & 63u) | ((unsigned)hn << 6);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j >= pwn) break;
                        // (an empty mask on either axis: n_h = 0 fails every code)
                        const bool any = rm != 0u && ((cpart[j] >> 16) & 15u) != 0u;
                        const unsigned whi = (any ? rpart : (rpart & 63u)) | cpart[j] |
                                             (((unsigned)(hs * lean_rw + cws[j]) & 1023u) << 20);
                        const unsigned wlo = e0 + (unsigned)((q * PW + j) * C);
                        if (pos < cap) slots[pos] = (unsigned long long)wlo | ((unsigned long long)whi << 32);
                        ++pos;
                    }
                }
                if (!good) atomicOr(&total[1], 2);          // a mask that is not an interval: cannot happen ("Lean decode" above); never silent
            } else
            for (int q = 0; q < phn; ++q) {
                const unsigned hq = ((unsigned)(rmask >> (8 * q)) & 0xffu) | (((unsigned)(rinfo >> (5 * q)) & 31u) << 16);
                for (int j = 0; j < pwn; ++j, ++pos) {
                    const unsigned whi = hq | (((unsigned)(cmask >> (8 * j)) & 0xffu) << 8) |
                                         (((unsigned)(cinfo >> (5 * j)) & 31u) << 21);
                    const unsigned wlo = e0 + (unsigned)((q * PW + j) * C);
                    if (pos < cap) slots[pos] = (unsigned long long)wlo | ((unsigned long long)whi << 32);
                }
            }
        }
        run += tot;
    }
    for (int i = nslots + (int)threadIdx.x; i < nrec * WALK_SLOTS; i += FILL_BLOCK)      // pad the last record
        if (first + i < cap) slots[first + i] = (unsigned long long)total_elems;
    if (threadIdx.x == 0 && (run != nslots || first + (long long)nrec * WALK_SLOTS > cap)) atomicOr(&total[1], 1);
}

// ---- 3. launch order: tiles sorted by work, heaviest first ---------------------------------------
__global__ __launch_bounds__(1024) void walk_order_kernel(const int *__restrict__ tile_slots, int items,
                                                          int *__restrict__ tile_off, int *__restrict__ order,
                                                          int *__restrict__ total) {
    __shared__ int wave_sums[16];
    __shared__ int hist[1024];
    __shared__ int s_max;
    const int t = threadIdx.x;
    if (t == 0) s_max = 0;
    hist[t] = 0;
    __syncthreads();
    // offsets: exclusive prefix sum of the records per tile
    int run = 0, mx = 0;
    for (int base = 0; base < items; base += 1024) {
        const int i = base + t;
        const int v = i < items ? tile_slots[i] : 0;
        int tot;
        const int ex = block_exclusive_scan<1024>((v + WALK_SLOTS - 1) / WALK_SLOTS, wave_sums, &tot);
        if (i < items) tile_off[i] = run + ex;
        run += tot;
        mx = max(mx, v);
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    if (t == 0) total[0] = run;
    // bucket sort (the order only decides when a tile is launched)
    const long long top = max(s_max, 1);
    for (int i = t; i < items; i += 1024)
        atomicAdd(&hist[1023 - (int)((long long)tile_slots[i] * 1023 / top)], 1);
    __syncthreads();
    {
        const int v = hist[t];
        int tot;
        const int ex = block_exclusive_scan<1024>(v, wave_sums, &tot);
        __syncthreads();
        hist[t] = ex;
    }
    __syncthreads();
    for (int i = t; i < items; i += 1024) {
        const int pos = atomicAdd(&hist[1023 - (int)((long long)tile_slots[i] * 1023 / top)], 1);
        order[pos] = i;
    }
}

// ---- 4. walk ----------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));

struct SlotRec {                 // one record = 8 slots, wave-uniform (scalar registers)
    unsigned lo[WALK_SLOTS];     // element offset of the bin
    unsigned hi[WALK_SLOTS];     // masks and window offsets
};

template <int CPL>
struct SlotData {                // what a lane holds of one record
    unsigned a[WALK_SLOTS][2];   // [0]: CPL 1-byte codes; I32: the CPL flat indices in [0], [1]
    float td[WALK_SLOTS][CPL];   // CPL top_diff values
};

// I32 (round 4): the arg-max in the reference op's own layout (i32 flat NHWC index inside the image, -1 = empty
// bin: roi_pooling_op_gpu.cu.cc:71-79) instead of the 1-byte codes -- wssdl_roi_pool_backward_ws.  A lane loads
// CPL i32 values per slot; index -> cell by a shift (C is a power of two on this path), cell -> (h, w) by one
// 24-bit multiply (exhaustively verified for every cell of the map on the host).
struct WalkI32 {
    int cshift;            // log2(C)
    unsigned magic, shift; // cell / W = (cell * magic) >> shift
};

// a record is fetched by lanes 0..15 (one dword each) and broadcast with v_readlane: that keeps the
// fetch in the vector-memory queue, where the compiler can count it (a scalar load would be waited
// for at the next LDS access, i.e. immediately)
__device__ __forceinline__ unsigned fetch_rec(const unsigned *__restrict__ recs, int i, int nrec, int lane) {
    const int k = min(i, max(nrec - 1, 0));
    return recs[(size_t)k * 16 + (lane & 15)];
}

__device__ __forceinline__ SlotRec spread_rec(unsigned v, bool valid, unsigned total_elems) {
    SlotRec r;
#pragma unroll
    for (int s = 0; s < WALK_SLOTS; ++s) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, 2 * s);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)v, 2 * s + 1);
        r.lo[s] = valid ? lo : total_elems;
        r.hi[s] = valid ? hi : 0u;
    }
    return r;
}

template <int CPL, bool I32, int AUX = 0 /* cache policy bits of the data loads (gfx950: 1 = sc0, 2 = nt, 16 = sc1) */>
__device__ __forceinline__ void issue_rec(SlotData<CPL> &d, const SlotRec &r, __amdgpu_buffer_rsrc_t ra,
                                          __amdgpu_buffer_rsrc_t rt, int voff8, int voff) {
#pragma unroll
    for (int s = 0; s < WALK_SLOTS; ++s) {
        if (I32) {
            if (CPL == 2) {
                typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                const uint2v a = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(ra, voff, (int)(r.lo[s] << 2), AUX));
                d.a[s][0] = a.x;  d.a[s][1] = a.y;
                const float2v t = __builtin_bit_cast(float2v, __builtin_amdgcn_raw_buffer_load_b64(rt, voff, (int)(r.lo[s] << 2), AUX));
                d.td[s][0] = t.x;  d.td[s][CPL - 1] = t.y;
            } else {
                d.a[s][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, voff, (int)(r.lo[s] << 2), AUX);
                d.td[s][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, voff, (int)(r.lo[s] << 2), AUX));
            }
        } else if (CPL == 2) {
            d.a[s][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(ra, voff8, (int)r.lo[s], AUX);
            const float2v t = __builtin_bit_cast(float2v, __builtin_amdgcn_raw_buffer_load_b64(rt, voff, (int)(r.lo[s] << 2), AUX));
            d.td[s][0] = t.x;  d.td[s][CPL - 1] = t.y;
        } else {
            d.a[s][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(ra, voff8, (int)r.lo[s], AUX);
            d.td[s][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, voff, (int)(r.lo[s] << 2), AUX));
        }
    }
}

template <int TH, int TW, int CPL, bool I32>
__device__ __forceinline__ void process_rec(const SlotData<CPL> &d, const SlotRec &r, float *acc, int lane, bool lane_ok,
                                            const WalkI32 &q, int h0, int w0, int W) {
    constexpr int DUMMY = TH * TW;
#pragma unroll
    for (int s = 0; s < WALK_SLOTS; ++s) {
        const unsigned w = r.hi[s];
        const unsigned rm = w & 0xffu, cm = (w >> 8) & 0xffu;
        const int hs = ((int)(w << 11)) >> 27, ws = ((int)(w << 6)) >> 27;     // sign-extended 5-bit fields
        const unsigned a = d.a[s][0];
        int idx[CPL];
        float val[CPL];
#pragma unroll
        for (int p = 0; p < CPL; ++p) {
            int th, tw;
            bool live;
            if (I32) {
                // flat index -> cell -> (h, w); -1 (empty bin) and anything outside the tile fail the range tests
                const int fi = (int)d.a[s][p];
                const unsigned cellg = (unsigned)fi >> q.cshift;
                const unsigned hh = __umul24(cellg, q.magic) >> q.shift;
                th = (int)hh - h0;
                tw = (int)(cellg - hh * (unsigned)W) - w0;
                live = (fi >= 0) & ((unsigned)th < (unsigned)TH) & ((unsigned)tw < (unsigned)TW);
            } else {
                const unsigned code = (a >> (8 * p)) & 0xffu;
                th = hs + (int)(code >> 4);
                tw = ws + (int)(code & 15u);
                live = code != ARG8_EMPTY_W;
            }
            // tile, in_roi and candidate-bin tests: the masks have no bits where a cell outside the
            // tile would index (th, tw in [-16, 30]; shifts use the low 5 bits)
            const unsigned bits = (rm >> (th & 31)) & (cm >> (tw & 31)) & 1u;
            const bool ok = (bits != 0u) & live & lane_ok;
            const int cell = ok ? th * TW + tw : DUMMY;
            idx[p] = (cell * CPL + p) * 64 + lane;
            val[p] = ok ? d.td[s][p] : 0.0f;
        }
        float v[CPL];
#pragma unroll
        for (int p = 0; p < CPL; ++p) v[p] = acc[idx[p]];
#pragma unroll
        for (int p = 0; p < CPL; ++p) acc[idx[p]] = v[p] + val[p];
    }
}

// the lean decode of a record (owner plans, 1-byte codes, every lane inside C): the word is described under "Lean decode"
template <int TH, int TW, int CPL>
__device__ __forceinline__ void process_rec_lean(const SlotData<CPL> &d, const SlotRec &r, float *acc, int lane) {
    constexpr int DUMMY = TH * TW;
#pragma unroll
    for (int s = 0; s < WALK_SLOTS; ++s) {
        const unsigned w = r.hi[s];
        const int lo_h = ((int)(w << 26)) >> 26, lo_w = ((int)(w << 16)) >> 26, base = ((int)(w << 2)) >> 22;
        const unsigned n_h = (w >> 6) & 15u, n_w = (w >> 16) & 15u;
        const unsigned a = d.a[s][0];
        int idx[CPL];
        float val[CPL];
#pragma unroll
        for (int p = 0; p < CPL; ++p) {
            const unsigned ch = (a >> (8 * p + 4)) & 15u, cw = (a >> (8 * p)) & 15u;
            const bool ok = ((unsigned)((int)ch - lo_h) < n_h) & ((unsigned)((int)cw - lo_w) < n_w);
            const int cell = ok ? (int)__umul24(ch, (unsigned)TW) + (int)cw + base : DUMMY;
            idx[p] = (cell * CPL + p) * 64 + lane;
            val[p] = ok ? d.td[s][p] : 0.0f;
        }
        float v[CPL];
#pragma unroll
        for (int p = 0; p < CPL; ++p) v[p] = acc[idx[p]];
#pragma unroll
        for (int p = 0; p < CPL; ++p) acc[idx[p]] = v[p] + val[p];
    }
}

template <int TH, int TW, int DEPTH, int MINW, int CPL, bool I32, bool OWN = false, int AUX = 0>
__global__ __launch_bounds__(64, MINW) void roi_pool_bwd_walk_kernel(
    const float *__restrict__ top_diff, const unsigned char *__restrict__ arg8 /* I32: the i32 arg-max */,
    const unsigned *__restrict__ slots, const int *__restrict__ tile_off, const int *__restrict__ tile_slots,
    const int *__restrict__ order, int items, int tiles_w, int tiles, int G, int H, int W, int C,
    unsigned total_elems, float *__restrict__ bottom_diff, int nseg, float *__restrict__ partial,
    unsigned long long seg_stride, WalkI32 q, int own_sh, int own_sw) {
    // OWN (bin-owner form): TH x TW is the wave's REGION; tiles repeat every own_sh x own_sw cells; `partial` is the
    // halo buffer [items][TH * TW][C] (region cells outside the tile's own cells; walk_merge_kernel adds them up)
    static_assert(TH <= 8 && TW <= 8 && DEPTH >= 2 && DEPTH <= 4, "8-bit masks; 2..4 records in flight");
    __shared__ float acc[(TH * TW + 1) * CPL * 64];
    // blockIdx -> (position k in the launch order, channel group g).  Workgroups are dealt
    // round-robin over the 8 XCDs (blockIdx % 8; observed, speed only): with 8 or more channel
    // groups every XCD serves its own groups for all tiles, so the border bins neighbouring tiles
    // re-read meet in one L2.
    const int b = blockIdx.x;
    int k, g;
    if ((G & 7) == 0) {
        // consecutive channel groups on one XCD (they share the 128-byte lines of the codes when a
        // group is only 64 channels wide)
        const int per = G >> 3, q = b >> 3;
        g = (b & 7) * per + (q % per);
        k = q / per;
    } else if (G < 8 && (8 % G) == 0) {
        const int share = 8 / G, x = b & 7;
        g = x % G;
        k = (b >> 3) * share + x / G;
    } else {
        g = b % G;
        k = b / G;
    }
    if (k >= items) return;
    const int item = order[k];
    const int n = item / tiles, tile = item - n * tiles;
    const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
    const int h0 = ty * (OWN ? own_sh : TH), w0 = tx * (OWN ? own_sw : TW);
    const int lane = threadIdx.x;
    const int c0 = g * (64 * CPL) + CPL * lane;
    const bool lane_ok = c0 < C;
    const int cl = lane_ok ? c0 : 0;
    const bool lean = (C % (64 * CPL)) == 0;          // the lists of a lean plan are in the lean format exactly then

#pragma unroll
    for (int i = 0; i < (TH * TW + 1) * CPL; ++i) acc[i * 64 + lane] = 0.0f;

    // Split form (wssdl_roi_pool_backward_compact_split): blockIdx.y = segment s of nseg walks the records
    // [s, s + 1) * nrec / nseg of the tile's stream into a tile of its own; segment 0 writes bottom_diff, the
    // others partial[s - 1] (same layout), and walk_combine_kernel adds them in segment order.  nseg == 1 is the
    // reference's summation order, bit for bit; nseg > 1 is deterministic but associates the sum differently.
    const int seg = blockIdx.y;
    const int nrec_all = (tile_slots[item] + WALK_SLOTS - 1) / WALK_SLOTS;
    const int rec0 = (int)((long long)nrec_all * seg / nseg), rec1 = (int)((long long)nrec_all * (seg + 1) / nseg);
    const int nrec = rec1 - rec0;
    const unsigned *rp = slots + ((size_t)tile_off[item] + rec0) * 16;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(arg8), 0, (int)(I32 ? total_elems << 2 : total_elems), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(top_diff), 0, (int)(total_elems << 2), 0x00020000);
    const int voff8 = cl, voff = cl * 4;

    // DEPTH records in flight: while record i is accumulated the data of records i+1 .. i+DEPTH-1
    // and the descriptor of record i+DEPTH travel.  r[j] describes the data in d[j].
    SlotRec r[DEPTH];
    SlotData<CPL> d[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) {
        r[j] = spread_rec(fetch_rec(rp, j, nrec, lane), j < nrec, total_elems);
        issue_rec<CPL, I32, (AUX & 31)>(d[j], r[j], ra, rt, voff8, voff);
    }
    unsigned pending = fetch_rec(rp, DEPTH - 1, nrec, lane);
    for (int i = 0; i < nrec; i += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int x = (j + DEPTH - 1) % DEPTH;
            const unsigned next = fetch_rec(rp, i + j + DEPTH, nrec, lane);
            r[x] = spread_rec(pending, i + j + DEPTH - 1 < nrec, total_elems);
            issue_rec<CPL, I32, (AUX & 31)>(d[x], r[x], ra, rt, voff8, voff);
            if (OWN && !I32 && (AUX & 128) && lean) process_rec_lean<TH, TW, CPL>(d[j], r[j], acc, lane);
            else process_rec<TH, TW, CPL, I32>(d[j], r[j], acc, lane, lane_ok, q, h0, w0, W);
            pending = next;
        }
    }

    if (lane_ok) {
        float *img = ((OWN || seg == 0) ? bottom_diff : partial + (size_t)(seg - 1) * seg_stride) +
                     (size_t)n * H * W * C;
        // OWN with segments (round 6, wssdl_roi_pool_backward_compact_owner_split): a tile's stream is cut into nseg
        // pieces as in the split form, each walked by a wave of its own into a region of its own; EVERY cell of the
        // region then goes to the segment's slice of the halo buffer, [seg][item][region cell][C], and
        // walk_merge_split_kernel writes bottom_diff = sum over segments of (own cells + the neighbours' halos).
        float *halo = OWN ? partial + ((size_t)seg * items + item) * (TH * TW) * C : nullptr;
        const bool all_to_halo = OWN && nseg > 1;
#pragma unroll
        for (int i = 0; i < TH * TW; ++i) {
            const int h = h0 + i / TW, w = w0 + i % TW;
            if (h < H && w < W) {
                float *dst = img + ((size_t)h * W + w) * C + c0;
                if (OWN && (all_to_halo || i / TW >= own_sh || i % TW >= own_sw)) dst = halo + (size_t)i * C + c0;
                if (CPL == 2) {
                    float2v o;
                    o.x = acc[(i * CPL) * 64 + lane];
                    o.y = acc[(i * CPL + CPL - 1) * 64 + lane];
                    if (AUX & 64) __builtin_nontemporal_store(o, reinterpret_cast<float2v *>(dst));
                    else *reinterpret_cast<float2v *>(dst) = o;
                } else {
                    if (AUX & 64) __builtin_nontemporal_store(acc[i * 64 + lane], dst);
                    else *dst = acc[i * 64 + lane];
                }
            }
        }
    }
}

// bin-owner form: bottom_diff (the tiles' own cells, written by the walk) += the halos of the left, upper and
// upper-left neighbours, in that order.  One thread = 4 channels of one cell that can receive a halo (the first
// RH - SH rows and RW - SW columns of a tile); the other cells are final after the walk.
__global__ __launch_bounds__(256) void walk_merge_kernel(float *__restrict__ bottom_diff, const float *__restrict__ halo,
                                                          int H, int W, int C4, int tiles_h, int tiles_w, int RH, int RW,
                                                          int SH, int SW) {
    const int HH = RH - SH, HW = RW - SW;
    const int ncell = SH * SW - (SH - HH) * (SW - HW);          // receiving cells per tile
    const int item = blockIdx.x;
    const int tiles = tiles_h * tiles_w;
    const int n = item / tiles, tile = item - n * tiles;
    const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
    const size_t reg = (size_t)RH * RW * C4;                     // float4 per item of the halo buffer
    const float4v *hb = reinterpret_cast<const float4v *>(halo);
    float4v *out = reinterpret_cast<float4v *>(bottom_diff) + (size_t)n * H * W * C4;
    // blockIdx.y = receiving cell: the first HH rows whole, then the first HW columns of the other rows
    const int cell = blockIdx.y;
    if (cell >= ncell) return;
    int lh, lw;
    if (cell < HH * SW) { lh = cell / SW;  lw = cell - lh * SW; }
    else { const int j = cell - HH * SW;  lh = HH + j / HW;  lw = j - (j / HW) * HW; }
    const int h = ty * SH + lh, w = tx * SW + lw;
    if (h >= H || w >= W) return;
    const bool left = lw < HW && tx > 0, up = lh < HH && ty > 0;
    for (int c = threadIdx.x; c < C4; c += 256) {
        float4v v = out[((size_t)h * W + w) * C4 + c];
        if (left) {
            const float4v a = hb[(size_t)(item - 1) * reg + (size_t)(lh * RW + lw + SW) * C4 + c];
            v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
        }
        if (up) {
            const float4v a = hb[(size_t)(item - tiles_w) * reg + (size_t)((lh + SH) * RW + lw) * C4 + c];
            v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
        }
        if (left && up) {
            const float4v a = hb[(size_t)(item - tiles_w - 1) * reg + (size_t)((lh + SH) * RW + lw + SW) * C4 + c];
            v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
        }
        out[((size_t)h * W + w) * C4 + c] = v;
    }
}

// bin-owner form with segments: bottom_diff (written whole, no read) = for every cell, over the segments in order: the
// owning tile's region cell + the left, upper and upper-left neighbours' halo cells.  One thread = 4 channels of a cell.
__global__ __launch_bounds__(256) void walk_merge_split_kernel(float *__restrict__ bottom_diff, const float *__restrict__ halo,
                                                                int N, int H, int W, int C4, int tiles_h, int tiles_w, int RH,
                                                                int RW, int SH, int SW, int nseg) {
    const int HH = RH - SH, HW = RW - SW;
    const int tiles = tiles_h * tiles_w, items = N * tiles;
    const long long cell = blockIdx.x;                           // (image, h, w)
    const int n = (int)(cell / ((long long)H * W)), hw = (int)(cell - (long long)n * H * W);
    const int h = hw / W, w = hw - h * W;
    const int ty = h / SH, tx = w / SW, lh = h - ty * SH, lw = w - tx * SW;
    const int item = n * tiles + ty * tiles_w + tx;
    const size_t reg = (size_t)RH * RW * C4;
    const float4v *hb = reinterpret_cast<const float4v *>(halo);
    const bool left = lw < HW && tx > 0, up = lh < HH && ty > 0;
    for (int c = threadIdx.x; c < C4; c += 256) {
        float4v v = (float4v)(0.0f);
        for (int s = 0; s < nseg; ++s) {
            const float4v *hs = hb + (size_t)s * items * reg;
            const float4v o = hs[(size_t)item * reg + (size_t)(lh * RW + lw) * C4 + c];
            v.x = v.x + o.x;  v.y = v.y + o.y;  v.z = v.z + o.z;  v.w = v.w + o.w;
            if (left) {
                const float4v a = hs[(size_t)(item - 1) * reg + (size_t)(lh * RW + lw + SW) * C4 + c];
                v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
            }
            if (up) {
                const float4v a = hs[(size_t)(item - tiles_w) * reg + (size_t)((lh + SH) * RW + lw) * C4 + c];
                v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
            }
            if (left && up) {
                const float4v a = hs[(size_t)(item - tiles_w - 1) * reg + (size_t)((lh + SH) * RW + lw + SW) * C4 + c];
                v.x = v.x + a.x;  v.y = v.y + a.y;  v.z = v.z + a.z;  v.w = v.w + a.w;
            }
        }
        reinterpret_cast<float4v *>(bottom_diff)[((size_t)n * H * W + hw) * C4 + c] = v;
    }
}

// A plan = tile shape, records in flight, the waves per SIMD the launch bounds ask for and the
// channels per lane.  wssdl_set_tuning("roi_bwd_plan", id) overrides the choice of walk_plan_auto.
struct WalkPlan {
    int th, tw, depth, minw, cpl;
};

#define WSSDL_WALK_PLANS(X) \
    X(0, 4, 4, 2, 5, 2)  \
    X(1, 4, 8, 2, 2, 2)  \
    X(2, 8, 8, 3, 1, 2)  \
    X(3, 4, 8, 3, 2, 2)  \
    X(4, 8, 8, 4, 1, 2)  \
    X(5, 4, 4, 3, 4, 2)  \
    X(6, 8, 8, 4, 2, 1)  /* one channel per lane: 8x8 tiles at 4x8's LDS */ \
    X(7, 8, 8, 3, 2, 1)  \
    X(8, 8, 8, 2, 2, 1)  \
    X(9, 6, 8, 2, 1, 2)  /* fewer border re-reads (1.52 against 1.68), 6 waves per CU */ \
    X(10, 5, 8, 2, 1, 2) /* 1.58, 7 waves per CU */ \
    X(11, 6, 6, 2, 2, 2) /* 1.60, 8 waves per CU */ \
    X(12, 6, 8, 3, 1, 2) \
    X(13, 6, 6, 3, 2, 2) \
    X(14, 6, 7, 2, 1, 2) \
    X(15, 7, 6, 2, 1, 2) \
    X(16, 5, 7, 2, 2, 2) \
    X(17, 7, 7, 2, 1, 2) \
    X(18, 2, 4, 3, 4, 2) /* small launches: short slot chains per wave */ \
    X(19, 2, 2, 3, 4, 2) \
    X(20, 3, 4, 3, 4, 2) \
    X(21, 2, 2, 3, 4, 1) /* small launches, 64-channel waves: twice the waves */ \
    X(22, 2, 4, 3, 4, 1) \
    X(23, 4, 4, 3, 4, 1) \
    X(24, 3, 4, 3, 4, 1) \
    X(25, 6, 6, 3, 4, 1)

static const WalkPlan kWalkPlans[] = {
#define WSSDL_X(ID, TH, TW, D, MW, CPL) {TH, TW, D, MW, CPL},
    WSSDL_WALK_PLANS(WSSDL_X)
#undef WSSDL_X
};
constexpr int WALK_PLANS = sizeof(kWalkPlans) / sizeof(kWalkPlans[0]);

int walk_plan_count() { return WALK_PLANS; }

// The tile shape trades border re-reads (large tiles: fewer bytes) against the length of the slot
// chain one wave walks alone (small tiles: more, shorter chains).  A train-sized launch is
// bandwidth-bound and wants 6x6; a launch with few waves (few images or channel groups) is bound by
// its longest chain: 2 images x 256 channels x 4000 RoIs take 0.44 ms with 6x6 tiles, 0.24 with 4x4,
// 0.12 with 2x2 (tools/bwd_plan_sweep.py).  Rule: walk down the list below -- tile area descending
// -- and take the first plan that gives 2048 waves (one per wave slot of the chip at 8
// per CU).  Measured at the shapes of the other bench workloads (round 3): 3 images x 512 channels x
// 6000 RoIs 0.27 -> 0.22 ms (4x4 / 64 channels instead of 2x4 / 128), 2 x 256 x 4000 0.12 -> 0.11,
// 1 x 1024 x 300 0.042 -> 0.040.
static int walk_plan_auto(int N, int H, int W, int C) {
    // (round 4: 4x4 / 128 channels -- plan 5 -- left the list: wherever it was the first plan with 2048 waves, 4x4 /
    // 64 channels was faster: two weak images x 1024 channels 0.344 -> 0.298 ms on the alternating workload's own set
    // and 0.313 -> 0.289 on two weak images of the default set, 3 images x 512 channels 0.47 -> 0.34)
    static const int order[] = {11, 23, 18, 22, 19, 21};
    for (int id : order) {
        const WalkPlan &p = kWalkPlans[id];
        if ((long long)N * cdiv(H, p.th) * cdiv(W, p.tw) * cdiv(C, 64 * p.cpl) >= 2048) return id;
    }
    return 21;
}

int walk_plan_auto_id(int N, int H, int W, int C) { return walk_plan_auto(N, H, W, C); }

static int walk_plan_choice(int N, int H, int W, int C) {
    const int v = tuning().roi_bwd_plan;
    if (v >= 0 && v < WALK_PLANS) return v;
    return walk_plan_auto(N, H, W, C);
}

// Split form: a launch with few images is bound by the longest slot chain one wave walks alone (every RoI of a
// 2000-RoI image touches the tiles at the image's centre: ~9000 dependent LDS read-add-write steps at ~35 ns,
// whatever the tile shape), not by bandwidth.  Cutting every tile's stream into `segments` pieces walked by
// separate waves shortens the chain by that factor; the pieces' tiles are then added in segment order, which
// associates the f32 sum differently from the reference (roi_pooling_op_gpu.cu.cc:132-186 sums roi^, ph^, pw^):
// deterministic, within ~1e-7 relative of the exact walk, NOT bit-identical -- so it is a separate entry point
// (wssdl_roi_pool_backward_compact_split) and the exact walk stays the contract of ..._backward_compact.
// Suggested only where it pays (extra traffic: (segments - 1) * 2 * N*H*W*C*4 bytes): at most 4 images and at
// least 1000 RoIs per image (the weak images of the reference's default 1 + 2 batch and of the alternating mode).
// Measured (tools/bwd_fixed_sweep.py, profiles/r04_split_walk_sweeps.log): VGG-16's own 1 + 2 set (R = 4128, C = 512,
// RoIs of ~24 x 24 cells) exact 0.31-0.34 ms (any plan) -> 0.246 with 8 segments on 8x8 / 64-channel tiles (plan 7);
// one 2000-RoI image x 1024 channels 0.186 -> 0.155; two such images x 1024 channels (the alternating mode's weak
// step) are NOT chain-bound any more (exact 0.289 ms = 0.44 of peak on moved bytes, split 0.285-0.30 on that set): they
// get 4 segments for the bytes the larger tiles save (see below); more than 2048 (image, channel) pairs: the exact walk.  Beyond the chain the launch is bound by the bytes neighbouring tiles
// re-read (windows of 4.4 x 4.5 cells against 8 x 8 tiles: 2 x), which no segment count changes.
int walk_split_segments(int R, int N, int H, int W, int C) {
    if (N < 1 || N > 4 || (C & 3) || (long long)R < 1000LL * N || (long long)N * C > 3072) return 1;
    // exactly 2048 pairs (two weak images x 1024 channels, the alternating mode's weak step): with the small RoIs of one
    // saved set the exact walk on 4x4 tiles tied (0.306-0.315 ms against 0.289-0.31 inside the roofline leg), with the
    // larger ones other runs of the same untrained network propose (21 x 17 cells instead of 13 x 12) it loses to the
    // split form's 8x8 tiles: 0.40 against 0.328 (4 segments) / 0.339 (8) -- tools/probes/alter_leg_split.py
    // (measured for two images; four images x 512 channels keep the exact walk)
    // 2049 .. 3072 pairs = three images x 1024 channels (the reference's default 1 + 2 batch on a ResNet): 4 segments tie with
    // the exact walk on the default set's small RoIs (0.312 against 0.315 ms) and win on 24 x 24-cell ones (0.425 against
    // 0.4825; 8 segments 0.418 there but 0.336 on the small ones); four images x 1024 channels: the exact walk wins (0.50-0.61
    // against 0.54-0.60) -- profiles/r04_alter_leg_split.log
    const long long pairs = (long long)N * C;
    if (pairs >= 2048) return (N <= 2 || (N == 3 && pairs > 2048)) ? 4 : 1;
    return 8;
}
// the plan the split form wants (chains no longer matter: larger tiles, fewer re-read bytes)
int walk_split_plan() { return 7; }

bool walk_supported(int R, int N, int H, int W, int C, int PH, int PW) {
    if (PH > 8 || PW > 8 || (C & 1)) return false;
    const long long elems = (long long)R * PH * PW * C;
    return elems * 4 < 0xffffffffLL && (long long)N * H * W * C < 0x7fffffffLL;
}

size_t walk_workspace_bytes(int R, int N, int H, int W, int PH, int PW) {
    // sized for the smallest tiles (most records), so that every plan fits
    return carve_walk(nullptr, R, N, cdiv(H, 2), cdiv(W, 2), walk_record_bound(R, N, H, W, PH, PW, 2, 2), nullptr);
}

size_t walk_flags_offset(int R, int N, int H, int W, int PH, int PW) {
    WalkWs ws;
    carve_walk(reinterpret_cast<void *>(0x1000), R, N, 1, 1, 0, &ws);       // the head of the carving does not depend on the plan
    return (size_t)(reinterpret_cast<char *>(ws.total) - reinterpret_cast<char *>(0x1000));
}

template <int TH, int TW>
static int prepare_t(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
                     int rounding, void *workspace, size_t workspace_bytes, hipStream_t st) {
    const int tiles_h = cdiv(H, TH), tiles_w = cdiv(W, TW), tiles = tiles_h * tiles_w;
    const int items = N * tiles;
    const long long cap = walk_record_bound(R, N, H, W, PH, PW, TH, TW);
    WalkWs ws;
    if (carve_walk(workspace, R, N, tiles_h, tiles_w, cap, &ws) > workspace_bytes) return WSSDL_ERR_WORKSPACE;
    const unsigned total_elems = (unsigned)((long long)R * PH * PW * C);
    // img_span, total = 0: one contiguous region at the head of the workspace
    hipError_t e = hipMemsetAsync(ws.img_span, 0, (char *)ws.tile_slots - (char *)ws.img_span, st);
    if (e != hipSuccess) { set_last_error(e);  return WSSDL_ERR_LAUNCH; }
    if (R > 0) {
        hipLaunchKernelGGL(walk_span_kernel, dim3(cdiv(R, 256)), dim3(256), 0, st, rois, R, N, ws.img_span);
        hipLaunchKernelGGL((walk_axes_kernel<TH, TW>), dim3(cdiv((long long)R * (tiles_h + tiles_w), 256)), dim3(256), 0,
                           st, rois, R, N, H, W, PH, PW, scale, rounding, tiles_h, tiles_w, ws.rowtab, ws.coltab);
    }
    hipLaunchKernelGGL(walk_count_kernel, dim3(items), dim3(FILL_BLOCK), 0, st, rois, R, tiles_h, tiles_w,
                       ws.img_span, ws.rowtab, ws.coltab, ws.tile_slots);
    hipLaunchKernelGGL(walk_order_kernel, dim3(1), dim3(1024), 0, st, ws.tile_slots, items, ws.tile_off, ws.order,
                       ws.total);
    hipLaunchKernelGGL(walk_fill_kernel, dim3(items), dim3(FILL_BLOCK), 0, st, rois, C, PW, PH * PW, R, tiles_h,
                       tiles_w, ws.img_span, ws.rowtab, ws.coltab, ws.tile_off, ws.tile_slots, ws.slots, cap,
                       total_elems, ws.total, 0);
    return check_launch();
}

int walk_prepare(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale, int rounding,
                 void *workspace, size_t workspace_bytes, int *plan_out, hipStream_t st, int force_plan) {
    const int plan = (force_plan >= 0 && force_plan < WALK_PLANS) ? force_plan : walk_plan_choice(N, H, W, C);
    const WalkPlan &p = kWalkPlans[plan];
    int rc = WSSDL_ERR_INVALID_ARGUMENT;
#define WSSDL_PREP(TH, TW) \
    if (p.th == TH && p.tw == TW) \
        rc = prepare_t<TH, TW>(rois, R, N, H, W, C, PH, PW, scale, rounding, workspace, workspace_bytes, st);
    WSSDL_PREP(2, 2) WSSDL_PREP(2, 4) WSSDL_PREP(3, 4) WSSDL_PREP(4, 4) WSSDL_PREP(4, 8) WSSDL_PREP(5, 7)
    WSSDL_PREP(5, 8) WSSDL_PREP(6, 6) WSSDL_PREP(6, 7) WSSDL_PREP(6, 8) WSSDL_PREP(7, 6) WSSDL_PREP(7, 7)
    WSSDL_PREP(8, 8)
#undef WSSDL_PREP
    if (rc == WSSDL_OK && plan_out) *plan_out = plan;
    return rc;
}

// partial sums of the split form: bottom_diff += partial[0] + partial[1] + ... in that order (4 channels per thread)
__global__ __launch_bounds__(256) void walk_combine_kernel(float *__restrict__ bottom_diff, const float *__restrict__ partial,
                                                           long long n4, int extra, unsigned long long seg_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4v o = reinterpret_cast<const float4v *>(bottom_diff)[i];
    for (int s = 0; s < extra; ++s) {
        const float4v p = reinterpret_cast<const float4v *>(partial + (size_t)s * seg_stride)[i];
        o.x = o.x + p.x;  o.y = o.y + p.y;  o.z = o.z + p.z;  o.w = o.w + p.w;
    }
    reinterpret_cast<float4v *>(bottom_diff)[i] = o;
}

// cell / W by one 24-bit multiply for every cell of the map (as in roi_pool.hip: launch_bwd), C = 2^cshift
static bool walk_i32_params(int H, int W, int C, WalkI32 *q) {
    int cshift = 0;
    while ((1 << cshift) < C) ++cshift;
    if ((1 << cshift) != C || (long long)H * W >= (1 << 16)) return false;
    const unsigned cells = (unsigned)H * (unsigned)W;
    for (unsigned sft = 8; sft <= 24; ++sft) {
        const unsigned long long mg = ((1ULL << sft) + (unsigned)W - 1) / (unsigned)W;
        if (mg >= (1ULL << 24) || mg * (cells ? cells - 1 : 0) >= (1ULL << 32)) continue;
        bool exact = true;
        for (unsigned n = 0; n < cells && exact; ++n) exact = (unsigned)((n * mg) >> sft) == n / (unsigned)W;
        if (exact) {
            q->cshift = cshift;  q->magic = (unsigned)mg;  q->shift = sft;
            return true;
        }
    }
    return false;
}

// the plans the i32 form of the walk is instantiated for (launch_walk_t: I32_BUILT)
bool walk_i32_plan_built(int id) {
    return id == 11 || id == 5 || id == 23 || id == 18 || id == 22 || id == 19 || id == 21 || id == 9 || id == 12 ||
           id == 17 || id == 4 || id == 13 || id == 14;
}

bool walk_i32_supported(int R, int N, int H, int W, int C, int PH, int PW) {
    WalkI32 q;
    return walk_supported(R, N, H, W, C, PH, PW) && walk_i32_params(H, W, C, &q);
}

template <int ID, int TH, int TW, int DEPTH, int MINW, int CPL>
static int launch_walk_t(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C,
                         int PH, int PW, float *bottom_diff, void *workspace, size_t workspace_bytes,
                         hipStream_t st, int nseg, float *partial, bool i32) {
    const int tiles_h = cdiv(H, TH), tiles_w = cdiv(W, TW), tiles = tiles_h * tiles_w;
    const int items = N * tiles;
    WalkWs ws;
    if (carve_walk(workspace, R, N, tiles_h, tiles_w, walk_record_bound(R, N, H, W, PH, PW, TH, TW), &ws) > workspace_bytes)
        return WSSDL_ERR_WORKSPACE;
    const unsigned total_elems = (unsigned)((long long)R * PH * PW * C);
    const int G = cdiv(C, 64 * CPL);
    long long blocks;
    if ((G & 7) == 0) blocks = (long long)items * G;
    else if (G < 8 && (8 % G) == 0) blocks = 8LL * cdiv(items, 8 / G);
    else blocks = (long long)items * G;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    const unsigned long long seg_stride = (unsigned long long)N * H * W * C;
    WalkI32 q = {0, 0u, 0u};
    // (the i32 form is only built for the plans walk_plan_auto can pick)
    constexpr bool I32_BUILT = ID == 11 || ID == 5 || ID == 23 || ID == 18 || ID == 22 || ID == 19 || ID == 21 ||
                               ID == 9 || ID == 12 || ID == 17 || ID == 4 || ID == 13 || ID == 14;      // == walk_i32_plan_built(ID)
    if (i32 && !I32_BUILT) return WSSDL_ERR_INVALID_ARGUMENT;
    if constexpr (I32_BUILT) if (i32) {
        if (!walk_i32_params(H, W, C, &q)) return WSSDL_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL((roi_pool_bwd_walk_kernel<TH, TW, DEPTH, MINW, CPL, true>), dim3((unsigned)blocks, (unsigned)nseg),
                           dim3(64), 0, st, top_diff, arg8, reinterpret_cast<const unsigned *>(ws.slots), ws.tile_off,
                           ws.tile_slots, ws.order, items, tiles_w, tiles, G, H, W, C, total_elems, bottom_diff, nseg, partial,
                           seg_stride, q, TH, TW);
    }
    if (!i32)
    hipLaunchKernelGGL((roi_pool_bwd_walk_kernel<TH, TW, DEPTH, MINW, CPL, false>), dim3((unsigned)blocks, (unsigned)nseg), dim3(64),
                       0, st, top_diff, arg8, reinterpret_cast<const unsigned *>(ws.slots), ws.tile_off, ws.tile_slots,
                       ws.order, items, tiles_w, tiles, G, H, W, C, total_elems, bottom_diff, nseg, partial, seg_stride, q, TH, TW);
    if (nseg > 1) {
        const long long n4 = (long long)(seg_stride / 4);          // C is even and walk_split_supported asks C % 4 == 0
        hipLaunchKernelGGL(walk_combine_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, bottom_diff, partial, n4,
                           nseg - 1, seg_stride);
    }
    return check_launch();
}

int launch_walk(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C, int PH,
                int PW, float *bottom_diff, void *workspace, size_t workspace_bytes, int plan, hipStream_t st,
                int nseg, float *partial, bool i32) {
    if (nseg < 1 || nseg > WALK_MAX_SEGMENTS || (nseg > 1 && (!partial || (C & 3)))) return WSSDL_ERR_INVALID_ARGUMENT;
    switch (plan) {
#define WSSDL_X(ID, TH, TW, D, MW, CPL) \
        case ID: return launch_walk_t<ID, TH, TW, D, MW, CPL>(top_diff, arg8, R, N, H, W, C, PH, PW, bottom_diff, workspace, \
                                                         workspace_bytes, st, nseg, partial, i32);
        WSSDL_WALK_PLANS(WSSDL_X)
#undef WSSDL_X
        default: return WSSDL_ERR_INVALID_ARGUMENT;
    }
}


// ---- bin-owner form: plans, prepare, launch ----------------------------------------------------
//   X(id, region h, region w, tile h, tile w, records in flight, waves per SIMD asked for, channels per lane, cache policy: bits 0-4 of the data loads (gfx950: 1 = sc0, 2 = nt, 16 = sc1), 64 = non-temporal stores, 128 = lean decode)
#define WSSDL_OWNER_PLANS(X) \
    X(0, 6, 7, 4, 5, 2, 1, 2, 66)  /* the default: 128-channel waves, 21.5 KiB of LDS (7 waves per CU), halo 2 x 2 */ \
    X(1, 6, 6, 4, 4, 2, 2, 2, 66)  /* at the exact walk's LDS (18.5 KiB) */ \
    X(2, 8, 8, 6, 6, 3, 2, 1, 66)  /* 64-channel waves, 16.25 KiB: the fewest bytes (1.20 x moved) */ \
    X(3, 7, 8, 5, 6, 2, 2, 1, 66)  \
    X(4, 6, 6, 4, 4, 2, 2, 2, 0)   /* plan 1 with plain loads and stores (the A/B of the cache policy) */ \
    X(5, 7, 7, 5, 5, 2, 1, 2, 66)  /* 25 KiB: 6 waves per CU */ \
    X(6, 5, 5, 4, 4, 2, 2, 2, 66)  /* halo 1: 13.3 KiB, 12 waves per CU */ \
    X(7, 6, 7, 5, 5, 2, 1, 2, 66)  /* halo 1 x 2 */ \
    X(8, 6, 7, 4, 5, 2, 1, 2, 66 + 128)  /* plan 0 with the lean decode (intervals in code space) */ \
    X(9, 6, 6, 4, 4, 2, 2, 2, 66 + 128)  /* plan 1 ... */ \
    X(10, 6, 7, 4, 5, 3, 1, 2, 66 + 128) \
    X(11, 7, 7, 5, 5, 2, 1, 2, 66 + 128)

struct OwnerPlan {
    int rh, rw, sh, sw, depth, minw, cpl;
};
static const OwnerPlan kOwnerPlans[] = {
#define WSSDL_X(ID, RH, RW, SH, SW, D, MW, CPL, AUX) {RH, RW, SH, SW, D, MW, CPL},
    WSSDL_OWNER_PLANS(WSSDL_X)
#undef WSSDL_X
};
constexpr int OWNER_PLANS = sizeof(kOwnerPlans) / sizeof(kOwnerPlans[0]);

int owner_plan_count() { return OWNER_PLANS; }

// Which launches take the owner form (-1: keep the exact walk / the split form).  Measured on subsets and copies of the
// default workload's fixed RoI set (tools/owner_ab.sh, profiles/r05_owner_ab.log; exact = the faster of plans 11 / 13 / 23,
// split = its best segment count):
//   8 images x 1024 channels (the default launch)  exact 0.52   owner 0.47 (plan 0)      16 images: 1.01 -> 0.96
//   8 x 512                                         exact 0.34   owner 0.24               8 x 2048: 1.00 -> 0.95
//   4 weak + 1 / 4 weak / 4 sup + 2 weak x 1024     0.51 / 0.51 / 0.34 -> 0.43 / 0.42 / 0.27
//   2 weak x 1024 (alternating mode's weak step)    split 0.29   owner 0.22
//   1 + 2 images x 1024 (reference's default batch) split 0.31   owner 0.23
//   1 + 2 x 512 (VGG-16) / 1 weak x 1024            split 0.18 / 0.16   owner 0.17 / 0.16 (plan 1: 4x4 tiles)
//   2 weak x 256 (ResNet-18)                        split 0.10   owner 0.16   -> not taken
//   2 supervised images (R = 256)                   exact 0.03   owner 0.04   -> not taken
int owner_plan_auto(int R, int N, int H, int W, int C) {
    const int v = tuning().roi_bwd_owner;
    if (v >= 0 && v < OWNER_PLANS) return v;
    if (v < -1) return -1;                                       // -2: never (tools, A/B runs)
    const long long pairs = (long long)N * C;
    // (1024-2047 pairs -- VGG-16's 1 + 2 x 512, one weak image x 1024 -- stay on the split form: on the default set's
    // small RoIs owner plan 9 ties with it (0.17 / 0.16 against 0.18 / 0.16 ms), on VGG-16's own 24 x 24-cell proposals it
    // loses a little (0.257 against 0.245) and the plain 4x4-tile plan 1 a lot in the bench's leg (0.40 against 0.28): windows of
    // 4-5 cells do not fit a 6x6 region and are listed in chains.  From 2048 pairs on plan 8 (6x7 regions, lean decode) wins on
    // every set measured, large proposals included: alternating weak step 0.288 -> 0.223 on 21 x 17-cell RoIs.)
    // Round 6: below 2048 pairs the owner form runs with TWO waves per tile stream (owner_split_segments) and then beats the
    // split form it used to lose to or tie with (profiles/r06_owner_split_ab.log; owner 8 / 9 x 2 segments against split
    // plan 7 x 8 segments): VGG-16's own proposals 0.199 against 0.265 ms, 3 x 512 / 1 x 1024 / 2 x 512 / 4 x 256 on the
    // default set's small RoIs 0.150 / 0.131 / 0.135 / 0.135 against 0.207 / 0.182 / 0.181 / 0.183, 2 x 256 and 1 x 512
    // 0.086 / 0.085 (plan 9) against 0.103 / 0.102.  Below 512 pairs nothing was measured: the split form / exact walk stay.
    if (R < 1536 || N < 1 || (C & 127) || pairs < 512) return -1;
    return pairs >= 1024 ? 8 : 9;
}

bool owner_supported(int R, int N, int H, int W, int C, int PH, int PW) {
    return walk_supported(R, N, H, W, C, PH, PW) && (C & 3) == 0;
}

size_t owner_scratch_bytes(int N, int H, int W, int C, int plan, int nseg) {
    if (plan < 0 || plan >= OWNER_PLANS || N < 1 || H < 1 || W < 1 || C < 1 || nseg < 1 || nseg > WALK_MAX_SEGMENTS) return 0;
    const OwnerPlan &p = kOwnerPlans[plan];
    return (size_t)nseg * N * cdiv(H, p.sh) * cdiv(W, p.sw) * p.rh * p.rw * (size_t)C * sizeof(float);
}

// How many segments the owner form cuts a tile's stream into (1: the plain owner form).  The owner walk is one wave per
// (image, tile, 128 channels): 2 images x 1024 channels are 2080 waves of very unequal length on 1024 SIMDs, 3 x 512
// are 1560 -- such launches wait for their longest chains, and several waves per stream shorten them; what it costs is
// a region buffer per segment and a merge pass over all of it.  Measured: profiles/r06_owner_split_ab.log.
int owner_split_segments(int R, int N, int H, int W, int C) {
    const int v = tuning().roi_bwd_owner_segments;
    if (v >= 1 && v <= WALK_MAX_SEGMENTS) return v;
    // two from 512 to 2047 pairs (see owner_plan_auto); from 2048 pairs on a second wave per stream only adds the merge:
    // 2 x 1024 0.266 -> 0.279 ms, 3 x 1024 0.329 -> 0.346, 3 x 768 0.210 -> 0.216; three and more segments lose everywhere
    // but 8 x 4 segments on 512 pairs (0.087, a tie with 9 x 2)
    return (long long)N * C < 2048 ? 2 : 1;
}

template <int RH, int RW, int SH, int SW>
static int prepare_own_t(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale,
                         int rounding, void *workspace, size_t workspace_bytes, hipStream_t st, bool lean) {
    const int tiles_h = cdiv(H, SH), tiles_w = cdiv(W, SW), tiles = tiles_h * tiles_w;
    const int items = N * tiles;
    const long long cap = walk_record_bound(R, N, H, W, PH, PW, SH, SW);      // a chain lists a bin no more often than the exact form
    WalkWs ws;
    if (carve_walk(workspace, R, N, tiles_h, tiles_w, cap, &ws) > workspace_bytes) return WSSDL_ERR_WORKSPACE;
    const unsigned total_elems = (unsigned)((long long)R * PH * PW * C);
    hipError_t e = hipMemsetAsync(ws.img_span, 0, (char *)ws.tile_slots - (char *)ws.img_span, st);
    if (e != hipSuccess) { set_last_error(e);  return WSSDL_ERR_LAUNCH; }
    if (R > 0) {
        hipLaunchKernelGGL(walk_span_kernel, dim3(cdiv(R, 256)), dim3(256), 0, st, rois, R, N, ws.img_span);
        hipLaunchKernelGGL((walk_axes_own_kernel<RH, RW, SH, SW>), dim3(cdiv((long long)R * (tiles_h + tiles_w), 256)),
                           dim3(256), 0, st, rois, R, N, H, W, PH, PW, scale, rounding, tiles_h, tiles_w, ws.rowtab, ws.coltab);
    }
    hipLaunchKernelGGL(walk_count_kernel, dim3(items), dim3(FILL_BLOCK), 0, st, rois, R, tiles_h, tiles_w,
                       ws.img_span, ws.rowtab, ws.coltab, ws.tile_slots);
    hipLaunchKernelGGL(walk_order_kernel, dim3(1), dim3(1024), 0, st, ws.tile_slots, items, ws.tile_off, ws.order,
                       ws.total);
    hipLaunchKernelGGL(walk_fill_kernel, dim3(items), dim3(FILL_BLOCK), 0, st, rois, C, PW, PH * PW, R, tiles_h,
                       tiles_w, ws.img_span, ws.rowtab, ws.coltab, ws.tile_off, ws.tile_slots, ws.slots, cap,
                       total_elems, ws.total, lean ? RW : 0);
    return check_launch();
}

int owner_prepare(const float *rois, int R, int N, int H, int W, int C, int PH, int PW, float scale, int rounding,
                  void *workspace, size_t workspace_bytes, int plan, hipStream_t st) {
    switch (plan) {
#define WSSDL_X(ID, RH, RW, SH, SW, D, MW, CPL, AUX) \
        case ID: return prepare_own_t<RH, RW, SH, SW>(rois, R, N, H, W, C, PH, PW, scale, rounding, workspace, workspace_bytes, st, \
                                                      ((AUX) & 128) != 0 && (C % (64 * CPL)) == 0);
        WSSDL_OWNER_PLANS(WSSDL_X)
#undef WSSDL_X
        default: return WSSDL_ERR_INVALID_ARGUMENT;
    }
}

template <int ID, int RH, int RW, int SH, int SW, int DEPTH, int MINW, int CPL, int AUX>
static int launch_owner_t(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C, int PH,
                          int PW, float *bottom_diff, void *workspace, size_t workspace_bytes, float *halo, hipStream_t st,
                          bool i32, int nseg) {
    const int tiles_h = cdiv(H, SH), tiles_w = cdiv(W, SW), tiles = tiles_h * tiles_w;
    const int items = N * tiles;
    if (nseg < 1 || nseg > WALK_MAX_SEGMENTS || (i32 && nseg != 1)) return WSSDL_ERR_INVALID_ARGUMENT;
    WalkWs ws;
    if (carve_walk(workspace, R, N, tiles_h, tiles_w, walk_record_bound(R, N, H, W, PH, PW, SH, SW), &ws) > workspace_bytes)
        return WSSDL_ERR_WORKSPACE;
    const unsigned total_elems = (unsigned)((long long)R * PH * PW * C);
    const int G = cdiv(C, 64 * CPL);
    long long blocks;
    if ((G & 7) == 0) blocks = (long long)items * G;
    else if (G < 8 && (8 % G) == 0) blocks = 8LL * cdiv(items, 8 / G);
    else blocks = (long long)items * G;
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    WalkI32 q = {0, 0u, 0u};
    // (the i32 arg-max -- the reference op's own layout -- is built for owner plans 0 and 1: tools/bwd_fixed_sweep.py --i32-owner)
    constexpr bool I32_BUILT = ID == 0 || ID == 1;
    if (i32 && !I32_BUILT) return WSSDL_ERR_INVALID_ARGUMENT;
    if constexpr (I32_BUILT) if (i32) {
        if (!walk_i32_params(H, W, C, &q)) return WSSDL_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL((roi_pool_bwd_walk_kernel<RH, RW, DEPTH, MINW, CPL, true, true, AUX>), dim3((unsigned)blocks, 1u), dim3(64),
                           0, st, top_diff, arg8, reinterpret_cast<const unsigned *>(ws.slots), ws.tile_off, ws.tile_slots,
                           ws.order, items, tiles_w, tiles, G, H, W, C, total_elems, bottom_diff, 1, halo, 0ull, q, SH, SW);
    }
    if (!i32)
    hipLaunchKernelGGL((roi_pool_bwd_walk_kernel<RH, RW, DEPTH, MINW, CPL, false, true, AUX>), dim3((unsigned)blocks, (unsigned)nseg), dim3(64),
                       0, st, top_diff, arg8, reinterpret_cast<const unsigned *>(ws.slots), ws.tile_off, ws.tile_slots,
                       ws.order, items, tiles_w, tiles, G, H, W, C, total_elems, bottom_diff, nseg, halo, 0ull, q, SH, SW);
    if (nseg > 1) {
        const long long cells = (long long)N * H * W;
        if (cells > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(walk_merge_split_kernel, dim3((unsigned)cells), dim3(256), 0, st, bottom_diff, halo, N, H, W, C / 4,
                           tiles_h, tiles_w, RH, RW, SH, SW, nseg);
        return check_launch();
    }
    constexpr int HH = RH - SH, HW = RW - SW;
    constexpr int ncell = SH * SW - (SH - HH) * (SW - HW);
    if (ncell > 0)
        hipLaunchKernelGGL(walk_merge_kernel, dim3((unsigned)items, (unsigned)ncell), dim3(256), 0, st, bottom_diff, halo, H, W,
                           C / 4, tiles_h, tiles_w, RH, RW, SH, SW);
    return check_launch();
}

int launch_owner(const float *top_diff, const unsigned char *arg8, int R, int N, int H, int W, int C, int PH, int PW,
                 float *bottom_diff, void *workspace, size_t workspace_bytes, int plan, float *halo, hipStream_t st, bool i32,
                 int nseg) {
    switch (plan) {
#define WSSDL_X(ID, RH, RW, SH, SW, D, MW, CPL, AUX) \
        case ID: return launch_owner_t<ID, RH, RW, SH, SW, D, MW, CPL, AUX>(top_diff, arg8, R, N, H, W, C, PH, PW, bottom_diff, workspace, \
                                                                workspace_bytes, halo, st, i32, nseg);
        WSSDL_OWNER_PLANS(WSSDL_X)
#undef WSSDL_X
        default: return WSSDL_ERR_INVALID_ARGUMENT;
    }
}

}  // namespace wssdl
