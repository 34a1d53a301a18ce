// RoI max-pooling forward (training path, 1-byte arg-max) for LARGE, heavily overlapping bin windows: per-step
// block-maximum tables.
//
// The rows kernel (roi_pool_compact.hip) visits every cell of every bin window: its work is sum(window area) x C at
// ~3 vector instructions per cell and channel, and the 2000 proposals of a weak image re-scan the same cells hundreds
// of times (alternating weak step: 13 cells per bin, VGG-16's proposals: 18).  Here the feature map is reduced ONCE
// per step to three tables -- for every cell (h, w) the first maximum of the k x k block whose top-left corner it is,
// k = 2, 3, 4 -- and a bin whose window is a x b cells with k <= min(a, b) and max(a, b) <= 2k is the union of the four
// (overlapping) k x k blocks anchored at the window's corners: four table reads instead of a x b cells.
//
// Bit-exactness.  The reference scans a window in (h, w) order and keeps a cell only when it is strictly greater
// than the running maximum started at -FLT_MAX (roi_pooling_op_gpu.cu.cc:66-79): the result is the maximum value and,
// among the cells that hold it, the one with the smallest (h, w) -- "minimum flat index among the maxima", which is
// associative and idempotent, so it may be evaluated block by block in any order and over overlapping blocks.  A
// table entry is (value, dh << 4 | dw) of its block's first maximum by exactly that scan; the merge takes the maximum
// of the four values and the smallest window-relative code among the entries that hold it.  Cells that never pass the
// scan (NaN, -inf, -FLT_MAX) are skipped by the table scan like by the reference's; a block with no passing cell has
// value -FLT_MAX, and a bin whose merged maximum is -FLT_MAX is the reference's "no arg-max" (-1 / code 0xff).
// +0.0 and -0.0 compare equal in the reference's scan but v_max_f32 has to pick one: the table build raises a flag
// when the map holds a -0.0 anywhere and the pooling kernel then takes the cell scan for every bin (correct, slow).
// Bins the blocks do not cover (one-cell-wide windows, a > 2b, ...) take the cell scan, as do all bins of a bin row
// whose window table entry says so (word 6 of the entry, roi_windows_kernel).
//
// L2.  Three tables of 5 bytes per element are 3.75 x the map: one image's 256-channel slice is 9 MB, an XCD's L2 holds
// 4.  The bin rows are therefore walked in the order of (image, first window row) -- a counting sort of the window
// table's keys -- so that the waves in flight on an XCD read a band of ~10 rows of the tables.
//
// Launches: roi_windows (window table; clears the sort's counters), blocks_build (map -> tables; counts the sort keys),
// rows_scatter (order), roi_pool_fwd_blocks (the pooling).
#include "roi_pool.hip.h"

#include <type_traits>

namespace wssdl {

constexpr int BLK_TABLES = 3;          // block sizes 2, 3, 4
constexpr int ORDER_SUBS = 64;         // counters per sort key (see blocks_build_kernel)

struct BlocksLayout {
    float *values;                     // [N][3][H][W][C] f32
    unsigned char *codes;              // [N][3][H][W][C] u8: dh << 4 | dw of the block's first maximum
    unsigned *order;                   // [R * 7] bin rows sorted by key
    unsigned *hist;                    // [N * H][ORDER_SUBS] counts
    unsigned *cursor;                  // [N * H][ORDER_SUBS] places handed out
    unsigned *flags;                   // [0]: the map holds a -0.0
    size_t bytes;
};

static BlocksLayout blocks_layout(void *ws, int R, int N, int H, int W, int C) {
    Carver cv(ws);
    BlocksLayout L;
    const size_t elems = (size_t)N * BLK_TABLES * H * W * C;
    L.values = cv.take<float>(elems);
    L.codes = cv.take<unsigned char>(elems);
    L.order = cv.take<unsigned>((size_t)R * 7);
    L.hist = cv.take<unsigned>((size_t)N * H * ORDER_SUBS);
    L.cursor = cv.take<unsigned>((size_t)N * H * ORDER_SUBS);
    L.flags = cv.take<unsigned>(4);
    L.bytes = cv.off;
    return L;
}

// the counters and flags a call sequence starts from zeroed (adjacent slices of the workspace)
void blocks_zero_region(void *ws, int R, int N, int H, int W, int C, unsigned **ptr, int *words) {
    const BlocksLayout L = blocks_layout(ws, R, N, H, W, C);
    *ptr = L.hist;
    *words = (int)((reinterpret_cast<char *>(L.flags) - reinterpret_cast<char *>(L.hist)) / 4) + 4;
}

bool blocks_supported(int R, int N, int H, int W, int C, int PH, int PW) {
    if (PH != 7 || PW != 7 || R < 1 || N < 1 || H < 4 || W < 4 || H > 255 || W > 255) return false;
    if (C % 256 != 0) return false;
    const int slices = C / 256;
    if (!((slices >= 8 && slices % 8 == 0) || (slices < 8 && 8 % slices == 0))) return false;
    if ((long long)BLK_TABLES * H * W * C * 4 >= 0x7fffffffLL) return false;       // one buffer resource per image
    if ((long long)N * H * W * C * 4 >= 0x7fffffffLL) return false;                    // ... and one over the map (table build)
    if ((long long)R * 7 >= 0x7fffffffLL || (long long)N * H > (1 << 16)) return false;
    // the 1-byte code needs every window inside 15 x 16 cells (compact_supported's rule)
    return cdiv(H + 1, PH) + 1 <= ARG8_MAX_WIN_H && cdiv(W + 1, PW) + 1 <= ARG8_MAX_WIN_W;
}

// ------------------------------------------------------------------ tables ---
// One lane = 4 channels of one cell (h, w); it reads the 4 x 4 cells below / right of it row by row.  Per row the
// first maxima of its first 2, 3, 4 cells (one scan, strict >), then each block's accumulator takes the row when it
// is strictly greater: rows in order and strict > again, so a block's entry is its first maximum in (h, w) order.
// Cells past the border are clamped duplicates (they come later in the scan and never pass the strict >); entries whose
// block leaves the map are never read (a window lies inside the map) and not written.
__global__ __launch_bounds__(256) void blocks_build_kernel(const float *__restrict__ bottom, int N, int H, int W, int C,
                                                           float *__restrict__ values, unsigned *__restrict__ codes4,
                                                           unsigned *__restrict__ flags, const unsigned *__restrict__ table,
                                                           int items, int hist_blocks, unsigned *__restrict__ hist) {
    // the first workgroups count the sort keys of the bin rows (independent of the tables: it rides along in this
    // launch instead of waiting for it).  A key owns ORDER_SUBS counters and a row uses the one of its lane: with one
    // counter per key ~400 atomics queue on each address (33 us, measured).
    if ((int)blockIdx.x < hist_blocks) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < items) {
            const unsigned key = min(table[(size_t)i * WIN_ENTRY_WORDS + 7], (unsigned)(N * H - 1));
            atomicAdd(&hist[key * ORDER_SUBS + (threadIdx.x & (ORDER_SUBS - 1))], 1u);
        }
        return;
    }
    const int lanes_per_cell = C >> 2;
    const long long unit = (long long)(blockIdx.x - hist_blocks) * 256 + threadIdx.x;
    const long long cell = unit / lanes_per_cell;
    if (cell >= (long long)N * H * W) return;
    const int c0 = (int)(unit - cell * lanes_per_cell) * 4;
    const int n = (int)(cell / (H * W)), hw = (int)(cell - (long long)n * H * W);
    const int h = hw / W, w = hw - h * W;
    // (one buffer resource over the whole tensor, 32-bit byte offsets: a 64-bit address per load costs 32 VGPRs here)
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bottom), 0, N * H * W * C * 4, 0x00020000);
    int ro[4], co[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ro[i] = ((n * H + min(h + i, H - 1)) * W * C + c0) * 4;
        co[i] = min(w + i, W - 1) * C * 4;
    }
    float4v acc[BLK_TABLES];
    unsigned ac[BLK_TABLES][4];
#pragma unroll
    for (int t = 0; t < BLK_TABLES; ++t) {
        acc[t] = (float4v)(-FLT_MAX);
#pragma unroll
        for (int k = 0; k < 4; ++k) ac[t][k] = 0u;
    }
    // two rows of cells in flight (the scheduler would hoist all 16 loads: 96 VGPRs, 5 waves per SIMD)
    float4v v[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[0][j] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rb, ro[0] + co[j], 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[(i + 1) & 1][j] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rb, ro[i + 1] + co[j], 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) {
            bool neg_zero = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) neg_zero |= __float_as_uint(v[0][0][k]) == 0x80000000u;
            if (neg_zero) atomicOr(flags, 1u);
        }
        float4v pv = (float4v)(-FLT_MAX);          // first maximum of the row's cells 0..j, code i << 4 | j
        unsigned pc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[i & 1][j][k] > pv[k]) { pv[k] = v[i & 1][j][k];  pc[k] = (unsigned)(i << 4 | j); }
            // the row's first j + 1 cells are row i of the (j + 1)-blocks (tables j - 1) that have more than i rows
            if (j >= 1 && i <= j) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (pv[k] > acc[j - 1][k]) { acc[j - 1][k] = pv[k];  ac[j - 1][k] = pc[k]; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < BLK_TABLES; ++t) {
        const int B = t + 2;
        if (h + B > H || w + B > W) continue;
        const size_t o = (((size_t)n * BLK_TABLES + t) * H * W + hw) * C + c0;
        *reinterpret_cast<float4v *>(values + o) = acc[t];
        codes4[o >> 2] = ac[t][0] | (ac[t][1] << 8) | (ac[t][2] << 16) | (ac[t][3] << 24);
    }
}

// ------------------------------------------------------------------- order ---
// Counting sort of the bin rows by the key in word 7 of their window-table entry: the counts come from the first
// workgroups of blocks_build_kernel, then one scatter launch.
// Every workgroup scans the counters for itself (a thread sums a contiguous chunk, the chunk sums are scanned in LDS, a
// row's base = its chunk's prefix + the counters in front of it inside the chunk): a one-workgroup scan kernel between
// the count and the scatter took 6.5 us, longer than either.  The counts stay read-only; the rows of a (key, lane)
// counter take their places from a cursor array of the same shape.
__global__ __launch_bounds__(256) void rows_scatter_kernel(const unsigned *__restrict__ table, int items, int keys,
                                                           const unsigned *__restrict__ hist, unsigned *__restrict__ cursor,
                                                           unsigned *__restrict__ order) {
    __shared__ unsigned part[256];
    const int n = keys * ORDER_SUBS;
    const int per = (n + 255) / 256;
    const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
    unsigned sum = 0u;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned t = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const unsigned key = min(table[(size_t)i * WIN_ENTRY_WORDS + 7], (unsigned)(keys - 1));
    // (same thread <-> counter mapping as the counting workgroups; the order inside a counter only moves work around)
    const int idx = (int)key * ORDER_SUBS + (threadIdx.x & (ORDER_SUBS - 1));
    const int chunk = idx / per;
    unsigned base = chunk > 0 ? part[chunk - 1] : 0u;
    for (int j = chunk * per; j < idx; ++j) base += hist[j];
    const unsigned pos = base + atomicAdd(&cursor[idx], 1u);
    if (pos < (unsigned)items) order[pos] = (unsigned)i;
}

// ----------------------------------------------------------------- pooling ---
// One wave = one (roi, ph) bin row x 256 channels, windows from the table of roi_windows_kernel (scalar loads), like
// roi_pool_fwd_rows_kernel<4, 4, 7, false, true>; blockIdx % 8 <-> channel slice as there.
template <int RPW, int PARTS /* waves that share a bin row: 1 = all 7 bins, 2 = bins 0..3 / 4..6, 4 = 0..1 / 2..3 / 4..5 / 6, 7 = a bin each */>
__global__ __launch_bounds__(64 * RPW) void roi_pool_fwd_blocks_kernel(
    const float *__restrict__ bottom, int N, int H, int W, int C, int R, float *__restrict__ top,
    unsigned char *__restrict__ arg8, int slices, const unsigned *__restrict__ table,
    const unsigned *__restrict__ order /* or NULL: RoI order */, const float *__restrict__ blkv,
    const unsigned char *__restrict__ blkc, const unsigned *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int sh = 31 - __builtin_clz(slices);
    int slice, group;
    if (slices >= 8) {
        const int per = slices >> 3;
        if ((slices & (slices - 1)) == 0) { slice = xcd + 8 * (q & (per - 1));  group = q >> (sh - 3); }
        else { slice = xcd + 8 * (q % per);  group = q / per; }
    } else {       // 1, 2 or 4 slices
        slice = xcd & (slices - 1);
        group = (q << (3 - sh)) + (xcd >> sh);
    }
    const long long widx = (long long)group * RPW + wave;
    if (widx >= (long long)R * 7 * PARTS) return;
    const long long idx = widx / PARTS;
    const int part = (int)(widx - idx * PARTS);
    const int pw_lo = (part * 7 + PARTS - 1) / PARTS, pw_hi = ((part + 1) * 7 + PARTS - 1) / PARTS;
    // (clamped: an order built on counters that were not cleared -- the calls out of sequence -- must not turn into a read
    // outside the window table)
    const int item = order ? (int)min(order[idx], (unsigned)(R * 7 - 1)) : (int)idx;
    const unsigned *e = table + (size_t)item * WIN_ENTRY_WORDS;
    const int batch = (int)e[0];
    const int hs = (int)(e[1] & 0xffu), he = (int)((e[1] >> 8) & 0xffu);
    const unsigned t_ws0 = e[2], t_ws1 = e[3], t_we0 = e[4], t_we1 = e[5];
    const unsigned blk = flags[0] != 0u ? 0u : e[6];        // a -0.0 in the map: cell scans only (see the header)
    const bool bad = batch < 0 || batch >= N;
    const bool row_dead = (he <= hs) || bad;
    const int c0 = (slice * 64 + lane) * 4;
    const int voff = c0 * 4;
    const int cell_bytes = C * 4;
    const int HWC = H * W * C;
    const size_t img = (size_t)(bad ? 0 : batch);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(bottom + img * HWC), 0, HWC * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(blkv + img * BLK_TABLES * HWC), 0, BLK_TABLES * HWC * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(blkc + img * BLK_TABLES * HWC), 0, BLK_TABLES * HWC, 0x00020000);
    const size_t o_row = (size_t)item * 7 * C + c0;

    float4v res[7];
    unsigned codes[7];
    int wss[7], wes[7], ks[7];
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
        wss[pw] = (int)(((pw < 4 ? t_ws0 : t_ws1) >> (8 * (pw & 3))) & 0xffu);
        wes[pw] = (int)(((pw < 4 ? t_we0 : t_we1) >> (8 * (pw & 3))) & 0xffu);
        ks[pw] = (row_dead || wes[pw] <= wss[pw]) ? -1 : (int)((blk >> (2 * pw)) & 3u);       // -1 empty, 0 cell scan, 1..3 blocks
    }
    // k x k blocks anchored at the window's corners: one when the window is k x k, two when one side is k, else four.
    // (Reading the coinciding blocks twice -- branch-free, four reads always -- was measured: the pooling kernel is
    // bound by the bytes it pulls through the L1s, ~21 TB/s, and 40 % of them were such duplicates.)
    auto block_bin = [&](int pw, auto ncand) {
        constexpr int NC = decltype(ncand)::value;             // 1, 2 (side by side), 3 (= 2, one above the other), 4
        const int ws = wss[pw], we = wes[pw], k = ks[pw];
        const int bs = k + 1;
        const int bh1 = he - bs, bw1 = we - bs;
        const int tb = (k - 1) * HWC;
        const int e00 = tb + (hs * W + ws) * C;
        const int e1 = NC == 3 ? tb + (bh1 * W + ws) * C : tb + (hs * W + bw1) * C;
        const int e10 = tb + (bh1 * W + ws) * C, e11 = tb + (bh1 * W + bw1) * C;
        float4v v[4];
        unsigned q[4];
        v[0] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, e00 * 4, 0));
        q[0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, c0, e00, 0);
        if (NC >= 2) {
            v[1] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, e1 * 4, 0));
            q[1] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, c0, e1, 0);
        }
        if (NC == 4) {
            v[2] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, e10 * 4, 0));
            v[3] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, e11 * 4, 0));
            q[2] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, c0, e10, 0);
            q[3] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rc, c0, e11, 0);
        }
        // block-relative -> window-relative codes, four channels at a time (dh + offset <= 14, dw + offset <= 15: no carry)
        const unsigned a01 = (unsigned)(bw1 - ws) * 0x01010101u, a10 = ((unsigned)(bh1 - hs) << 4) * 0x01010101u;
        if (NC == 2) q[1] += a01;
        if (NC == 3) q[1] += a10;
        if (NC == 4) { q[1] += a01;  q[2] += a10;  q[3] += a10 + a01; }
        float4v mv;
        unsigned packed = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m;
            unsigned code;
            if (NC == 1) {
                m = v[0][j];
                code = (q[0] >> (8 * j)) & 0xffu;
            } else if (NC == 4) {
                // (v_max3 / v_max written out: fmaxf puts an sNaN-quieting v_max x, x, x in front of every loaded operand --
                // four instructions instead of two per channel; the tables hold no NaN)
                float m3;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(v[0][j]), "v"(v[1][j]), "v"(v[2][j]));
                asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m3), "v"(v[3][j]));
                const unsigned u0 = (q[0] >> (8 * j)) & 0xffu, u1 = (q[1] >> (8 * j)) & 0xffu;
                const unsigned u2 = (q[2] >> (8 * j)) & 0xffu, u3 = (q[3] >> (8 * j)) & 0xffu;
                const unsigned s0 = v[0][j] == m ? u0 : 0x1ffu, s1 = v[1][j] == m ? u1 : 0x1ffu;
                const unsigned s2 = v[2][j] == m ? u2 : 0x1ffu, s3 = v[3][j] == m ? u3 : 0x1ffu;
                code = min(min(s0, s1), min(s2, s3));
            } else {
                // two blocks: the second wins when it is greater, or equal with the smaller code
                const unsigned u0 = (q[0] >> (8 * j)) & 0xffu, u1 = (q[1] >> (8 * j)) & 0xffu;
                const bool second = v[1][j] > v[0][j] || (v[1][j] == v[0][j] && u1 < u0);
                m = second ? v[1][j] : v[0][j];
                code = second ? u1 : u0;
            }
            code = m == -FLT_MAX ? ARG8_EMPTY : code;          // no cell passed the scan
            mv[j] = m;
            packed |= code << (8 * j);
        }
        res[pw] = mv;
        codes[pw] = packed;
    };
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
        if (PARTS > 1 && (pw < pw_lo || pw >= pw_hi)) continue;
        const int ws = wss[pw], we = wes[pw], k = ks[pw];
        if (k < 0) {
            res[pw] = (float4v)(0.0f);
            codes[pw] = 0xffffffffu;
        } else if (k == 0) {
            // the reference's scan (h ascending, w ascending, strict >: roi_pooling_op_gpu.cu.cc:66-79), two cells in flight
            float4v mv = (float4v)(-FLT_MAX);
            unsigned mi[4] = {ARG8_EMPTY, ARG8_EMPTY, ARG8_EMPTY, ARG8_EMPTY};
            for (int h = hs; h < he; ++h) {
                const int so_row = h * W * cell_bytes;
                const unsigned rcode = (unsigned)(h - hs) << 4;
                int w = ws;
                for (; w + 1 < we; w += 2) {
                    const float4v v0 = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so_row + w * cell_bytes, 0));
                    const float4v v1 = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so_row + (w + 1) * cell_bytes, 0));
                    const unsigned code0 = rcode + (unsigned)(w - ws), code1 = code0 + 1u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (v0[j] > mv[j]) { mv[j] = v0[j];  mi[j] = code0; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (v1[j] > mv[j]) { mv[j] = v1[j];  mi[j] = code1; }
                }
                if (w < we) {
                    const float4v v0 = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so_row + w * cell_bytes, 0));
                    const unsigned code0 = rcode + (unsigned)(w - ws);
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (v0[j] > mv[j]) { mv[j] = v0[j];  mi[j] = code0; }
                }
            }
            res[pw] = mv;
            codes[pw] = mi[0] | (mi[1] << 8) | (mi[2] << 16) | (mi[3] << 24);
        } else {
            const bool two_h = he - hs != k + 1, two_w = we - ws != k + 1;
            if (two_h && two_w) block_bin(pw, std::integral_constant<int, 4>());
            else if (two_w) block_bin(pw, std::integral_constant<int, 2>());
            else if (two_h) block_bin(pw, std::integral_constant<int, 3>());
            else block_bin(pw, std::integral_constant<int, 1>());
        }
    }
    // (the stores of the row together behind its last bin: a store per bin would put a store acknowledgement on the
    // wave's critical path per bin -- stores and loads share the in-order vmcnt counter)
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
        if (PARTS > 1 && (pw < pw_lo || pw >= pw_hi)) continue;
        const size_t o = o_row + (size_t)pw * C;
        __builtin_nontemporal_store(res[pw], reinterpret_cast<float4v *>(top + o));
        __builtin_nontemporal_store(codes[pw], reinterpret_cast<unsigned *>(arg8 + o));
    }
}

}  // namespace wssdl

using namespace wssdl;

size_t wssdl::blocks_workspace_bytes(int R, int N, int H, int W, int C) { return blocks_layout(nullptr, R, N, H, W, C).bytes; }

extern "C" size_t wssdl_roi_pool_forward_blocks_bytes(int R, int N, int H, int W, int C, int pooled_h, int pooled_w) {
    if (!blocks_supported(R, N, H, W, C, pooled_h, pooled_w)) return 0;
    return blocks_layout(nullptr, R, N, H, W, C).bytes;
}

// Rule by launch shape (the window sizes are device data): the tables cost ~4 passes over the map per call and pay
// when an image carries many proposals -- train-sized lists on few images.  0 = keep the rows kernel.
extern "C" int wssdl_roi_pool_forward_blocks_auto(int R, int N, int H, int W, int C, int pooled_h, int pooled_w) {
    if (!blocks_supported(R, N, H, W, C, pooled_h, pooled_w)) return 0;
    const int v = tuning().roi_fwd_blocks;
    if (v >= 0) return v != 0 ? 1 : 0;
    // measured (profiles/r06_fwd_blocks_ab_*.log): 2 x 2000 proposals 0.39 -> 0.25 ms, VGG-16's 1 + 2 batch 0.26 -> 0.15,
    // the 4 + 4 batch (1064 RoIs per image, small windows, 8 images of tables) 0.47 -> 0.75
    return (R >= 2048 && (long long)R >= 1200LL * N) ? 1 : 0;
}

extern "C" int wssdl_roi_pool_forward_blocks_prepare(const float *bottom, int N, int H, int W, int C, int R,
                                                     int pooled_h, int pooled_w, const void *table, void *blocks,
                                                     size_t blocks_bytes, wssdl_stream_t stream) {
    if (!blocks_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!bottom || !table || !blocks || (reinterpret_cast<uintptr_t>(bottom) & 15) ||
        (reinterpret_cast<uintptr_t>(blocks) & 255))
        return WSSDL_ERR_INVALID_ARGUMENT;
    const BlocksLayout L = blocks_layout(blocks, R, N, H, W, C);
    if (blocks_bytes < L.bytes) return WSSDL_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int items = R * 7, keys = N * H;
    // (hist and flags were zeroed by wssdl_roi_pool_forward_windows_blocks)
    const long long units = (long long)N * H * W * (C >> 2);
    const int hist_blocks = cdiv(items, 256);
    hipLaunchKernelGGL(blocks_build_kernel, dim3((unsigned)(hist_blocks + cdiv(units, 256))), dim3(256), 0, st, bottom, N, H, W,
                       C, L.values, reinterpret_cast<unsigned *>(L.codes), L.flags, static_cast<const unsigned *>(table), items,
                       hist_blocks, L.hist);
    hipLaunchKernelGGL(rows_scatter_kernel, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, st,
                       static_cast<const unsigned *>(table), items, keys, L.hist, L.cursor, L.order);
    return check_launch();
}

extern "C" int wssdl_roi_pool_forward_compact_blocks(const float *bottom, int N, int H, int W, int C, int R,
                                                     int pooled_h, int pooled_w, const void *table, const void *blocks,
                                                     size_t blocks_bytes, float *top, uint8_t *argmax8,
                                                     wssdl_stream_t stream) {
    if (!blocks_supported(R, N, H, W, C, pooled_h, pooled_w)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!bottom || !table || !blocks || !top || !argmax8) return WSSDL_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(bottom) & 15) || (reinterpret_cast<uintptr_t>(top) & 15) ||
        (reinterpret_cast<uintptr_t>(argmax8) & 3) || (reinterpret_cast<uintptr_t>(blocks) & 255))
        return WSSDL_ERR_INVALID_ARGUMENT;
    const BlocksLayout L = blocks_layout(const_cast<void *>(blocks), R, N, H, W, C);
    if (blocks_bytes < L.bytes) return WSSDL_ERR_WORKSPACE;
    // two waves per bin row (bins 0..3 / 4..6): a wave's chain of dependent reads is what a launch of this size waits
    // for -- 0.118 -> 0.109 ms on VGG-16's 1 + 2 batch, 0.207 -> 0.201 on the alternating weak step
    const int tp = tuning().roi_fwd_blocks_parts;
    const int parts = (tp == 1 || tp == 4 || tp == 7) ? tp : 2;
    const int slices = C / 256, rpw = 4;
    const long long groups = ((long long)R * 7 * parts + rpw - 1) / rpw;
    const long long nblocks = slices >= 8 ? groups * slices : 8 * ((groups + 8 / slices - 1) / (8 / slices));
    if (nblocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    const unsigned *order = tuning().roi_fwd_blocks_sort != 0 ? L.order : nullptr;
#define WSSDL_BLOCKS_LAUNCH(P) \
    hipLaunchKernelGGL((roi_pool_fwd_blocks_kernel<4, P>), dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), bottom, \
                       N, H, W, C, R, top, argmax8, slices, static_cast<const unsigned *>(table), order, L.values, L.codes, L.flags)
    if (parts == 2) WSSDL_BLOCKS_LAUNCH(2);
    else if (parts == 4) WSSDL_BLOCKS_LAUNCH(4);
    else if (parts == 7) WSSDL_BLOCKS_LAUNCH(7);
    else
        hipLaunchKernelGGL((roi_pool_fwd_blocks_kernel<4, 1>), dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), bottom,
                           N, H, W, C, R, top, argmax8, slices, static_cast<const unsigned *>(table), order, L.values, L.codes,
                           L.flags);
    return check_launch();
}
