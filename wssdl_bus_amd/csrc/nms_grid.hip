// Greedy NMS without the serial walk (round 6): neighbour lists from a spatial join + the greedy rule iterated to its
// fixed point.  Same keep lists as cpu_nms.pyx:17-68, bit for bit; taken by launch_nms_two_pass / wssdl_nms for
// thresholds >= 0.6 when wssdl_set_tuning("nms_grid", 1).
//
// Why it is the same result.  Greedy NMS keeps box i (score order) iff no KEPT box j < i has ovr(i, j) >= thresh
// (cpu_nms.pyx:43-66).  That is a recursion on the index, so the kept set is the one fixed point of
//     kept(i)    <=> every j < i with ovr(i, j) >= thresh is removed
//     removed(i) <=> some  j < i with ovr(i, j) >= thresh is kept
// and evaluating the two rules for all undecided boxes at once, again and again, reaches it: a decision, once made,
// rests on facts that never change, and the lowest undecided index can always be decided.  On proposal sets that takes
// 10-14 rounds (tools/probes/nms_rounds_model.py) instead of a walk over 188 chunks of 64.
//
// Why a join.  The rounds need, for every box, ALL its higher-scored neighbours (the walk only ever reads the rows of
// kept boxes): 72 M pair tests per image as a dense lower triangle.  But ovr >= t bounds the pair's geometry --
// |cx_i - cx_j| <= k (w_i + w_j), k = 1/2 - t / (1 + t) (the mask kernel's prefilter, nms.hip) and w_j <= w_i / f,
// f = t / (1 + t) -- so a box only has to look at the boxes whose centres lie within (1 + 1/f) k w_i of its own:
// boxes are binned by centre into a uniform grid (counting sort), a wave takes 64 boxes of one size class that are
// neighbours in the grid and runs the mask kernel's pair code (prefilter, then the exact test in cpu_nms.pyx's f32
// operation order) against the cells of their common window only.  At t = 0.7 that is ~1-2 % of the pairs.
//
//   nms_grid_build_kernel    one workgroup per image: geometry (cx, cy, k w + slack, k h + slack) per box, grid over the
//                            centres, boxes ordered by cell (candidates) and by (size class, cell) (rows)
//   nms_grid_join_kernel     one workgroup per 64 rows, 4 waves sharing the window's cell rows: lists of the
//                            higher-scored neighbours of every box, u16 indices, CAP per box
//   nms_grid_fixpoint_kernel one workgroup per image: statuses in LDS, rounds until none is undecided, then the first
//                            max_keep kept boxes in score order -> keep / rois_padded / num_keep
// A box with more than CAP neighbours (or without a sane width / height) is not an error: its list is marked
// overflowed and the fixed-point kernel tests it against every higher-scored box each round (slow, exact).
#include "nms.hip.h"

namespace wssdl {

constexpr int GRID_DIM = 64;                          // cells per axis at most
constexpr int GRID_CELLS = GRID_DIM * GRID_DIM + 1;   // + the "everywhere" cell (boxes without a sane geometry)
constexpr int GRID_BLOCK = 1024;
constexpr int GRID_MAX_N = 16384;
constexpr int GRID_CLASSES = 4;
constexpr float GRID_CELL_PX = 32.0f;
constexpr int GRID_CAP_MAX = 512;

struct GridHeader {      // per image (256 bytes reserved)
    float min_x, min_y, inv_x, inv_y;
    int gx, gy, n, pad;
};

struct GridLayout {      // per-image slices inside the image's part of the suppression-matrix workspace
    size_t header, cell_start, items, rows, geo, cnt, lists, total;
    int cap;
};

// bytes_per_image = what the caller's matrix gives an image: n_max * pitch * 8
static GridLayout grid_layout(int n_max, size_t bytes_per_image) {
    GridLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off;  off += (bytes + 255) & ~size_t(255);  return o; };
    L.header = take(256);
    L.cell_start = take(sizeof(int) * (GRID_CELLS + 1));
    L.items = take(sizeof(unsigned short) * (size_t)n_max);
    L.rows = take(sizeof(unsigned short) * (size_t)n_max);
    L.geo = take(sizeof(nms_float4v) * (size_t)n_max);
    L.cnt = take(sizeof(int) * (size_t)n_max);
    long long room = (long long)bytes_per_image - (long long)off - 256;
    long long cap = room > 0 ? room / (2LL * n_max) : 0;
    cap &= ~7LL;                                    // whole 16-byte groups
    if (cap > GRID_CAP_MAX) cap = GRID_CAP_MAX;
    L.cap = (int)cap;
    L.lists = take(sizeof(unsigned short) * (size_t)n_max * (size_t)(cap > 0 ? cap : 0));
    L.total = off;
    return L;
}

struct GridArgs {
    const float *boxes;  int box_stride_img;  const int *n_dev;  int n_max;
    char *ws;  size_t ws_stride_img;  GridLayout L;
    float kq, reach, t_lo, t_hi;  double thresh;
    int max_keep;  const int *order;  int order_stride_img;  int *keep, *num_keep;  float *rois_padded;
};

__device__ __forceinline__ int grid_coord(float c, float mn, float inv, int g) {
    const float f = (c - mn) * inv;                 // monotone in c
    if (!(f > 0.0f)) return 0;                      // (NaN too)
    if (f >= (float)g) return g - 1;
    return (int)f;
}

// (cx, cy, rx, ry) as the mask kernel's prefilter defines them (nms.hip nms_mask_block); `sane` = the box has a
// positive width and height and finite numbers: only then do the geometric bounds hold
__device__ __forceinline__ nms_float4v grid_geometry(float x1, float y1, float x2, float y2, float kq, bool *sane) {
    float w = x2 - x1;  w = w + 1.0f;
    float h = y2 - y1;  h = h + 1.0f;
    nms_float4v g;
    g.x = x1 + 0.5f * w;
    g.y = y1 + 0.5f * h;
    const bool pos = w > 0.0f && h > 0.0f;
    g.z = pos ? w * kq + 1e-3f : INFINITY;
    g.w = pos ? h * kq + 1e-3f : INFINITY;
    *sane = pos && __builtin_isfinite(g.x) && __builtin_isfinite(g.y) && __builtin_isfinite(g.z) && __builtin_isfinite(g.w);
    return g;
}

__device__ __forceinline__ int grid_class(const nms_float4v &g, bool sane) {
    if (!sane) return GRID_CLASSES - 1;
    const float r = g.z > g.w ? g.z : g.w;
    return r < 8.0f ? 0 : (r < 16.0f ? 1 : (r < 32.0f ? 2 : 3));
}

// the exact decision of a pair, cpu_nms.pyx:43-66 in f32 with the division only near the threshold (nms.hip)
__device__ __forceinline__ bool grid_pair_hit(float ix1, float iy1, float ix2, float iy2, float iarea, float x1, float y1,
                                              float x2, float y2, float carea, float t_lo, float t_hi, double thresh) {
    const float xx1 = fmax_ref(ix1, x1);
    const float yy1 = fmax_ref(iy1, y1);
    const float xx2 = fmin_ref(ix2, x2);
    const float yy2 = fmin_ref(iy2, y2);
    float w = xx2 - xx1;  w = fmax0_ref(w + 1.0f);
    float h = yy2 - yy1;  h = fmax0_ref(h + 1.0f);
    const float inter = w * h;
    float den = iarea + carea;
    den = den - inter;
    const bool yes = inter > den * t_hi, no = inter < den * t_lo;
    if ((den > 0.0f) & (yes | no)) return yes;
    return (double)(inter / den) >= thresh;
}

template <typename T>
__device__ __forceinline__ T wave_min_i(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const T o = __shfl_xor(v, off, 64);  v = o < v ? o : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max_i(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const T o = __shfl_xor(v, off, 64);  v = o > v ? o : v; }
    return v;
}

// block-wide exclusive scan of one int per thread (GRID_BLOCK threads); returns the exclusive prefix, *total = the sum
__device__ __forceinline__ int block_excl_scan(int v, int *wsum /* LDS [GRID_BLOCK / 64 + 1] */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();                                  // (wsum may still be read from a previous scan)
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < GRID_BLOCK / 64; ++w) { const int s = wsum[w];  base += (w < wave) ? s : 0;  all += s; }
    *total = all;
    return base + inc - v;
}

__global__ __launch_bounds__(GRID_BLOCK) void nms_grid_build_kernel(GridArgs A) {
    __shared__ int s_hist[GRID_CELLS + 1];
    __shared__ unsigned short s_items[GRID_MAX_N];
    __shared__ float s_red[4][GRID_BLOCK / 64];
    __shared__ int s_wsum[GRID_BLOCK / 64 + 1];
    __shared__ GridHeader s_h;
    const int img = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = min(A.n_dev[img], A.n_max);
    const float *b = A.boxes + (size_t)img * A.box_stride_img;
    char *ws = A.ws + (size_t)img * A.ws_stride_img;
    GridHeader *hdr = reinterpret_cast<GridHeader *>(ws + A.L.header);
    int *cell_start = reinterpret_cast<int *>(ws + A.L.cell_start);
    unsigned short *items = reinterpret_cast<unsigned short *>(ws + A.L.items);
    unsigned short *rows = reinterpret_cast<unsigned short *>(ws + A.L.rows);
    nms_float4v *geo = reinterpret_cast<nms_float4v *>(ws + A.L.geo);
    // 1. geometry, extent of the sane centres
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = t; i < n; i += GRID_BLOCK) {
        const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)i * 4);
        bool sane;
        nms_float4v g = grid_geometry(v.x, v.y, v.z, v.w, A.kq, &sane);
        if (!sane) g.z = g.w = INFINITY;             // (every pair with it goes to the exact test, as in the mask kernel)
        geo[i] = g;
        if (sane) { mnx = fminf(mnx, g.x);  mny = fminf(mny, g.y);  mxx = fmaxf(mxx, g.x);  mxy = fmaxf(mxy, g.y); }
    }
    mnx = wave_min_i(mnx);  mny = wave_min_i(mny);  mxx = wave_max_i(mxx);  mxy = wave_max_i(mxy);
    if (lane == 0) { s_red[0][wave] = mnx;  s_red[1][wave] = mny;  s_red[2][wave] = mxx;  s_red[3][wave] = mxy; }
    for (int c = t; c <= GRID_CELLS; c += GRID_BLOCK) s_hist[c] = 0;
    __syncthreads();
    if (t == 0) {
        float a = INFINITY, bb = INFINITY, c = -INFINITY, d = -INFINITY;
        for (int w = 0; w < GRID_BLOCK / 64; ++w) {
            a = fminf(a, s_red[0][w]);  bb = fminf(bb, s_red[1][w]);  c = fmaxf(c, s_red[2][w]);  d = fmaxf(d, s_red[3][w]);
        }
        if (!(c >= a)) { a = bb = c = d = 0.0f; }    // no sane box
        const float ex = c - a, ey = d - bb;
        const float cw = fmaxf(GRID_CELL_PX, ex / (float)(GRID_DIM - 1)), ch = fmaxf(GRID_CELL_PX, ey / (float)(GRID_DIM - 1));
        GridHeader h;
        h.min_x = a;  h.min_y = bb;  h.inv_x = 1.0f / cw;  h.inv_y = 1.0f / ch;
        h.gx = min(GRID_DIM, (int)(ex * h.inv_x) + 1);
        h.gy = min(GRID_DIM, (int)(ey * h.inv_y) + 1);
        h.n = n;  h.pad = 0;
        s_h = h;
        *hdr = h;
    }
    __syncthreads();
    const GridHeader h = s_h;
    const int ncell = h.gx * h.gy;                   // index of the "everywhere" cell
    auto cell_of = [&](const nms_float4v &g) {
        if (!(g.z < INFINITY)) return ncell;
        return grid_coord(g.y, h.min_y, h.inv_y, h.gy) * h.gx + grid_coord(g.x, h.min_x, h.inv_x, h.gx);
    };
    // 2. counting sort by cell
    for (int i = t; i < n; i += GRID_BLOCK) atomicAdd(&s_hist[cell_of(geo[i])], 1);
    __syncthreads();
    {
        constexpr int PER = (GRID_CELLS + 1 + GRID_BLOCK - 1) / GRID_BLOCK;
        int loc[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = t * PER + k;
            loc[k] = (c <= ncell) ? s_hist[c] : 0;
            sum += loc[k];
        }
        int total;
        int base = block_excl_scan(sum, s_wsum, &total);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = t * PER + k;
            if (c <= GRID_CELLS) {         // (entries past the "everywhere" cell: the end of the list)
                s_hist[c] = base;
                cell_start[c] = base;
            }
            base += loc[k];
        }
    }
    __syncthreads();
    for (int i = t; i < n; i += GRID_BLOCK) {
        const int p = atomicAdd(&s_hist[cell_of(geo[i])], 1);
        s_items[p] = (unsigned short)i;
        items[p] = (unsigned short)i;
    }
    __syncthreads();
    // 3. the rows' order: by size class, inside a class in cell order (a stable partition of the cell order)
    const int per = (n + GRID_BLOCK - 1) / GRID_BLOCK;
    const int p0 = min(n, t * per), p1 = min(n, p0 + per);
    int ccount[GRID_CLASSES] = {0, 0, 0, 0};
    for (int p = p0; p < p1; ++p) {
        const nms_float4v g = geo[s_items[p]];
        const int c = grid_class(g, g.z < INFINITY);
#pragma unroll
        for (int k = 0; k < GRID_CLASSES; ++k) ccount[k] += (c == k) ? 1 : 0;
    }
    int coff[GRID_CLASSES], cbase = 0;
#pragma unroll
    for (int k = 0; k < GRID_CLASSES; ++k) {
        int total;
        coff[k] = cbase + block_excl_scan(ccount[k], s_wsum, &total);
        cbase += total;
    }
    for (int p = p0; p < p1; ++p) {
        const unsigned short it = s_items[p];
        const nms_float4v g = geo[it];
        const int c = grid_class(g, g.z < INFINITY);
#pragma unroll
        for (int k = 0; k < GRID_CLASSES; ++k)
            if (c == k) rows[coff[k]++] = it;
    }
}

constexpr int JOIN_WAVES = 4;

__global__ __launch_bounds__(64 * JOIN_WAVES) void nms_grid_join_kernel(GridArgs A) {
    __shared__ nms_float4v s_geo[JOIN_WAVES][64];
    __shared__ float s_box[JOIN_WAVES][5][64];
    __shared__ int s_idx[JOIN_WAVES][64];
    __shared__ int s_cnt[64];
    const int img = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *ws = A.ws + (size_t)img * A.ws_stride_img;
    const GridHeader h = *reinterpret_cast<const GridHeader *>(ws + A.L.header);
    const int n = h.n;
    const int p = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;
    const int *cell_start = reinterpret_cast<const int *>(ws + A.L.cell_start);
    const unsigned short *items = reinterpret_cast<const unsigned short *>(ws + A.L.items);
    const unsigned short *rows = reinterpret_cast<const unsigned short *>(ws + A.L.rows);
    const nms_float4v *geo = reinterpret_cast<const nms_float4v *>(ws + A.L.geo);
    int *cnt = reinterpret_cast<int *>(const_cast<char *>(ws) + A.L.cnt);
    unsigned short *lists = reinterpret_cast<unsigned short *>(const_cast<char *>(ws) + A.L.lists);
    const float *b = A.boxes + (size_t)img * A.box_stride_img;
    const int cap = A.L.cap;
    const bool row_ok = p < n;
    const int i = row_ok ? (int)rows[p] : -1;
    float ix1 = 0.f, iy1 = 0.f, ix2 = 0.f, iy2 = 0.f;
    nms_float4v ig = {__builtin_nanf(""), 0.f, 0.f, 0.f};      // (a lane without a row never passes the prefilter)
    if (row_ok) {
        const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)i * 4);
        ix1 = v.x; iy1 = v.y; ix2 = v.z; iy2 = v.w;
        ig = geo[i];
    }
    const float iarea = box_area_ref(ix1, iy1, ix2, iy2);
    // the window of the row's possible partners: |dc| <= r_i + r_j and r_j <= r_i / f (+ slack) -> reach = 1 + 1 / f'
    int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
    if (row_ok) {
        if (ig.z < INFINITY) {
            const float rx = ig.z * A.reach + 0.5f, ry = ig.w * A.reach + 0.5f;
            x0 = grid_coord(ig.x - rx, h.min_x, h.inv_x, h.gx);  x1 = grid_coord(ig.x + rx, h.min_x, h.inv_x, h.gx);
            y0 = grid_coord(ig.y - ry, h.min_y, h.inv_y, h.gy);  y1 = grid_coord(ig.y + ry, h.min_y, h.inv_y, h.gy);
        } else {
            x0 = 0;  x1 = h.gx - 1;  y0 = 0;  y1 = h.gy - 1;
        }
    }
    x0 = wave_min_i(x0);  y0 = wave_min_i(y0);  x1 = wave_max_i(x1);  y1 = wave_max_i(y1);
    // (all four waves hold the same rows: the window is the same in each)
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int ux0 = __builtin_amdgcn_readfirstlane(x0), ux1 = __builtin_amdgcn_readfirstlane(x1);
    const int uy0 = __builtin_amdgcn_readfirstlane(y0), uy1 = __builtin_amdgcn_readfirstlane(y1);
    const nms_float2v ic = {ig.x, ig.y}, ir = {ig.z, ig.w};
    const int ncell = h.gx * h.gy;
    // cell rows uy0 .. uy1 of the window, then the "everywhere" cell, dealt to the waves in turn
    for (int cy = uy0 + wave; cy <= uy1 + 1; cy += JOIN_WAVES) {
        int run_lo, run_hi;
        if (cy <= uy1) { run_lo = cell_start[cy * h.gx + ux0];  run_hi = cell_start[cy * h.gx + ux1 + 1]; }
        else { run_lo = cell_start[ncell];  run_hi = cell_start[ncell + 1]; }
        run_lo = __builtin_amdgcn_readfirstlane(run_lo);  run_hi = __builtin_amdgcn_readfirstlane(run_hi);
        for (int q0 = run_lo; q0 < run_hi; q0 += 64) {
            const int q = q0 + lane;
            const bool cand_ok = q < run_hi;
            const int j = cand_ok ? (int)items[q] : 0;
            nms_float4v cg = {__builtin_nanf(""), 0.f, 0.f, 0.f};
            float cx1 = 0.f, cy1 = 0.f, cx2 = 0.f, cy2 = 0.f;
            if (cand_ok) {
                cg = geo[j];
                const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)j * 4);
                cx1 = v.x; cy1 = v.y; cx2 = v.z; cy2 = v.w;
            }
            s_geo[wave][lane] = cg;
            s_box[wave][0][lane] = cx1;  s_box[wave][1][lane] = cy1;  s_box[wave][2][lane] = cx2;  s_box[wave][3][lane] = cy2;
            s_box[wave][4][lane] = box_area_ref(cx1, cy1, cx2, cy2);
            s_idx[wave][lane] = cand_ok ? j : 0x7fffffff;
            __builtin_amdgcn_wave_barrier();
            // the mask kernel's prefilter: candidate column k's verdict enters the word through the carry
            unsigned u_lo = 0u, u_hi = 0u;
            auto near = [&](int k, unsigned u) -> unsigned {
                const nms_float4v qg = s_geo[wave][k];
                const nms_float2v d = ic - qg.xy;
                const nms_float2v r = ir + qg.zw;
                const unsigned long long px = __builtin_amdgcn_fcmpf(__builtin_fabsf(d.x), r.x, 5 /* ole */);
                const unsigned long long py = __builtin_amdgcn_fcmpf(__builtin_fabsf(d.y), r.y, 5 /* ole */);
                const unsigned long long both = px & py;
                unsigned long long carry_out;
                asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(u), "=s"(carry_out) : "v"(u), "s"(both));
                return u;
            };
#pragma unroll 8
            for (int k = 31; k >= 0; --k) u_lo = near(k, u_lo);
#pragma unroll 8
            for (int k = 63; k >= 32; --k) u_hi = near(k, u_hi);
            unsigned long long todo = ((unsigned long long)u_hi << 32) | u_lo;
            while (todo != 0ull) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int jj = s_idx[wave][k];
                if (jj >= i) continue;                       // only higher-scored partners (and not the box itself)
                if (grid_pair_hit(ix1, iy1, ix2, iy2, iarea, s_box[wave][0][k], s_box[wave][1][k], s_box[wave][2][k],
                                  s_box[wave][3][k], s_box[wave][4][k], A.t_lo, A.t_hi, A.thresh)) {
                    const int slot = atomicAdd(&s_cnt[lane], 1);
                    if (slot < cap) lists[(size_t)i * cap + slot] = (unsigned short)jj;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && row_ok) cnt[i] = s_cnt[threadIdx.x];
}

enum { ST_UNKNOWN = 0, ST_KEPT = 1, ST_REMOVED = 2 };

__global__ __launch_bounds__(GRID_BLOCK) void nms_grid_fixpoint_kernel(GridArgs A) {
    __shared__ unsigned char s_st[GRID_MAX_N];
    __shared__ int s_open;
    __shared__ int s_wsum[GRID_BLOCK / 64 + 1];
    const int img = blockIdx.x, t = threadIdx.x;
    const char *ws = A.ws + (size_t)img * A.ws_stride_img;
    const int n = min(A.n_dev[img], A.n_max);
    const int *cnt = reinterpret_cast<const int *>(ws + A.L.cnt);
    const unsigned short *lists = reinterpret_cast<const unsigned short *>(ws + A.L.lists);
    const float *b = A.boxes + (size_t)img * A.box_stride_img;
    const int cap = A.L.cap;
    // round 0: a box without higher-scored neighbours is kept
    for (int i = t; i < n; i += GRID_BLOCK) s_st[i] = cnt[i] == 0 ? ST_KEPT : ST_UNKNOWN;
    for (;;) {
        __syncthreads();
        if (t == 0) s_open = 0;
        __syncthreads();
        int open = 0;
        for (int i = t; i < n; i += GRID_BLOCK) {
            if (s_st[i] != ST_UNKNOWN) continue;
            const int c = cnt[i];
            bool removed = false, unknown = false;
            if (c <= cap) {
                const unsigned short *li = lists + (size_t)i * cap;
                for (int e0 = 0; e0 < c && !removed; e0 += 8) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(li + e0);
                    const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (e0 + k < c) {
                            const unsigned j = (wv[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                            const unsigned s = s_st[j];
                            removed |= s == ST_KEPT;
                            unknown |= s == ST_UNKNOWN;
                        }
                    }
                }
            } else {
                // the list overflowed: every higher-scored box, tested here (exact, slow, rare)
                const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)i * 4);
                const float iarea = box_area_ref(v.x, v.y, v.z, v.w);
                for (int j = 0; j < i && !removed; ++j) {
                    const unsigned s = s_st[j];
                    if (s == ST_REMOVED) continue;
                    const nms_float4v u = *reinterpret_cast<const nms_float4v *>(b + (size_t)j * 4);
                    if (grid_pair_hit(v.x, v.y, v.z, v.w, iarea, u.x, u.y, u.z, u.w, box_area_ref(u.x, u.y, u.z, u.w), A.t_lo,
                                      A.t_hi, A.thresh)) {
                        removed |= s == ST_KEPT;
                        unknown |= s == ST_UNKNOWN;
                    }
                }
            }
            // (a status is written by its own thread only and read by others whenever: any value they see is final or
            // "unknown", and a decision taken on final values is the right one)
            if (removed) s_st[i] = ST_REMOVED;
            else if (!unknown) s_st[i] = ST_KEPT;
            else open = 1;
        }
        if (open) s_open = 1;
        __syncthreads();
        if (!s_open) break;
    }
    // the first max_keep kept boxes, in score order
    const int per = (n + GRID_BLOCK - 1) / GRID_BLOCK;
    const int i0 = min(n, t * per), i1 = min(n, i0 + per);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += s_st[i] == ST_KEPT ? 1 : 0;
    int total;
    int pos = block_excl_scan(mine, s_wsum, &total);
    for (int i = i0; i < i1 && pos < A.max_keep; ++i) {
        if (s_st[i] != ST_KEPT) continue;
        if (A.keep) A.keep[(size_t)img * A.max_keep + pos] = A.order ? A.order[(size_t)img * A.order_stride_img + i] : i;
        if (A.rois_padded) {
            const float *bx = b + (size_t)i * 4;
            float *o = A.rois_padded + ((size_t)img * A.max_keep + pos) * 5;
            o[0] = (float)img; o[1] = bx[0]; o[2] = bx[1]; o[3] = bx[2]; o[4] = bx[3];
        }
        ++pos;
    }
    if (t == 0) A.num_keep[img] = min(total, A.max_keep);
}

bool nms_grid_supported(int n_max, int n_images, double thresh, int max_keep, size_t ws_bytes_per_image) {
    if (n_max < 64 || n_max > GRID_MAX_N || n_images < 1 || max_keep < 1) return false;
    if (!(thresh >= 0.6 && thresh < 1.0)) return false;        // (below, the window of a box is most of the image)
    const GridLayout L = grid_layout(n_max, ws_bytes_per_image);
    return L.cap >= 32 && L.total <= ws_bytes_per_image;
}

int launch_nms_grid(const float *boxes, int box_stride_img, const int *n_dev, int n_max, int n_images, double thresh,
                    void *ws, size_t ws_bytes_per_image, int max_keep, const int *order, int order_stride_img, int *keep,
                    int *num_keep, float *rois_padded, hipStream_t st) {
    if (!nms_grid_supported(n_max, n_images, thresh, max_keep, ws_bytes_per_image)) return WSSDL_ERR_INVALID_ARGUMENT;
    GridArgs A;
    A.boxes = boxes;  A.box_stride_img = box_stride_img;  A.n_dev = n_dev;  A.n_max = n_max;
    A.ws = static_cast<char *>(ws);  A.ws_stride_img = ws_bytes_per_image;  A.L = grid_layout(n_max, ws_bytes_per_image);
    const double fq = thresh / (1.0 + thresh);
    A.kq = (float)((0.5 - fq) * (1.0 + 1e-3));                 // as in nms_mask_block
    A.reach = (float)(1.0 + 1.0 / (fq * 0.98));                // r_j <= r_i / f: 2 % of slack on f, half a pixel on the reach
    A.t_lo = (float)(thresh * (1.0 - 1e-4));  A.t_hi = (float)(thresh * (1.0 + 1e-4));  A.thresh = thresh;
    A.max_keep = max_keep;  A.order = order;  A.order_stride_img = order_stride_img;  A.keep = keep;  A.num_keep = num_keep;
    A.rois_padded = rois_padded;
    hipLaunchKernelGGL(nms_grid_build_kernel, dim3(n_images), dim3(GRID_BLOCK), 0, st, A);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(nms_grid_join_kernel, dim3(cdiv(n_max, 64), n_images), dim3(64 * JOIN_WAVES), 0, st, A);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(nms_grid_fixpoint_kernel, dim3(n_images), dim3(GRID_BLOCK), 0, st, A);
    return check_launch();
}

}  // namespace wssdl
