// Greedy NMS + score ranking for gfx950 (MI355X), batched over images.
//
// Reference: code/lib/nms/cpu_nms.pyx:17-68 (the NMS the RPN actually uses:
// fast_rcnn/config.py:321 USE_GPU_NMS=False) via fast_rcnn/nms_wrapper.py:13-21;
// sort: rpn_msr/proposal_layer_tf_bus.py:129-133.
//
// Three kernels, each launched once for ALL images of a step:
//   rank_topk   : position of every candidate in descending score order by
//                 counting (key_j > key_i) over LDS-staged key tiles; 64-bit keys
//                 = (order-preserving score bits << 32 | index) make the order
//                 total (ties: higher index first) and the result deterministic.
//   nms_mask    : 64x64 tiles of the upper triangle of the suppression matrix,
//                 one u64 word per (row box, column block): 64 column boxes in
//                 LDS, one row box per lane.  f32 arithmetic in cpu_nms.pyx's
//                 operation order, test (double)iou >= thresh (vendored
//                 cpu_nms.c:2495 compares PyFloat objects).
//   nms_sweep   : one workgroup per image walks the 64-row chunks in order.
//                 Wave 0 resolves a chunk against its diagonal word entirely in
//                 scalar registers (v_readlane), appends the kept rows, then all
//                 waves OR the kept rows' mask words into the LDS-resident
//                 `removed` bitmap.  Stops as soon as max_keep boxes are kept
//                 (the reference's caller truncates keep[:post_nms_topN]).
#include "nms.hip.h"
#include "select.hip.h"

namespace wssdl {

// ---------------------------------------------------------------- rank/top-k ---
// Descending order of the topn largest 64-bit keys of each image, in three steps:
//   topk_threshold : the topn-th largest key by MSB-first radix select (one workgroup per
//                    image, 8 histogram passes over the keys; keys are unique);
//   topk_compact   : keys >= threshold -> dense candidate array (any order);
//   rank_topk      : position of every candidate = number of greater candidates, counted over
//                    LDS-staged tiles -- O(topn^2) instead of O(M^2) compares.
// 64-bit keys = (order-preserving score bits << 32 | index) make the order total (ties:
// higher index first) and the result deterministic.
constexpr int SEL_BLOCK = 1024;
constexpr int SEL_LIST = 512;

__global__ __launch_bounds__(SEL_BLOCK) void topk_threshold_kernel(
    const unsigned long long *__restrict__ keys, int M, int topn,
    unsigned long long *__restrict__ thresh, int *__restrict__ n_sorted) {
    __shared__ SelectScratch<SEL_LIST> sc;
    const int img = blockIdx.x, t = threadIdx.x;
    const unsigned long long *k = keys + (size_t)img * M;
    int valid = 0;
    // topn-th largest key; with no more than topn valid keys every one of them is a candidate
    const unsigned long long th = block_radix_select<SEL_BLOCK, SEL_LIST, true>(
        [k](int i, unsigned long long &v) { v = k[i]; return v != 0ull; }, M,
        [topn](int members) { return members > topn ? topn : 0; }, sc, &valid);
    if (t == 0) {
        n_sorted[img] = min(valid, topn);
        thresh[img] = valid > topn ? th : 1ull;
    }
}

__global__ __launch_bounds__(256) void topk_compact_kernel(
    const unsigned long long *__restrict__ keys, int M, int topn,
    const unsigned long long *__restrict__ thresh, unsigned long long *__restrict__ cand,
    int *__restrict__ cand_fill) {
    const int img = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned long long v = keys[(size_t)img * M + i];
    if (v != 0ull && v >= thresh[img]) {
        const int pos = atomicAdd(&cand_fill[img], 1);
        if (pos < topn) cand[(size_t)img * topn + pos] = v;
    }
}

constexpr int RANK_BLOCK = 256;
constexpr int RANK_TILE = 1024;

__global__ __launch_bounds__(RANK_BLOCK) void rank_topk_kernel(
    const unsigned long long *__restrict__ cand, const int *__restrict__ n_cand, int topn,
    int *__restrict__ sorted_index) {
    __shared__ unsigned long long tile[RANK_TILE];
    const int img = blockIdx.y;
    const int n = min(n_cand[img], topn);
    if (blockIdx.x * RANK_BLOCK >= n) return;
    const unsigned long long *k = cand + (size_t)img * topn;
    const int i = blockIdx.x * RANK_BLOCK + threadIdx.x;
    const unsigned long long mine = (i < n) ? k[i] : ~0ull;
    int cnt = 0;
    for (int j0 = 0; j0 < n; j0 += RANK_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < RANK_TILE; t += RANK_BLOCK)
            tile[t] = (j0 + t < n) ? k[j0 + t] : 0ull;
        __syncthreads();
        const int tn = min(RANK_TILE, n - j0);
        int t = 0;
        for (; t + 8 <= tn; t += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt += (tile[t + u] > mine) ? 1 : 0;
        }
        for (; t < tn; ++t) cnt += (tile[t] > mine) ? 1 : 0;
    }
    if (i < n) sorted_index[(size_t)img * topn + cnt] = (int)(unsigned)(mine & 0xffffffffull);
}

// Bucketed ranking (sample sort) for large topn: rank_topk above compares every candidate with
// every other (12000^2 per image); here 255 splitters taken from a sorted sample of 256
// candidates cut them into 256 buckets first, and a candidate is only compared with its own bucket:
//   rank_bucketize : one workgroup per image: sample -> splitters -> bucket sizes -> candidates
//                    regrouped by bucket (any order inside a bucket) + bucket offsets;
//   rank_in_bucket : position = bucket offset + number of greater keys in the bucket.
// Keys are unique, so positions are a permutation and the result is the same total order.
constexpr int RB_BLOCK = 1024;
constexpr int RB_SAMPLES = 256;
constexpr int RB_BUCKETS = 256;

__device__ __forceinline__ int rb_bucket_of(const unsigned long long *split, int nb,
                                            unsigned long long key) {
    // splitters descending; bucket = number of splitters strictly greater than key
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (split[mid] > key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(RB_BLOCK) void rank_bucketize_kernel(
    const unsigned long long *__restrict__ cand, const int *__restrict__ n_cand, int topn,
    unsigned long long *__restrict__ grouped, int *__restrict__ bucket_off) {
    __shared__ unsigned long long samp[RB_SAMPLES];
    __shared__ unsigned long long ssort[RB_SAMPLES];
    __shared__ unsigned long long split[RB_BUCKETS];
    __shared__ int bcount[RB_BUCKETS];
    __shared__ int bfill[RB_BUCKETS];
    __shared__ int wsum[4];
    const int img = blockIdx.x, t = threadIdx.x;
    const int n = min(n_cand[img], topn);
    const unsigned long long *k = cand + (size_t)img * topn;
    unsigned long long *g = grouped + (size_t)img * topn;
    int *boff = bucket_off + (size_t)img * (RB_BUCKETS + 1);
    const int S = min(n, RB_SAMPLES);
    const int nb = min(RB_BUCKETS, max(S, 1));
    if (t < S) samp[t] = k[(long long)t * n / S];
    if (t < RB_BUCKETS) bcount[t] = 0;
    __syncthreads();
    if (t < S) {
        const unsigned long long mine = samp[t];
        int r = 0;
        for (int j = 0; j < S; ++j) r += (samp[j] > mine) ? 1 : 0;
        ssort[r] = mine;                           // sample positions are distinct -> keys distinct
    }
    __syncthreads();
    if (t < nb - 1) split[t] = ssort[(long long)(t + 1) * S / nb];
    __syncthreads();
    for (int i = t; i < n; i += RB_BLOCK) atomicAdd(&bcount[rb_bucket_of(split, nb, k[i])], 1);
    __syncthreads();
    {   // bucket offsets: 256-wide exclusive scan by the first four waves
        static_assert(RB_BUCKETS == 256, "scan below is written for 4 waves");
        const int lane = t & 63, wave = t >> 6;
        int h = 0, inc = 0;
        if (t < RB_BUCKETS) {
            h = bcount[t];
            inc = h;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (t < RB_BUCKETS) {
            int basev = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) basev += (w < wave) ? wsum[w] : 0;
            inc += basev;
            boff[t] = inc - h;
            bfill[t] = inc - h;
            if (t == RB_BUCKETS - 1) boff[RB_BUCKETS] = inc;
        }
    }
    __syncthreads();
    for (int i = t; i < n; i += RB_BLOCK) {
        const unsigned long long v = k[i];
        g[atomicAdd(&bfill[rb_bucket_of(split, nb, v)], 1)] = v;
    }
}

constexpr int RIB_BLOCK = 256;
constexpr int RIB_TILE = 1024;

__global__ __launch_bounds__(RIB_BLOCK) void rank_in_bucket_kernel(
    const unsigned long long *__restrict__ grouped, const int *__restrict__ n_cand, int topn,
    const int *__restrict__ bucket_off, int *__restrict__ sorted_index) {
    __shared__ int s_boff[RB_BUCKETS + 1];
    __shared__ unsigned long long tile[RIB_TILE];
    const int img = blockIdx.y;
    const int n = min(n_cand[img], topn);
    const int p0 = blockIdx.x * RIB_BLOCK;
    if (p0 >= n) return;
    const unsigned long long *g = grouped + (size_t)img * topn;
    for (int i = threadIdx.x; i <= RB_BUCKETS; i += RIB_BLOCK)
        s_boff[i] = bucket_off[(size_t)img * (RB_BUCKETS + 1) + i];
    __syncthreads();
    const int p = p0 + threadIdx.x;
    const bool live = p < n;
    const unsigned long long mine = live ? g[p] : ~0ull;
    // bucket of a position: last b with boff[b] <= p
    auto bucket_at = [&](int pos) {
        int lo = 0, hi = RB_BUCKETS - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_boff[mid] <= pos) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    const int b = bucket_at(live ? p : n - 1);
    const int lo = s_boff[b], hi = s_boff[b + 1];
    // the workgroup's positions span the buckets of p0 .. min(p0 + 255, n - 1)
    const int wlo = s_boff[bucket_at(p0)];
    const int whi = s_boff[bucket_at(min(p0 + RIB_BLOCK, n) - 1) + 1];
    int cnt = 0;
    for (int j0 = wlo; j0 < whi; j0 += RIB_TILE) {
        __syncthreads();
        for (int u = threadIdx.x; u < RIB_TILE; u += RIB_BLOCK) tile[u] = (j0 + u < whi) ? g[j0 + u] : 0ull;
        __syncthreads();
        const int a = max(lo, j0) - j0, z = min(hi, j0 + RIB_TILE) - j0;
        for (int u = a; u < z; ++u) cnt += (tile[u] > mine) ? 1 : 0;
    }
    if (live) sorted_index[(size_t)img * topn + lo + cnt] = (int)(unsigned)(mine & 0xffffffffull);
}

size_t rank_topk_scratch_bytes(int n_images, int topn) {
    return (size_t)n_images * topn * sizeof(unsigned long long) +
           (size_t)n_images * (RB_BUCKETS + 1) * sizeof(int);
}

int launch_rank_topk(const unsigned long long *keys, int M, int n_images, int topn,
                     unsigned long long *cand, unsigned long long *thresh, int *cand_fill,
                     int *sorted_index, int *n_sorted, void *scratch, size_t scratch_bytes,
                     hipStream_t st) {
    // sorted_index must be pre-filled with -1 and cand_fill with 0 by the caller
    hipLaunchKernelGGL(topk_threshold_kernel, dim3(n_images), dim3(SEL_BLOCK), 0, st, keys, M, topn,
                       thresh, n_sorted);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(topk_compact_kernel, dim3(cdiv(M, 256), n_images), dim3(256), 0, st, keys, M,
                       topn, thresh, cand, cand_fill);
    if ((rc = check_launch())) return rc;
    if (topn >= 4096 && scratch && scratch_bytes >= rank_topk_scratch_bytes(n_images, topn)) {
        unsigned long long *grouped = static_cast<unsigned long long *>(scratch);
        int *boff = reinterpret_cast<int *>(grouped + (size_t)n_images * topn);
        hipLaunchKernelGGL(rank_bucketize_kernel, dim3(n_images), dim3(RB_BLOCK), 0, st, cand, n_sorted,
                           topn, grouped, boff);
        if ((rc = check_launch())) return rc;
        hipLaunchKernelGGL(rank_in_bucket_kernel, dim3(cdiv(topn, RIB_BLOCK), n_images), dim3(RIB_BLOCK),
                           0, st, grouped, n_sorted, topn, boff, sorted_index);
        return check_launch();
    }
    hipLaunchKernelGGL(rank_topk_kernel, dim3(cdiv(topn, RANK_BLOCK), n_images), dim3(RANK_BLOCK), 0,
                       st, cand, n_sorted, topn, sorted_index);
    return check_launch();
}

// ------------------------------------------------------------------ nms mask ---
__device__ __forceinline__ float fmax_ref(float a, float b) { return a >= b ? a : b; }  // cpu_nms.pyx:11
__device__ __forceinline__ float fmin_ref(float a, float b) { return a <= b ? a : b; }  // cpu_nms.pyx:14

__device__ __forceinline__ float box_area_ref(float x1, float y1, float x2, float y2) {
    float w = x2 - x1;  w = w + 1.0f;       // numpy f32: (x2 - x1 + 1) * (y2 - y1 + 1), cpu_nms.pyx:24
    float h = y2 - y1;  h = h + 1.0f;
    return w * h;
}

constexpr int MASK_WAVES = 4;

constexpr int MASK_SEG = 16;     // column blocks per workgroup

// One workgroup per (64-row block, segment of 16 column blocks); its 4 waves stride over the
// segment's column blocks cb >= rb (upper triangle only).  A lane keeps its row box in
// registers; the 64 column boxes of the wave's current block sit in that wave's LDS slice and
// are read as broadcasts.
__global__ __launch_bounds__(64 * MASK_WAVES) void nms_mask_kernel(
    const float *__restrict__ boxes, int box_stride_img, const int *__restrict__ n_dev, int n_max,
    double thresh, unsigned long long *__restrict__ mask, int ncb) {
    const int rb = blockIdx.x, seg = blockIdx.y, img = blockIdx.z;
    const int n = min(n_dev[img], n_max);
    if (rb * 64 >= n || (seg + 1) * MASK_SEG <= rb || seg * MASK_SEG * 64 >= n) return;
    __shared__ float cbox[MASK_WAVES][5][64];
    const float *b = boxes + (size_t)img * box_stride_img;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = rb * 64 + lane;
    const bool row_ok = i < n;
    float ix1 = 0.f, iy1 = 0.f, ix2 = 0.f, iy2 = 0.f;
    if (row_ok) { ix1 = b[i * 4 + 0]; iy1 = b[i * 4 + 1]; ix2 = b[i * 4 + 2]; iy2 = b[i * 4 + 3]; }
    const float iarea = box_area_ref(ix1, iy1, ix2, iy2);
    const float t_lo = (float)(thresh * (1.0 - 1e-4)), t_hi = (float)(thresh * (1.0 + 1e-4));
    const int cb_end = min((n + 63) / 64, (seg + 1) * MASK_SEG);
    for (int cb = max(rb, seg * MASK_SEG) + wave; cb < cb_end; cb += MASK_WAVES) {
        const int col = cb * 64 + lane;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        if (col < n) { x1 = b[col * 4 + 0]; y1 = b[col * 4 + 1]; x2 = b[col * 4 + 2]; y2 = b[col * 4 + 3]; }
        // the wave's own slice: wave-synchronous, no workgroup barrier needed
        cbox[wave][0][lane] = x1; cbox[wave][1][lane] = y1; cbox[wave][2][lane] = x2;
        cbox[wave][3][lane] = y2; cbox[wave][4][lane] = box_area_ref(x1, y1, x2, y2);
        __builtin_amdgcn_wave_barrier();
        const int jn = min(64, n - cb * 64);
        unsigned long long bits = 0ull;
        for (int j = 0; j < jn; ++j) {
            float xx1 = fmax_ref(ix1, cbox[wave][0][j]);
            float yy1 = fmax_ref(iy1, cbox[wave][1][j]);
            float xx2 = fmin_ref(ix2, cbox[wave][2][j]);
            float yy2 = fmin_ref(iy2, cbox[wave][3][j]);
            float w = xx2 - xx1;  w = fmax_ref(0.0f, w + 1.0f);
            float h = yy2 - yy1;  h = fmax_ref(0.0f, h + 1.0f);
            float inter = w * h;
            float den = iarea + cbox[wave][4][j];
            den = den - inter;
            // (double)(inter / den) >= thresh, deciding without the IEEE division whenever the
            // quotient is at least 1e-4 (relative) away from the threshold -- the f32 rounding
            // of the quotient (2^-24) cannot cross that margin; otherwise the exact test
            bool sup;
            if (den > 0.0f && inter < den * t_lo) sup = false;
            else if (den > 0.0f && inter > den * t_hi) sup = true;
            else sup = (double)(inter / den) >= thresh;
            if (sup && cb * 64 + j > i) bits |= 1ull << j;
        }
        if (row_ok) mask[((size_t)img * n_max + i) * ncb + cb] = bits;
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_nms_mask(const float *boxes, int box_stride_img, const int *n_dev, int n_max,
                    int n_images, double thresh, unsigned long long *mask, hipStream_t st) {
    int ncb = cdiv(n_max, 64);
    if (ncb == 0 || n_images == 0) return WSSDL_OK;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(ncb, cdiv(ncb, MASK_SEG), n_images), dim3(64 * MASK_WAVES), 0, st, boxes,
                       box_stride_img, n_dev, n_max, thresh, mask, ncb);
    return check_launch();
}

// ----------------------------------------------------------------- nms sweep ---
#ifndef WSSDL_SWEEP_BLOCK
#define WSSDL_SWEEP_BLOCK 1024
#endif
constexpr int SWEEP_BLOCK = WSSDL_SWEEP_BLOCK;
constexpr size_t SWEEP_LDS_LIMIT = 60 * 1024;     // kept list in LDS up to ~15k entries

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
    unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, lane);
    unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// OR over the 64 lanes of a wave with DPP moves (VALU latency; a ds_bpermute butterfly costs six
// dependent LDS round trips, which sat on the per-chunk critical path of the sweep).  Shifts
// within each row of 16 bring the row's OR to its lane 15, the two row broadcasts carry it to
// lane 63; OR is idempotent so the overlapping shifts need no masking.  Uniform result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_or(unsigned v) {
    return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v = dpp_or<0x111, 0xf>(v);      // row_shr:1
    v = dpp_or<0x112, 0xf>(v);      // row_shr:2
    v = dpp_or<0x114, 0xf>(v);      // row_shr:4
    v = dpp_or<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of each row holds the row's OR
    v = dpp_or<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_or<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds all
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    unsigned lo = wave_or_u32((unsigned)v);
    unsigned hi = wave_or_u32((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Greedy sweep over the suppression matrix, one workgroup per image, pull formulation: the
// word that chunk c needs is  removed_c = OR over every box kept in chunks < c of mask[box][c].
// Per chunk, wave 0 (critical path)
//   resolves the 64-row chunk against its diagonal word in scalar registers, visiting only
//   the surviving boxes, appends them to the kept list, then fetches word c+1 of the boxes it
//   just kept (one load per lane, wave OR-reduce);
// while the other waves, overlapped with that,
//   OR word c+1 of every box kept in EARLIER chunks (list in LDS, loads all independent).
// One barrier per chunk joins the two halves.  Stops as soon as max_keep boxes are kept (the
// reference's caller truncates keep[:post_nms_topN]).
__global__ __launch_bounds__(SWEEP_BLOCK) void nms_sweep_kernel(
    const unsigned long long *__restrict__ mask, const int *__restrict__ n_dev, int n_max, int ncb,
    int max_keep, const int *__restrict__ order, int order_stride_img,
    int *__restrict__ keep, int *__restrict__ num_keep,
    const float *__restrict__ boxes, int box_stride_img, float *__restrict__ rois_padded,
    int *__restrict__ kept_scratch) {
    extern __shared__ int kept_lds[];                // [max_keep + 64] rows kept so far, in order
    // (global scratch instead when the list does not fit in LDS: same-CU visibility after the
    // workgroup barrier is all that is needed)
    int *kept_rows = kept_scratch ? kept_scratch + (size_t)blockIdx.x * (max_keep + 64) : kept_lds;
    __shared__ unsigned long long s_part[2][SWEEP_BLOCK / 64];   // helpers' partial ORs, by parity
    __shared__ unsigned long long s_own[2];          // wave 0's contribution for the next chunk
    __shared__ int s_count;
    const int img = blockIdx.x;
    const int n = min(n_dev[img], n_max);
    const unsigned long long *m = mask + (size_t)img * n_max * ncb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NHELP = SWEEP_BLOCK - 64;
    if (tid == 0) { s_count = 0; s_own[0] = s_own[1] = 0ull; }
    if (tid < 2 * (SWEEP_BLOCK / 64)) s_part[tid / (SWEEP_BLOCK / 64)][tid % (SWEEP_BLOCK / 64)] = 0ull;
    __syncthreads();
    int count = 0;                                    // boxes kept in chunks < c (all threads agree)
    const int nchunks = (n + 63) / 64;
    // wave 0 keeps two words per lane in flight one chunk ahead: the diagonal word of its row and
    // the word right of it (needed only if the row survives: loaded speculatively so that no
    // memory latency sits between resolving a chunk and handing its result on)
    unsigned long long diag = 0ull, right = 0ull;
    if (wave == 0 && nchunks > 0 && lane < n) {
        diag = m[(size_t)lane * ncb];
        if (nchunks > 1) right = m[(size_t)lane * ncb + 1];
    }
    for (int c = 0; c < nchunks; ++c) {
        const int par = c & 1;
        if (wave == 0) {
            // s_part[par][0] is unused by the helpers (wave 0 is not one): lane 0 reads s_own there
            unsigned long long rem = (lane == 0) ? s_own[par]
                                   : (lane < SWEEP_BLOCK / 64) ? s_part[par][lane] : 0ull;
            rem = wave_or_u64(rem);
            const int row = c * 64 + lane;
            const int nv = n - c * 64;
            const unsigned long long valid = (nv >= 64) ? ~0ull : ((1ull << nv) - 1ull);
            unsigned long long cur = rem | ~valid;
            unsigned long long kept = 0ull;
            unsigned long long cand = ~cur;
            while (cand != 0ull) {
                const int bsel = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
                kept |= 1ull << bsel;
                unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)diag, bsel);
                unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), bsel);
                cur |= (((unsigned long long)hi << 32) | lo) | (1ull << bsel);
                cand = ~cur & ((bsel == 63) ? 0ull : (~0ull << (bsel + 1)));
            }
            const bool mine = (kept >> lane) & 1ull;
            if (mine) {
                const int pos = count + __popcll(kept & ((1ull << lane) - 1ull));
                if (pos < max_keep) {
                    kept_rows[pos] = row;
                    if (keep)
                        keep[(size_t)img * max_keep + pos] =
                            order ? order[(size_t)img * order_stride_img + row] : row;
                    if (rois_padded) {
                        const float *bx = boxes + (size_t)img * box_stride_img + (size_t)row * 4;
                        float *o = rois_padded + ((size_t)img * max_keep + pos) * 5;
                        o[0] = (float)img; o[1] = bx[0]; o[2] = bx[1]; o[3] = bx[2]; o[4] = bx[3];
                    }
                }
            }
            // word c+1 of the boxes kept in this chunk (already in registers), then prefetch the
            // two words of the next chunk's rows
            unsigned long long nxt = wave_or_u64(mine ? right : 0ull);
            diag = 0ull;
            right = 0ull;
            const int nrow = (c + 1) * 64 + lane;
            if (c + 1 < nchunks && nrow < n) {
                diag = m[(size_t)nrow * ncb + c + 1];
                if (c + 2 < nchunks) right = m[(size_t)nrow * ncb + c + 2];
            }
            if (lane == 0) {
                s_own[par ^ 1] = nxt;
                s_count = count + __popcll(kept);
            }
        } else if (c + 1 < nchunks) {
            // helpers: word c+1 of every box kept before this chunk
            unsigned long long acc = 0ull;
            const int lim = min(count, max_keep);
            for (int i = tid - 64; i < lim; i += NHELP) acc |= m[(size_t)kept_rows[i] * ncb + c + 1];
            acc = wave_or_u64(acc);
            if (lane == 0) s_part[par ^ 1][wave] = acc;
        }
        __syncthreads();
        count = s_count;
        if (count >= max_keep) break;
    }
    if (tid == 0) num_keep[img] = min(count, max_keep);
}


// ------------------------------------------------- nms sweep, latency-pipelined ---
// Same greedy sweep, restructured so that no global-memory latency sits on the per-chunk
// critical path (the version above spends ~1.6 us per 64-box chunk, most of it waiting for one
// dependent load; 188 chunks at 12000 boxes).  Contributions to removed[w] are split by age:
//   * boxes kept in chunks w-3..w-1: wave 0 holds words c+1..c+3 of every row of chunk c in
//     registers (loaded two chunks ahead, together with the diagonal word) and ORs the
//     survivors' words into the ring slots of words c+1..c+3 right after resolving chunk c;
//   * boxes kept in chunks < w-3: the helper waves issue word c+3 of every box kept before
//     chunk c during iteration c and consume it two iterations later, so each of their loads
//     has two full iterations to complete.
// removed[] lives in a 4-slot LDS ring (slot = word & 3) updated with ds_or_b64.  The barrier
// waits for LDS traffic only, so the loads stay in flight across it.  Needs the kept list in
// LDS and max_keep <= LH * (SWEEP_BLOCK - 64); the kernel above is the general fallback.
constexpr int SWEEP_LH = 3;

__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct SweepRows {                  // wave 0: one row per lane, words c..c+3 of chunk c
    unsigned long long w[4];
};
struct SweepPend {                  // helpers: loads in flight for one future word
    unsigned long long v[SWEEP_LH];
};

struct SweepCtx {
    const unsigned long long *m;
    int n, ncb, nchunks, max_keep, img;
    const int *order; int order_stride_img;
    int *keep; const float *boxes; int box_stride_img; float *rois_padded;
    int *kept_rows; unsigned long long *ring; int *s_count;
    int tid, lane, wave;
};

__device__ __forceinline__ void sweep_load_rows(const SweepCtx &k, int chunk, SweepRows &r) {
    const int row = chunk * 64 + k.lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.w[j] = 0ull;
        if (row < k.n && chunk + j < k.nchunks) r.w[j] = k.m[(size_t)row * k.ncb + chunk + j];
    }
}

// one chunk; `rows` holds chunk c (refilled with chunk c+2), `pend` holds the helpers' word c+1
// loads issued at iteration c-2 (refilled with word c+3).  Returns the kept count after c.
__device__ __forceinline__ int sweep_step(const SweepCtx &k, int c, int count, SweepRows &rows,
                                          SweepPend &pend) {
    constexpr int NHELP = SWEEP_BLOCK - 64;
    if (k.wave == 0) {
        unsigned long long rem = 0ull;
        if (k.lane == 0) { rem = k.ring[c & 3]; k.ring[c & 3] = 0ull; }   // slot reused by word c+4
        rem = readlane_u64(rem, 0);
        const int row = c * 64 + k.lane;
        const int nv = k.n - c * 64;
        const unsigned long long valid = (nv >= 64) ? ~0ull : ((1ull << nv) - 1ull);
        unsigned long long cur = rem | ~valid;
        unsigned long long kept = 0ull;
        unsigned long long cand = ~cur;
        const unsigned long long diag = rows.w[0];
        while (cand != 0ull) {
            const int bsel = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
            kept |= 1ull << bsel;
            unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)diag, bsel);
            unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), bsel);
            cur |= (((unsigned long long)hi << 32) | lo) | (1ull << bsel);
            cand = ~cur & ((bsel == 63) ? 0ull : (~0ull << (bsel + 1)));
        }
        const bool mine = (kept >> k.lane) & 1ull;
        // hand the survivors' next three words on first: this is what the next chunk waits for
        const unsigned long long n1 = wave_or_u64(mine ? rows.w[1] : 0ull);
        const unsigned long long n2 = wave_or_u64(mine ? rows.w[2] : 0ull);
        const unsigned long long n3 = wave_or_u64(mine ? rows.w[3] : 0ull);
        if (k.lane == 0) {
            atomicOr(&k.ring[(c + 1) & 3], n1);
            atomicOr(&k.ring[(c + 2) & 3], n2);
            atomicOr(&k.ring[(c + 3) & 3], n3);
            *k.s_count = count + __popcll(kept);
        }
        if (mine) {
            const int pos = count + __popcll(kept & ((1ull << k.lane) - 1ull));
            if (pos < k.max_keep) {
                k.kept_rows[pos] = row;
                if (k.keep)
                    k.keep[(size_t)k.img * k.max_keep + pos] =
                        k.order ? k.order[(size_t)k.img * k.order_stride_img + row] : row;
                if (k.rois_padded) {
                    const float *bx = k.boxes + (size_t)k.img * k.box_stride_img + (size_t)row * 4;
                    float *o = k.rois_padded + ((size_t)k.img * k.max_keep + pos) * 5;
                    o[0] = (float)k.img; o[1] = bx[0]; o[2] = bx[1]; o[3] = bx[2]; o[4] = bx[3];
                }
            }
        }
        sweep_load_rows(k, c + 2, rows);
    } else {
        // consume word c+1 (issued at iteration c-2: boxes kept before chunk c-2)
        unsigned long long acc = 0ull;
#pragma unroll
        for (int j = 0; j < SWEEP_LH; ++j) acc |= pend.v[j];
        acc = wave_or_u64(acc);
        if (k.lane == 0 && acc != 0ull) atomicOr(&k.ring[(c + 1) & 3], acc);
        // issue word c+3 of every box kept before chunk c
        const int lim = min(count, k.max_keep);
#pragma unroll
        for (int j = 0; j < SWEEP_LH; ++j) {
            const int i = k.tid - 64 + j * NHELP;
            pend.v[j] = 0ull;
            if (i < lim && c + 3 < k.nchunks) pend.v[j] = k.m[(size_t)k.kept_rows[i] * k.ncb + c + 3];
        }
    }
    lds_only_barrier();
    return *k.s_count;
}

__global__ __launch_bounds__(SWEEP_BLOCK) void nms_sweep_pipelined_kernel(
    const unsigned long long *__restrict__ mask, const int *__restrict__ n_dev, int n_max, int ncb,
    int max_keep, const int *__restrict__ order, int order_stride_img,
    int *__restrict__ keep, int *__restrict__ num_keep,
    const float *__restrict__ boxes, int box_stride_img, float *__restrict__ rois_padded) {
    extern __shared__ int kept_lds[];                // [max_keep + 64]
    __shared__ unsigned long long s_ring[4];
    __shared__ int s_count;
    SweepCtx k;
    k.img = blockIdx.x;
    k.n = min(n_dev[k.img], n_max);
    k.ncb = ncb;
    k.nchunks = (k.n + 63) / 64;
    k.max_keep = max_keep;
    k.m = mask + (size_t)k.img * n_max * ncb;
    k.order = order; k.order_stride_img = order_stride_img;
    k.keep = keep; k.boxes = boxes; k.box_stride_img = box_stride_img; k.rois_padded = rois_padded;
    k.kept_rows = kept_lds; k.ring = s_ring; k.s_count = &s_count;
    k.tid = threadIdx.x; k.lane = k.tid & 63; k.wave = k.tid >> 6;
    if (k.tid < 4) s_ring[k.tid] = 0ull;
    if (k.tid == 0) s_count = 0;
    __syncthreads();
    SweepRows rows0, rows1;
    SweepPend pend0, pend1;
#pragma unroll
    for (int j = 0; j < 4; ++j) rows0.w[j] = rows1.w[j] = 0ull;
#pragma unroll
    for (int j = 0; j < SWEEP_LH; ++j) pend0.v[j] = pend1.v[j] = 0ull;
    if (k.wave == 0) {
        sweep_load_rows(k, 0, rows0);
        sweep_load_rows(k, 1, rows1);
    }
    int count = 0;
    for (int c = 0; c < k.nchunks; c += 2) {
        count = sweep_step(k, c, count, rows0, pend0);
        if (count >= max_keep || c + 1 >= k.nchunks) break;
        count = sweep_step(k, c + 1, count, rows1, pend1);
        if (count >= max_keep) break;
    }
    if (k.tid == 0) num_keep[k.img] = min(count, max_keep);
}

int launch_nms_sweep(const unsigned long long *mask, const int *n_dev, int n_max, int n_images,
                     int max_keep, const int *order, int order_stride_img, int *keep,
                     int *num_keep, const float *boxes, int box_stride_img, float *rois_padded,
                     int *kept_scratch, hipStream_t st) {
    int ncb = cdiv(n_max, 64);
    if (n_images == 0) return WSSDL_OK;
    size_t lds = ((size_t)max_keep + 64) * sizeof(int);
    if (lds <= SWEEP_LDS_LIMIT && max_keep <= SWEEP_LH * (SWEEP_BLOCK - 64)) {
        hipLaunchKernelGGL(nms_sweep_pipelined_kernel, dim3(n_images), dim3(SWEEP_BLOCK), lds, st,
                           mask, n_dev, n_max, ncb, max_keep, order, order_stride_img, keep,
                           num_keep, boxes, box_stride_img, rois_padded);
        return check_launch();
    }
    if (lds > SWEEP_LDS_LIMIT) {
        if (!kept_scratch) return WSSDL_ERR_WORKSPACE;
        lds = 0;
    } else {
        kept_scratch = nullptr;
    }
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(n_images), dim3(SWEEP_BLOCK), lds, st, mask, n_dev,
                       n_max, ncb, max_keep, order, order_stride_img, keep, num_keep, boxes,
                       box_stride_img, rois_padded, kept_scratch);
    return check_launch();
}

// ------------------------------------------------------- standalone nms entry ---
__global__ void nms_prepare_kernel(const float *__restrict__ dets, int n,
                                   unsigned long long *__restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = score_key(dets[(size_t)i * 5 + 4], (unsigned)i);
}

__global__ void nms_gather_kernel(const float *__restrict__ dets, const int *__restrict__ order,
                                  const int *__restrict__ n_sorted, int n,
                                  float *__restrict__ boxes) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && p < n_sorted[0]) {
        const float *d = dets + (size_t)order[p] * 5;
        boxes[p * 4 + 0] = d[0]; boxes[p * 4 + 1] = d[1]; boxes[p * 4 + 2] = d[2]; boxes[p * 4 + 3] = d[3];
    }
}

struct NmsWs {
    unsigned long long *keys, *cand, *thresh, *mask;
    int *order, *n_sorted, *cand_fill, *kept;
    float *boxes;
};

static size_t carve_nms(void *ws, int n, NmsWs *out) {
    Carver c(ws);
    int ncb = cdiv(n, 64);
    NmsWs w;
    w.keys = c.take<unsigned long long>(n);
    w.cand = c.take<unsigned long long>(n);
    w.thresh = c.take<unsigned long long>(32);
    w.order = c.take<int>(n);
    w.n_sorted = c.take<int>(64);
    w.cand_fill = c.take<int>(64);
    w.kept = c.take<int>((size_t)n + 64);
    w.boxes = c.take<float>((size_t)n * 4);
    w.mask = c.take<unsigned long long>((size_t)n * ncb);
    if (out) *out = w;
    return c.off;
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    return carve_nms(nullptr, n, nullptr);
}

extern "C" int wssdl_nms(const float *dets, int n, double thresh, int max_keep, int32_t *keep,
                         int32_t *num_keep, void *workspace, size_t workspace_bytes,
                         wssdl_stream_t stream) {
    if (n < 0 || max_keep < 0 || !num_keep) return WSSDL_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    if (n == 0 || max_keep == 0) {                     // nms_wrapper.py:16-17: empty -> []
        if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), st) != hipSuccess) return WSSDL_ERR_LAUNCH;
        return WSSDL_OK;
    }
    if (!dets || !keep || !workspace) return WSSDL_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < wssdl_nms_workspace_bytes(n)) return WSSDL_ERR_WORKSPACE;
    NmsWs w;
    carve_nms(workspace, n, &w);
    if (hipMemsetAsync(w.order, 0xff, sizeof(int) * (size_t)n, st) != hipSuccess ||
        hipMemsetAsync(w.n_sorted, 0, sizeof(int), st) != hipSuccess ||
        hipMemsetAsync(w.cand_fill, 0, sizeof(int), st) != hipSuccess)
        return WSSDL_ERR_LAUNCH;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dets, n, w.keys);
    int rc = check_launch();
    if (rc) return rc;
    rc = launch_rank_topk(w.keys, n, 1, n, w.cand, w.thresh, w.cand_fill, w.order, w.n_sorted, w.mask,
                          sizeof(unsigned long long) * (size_t)n * cdiv(n, 64), st);
    if (rc) return rc;
    hipLaunchKernelGGL(nms_gather_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dets, w.order,
                       w.n_sorted, n, w.boxes);
    rc = check_launch();
    if (rc) return rc;
    rc = launch_nms_mask(w.boxes, n * 4, w.n_sorted, n, 1, thresh, w.mask, st);
    if (rc) return rc;
    return launch_nms_sweep(w.mask, w.n_sorted, n, 1, max_keep, w.order, n, keep, num_keep,
                            nullptr, 0, nullptr, w.kept, st);
}
