// Greedy NMS + score ranking for gfx950 (MI355X), batched over images.
//
// Reference: code/lib/nms/cpu_nms.pyx:17-68 (the NMS the RPN actually uses:
// fast_rcnn/config.py:321 USE_GPU_NMS=False) via fast_rcnn/nms_wrapper.py:13-21;
// sort: rpn_msr/proposal_layer_tf_bus.py:129-133.
//
// Three stages, each launched once for ALL images of a step:
//   top-K rank  : 64-bit keys = (order-preserving score bits << 32 | index) make the
//                 order total (ties: higher index first) and the result
//                 deterministic.  Threshold by radix select (select.hip.h),
//                 compaction, then a sample sort: splitters from a sorted sample,
//                 position = bucket offset + number of greater keys in the bucket
//                 (all-pairs counting rank_topk_kernel remains for small topn).
//   nms_mask    : 64x64 tiles of the upper triangle of the suppression matrix,
//                 one u64 word per (row box, column block): 64 column boxes in
//                 LDS, one row box per lane.  f32 arithmetic in cpu_nms.pyx's
//                 operation order, test (double)iou >= thresh (vendored
//                 cpu_nms.c:2495 compares PyFloat objects).  The diagonal block
//                 is also emitted transposed (diag_t) for the sweep's resolver.
//   nms_sweep   : one workgroup per image walks the 64-row chunks in order and
//                 stops as soon as max_keep boxes are kept (the reference's caller
//                 truncates keep[:post_nms_topN]).  nms_sweep_pipelined_kernel
//                 (roles: resolver / scribes / stagers / helpers, see there) is
//                 the fast path; nms_sweep_kernel the general fallback (kept list
//                 too long for LDS or for the helpers' registers).
#include "nms.hip.h"
#include "select.hip.h"

namespace wssdl {

// ---------------------------------------------------------------- rank/top-k ---
// Descending order of the topn largest 64-bit keys of each image, in three steps:
//   topk_threshold : the topn-th largest key by MSB-first radix select (one workgroup per
//                    image; keys are unique), then keys >= threshold -> dense candidate
//                    array (any order) by the same workgroup;
//   rank_topk      : position of every candidate = number of greater candidates, counted over
//                    LDS-staged tiles -- O(topn^2) instead of O(M^2) compares.
// 64-bit keys = (order-preserving score bits << 32 | index) make the order total (ties:
// higher index first) and the result deterministic.
constexpr int SEL_BLOCK = 1024;
constexpr int SEL_LIST = 512;

__global__ __launch_bounds__(SEL_BLOCK) void topk_threshold_kernel(
    const unsigned long long *__restrict__ keys, int M, int topn,
    unsigned long long *__restrict__ thresh, int *__restrict__ n_sorted,
    unsigned long long *__restrict__ cand) {
    __shared__ SelectScratch<SEL_LIST> sc;
    __shared__ int s_fill;
    const int img = blockIdx.x, t = threadIdx.x;
    const unsigned long long *k = keys + (size_t)img * M;
    int valid = 0;
    // topn-th largest key; with no more than topn valid keys every one of them is a candidate
    const unsigned long long th = block_radix_select<SEL_BLOCK, SEL_LIST, true>(
        [k](int i, unsigned long long &v) { v = k[i]; return v != 0ull; }, M,
        [topn](int members) { return members > topn ? topn : 0; }, sc, &valid);
    const unsigned long long cut = valid > topn ? th : 1ull;
    if (t == 0) {
        n_sorted[img] = min(valid, topn);
        thresh[img] = cut;
        s_fill = 0;
    }
    __syncthreads();
    // compaction by the same workgroup: keys >= cut -> dense candidate array, any order (the
    // ranking that follows orders them).  One LDS atomic per wave and step; a separate kernel
    // with global atomics on one counter took 33 us for this.
    const int lane = t & 63;
    for (int i0 = 0; i0 < M; i0 += SEL_BLOCK) {
        const int i = i0 + t;
        const unsigned long long v = (i < M) ? k[i] : 0ull;
        const bool take = v != 0ull && v >= cut;
        const unsigned long long m = __ballot(take);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&s_fill, __popcll(m));
        base = __builtin_amdgcn_readlane(base, leader);
        if (take) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < topn) cand[(size_t)img * topn + pos] = v;
        }
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
    (void)cand_fill;
    hipLaunchKernelGGL(topk_threshold_kernel, dim3(n_images), dim3(SEL_BLOCK), 0, st, keys, M, topn,
                       thresh, n_sorted, cand);
    int rc = check_launch();
    if (rc) return rc;
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
// (fmax_ref / fmin_ref / fmax0_ref / box_area_ref: nms.hip.h)

// ---- shared by the mask kernel and the sweeps (the sweep section explains them) ----
constexpr int SWEEP_BLOCK = 1024;
constexpr size_t SWEEP_LDS_LIMIT = 60 * 1024;     // kept list in LDS up to ~15k entries
constexpr int SWEEP_LH = 7;
constexpr int SWEEP_FIRST_HELPER = 6;                                   // wave index
constexpr int SWEEP_GROUP = (SWEEP_BLOCK / 64 - SWEEP_FIRST_HELPER) / 2 * 64;   // threads per helper group
constexpr int SWEEP_AHEAD = 4;          // words right of the diagonal that wave 0 handles itself
// Band of mask words right of the diagonal that the pipelined sweep reads whatever the summary says:
// the stagers fetch words c+1..c+SWEEP_AHEAD of chunk c's rows.
constexpr int NMS_DENSE_AHEAD = SWEEP_AHEAD;
constexpr int SWEEP_MAX_CHUNKS = SWEEP_GROUP;     // a helper lane per chunk

// Row pitch of the suppression matrix in u64 words: whole 128-byte lines per segment of MASK_SEG = 16
// column blocks, so that no line holds words of two segments (the fused launch hands the matrix over
// segment by segment; a line fetched while a neighbour segment is still being written would go stale
// in the reader's L2).
int nms_mask_pitch(int n_max) { return (cdiv(n_max, 64) + 15) / 16 * 16; }

// Whether launch_nms_sweep will take the role-pipelined kernel (which reads mask words beyond the
// dense band only where the summary has a bit) or the general one (which reads every word).
static bool nms_sweep_is_pipelined(int n_max, int max_keep, const void *diag_t, const void *summ) {
    return diag_t && summ && cdiv(n_max, 64) <= SWEEP_MAX_CHUNKS && max_keep <= SWEEP_LH * SWEEP_GROUP && n_max < (1 << 24) &&
           (long long)n_max * nms_mask_pitch(n_max) < (1LL << 31);
}

constexpr int MASK_WAVES = 4;


constexpr int MASK_SEG = 16;     // column blocks per workgroup

// One "mask block" per (64-row block, segment of 16 column blocks); its 4 waves stride over the
// segment's column blocks cb >= rb (upper triangle only).  A lane keeps its row box in
// registers; the 64 column boxes of the wave's current block sit in that wave's LDS slice and
// are read as broadcasts.  The waves of a mask block are independent (no workgroup barrier), so the
// body below serves the stand-alone kernel (one block = 4 waves) and the fused mask + sweep kernel
// (16 waves of a workgroup = 4 mask blocks): `wave` = 0..3 inside the mask block, cbox_w / cgeo_w =
// the LDS slice of the calling wave.
struct MaskArgs {
    const float *boxes;  int box_stride_img;  const int *n_dev;  int n_max;
    double thresh;  unsigned long long *mask;  int ncb;          // ncb: row pitch of mask (nms_mask_pitch)
    unsigned long long *diag_t;  unsigned long long *summ;      // summ: [n_images][ncb column blocks][ncb row blocks]
    int n_limit;  int cb_min;  const int *done;  int dense_ahead;
    // fused launch: finished[img * finished_stride] != 0 once the image's sweep has all it wants (the block is
    // skipped); NULL otherwise
    const int *finished;  int finished_stride;
    // 0: the rule of cpu_nms.pyx / utils/nms.pyx `nms` (ovr >= thresh); 1: utils/nms.pyx:118-121 `nms_new`
    // (ovr >= thresh or inter / area_i > 0.95 or inter / area_j > 0.95)
    int rule;
    // fused launch: keptpub[(img * ncb + chunk) * 2 + h] = half h of the chunk's kept bitmask | (chunk + 1) << 32, written
    // by the image's sweep once the chunk is resolved (zeroed by the launcher); NULL otherwise
    const unsigned long long *keptpub;
    int sparse_max;       // most kept boxes of a row block for which the kept-rows form is taken
};

// COHERENT: the words are read by a sweep that runs beside this kernel, possibly on another XCD (whose
// L2 is not coherent with this one's for ordinary stores): they are stored write-through at agent scope,
// so that the release of the row-block counter needs no L2 write-back (a buffer_wbl2 per wave made the
// fused launch 4x slower than the two it replaces).
template <bool COHERENT>
__device__ __forceinline__ void nms_store_word(unsigned long long *p, unsigned long long v) {
    if (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Round 5: rows of KEPT boxes only.  Greedy NMS consults the rows of kept boxes and nothing else, and in the fused
// launch the sweep is usually PAST a row block by the time the mask role reaches the block's far column segments
// (at chunk c the sweep needs columns <= c + 7; segment s of row block rb with 16 s > rb + 4 is computed when the sweep
// is near chunk 16 s - 8).  The sweep's spare wave publishes every resolved chunk's kept bitmask (`keptpub`); a mask
// block that finds its row block resolved -- with at most `sparse_max` (16) of its 64 boxes kept: beyond ~40 % kept the
// dense block wins -- runs this TRANSPOSED form: column boxes in lanes, a scalar loop over the kept rows (the row boxes
// stay in the lanes' registers, lane = row, and reach the scalar registers with v_readlane: an LDS broadcast read per
// kept row put a dependent LDS round trip into the loop), the word of (row, column block) = one ballot.
// ~8 instructions per (kept row, column block) without a candidate (~45 with one) against ~17 per (row, column block)
// of the dense form, and 10-30 % of the rows are kept.  Same pair arithmetic (cpu_nms.pyx:43-66: every step is symmetric in the two boxes), so the
// words of kept rows are bit for bit those of the dense form; rows that were not kept get no words and no summary
// bits -- nothing reads them (the helpers walk the kept list, the dense band next to the diagonal stays dense).
template <bool COHERENT>
__device__ __forceinline__ void nms_mask_block_sparse(const MaskArgs &A, int rb, int seg, int img, int wave, int lane, int n,
                                                      unsigned long long kept, float ix1, float iy1, float ix2, float iy2) {
    const float *__restrict__ b = A.boxes + (size_t)img * A.box_stride_img;
    const int n_max = A.n_max, ncb = A.ncb;
    const double thresh = A.thresh;
    unsigned long long *__restrict__ mask = A.mask, *__restrict__ summ = A.summ;
    const float t_lo = (float)(thresh * (1.0 - 1e-4)), t_hi = (float)(thresh * (1.0 + 1e-4));
    const double fq = thresh / (1.0 + thresh);
    const bool contain = A.rule == 1;
    const bool prefilter = !contain && thresh >= 0.25 && thresh < 1.0;
    const float kq = (float)((0.5 - fq) * (1.0 + 1e-3));
    auto geometry = [&](float x1, float y1, float x2, float y2, bool live) -> nms_float4v {      // as in nms_mask_block
        float w = x2 - x1;  w = w + 1.0f;
        float h = y2 - y1;  h = h + 1.0f;
        const bool sane = w > 0.0f && h > 0.0f;
        nms_float4v g;
        g.x = live ? x1 + 0.5f * w : __builtin_nanf("");
        g.y = y1 + 0.5f * h;
        g.z = sane ? w * kq + 1e-3f : INFINITY;
        g.w = sane ? h * kq + 1e-3f : INFINITY;
        return g;
    };
    // the row boxes stay in the lanes' registers (lane = row); a kept row's values reach the scalar registers with
    // v_readlane (no LDS round trip inside the row loop)
    const float iarea = box_area_ref(ix1, iy1, ix2, iy2);
    const nms_float4v ig = geometry(ix1, iy1, ix2, iy2, true);
    auto bcast = [](float v, int r) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
    const int cb_first = max(rb, A.cb_min);
    const int cb_end = min((n + 63) / 64, (seg + 1) * MASK_SEG);
    for (int cb = max(cb_first, seg * MASK_SEG) + wave; cb < cb_end; cb += MASK_WAVES) {
        const int col = cb * 64 + lane;
        const bool col_ok = col < n;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        if (col_ok) {
            const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)col * 4);
            x1 = v.x; y1 = v.y; x2 = v.z; y2 = v.w;
        }
        const float carea = box_area_ref(x1, y1, x2, y2);
        const nms_float4v cg = geometry(x1, y1, x2, y2, col_ok);
        unsigned long long nz = 0ull, todo = kept;
        while (todo != 0ull) {
            const int r = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            unsigned long long cand;
            if (prefilter) {
                const float dx = cg.x - bcast(ig.x, r), dy = cg.y - bcast(ig.y, r);
                const float sx = cg.z + bcast(ig.z, r), sy = cg.w + bcast(ig.w, r);
                cand = __builtin_amdgcn_fcmpf(__builtin_fabsf(dx), sx, 5 /* ole */) &
                       __builtin_amdgcn_fcmpf(__builtin_fabsf(dy), sy, 5 /* ole */);
            } else {
                cand = __ballot(col_ok);
            }
            if (cand == 0ull) continue;
            const float rx1 = bcast(ix1, r), ry1 = bcast(iy1, r), rx2 = bcast(ix2, r), ry2 = bcast(iy2, r), rarea = bcast(iarea, r);
            bool hit = false;
            if ((cand >> lane) & 1ull) {
                const float xx1 = fmax_ref(rx1, x1);
                const float yy1 = fmax_ref(ry1, y1);
                const float xx2 = fmin_ref(rx2, x2);
                const float yy2 = fmin_ref(ry2, y2);
                float w = xx2 - xx1;  w = fmax0_ref(w + 1.0f);
                float h = yy2 - yy1;  h = fmax0_ref(h + 1.0f);
                const float inter = w * h;
                float den = rarea + carea;
                den = den - inter;
                const bool yes = inter > den * t_hi, no = inter < den * t_lo;
                if (contain) {
                    hit = ((den > 0.0f) & yes) || (double)(inter / den) >= thresh || (double)(inter / rarea) > 0.95 ||
                          (double)(inter / carea) > 0.95;
                } else if ((den > 0.0f) & (yes | no)) {
                    hit = yes;
                } else {
                    hit = (double)(inter / den) >= thresh;
                }
                hit = hit && col_ok;
            }
            const unsigned long long word = __ballot(hit);
            if (word != 0ull) {
                if (lane == 0) nms_store_word<COHERENT>(&mask[((size_t)img * n_max + rb * 64 + r) * ncb + cb], word);
                nz |= 1ull << r;
            }
        }
        if (summ && lane == 0) nms_store_word<COHERENT>(&summ[((size_t)img * ncb + cb) * ncb + rb], nz);
    }
}

template <bool COHERENT>
__device__ __forceinline__ void nms_mask_block(const MaskArgs &A, int rb, int seg, int img, int wave, int lane,
                                               float (*cbox_w)[64] /* [5][64] */, nms_float4v *cgeo_w /* [64] */) {
    const float *__restrict__ boxes = A.boxes;
    const int box_stride_img = A.box_stride_img, n_max = A.n_max, ncb = A.ncb, n_limit = A.n_limit,
              cb_min = A.cb_min, dense_ahead = A.dense_ahead;
    const double thresh = A.thresh;
    unsigned long long *__restrict__ mask = A.mask, *__restrict__ diag_t = A.diag_t, *__restrict__ summ = A.summ;
    // two-pass use (launch_nms_two_pass): the first pass covers the candidates below n_limit only,
    // the second one the column blocks >= cb_min of the images the first pass could not finish
    if (A.done && A.done[img]) return;
    const int n = min(min(A.n_dev[img], n_max), n_limit);
    const int cb_first = max(rb, cb_min);
    if (rb * 64 >= n || (seg + 1) * MASK_SEG <= cb_first || seg * MASK_SEG * 64 >= n) return;
    const float *b = boxes + (size_t)img * box_stride_img;
    const int i = rb * 64 + lane;
    const bool row_ok = i < n;
    // (the finished flag travels beside the row boxes: one exposed latency for both)
    int fin = 0;
    if (COHERENT && A.finished)
        fin = __hip_atomic_load(A.finished + (size_t)img * A.finished_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float ix1 = 0.f, iy1 = 0.f, ix2 = 0.f, iy2 = 0.f;
    if (row_ok) {
        const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)i * 4);
        ix1 = v.x; iy1 = v.y; ix2 = v.z; iy2 = v.w;
    }
    // (likewise the kept bitmask of this row block, if the sweep has published it: two tagged halves)
    unsigned long long kp0 = 0ull, kp1 = 0ull;
    const bool far = COHERENT && A.keptpub && dense_ahead >= 0 && seg * MASK_SEG > rb + dense_ahead && max(rb, cb_min) <= seg * MASK_SEG;
    if (far) {
        kp0 = __hip_atomic_load(A.keptpub + ((size_t)img * ncb + rb) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        kp1 = __hip_atomic_load(A.keptpub + ((size_t)img * ncb + rb) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (fin) return;
    if (far && (unsigned)(kp0 >> 32) == (unsigned)(rb + 1) && (unsigned)(kp1 >> 32) == (unsigned)(rb + 1)) {
        const unsigned long long kept = (kp0 & 0xffffffffull) | (kp1 << 32);
        // (a row loop per kept box pays off while few are kept; a chunk that kept most of its boxes takes the dense form)
        if (__popcll(kept) <= A.sparse_max) {
            nms_mask_block_sparse<COHERENT>(A, rb, seg, img, wave, lane, n, kept, ix1, iy1, ix2, iy2);
            return;
        }
    }
    const float iarea = box_area_ref(ix1, iy1, ix2, iy2);
    const float t_lo = (float)(thresh * (1.0 - 1e-4)), t_hi = (float)(thresh * (1.0 + 1e-4));
    // Prefilter.  ovr >= t  =>  inter >= f (area_i + area_j), f = t / (1 + t); with
    // inter = ow * oh and oh <= min(h_i, h_j):  ow >= f (w_i + w_j), and for two intervals
    // ow <= (w_i + w_j) / 2 - |cx_i - cx_j|.  So a pair can only be suppressed when
    //     |cx_i - cx_j| <= k (w_i + w_j)  and  |cy_i - cy_j| <= k (h_i + h_j),   k = 1/2 - f
    // (k = 0.088 at t = 0.7: about 1 % of the pairs of a proposal set pass).  Each side brings its
    // own radius (k w + slack, k h + slack), so a pair costs two packed adds, two compares and one
    // add-with-carry that shifts the verdict into the lane's candidate word -- 5 VALU against
    // ~28 for the decision itself.  k is widened by 1e-3 relative and each radius by 1e-3 px for the
    // f32 rounding of centres and radii; boxes with a non-positive width or height (for which the
    // reference's arithmetic has no geometric meaning) get an infinite radius and always pass.
    // Only used for 0.25 <= t < 1 (k > 0).
    const double fq = thresh / (1.0 + thresh);
    // nms_new's containment terms suppress a small box anywhere inside a large one: no centre-distance bound
    const bool contain = A.rule == 1;
    const bool prefilter = !contain && thresh >= 0.25 && thresh < 1.0;
    const float kq = (float)((0.5 - fq) * (1.0 + 1e-3));
    auto geometry = [&](float x1, float y1, float x2, float y2, bool live) -> nms_float4v {
        float w = x2 - x1;  w = w + 1.0f;
        float h = y2 - y1;  h = h + 1.0f;
        const bool sane = w > 0.0f && h > 0.0f;
        nms_float4v g;
        g.x = live ? x1 + 0.5f * w : __builtin_nanf("");      // columns past the end never pass
        g.y = y1 + 0.5f * h;
        g.z = sane ? w * kq + 1e-3f : INFINITY;
        g.w = sane ? h * kq + 1e-3f : INFINITY;
        return g;
    };
    const nms_float4v ig = geometry(ix1, iy1, ix2, iy2, true);
    const nms_float2v ic = {ig.x, ig.y}, ir = {ig.z, ig.w};
    const int cb_end = min((n + 63) / 64, (seg + 1) * MASK_SEG);
    for (int cb = max(cb_first, seg * MASK_SEG) + wave; cb < cb_end; cb += MASK_WAVES) {
        const int col = cb * 64 + lane;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        if (col < n) {
            const nms_float4v v = *reinterpret_cast<const nms_float4v *>(b + (size_t)col * 4);
            x1 = v.x; y1 = v.y; x2 = v.z; y2 = v.w;
        }
        // the wave's own slice: wave-synchronous, no workgroup barrier needed
        cbox_w[0][lane] = x1; cbox_w[1][lane] = y1; cbox_w[2][lane] = x2;
        cbox_w[3][lane] = y2; cbox_w[4][lane] = box_area_ref(x1, y1, x2, y2);
        if (prefilter) cgeo_w[lane] = geometry(x1, y1, x2, y2, col < n);
        __builtin_amdgcn_wave_barrier();
        const int jn = min(64, n - cb * 64);
        unsigned b_lo = 0u, b_hi = 0u, u_lo = 0u, u_hi = 0u;
        if (prefilter) {
            // candidates only: the exact loop below decides them.  Column j's verdict enters the
            // word through the carry (u = 2 u + verdict), highest column first.
            auto near = [&](int j, unsigned u) -> unsigned {
                const nms_float4v q = cgeo_w[j];
                const nms_float2v d = ic - q.xy;
                const nms_float2v r = ir + q.zw;
                const unsigned long long px = __builtin_amdgcn_fcmpf(__builtin_fabsf(d.x), r.x, 5 /* ole */);
                const unsigned long long py = __builtin_amdgcn_fcmpf(__builtin_fabsf(d.y), r.y, 5 /* ole */);
                const unsigned long long both = px & py;
                unsigned long long carry_out;
                asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(u), "=s"(carry_out) : "v"(u), "s"(both));
                return u;
            };
#pragma unroll 8
            for (int j = 31; j >= 0; --j) u_lo = near(j, u_lo);
#pragma unroll 8
            for (int j = 63; j >= 32; --j) u_hi = near(j, u_hi);
        } else {
            // Pass 1, straight-line per column box: decide (double)(inter / den) >= thresh without
            // the IEEE division whenever the quotient is at least 1e-4 (relative) away from the
            // threshold -- the f32 rounding of the quotient (2^-24) cannot cross that margin;
            // the undecided pairs (and den <= 0) are only marked.  Pass 2 settles the marked pairs
            // with the exact test; they are rare, so pass 1 carries no branch.
            // (two 32-column halves: 32-bit shift-or per flag instead of 64-bit shifts)
            auto pair_flags = [&](int j, unsigned &yes_bit, unsigned &und_bit) {
                const float xx1 = fmax_ref(ix1, cbox_w[0][j]);
                const float yy1 = fmax_ref(iy1, cbox_w[1][j]);
                const float xx2 = fmin_ref(ix2, cbox_w[2][j]);
                const float yy2 = fmin_ref(iy2, cbox_w[3][j]);
                float w = xx2 - xx1;  w = fmax0_ref(w + 1.0f);
                float h = yy2 - yy1;  h = fmax0_ref(h + 1.0f);
                const float inter = w * h;
                float den = iarea + cbox_w[4][j];
                den = den - inter;
                const bool yes = inter > den * t_hi;
                const bool no = inter < den * t_lo;
                bool sure = (den > 0.0f) & (yes | no);
                // nms_new: "no" is only final for pairs that do not intersect at all (inter == 0: the two
                // containment quotients are 0 or NaN); every other pair goes to the exact loop
                if (contain) sure = (den > 0.0f) & (yes | ((inter == 0.0f) & no));
                yes_bit = (sure & yes) ? 1u : 0u;
                und_bit = sure ? 0u : 1u;
            };
            const int j_lo = min(jn, 32);
            for (int j = 0; j < j_lo; ++j) {
                unsigned y, u;
                pair_flags(j, y, u);
                b_lo |= y << j;
                u_lo |= u << j;
            }
            for (int j = 32; j < jn; ++j) {
                unsigned y, u;
                pair_flags(j, y, u);
                b_hi |= y << (j - 32);
                u_hi |= u << (j - 32);
            }
        }
        unsigned long long bits = ((unsigned long long)b_hi << 32) | b_lo;
        unsigned long long undecided = ((unsigned long long)u_hi << 32) | u_lo;
        while (undecided != 0ull) {
            const int j = __ffsll((long long)undecided) - 1;
            undecided &= undecided - 1ull;
            const float xx1 = fmax_ref(ix1, cbox_w[0][j]);
            const float yy1 = fmax_ref(iy1, cbox_w[1][j]);
            const float xx2 = fmin_ref(ix2, cbox_w[2][j]);
            const float yy2 = fmin_ref(iy2, cbox_w[3][j]);
            float w = xx2 - xx1;  w = fmax0_ref(w + 1.0f);
            float h = yy2 - yy1;  h = fmax0_ref(h + 1.0f);
            const float inter = w * h;
            float den = iarea + cbox_w[4][j];
            den = den - inter;
            // the IEEE division (and the f64 compare) only where the quotient is within 1e-4 of the threshold:
            // further away the f32 rounding of the quotient cannot cross it (the loop took 28 % of the kernel with
            // a division per candidate)
            const bool yes = inter > den * t_hi, no = inter < den * t_lo;
            if (contain) {
                // utils/nms.pyx:118-120: ovr is a C float widened for the compare with the Python float `thresh`;
                // ovr1 / ovr2 are untyped, i.e. the f32 quotients as Python floats, compared with 0.95 in f64
                if (((den > 0.0f) & yes) || (double)(inter / den) >= thresh || (double)(inter / iarea) > 0.95 ||
                    (double)(inter / cbox_w[4][j]) > 0.95)
                    bits |= 1ull << j;
            } else if ((den > 0.0f) & (yes | no)) {
                if (yes) bits |= 1ull << j;
            } else if ((double)(inter / den) >= thresh) {
                bits |= 1ull << j;
            }
        }
        if (cb == rb) {
            // diagonal block.  The test is symmetric in the two boxes (max/min and the area sum
            // commute), so the same word also holds the transposed block: bits below the
            // lane's own index = the EARLIER boxes of this chunk that suppress box i (what the
            // sweep's resolver wants per lane); the mask proper keeps the bits above it.
            const unsigned long long below = (1ull << lane) - 1ull;
            if (row_ok && diag_t) nms_store_word<COHERENT>(&diag_t[(size_t)img * n_max + i], bits & below);
            bits &= ~(below | (1ull << lane));
        }
        // Over 99 % of the words are zero.  A reader that goes by the summary (the pipelined sweep)
        // never fetches those, so they are not stored either: dense_ahead >= 0 keeps only the
        // words up to dense_ahead blocks right of the diagonal unconditional (the sweep's stagers
        // read that band without looking at the summary); dense_ahead < 0 stores every word.
        if (row_ok && (bits != 0ull || dense_ahead < 0 || cb - rb <= dense_ahead))
            nms_store_word<COHERENT>(&mask[((size_t)img * n_max + i) * ncb + cb], bits);
        // summary: which words of this column block are non-zero at all (most are zero: a box overlaps
        // few others), one bit per row, one u64 per (column block, row block) -- every entry of the upper
        // triangle is written exactly once (nothing to zero, no atomics); the sweep's helpers fetch a
        // column's summary for all rows at once and only load the words it names
        if (summ) {
            const unsigned long long nz = __ballot(row_ok && bits != 0ull);
            if (lane == 0) nms_store_word<COHERENT>(&summ[((size_t)img * ncb + cb) * ncb + rb], nz);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64 * MASK_WAVES) void nms_mask_kernel(MaskArgs A) {
    __shared__ float cbox[MASK_WAVES][5][64];      // x1 y1 x2 y2 area
    __shared__ nms_float4v cgeo[MASK_WAVES][64];   // cx cy rx ry of the column boxes: one 16-byte read per pair
    // wave index through readfirstlane: the column-block loop and its trip counts are then
    // scalar (loop control on the SALU)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nms_mask_block<false>(A, blockIdx.x, blockIdx.y, blockIdx.z, wave, lane, cbox[wave], cgeo[wave]);
}

int launch_nms_mask(const float *boxes, int box_stride_img, const int *n_dev, int n_max,
                    int n_images, double thresh, unsigned long long *mask,
                    unsigned long long *diag_t, unsigned long long *summ, hipStream_t st,
                    int n_limit, int cb_min, const int *done, int max_keep, int rule) {
    if (n_max <= 0 || n_images == 0) return WSSDL_OK;
    const int ncb = nms_mask_pitch(n_max);
    // (the second pass of a two-pass run adds the column blocks >= cb_min to the summary of the first)
    const int ncb_eff = cdiv(min(n_max, n_limit), 64);      // row / column blocks this pass can touch
    const MaskArgs A = {boxes, box_stride_img, n_dev, n_max, thresh, mask, ncb, diag_t, summ, n_limit, cb_min, done,
                        nms_sweep_is_pipelined(n_max, max_keep, diag_t, summ) ? NMS_DENSE_AHEAD : -1, nullptr, 0, rule, nullptr, 0};
    hipLaunchKernelGGL(nms_mask_kernel, dim3(ncb_eff, cdiv(ncb_eff, MASK_SEG), n_images), dim3(64 * MASK_WAVES), 0, st, A);
    return check_launch();
}

// ----------------------------------------------------------------- nms sweep ---

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
    int *__restrict__ kept_scratch, int n_limit, const int *__restrict__ done_in, int *__restrict__ done_out) {
    extern __shared__ int kept_lds[];
    if (done_in && done_in[blockIdx.x]) return;                // [max_keep + 64] rows kept so far, in order
    // (global scratch instead when the list does not fit in LDS: same-CU visibility after the
    // workgroup barrier is all that is needed)
    int *kept_rows = kept_scratch ? kept_scratch + (size_t)blockIdx.x * (max_keep + 64) : kept_lds;
    __shared__ unsigned long long s_part[2][SWEEP_BLOCK / 64];   // helpers' partial ORs, by parity
    __shared__ unsigned long long s_own[2];          // wave 0's contribution for the next chunk
    __shared__ int s_count;
    const int img = blockIdx.x;
    const int n = min(min(n_dev[img], n_max), n_limit);
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
    if (tid == 0) {
        num_keep[img] = min(count, max_keep);
        if (done_out) done_out[img] = (count >= max_keep || n_dev[img] <= n_limit) ? 1 : 0;
    }
}


// ------------------------------------------- nms sweep, role-pipelined ---
// Same greedy sweep with the per-chunk critical path cut down to what is inherently serial.
// Measured on the version above (12000 boxes, 188 chunks): ~1.6 us per chunk, of which the
// resolver's scalar loop costs ~36 ns per kept box and the rest is wave 0 fetching mask words,
// storing results, reducing the next `removed` word, and global-memory latency between chunks.
// Here the 16 waves of the workgroup have four roles:
//   wave 0  (resolver) per chunk c: reads removed[c], the TRANSPOSED diagonal block (lane j:
//           which earlier boxes of the chunk suppress box j; written by the mask kernel) and
//           the chunk's words c+1..c+4 from LDS.  The kept set K is the unique fixed point of
//               K[j] = cand[j] and not (T[j] & K)
//           (a bit only depends on lower bits), reached by iterating K <- cand & ~ballot(T & K)
//           from K = cand: each round is 4 vector + 3 scalar instructions for the whole chunk and
//           fixes at least one more position; 2-4 rounds in practice instead of one scalar
//           round per kept box.  Then the survivors' words c+1..c+4 are ORed into the LDS ring
//           of future `removed` words (ds_or, no reduction) and the kept bitmask is published;
//   waves 1-2 (scribes) one chunk behind: expand the published bitmask into the kept list (LDS)
//           and fetch what the global outputs need (score-order index, box); the global stores
//           follow two iterations later;
//   waves 3-4 (stagers) bring T and the mask words c+1..c+4 of the rows of chunk c into LDS:
//           loaded into registers at iteration c-3, written to LDS at iteration c-1;
//   waves 6..15 (helpers) cover the boxes kept five or more chunks before a word: at iteration
//           c they issue word c+3 of every box in the kept list and consume it at iteration c+2.
// Scribes, stagers and helpers work in two alternating groups (one acts on even, one on odd
// iterations), so a wave never has a newer batch of loads in flight when it consumes an older
// one: the compiler cannot count loads in flight across loop iterations and waits for ALL of
// them (s_waitcnt vmcnt(0)); with interleaved batches that put a full memory latency into
// every iteration, with alternating groups the wait is for loads issued two iterations earlier.
// removed[w] = ring[w & 7]: wave 0 contributes chunks w-4..w-1, the helpers chunks <= w-5.
// The per-iteration barrier waits for LDS traffic only; global loads stay in flight across it.
// Needs the kept list in LDS and max_keep <= SWEEP_LH * SWEEP_GROUP; the kernel above is the
// general fallback.

__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct SweepShared {
    unsigned long long ring[8];
    unsigned long long rowbuf[2][SWEEP_AHEAD + 1][64];   // [parity][0 = T, j = word c+j][row]
    unsigned long long colsum[4][SWEEP_MAX_CHUNKS];      // [column block & 3][row block]: the column's summary
    struct __attribute__((aligned(16))) Publish {   // written by the resolver with one 16-byte store
        unsigned long long kept;                    // kept bitmask of the chunk
        int base;                                   // boxes kept before the chunk
        int count;                                  // boxes kept after it
    } pub[2];                                       // by chunk parity
    int timed_out;                                  // fused run: the mask kernel never finished a row block
};

struct SweepArgs {
    const unsigned long long *mask, *diag_t, *summ;
    const int *n_dev;  int n_max, ncb, max_keep;  const int *order;  int order_stride_img;     // ncb: row pitch
    int *keep, *num_keep;  const float *boxes;  int box_stride_img;  float *rois_padded;
    int n_limit;  const int *done_in;  int *done_out;
    // fused with the mask kernel (nms_mask_sweep_fused_kernel): segdone[img * ncb + s] counts the waves of
    // the mask blocks of column segment s (MASK_SEG column blocks, all row blocks above the diagonal) that
    // have finished; the segment is complete at MASK_WAVES * min((s + 1) * MASK_SEG, nrb).  NULL: the mask is
    // complete before the sweep starts.
    const int *segdone;  int nrb;
    // fused run: how long a stager waits for a column segment without progress before it gives up, in ticks of
    // the 100 MHz real-time counter (tuning "nms_wait_us", 50 ms, unless the fault-injection knob shortens it)
    unsigned long long wait_ticks;
    // fused run: where the spare wave publishes the kept bitmask of every resolved chunk for the mask role
    // (MaskArgs::keptpub); NULL: nobody reads it
    unsigned long long *keptpub;
    // 1: the roles run without the per-chunk workgroup barrier (nms_sweep_async_block); tuning "nms_sweep_async"
    int async;
};

// Wait (one wave, before it reads words of column segment `index`) until the mask blocks running beside
// this sweep have finished that segment.  Bounded: after ~50 ms ("nms_wait_us") without progress the wave gives up
// and raises *timed_out; the image then reports num_keep = -1 (WSSDL_NMS_TIMED_OUT), which every consumer of
// the counts treats as an error -- never a silent zero (the GPU shared with a long-running kernel of another
// process is the realistic cause: the mask blocks this sweep waits for are then not being scheduled).
__device__ __forceinline__ void sweep_wait_segment(const int *segdone, int expected, int index, int *timed_out,
                                                   unsigned long long wait_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    while (__hip_atomic_load(segdone + index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > wait_ticks) { *timed_out = 1;  break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// what a helper slot with no word to fetch loads (the pipelined sweep): eight zero bytes
__device__ const unsigned long long wssdl_sweep_zero_word[2] = {0ull, 0ull};

#ifdef WSSDL_SWEEP_PROFILE
// tools/nms_sweep_profile.py builds the library with this macro: per wave of an image's sweep, the shader-clock
// cycles between barriers (work) and inside them (wait), summed over the walk.  [image][wave][4]: work, wait,
// iterations, realtime ticks (100 MHz) of the whole walk.  Never defined in the product build.
__device__ unsigned long long wssdl_sweep_prof[64][SWEEP_BLOCK / 64][32];   // [8 + k]: the wave's work cycles in chunks 16k .. 16k + 15
#endif

__device__ __forceinline__ void nms_sweep_pipelined_block(const SweepArgs &A, int img, int *kept_rows /* LDS [max_keep + 64] */,
                                                          SweepShared &sh) {
    const unsigned long long *__restrict__ mask = A.mask, *__restrict__ diag_t = A.diag_t, *__restrict__ summ = A.summ;
    const int n_max = A.n_max, ncb = A.ncb, max_keep = A.max_keep, order_stride_img = A.order_stride_img,
              box_stride_img = A.box_stride_img, n_limit = A.n_limit;
    const int *__restrict__ n_dev = A.n_dev, *__restrict__ order = A.order, *__restrict__ done_in = A.done_in;
    int *__restrict__ keep = A.keep, *__restrict__ num_keep = A.num_keep, *__restrict__ done_out = A.done_out;
    const float *__restrict__ boxes = A.boxes;
    float *__restrict__ rois_padded = A.rois_padded;
    if (done_in && done_in[img]) return;
    if (threadIdx.x == 0) sh.timed_out = 0;
    const int n = min(min(n_dev[img], n_max), n_limit);
    const int nchunks = (n + 63) / 64;
    const unsigned long long *m = mask + (size_t)img * n_max * ncb;
    const unsigned long long *dt = diag_t + (size_t)img * n_max;
    const unsigned long long *cs = summ + (size_t)img * ncb * ncb;        // [column block][row block]
    const unsigned long long *zero_word = wssdl_sweep_zero_word;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 8) sh.ring[tid] = 0ull;
    if (tid < 2) { sh.pub[tid].kept = 0ull; sh.pub[tid].base = 0; sh.pub[tid].count = 0; }
    __syncthreads();        // sh.timed_out is reset before any stager's first wait can raise it

    // role and group of this wave; a group acts on the iterations with (c & 1) == group
    // (a SIMD-aware placement of the roles -- resolver alone with the light roles on its SIMD,
    // at most two active helpers per SIMD -- measured 10 % slower than this plain numbering)
    const bool scribe = wave == 1 || wave == 2;
    const bool stager = wave == 3 || wave == 4;
    const bool helper = wave >= SWEEP_FIRST_HELPER;
    const int role = wave == 5 ? 4 : 0;
    const int hw = wave - SWEEP_FIRST_HELPER;
    const int group = helper ? (hw & 1) : (wave & 1);
    // A wave has ONE role, so the loop-carried 64-bit registers of the roles share one array
    // (separate arrays are all live across the loop for every wave):
    //   stager  rows[0..4]    T + words of one chunk's rows
    //   scribe  column[0..4]  the summary of one column block (entries lane, lane + 64, ...), in flight
    constexpr int CS_PER_LANE = SWEEP_MAX_CHUNKS / 64;
    static_assert(SWEEP_AHEAD + 1 <= SWEEP_LH && CS_PER_LANE <= SWEEP_LH, "roles share pend[]");
    unsigned long long pend[SWEEP_LH];
    unsigned long long (&rows)[SWEEP_LH] = pend;
    unsigned long long (&column)[SWEEP_LH] = pend;
    // scribe: outputs of one kept box, fetched but not yet stored
    int out_pos = -1, out_idx = 0;
    float out_box[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SWEEP_LH; ++j) pend[j] = 0ull;

    // stagers (fused launch): the leading column segments known to be complete.  Everything the sweep reads
    // at iteration c lies in column blocks <= c + 3 + SWEEP_AHEAD, which the stager that loads chunk c + 3
    // waits for here; what the scribes fetch at iteration c (summary of column c + 6) was covered one
    // iteration earlier (chunk c + 2: column c + 6), a barrier ago.
    int seg_ready = 0;
    auto load_rows = [&](int chunk) {
        if (A.segdone && chunk < nchunks) {
            const int want = min(chunk + SWEEP_AHEAD, nchunks - 1) / MASK_SEG + 1;
            while (seg_ready < want) {
                sweep_wait_segment(A.segdone, MASK_WAVES * min((seg_ready + 1) * MASK_SEG, A.nrb), img * ncb + seg_ready,
                                   &sh.timed_out, A.wait_ticks);
                ++seg_ready;
            }
        }
        const int row = chunk * 64 + lane;
#pragma unroll
        for (int j = 0; j <= SWEEP_AHEAD; ++j) rows[j] = 0ull;
        if (row < n) {
            rows[0] = dt[row];
#pragma unroll
            for (int j = 1; j <= SWEEP_AHEAD; ++j)
                if (chunk + j < nchunks) rows[j] = m[(size_t)row * ncb + chunk + j];
        }
    };
    // kept bitmask of `chunk` -> positions in the output + fetch of what the global outputs need
    auto expand = [&](int chunk) {
        const unsigned long long kept = sh.pub[chunk & 1].kept;
        const int base = sh.pub[chunk & 1].base;
        out_pos = -1;
        if ((kept >> lane) & 1ull) {
            const int row = chunk * 64 + lane;
            const int pos = base + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < max_keep) {
                kept_rows[pos] = row;
                out_pos = pos;
                out_idx = row;
                if (keep && order) out_idx = order[(size_t)img * order_stride_img + row];
                if (rois_padded) {
                    const float *bx = boxes + (size_t)img * box_stride_img + (size_t)row * 4;
                    out_box[0] = bx[0]; out_box[1] = bx[1]; out_box[2] = bx[2]; out_box[3] = bx[3];
                }
            }
        }
    };
    // scribes also stage the column summaries for the helpers: the summary of column block cb (one u64 per row
    // block <= cb) is fetched at the scribe group's turn cb - 6, written to LDS at its next turn cb - 4 and read
    // by the helpers at iteration cb - 3 (the mask blocks have finished column cb before iteration cb - 6
    // starts: the stager of iteration cb - 7 waited for it)
    auto fetch_column = [&](int cb) {
#pragma unroll
        for (int q = 0; q < CS_PER_LANE; ++q) {
            const int rb = lane + 64 * q;
            column[q] = (cb < nchunks && rb <= cb) ? cs[(size_t)cb * ncb + rb] : 0ull;
        }
    };
    auto store_column = [&](int cb) {
#pragma unroll
        for (int q = 0; q < CS_PER_LANE; ++q) sh.colsum[cb & 3][lane + 64 * q] = column[q];
    };
    auto flush = [&]() {
        if (out_pos >= 0) {
            if (keep) keep[(size_t)img * max_keep + out_pos] = out_idx;
            if (rois_padded) {
                float *o = rois_padded + ((size_t)img * max_keep + out_pos) * 5;
                o[0] = (float)img; o[1] = out_box[0]; o[2] = out_box[1]; o[3] = out_box[2]; o[4] = out_box[3];
            }
        }
        out_pos = -1;
    };
    // stager `group` acts at even/odd c: writes chunk c+1, loads chunk c+3.  Prologue: chunk 0
    // straight into LDS (group 1 would have written it at c = -1), then chunk 1 (for c = 0) into
    // group 0's registers and chunk 2 (for c = 1) into group 1's.
    if (stager && group == 1) {
        load_rows(0);
#pragma unroll
        for (int j = 0; j <= SWEEP_AHEAD; ++j) sh.rowbuf[0][j][lane] = rows[j];
        load_rows(2);
    } else if (stager) {
        load_rows(1);
    }
    __syncthreads();
    if (scribe && group == 1) fetch_column(5);     // the scribe of the odd iterations: stored at c = 1, read at c = 2
    else if (scribe) {
#pragma unroll
        for (int q = 0; q < CS_PER_LANE; ++q) column[q] = 0ull;
    }

    // Helpers: word w of the boxes kept five or more chunks before it.  ALL ten helper waves act in EVERY
    // iteration, each lane on HELPER_SLOTS positions of the kept list (position h + j * 640): a turn is 4 slots
    // instead of the 7 that two alternating groups of five waves needed, and the helpers' turn was the longest
    // of an iteration (0.85 us against the resolver's 0.42).  A wave then has TWO batches of gather loads in
    // flight, and the one issued an iteration ago must not be waited for when the one issued two iterations
    // ago is consumed.  The compiler's wait insertion cannot express that across the loop (it waits for
    // everything), and it may COPY a loop-carried register whose load is still in flight if the load is hidden
    // in inline asm with an ordinary operand.  So the batches live in sixteen FIXED registers, v80-v95, that
    // only the asm statements below name (both kernels carry amdgpu_num_vgpr(80), which keeps the register
    // allocator below them -- a request, which tests/test_isa_reserved_registers.py checks in the built library;
    // 96 VGPRs in all -- see the fused kernel for why not the top of a 128 budget; round 4 tried them at v48-v63 and
    // v64-v79 for a smaller kernel: the allocator ignored amdgpu_num_vgpr(48) / (64) and used those registers, the
    // test caught it): always HELPER_SLOTS loads per turn (a slot with nothing to fetch loads a zero word),
    // consumed behind s_waitcnt vmcnt(HELPER_SLOTS).  Helper waves issue no other vector memory operation.
    constexpr int HELPER_SLOTS = 4;
    constexpr int HELPER_LANES = (SWEEP_BLOCK / 64 - SWEEP_FIRST_HELPER) * 64;
    static_assert(HELPER_SLOTS * HELPER_LANES >= SWEEP_LH * SWEEP_GROUP, "every list position has a slot");
    // take: the OR of the batch's four words, straight out of the fixed registers (a slot without a word to fetch
    // loaded a zero, see helper_turn -- no masks to keep, no moves: 4 instructions)
#define WSSDL_TAKE(R0, R1, R2, R3, R4, R5, R6, R7)                                                                      \
    [&]() {                                                                                                             \
        unsigned lo, hi;                                                                                                \
        asm volatile("s_waitcnt vmcnt(4)\n\tv_or3_b32 %0, " R0 ", " R2 ", " R4 "\n\tv_or3_b32 %1, " R1 ", " R3 ", " R5       \
                     "\n\tv_or_b32 %0, %0, " R6 "\n\tv_or_b32 %1, %1, " R7                                              \
                     : "=&v"(lo), "=&v"(hi)                                                                             \
                     :                                                                                                  \
                     : "memory");                                                                                       \
        return ((unsigned long long)hi << 32) | lo;                                                                     \
    }
#define WSSDL_ISSUE(P0, P1, P2, P3, C0, C1, C2, C3, C4, C5, C6, C7)                                                     \
    [&](int j, const unsigned long long *src) {                                                                         \
        if (j == 0) asm volatile("global_load_dwordx2 " P0 ", %0, off" : : "v"(src) : "memory", C0, C1);                \
        else if (j == 1) asm volatile("global_load_dwordx2 " P1 ", %0, off" : : "v"(src) : "memory", C2, C3);           \
        else if (j == 2) asm volatile("global_load_dwordx2 " P2 ", %0, off" : : "v"(src) : "memory", C4, C5);           \
        else asm volatile("global_load_dwordx2 " P3 ", %0, off" : : "v"(src) : "memory", C6, C7);                       \
    }
    auto take_a = WSSDL_TAKE("v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    auto take_b = WSSDL_TAKE("v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    auto issue_a = WSSDL_ISSUE("v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]", "v80", "v81", "v82", "v83", "v84",
                               "v85", "v86", "v87");
    auto issue_b = WSSDL_ISSUE("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v88", "v89", "v90", "v91", "v92",
                               "v93", "v94", "v95");
#undef WSSDL_TAKE
#undef WSSDL_ISSUE
    // (the first two turns take batches that were never issued: the registers start as zero words)
    if (helper)
        asm volatile("v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\t"
                     "v_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\tv_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\t"
                     "v_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\t"
                     "v_mov_b32 v95, 0" ::: "memory", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
                     "v90", "v91", "v92", "v93", "v94", "v95");
    // Round 6: a helper wave owns 256 CONSECUTIVE list positions (slot j of lane l: hw * 256 + j * 64 + l; it was
    // hw * 64 + l + j * 640, which puts a live slot into every wave as soon as the list holds 577 boxes), and a wave whose
    // positions all lie beyond the list skips its turn altogether -- no loads, nothing to take, its batch registers stay
    // the zero words they start as.  The list only grows, so a wave that has started never stops.  The sixteen waves
    // of the sweep share four SIMDs: an idle helper's ~90 instructions per chunk are issue slots for the others.
    const int wave_first = hw * (HELPER_SLOTS * 64);
    bool helper_live = false;
    int hposl[HELPER_SLOTS];                 // this lane's list positions; hpos = the same, held inside the LDS list
    int hpos[HELPER_SLOTS];
#pragma unroll
    for (int j = 0; j < HELPER_SLOTS; ++j) { hposl[j] = wave_first + j * 64 + lane;  hpos[j] = min(hposl[j], max_keep); }
    // The helpers' turn is the longest of most iterations (tools/nms_sweep_profile.py: the last helper waves arrive
    // last at the barrier in 50-90 % of them) and what it costs is issue slots: one CU issues for all sixteen
    // waves, and a turn took the same ~1100-1400 cycles with nothing to gather as with a full list.  So it is
    // written for instruction count (round 4, ~170 -> ~90): a slot with nothing to fetch loads a word that is zero
    // (no per-slot masks: the batch is ORed straight out of its registers), the wave's contribution goes to the
    // ring with one LDS atomic per lane that HAS one (a few lanes per wave; the 64-bit DPP reduction was 30
    // instructions for every wave, every turn), the list entries and the summary words are read in two rounds of
    // LDS loads instead of a dependent pair per slot, the summary as 32-bit halves.  Early-stop walks 0.165 ->
    // 0.157 ms (fused) and 0.243 -> 0.233 (two launches) for the 8-image layer; the bench's full walks, which wait
    // for the mask's last segments, did not move (profiles/r04_sweep_roles.log).
#ifdef WSSDL_SWEEP_PROFILE
    unsigned long long prof_mid = 0ull;      // helpers: cycles in take() (the wait for the batch); stagers: in the wait for their rows
#endif
#ifdef WSSDL_SWEEP_PROFILE_HELPER
    // (a second macro: with these stamps the register allocator of ROCm 7.2 goes above v79 -- the build is refused by the
    // ISA check, tools/nms_sweep_profile.py --build --define WSSDL_SWEEP_PROFILE_HELPER -- so they stay out of the profile build)
    unsigned prof_helper[4] = {0u, 0u, 0u, 0u};      // a helper turn by section (see helper_turn)
#endif
    auto helper_turn = [&](auto &take, auto &issue, int c) {
        const int lim = (c >= 1 && c + 3 < nchunks) ? min(sh.pub[(c - 1) & 1].base, max_keep) : 0;
        helper_live = helper_live || wave_first < lim;
        if (!helper_live) return;
        // consume word c+1: this batch was issued at iteration c-2; the one issued at c-1 may stay in flight
        // (the instruction itself: atomicOr() on one address becomes a loop over the active lanes)
#ifdef WSSDL_SWEEP_PROFILE
        const unsigned long long prof_m0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        const unsigned long long acc = take();
#ifdef WSSDL_SWEEP_PROFILE
        prof_mid += __builtin_amdgcn_s_memtime() - prof_m0;       // (incl. ~2 x 60 cycles of the stamps themselves)
#endif
        if (acc != 0ull)
            asm volatile("ds_or_b64 %0, %1" : : "v"((unsigned)(size_t)&sh.ring[(c + 1) & 7]), "v"(acc) : "memory");
#ifdef WSSDL_SWEEP_PROFILE_HELPER
        const unsigned long long prof_h1 = __builtin_amdgcn_s_memtime();
        prof_helper[0] += (unsigned)(prof_h1 - prof_m0);          // take + OR into the ring
#endif
        // issue word c+3 of every box in the kept list (chunks <= c-2) for which the column's summary has a
        // bit (the others are zero, and were not even stored); 32-bit word offsets (n_max * pitch < 2^31
        // checked by the launcher).  (A lane per CHUNK, walking the bits of kept & summary, needs no list -- but
        // a chunk can hold more such boxes than a lane has registers, and the overflow loads sat inside the
        // iteration: 0.49 against 0.37 ms in the step.)
        const unsigned *colsum_now = reinterpret_cast<const unsigned *>(sh.colsum[(c + 3) & 3]);
        unsigned row[HELPER_SLOTS], cw[HELPER_SLOTS];
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) row[j] = (unsigned)kept_rows[hpos[j]];
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) row[j] = (hposl[j] < lim) ? row[j] : 0u;
#ifdef WSSDL_SWEEP_PROFILE_HELPER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long prof_h2 = __builtin_amdgcn_s_memtime();
        prof_helper[1] += (unsigned)(prof_h2 - prof_h1);          // the list entries (LDS)
#endif
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) cw[j] = colsum_now[row[j] >> 5];
#ifdef WSSDL_SWEEP_PROFILE_HELPER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long prof_h3 = __builtin_amdgcn_s_memtime();
        prof_helper[2] += (unsigned)(prof_h3 - prof_h2);          // the summary words (LDS, dependent on the entries)
#endif
        const unsigned long long *src[HELPER_SLOTS];
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) {
            const unsigned hit = (unsigned)(hposl[j] < lim) & (cw[j] >> (row[j] & 31u)) & 1u;
            src[j] = hit ? m + (__umul24(row[j], (unsigned)ncb) + (unsigned)(c + 3)) : zero_word;
        }
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) issue(j, src[j]);
#ifdef WSSDL_SWEEP_PROFILE_HELPER
        prof_helper[3] += (unsigned)(__builtin_amdgcn_s_memtime() - prof_h3);  // addresses + the four loads issued
#endif
    };
    int count = 0, last = -1;
#ifdef WSSDL_SWEEP_PROFILE
    __shared__ unsigned long long prof_lds[SWEEP_BLOCK / 64][24];
    if (tid < (SWEEP_BLOCK / 64) * 24) prof_lds[tid / 24][tid % 24] = 0ull;
    __syncthreads();
    unsigned long long prof_work = 0ull, prof_wait = 0ull, prof_last = 0ull, prof_max = 0ull, prof_maxc = 0ull, prof_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int c = 0; c < nchunks; ++c) {
        if (wave == 0) {
            unsigned long long rem = 0ull;
            if (lane == 0) { rem = sh.ring[c & 7]; sh.ring[c & 7] = 0ull; }   // slot reused by word c+8
            unsigned long long w[SWEEP_AHEAD + 1];
#pragma unroll
            for (int j = 0; j <= SWEEP_AHEAD; ++j) w[j] = sh.rowbuf[c & 1][j][lane];
            rem = readlane_u64(rem, 0);
            const int nv = n - c * 64;
            const unsigned long long valid = (nv >= 64) ? ~0ull : ((1ull << nv) - 1ull);
            const unsigned long long cand = ~rem & valid;
            const unsigned long long t_mine = w[0];
            unsigned long long kept = cand;
            for (;;) {
                const unsigned long long nk = cand & ~__ballot((t_mine & kept) != 0ull);
                if (nk == kept) break;
                kept = nk;
            }
            if ((kept >> lane) & 1ull) {
#pragma unroll
                for (int j = 1; j <= SWEEP_AHEAD; ++j)
                    if (w[j] != 0ull) atomicOr(&sh.ring[(c + j) & 7], w[j]);
            }
            if (lane == 0) {
                SweepShared::Publish pr;
                pr.kept = kept;  pr.base = count;  pr.count = count + __popcll(kept);
                sh.pub[c & 1] = pr;
            }
            count += __popcll(kept);                   // the resolver keeps its own running count
        } else if (role == 4) {
            // the spare wave (fused launch): chunk c - 1 was resolved an iteration ago -- tell the mask role which of
            // its boxes survived, so that the far column segments of that row block are computed for those rows only
            // (nms_mask_block_sparse).  Two tagged 8-byte halves, each atomic on its own: no ordering, no wait.
            if (A.keptpub && c >= 1 && lane == 0) {
                const unsigned long long k = sh.pub[(c - 1) & 1].kept, tag = (unsigned long long)c << 32;
                unsigned long long *kp = A.keptpub + ((size_t)img * ncb + (c - 1)) * 2;
                __hip_atomic_store(kp, (k & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(kp + 1, (k >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (helper) {
#ifndef WSSDL_SWEEP_NULL_HELPERS      // (timing experiment of the profile build: what the iteration costs without them; wrong keeps)
            if (c & 1) helper_turn(take_b, issue_b, c);
            else helper_turn(take_a, issue_a, c);
#endif
        } else if ((c & 1) == group) {
            if (scribe) {
#ifdef WSSDL_SWEEP_PROFILE
                const unsigned long long prof_m0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                prof_mid += __builtin_amdgcn_s_memtime() - prof_m0;
#endif
                // the column first: its s_waitcnt vmcnt(0) (for the loads of two iterations ago) otherwise also waits
                // for the global stores flush() has just issued -- vmcnt counts stores -- and the scribe of the
                // iteration was the last wave at the barrier in 90 % of the chunks of the bench's steps, ~19-26 %
                // after the swap (tools/nms_sweep_profile.py, profiles/r04_sweep_roles.log)
                store_column(c + 4);                   // fetched two iterations ago
                flush();                               // outputs fetched two iterations ago
                if (c > 0) expand(c - 1);
                fetch_column(c + 6);
            } else if (stager) {
#ifdef WSSDL_SWEEP_PROFILE
                const unsigned long long prof_m0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                prof_mid += __builtin_amdgcn_s_memtime() - prof_m0;
#endif
#pragma unroll
                for (int j = 0; j <= SWEEP_AHEAD; ++j) sh.rowbuf[(c + 1) & 1][j][lane] = rows[j];
                load_rows(c + 3);
            }
        }
#ifdef WSSDL_SWEEP_PROFILE
        const unsigned long long prof_t1 = __builtin_amdgcn_s_memtime();
        lds_only_barrier();
        const unsigned long long prof_t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        prof_last += (prof_t2 - prof_t1 < 200ull) ? 1ull : 0ull;             // this wave was (nearly) the last to arrive
        if (lane == 0 && c < 24 * 16) prof_lds[wave][c >> 4] += prof_t1 - prof_prev;      // this wave's work, by 16-chunk bucket
        if (prof_t1 - prof_prev > prof_max) { prof_max = prof_t1 - prof_prev;  prof_maxc = (unsigned long long)c; }
        prof_work += prof_t1 - prof_prev;  prof_wait += prof_t2 - prof_t1;  prof_prev = prof_t2;
#else
        lds_only_barrier();
#endif
        if (wave != 0) count = sh.pub[c & 1].count;
        last = c;
        if (count >= max_keep) break;
    }
    if (helper) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (batches still in flight)
#ifdef WSSDL_SWEEP_PROFILE
    if (lane == 0 && img < 64) {
        unsigned long long *o = wssdl_sweep_prof[img][wave];
        o[0] = prof_work;  o[1] = prof_wait;  o[2] = (unsigned long long)(last + 1);
        o[3] = __builtin_amdgcn_s_memrealtime() - prof_rt0;  o[4] = prof_mid;  o[5] = prof_last;  o[6] = prof_max;  o[7] = prof_maxc;
        for (int k = 0; k < 20; ++k) o[8 + k] = prof_lds[wave][k];
#ifdef WSSDL_SWEEP_PROFILE_HELPER
        for (int k = 0; k < 4; ++k) o[28 + k] = prof_helper[k];
#endif
    }
#endif
    if (scribe && last >= 0) {
        // chunk `last` was resolved but not expanded yet: it belongs to the group of last + 1
        flush();
        if (((last + 1) & 1) == group) { expand(last); flush(); }
    }
    if (tid == 0) {
        num_keep[img] = sh.timed_out ? WSSDL_NMS_TIMED_OUT : min(count, max_keep);
        if (done_out) done_out[img] = (count >= max_keep || n_dev[img] <= n_limit) ? 1 : 0;
    }
}

// ------------------------------------------- nms sweep, roles without a barrier ---
// Round 6.  The role-pipelined sweep above joins its sixteen waves with ONE workgroup barrier per chunk, and an
// iteration (1.05 us, 188 of them in a full walk) turned out to wait for that barrier and the LDS hand-overs around
// it, not for any role's instructions (round 5: fewer helper instructions, same time).  Here every role runs ITS OWN
// loop over the chunks and waits only for what it reads, through sequence numbers in LDS that their single writers
// only ever count up:
//   resolved      chunks the resolver has resolved           (writer: resolver)
//   staged[s]     chunk + 1 of the last rows stager s wrote   (three stagers, chunk k belongs to stager k % 3)
//   expanded[g]   chunk + 1 of the last chunk scribe g expanded into the kept list (scribe g: turns t with t & 1 == g,
//                 turn t expands chunk t - 1)
//   colstored[g]  column + 1 of the last column summary scribe g stored
//   hdone[w & 7]  helper waves that have ORed their part of `removed` word w into the ring
// The schedule is the barrier version's (who computes what, how far ahead: resolver = the four chunks before a word
// from the staged rows, helpers = everything older from the kept list, two batches of gathers in flight); only the
// joins differ.  resolver at chunk c: staged[c % 3] > c and, from word 5 on, hdone[c & 7] = all helpers.  Stager of
// chunk k: the slot k & 3 is free once chunk k - 4 is resolved.  Scribe turn t: chunk t - 1 resolved.  Helpers at
// index c (take word c + 1, issue word c + 3): chunks <= c - 2 expanded, column c + 3 stored.  No cycle: everything
// waits for the resolver's past or for a role that does.  Every wait is bounded (spins, then `abort`: all roles leave,
// the image reports WSSDL_NMS_TIMED_OUT) and ends when the resolver has stopped (`stop` = chunks resolved in all).
// Results: the same kept sets by construction (same words, ORed in a different order); tests/test_gpu_parity.py
// (reference keep lists), tools/nms_fuzz.py, tools/nms_fused_stress.py.
constexpr int ASYNC_STAGERS = 3;
constexpr int ASYNC_ROW_SLOTS = 4;
constexpr int ASYNC_SPIN_LIMIT = 1 << 22;        // polls of ~100-200 cycles each: > 0.1 s

struct SweepAsyncShared {
    unsigned long long ring[8];
    unsigned long long rowbuf[ASYNC_ROW_SLOTS][SWEEP_AHEAD + 1][64];   // [chunk & 3][0 = T, j = word c+j][row]
    unsigned long long colsum[4][SWEEP_MAX_CHUNKS];                   // [column block & 3][row block]
    struct __attribute__((aligned(16))) Publish {
        unsigned long long kept;
        int base, count;
    } pub[8];                                                          // by chunk & 7
    int hdone[8];
    int resolved, stop, abort, timed_out;
    int staged[ASYNC_STAGERS + 1];
    int expanded[2], colstored[2];
};

// (every lane reads the same word; readfirstlane tells the compiler so -- a poll loop on a per-lane value becomes vector
// control flow with saved exec masks, and the kernel spilled 217 scalar registers to vector lanes)
__device__ __forceinline__ int lds_peek(const int *p) {
    return __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(p));
}
__device__ __forceinline__ void lds_post(int *p, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (LDS executes a wave's operations in order; this is for the compiler)
    *reinterpret_cast<volatile int *>(p) = v;
}
// until *p >= v.  false: it never will (the resolver has stopped, or somebody gave up)
template <int SLEEP>
__device__ __forceinline__ bool lds_wait_ge_raw(const int *p, int v, SweepAsyncShared &sh) {
    for (int spins = 0;; ++spins) {
        if (lds_peek(p) >= v) return true;
        if (lds_peek(&sh.abort)) return false;
        if (lds_peek(&sh.stop) != 0x7fffffff) return lds_peek(p) >= v;
        if (spins > ASYNC_SPIN_LIMIT) { *reinterpret_cast<volatile int *>(&sh.abort) = 1;  return false; }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}
#ifdef WSSDL_SWEEP_PROFILE
// profile build (tools/nms_sweep_profile.py --async): cycles a wave spends in each of its waits, by site
#define WSSDL_ASYNC_WAIT(SLEEP, P, V, SITE)                                                  \
    [&]() {                                                                                  \
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();                          \
        const bool r = lds_wait_ge_raw<SLEEP>(P, V, sh);                                     \
        prof_site[SITE] += __builtin_amdgcn_s_memtime() - w0;                                \
        return r;                                                                            \
    }()
#else
#define WSSDL_ASYNC_WAIT(SLEEP, P, V, SITE) lds_wait_ge_raw<SLEEP>(P, V, sh)
#endif

__device__ __forceinline__ void nms_sweep_async_block(const SweepArgs &A, int img, int *kept_rows /* LDS [max_keep + 64] */,
                                                      SweepAsyncShared &sh) {
    const unsigned long long *__restrict__ mask = A.mask, *__restrict__ diag_t = A.diag_t, *__restrict__ summ = A.summ;
    const int n_max = A.n_max, ncb = A.ncb, max_keep = A.max_keep, order_stride_img = A.order_stride_img,
              box_stride_img = A.box_stride_img, n_limit = A.n_limit;
    const int *__restrict__ n_dev = A.n_dev, *__restrict__ order = A.order, *__restrict__ done_in = A.done_in;
    int *__restrict__ keep = A.keep, *__restrict__ num_keep = A.num_keep, *__restrict__ done_out = A.done_out;
    const float *__restrict__ boxes = A.boxes;
    float *__restrict__ rois_padded = A.rois_padded;
    if (done_in && done_in[img]) return;
    const int n = min(min(n_dev[img], n_max), n_limit);
    const int nchunks = (n + 63) / 64;
    const unsigned long long *m = mask + (size_t)img * n_max * ncb;
    const unsigned long long *dt = diag_t + (size_t)img * n_max;
    const unsigned long long *cs = summ + (size_t)img * ncb * ncb;        // [column block][row block]
    const unsigned long long *zero_word = wssdl_sweep_zero_word;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 8) { sh.ring[tid] = 0ull;  sh.hdone[tid] = 0;  sh.pub[tid].kept = 0ull;  sh.pub[tid].base = 0;  sh.pub[tid].count = 0; }
    if (tid == 0) {
        sh.resolved = 0;  sh.stop = 0x7fffffff;  sh.abort = 0;  sh.timed_out = 0;
        for (int i = 0; i <= ASYNC_STAGERS; ++i) sh.staged[i] = 0;
        sh.expanded[0] = sh.expanded[1] = 0;  sh.colstored[0] = sh.colstored[1] = 0;
    }
    __syncthreads();
    constexpr int NHELPERS = SWEEP_BLOCK / 64 - SWEEP_FIRST_HELPER;
    int count = 0;
#ifdef WSSDL_SWEEP_PROFILE
    unsigned long long prof_site[4] = {0ull, 0ull, 0ull, 0ull};      // [0], [1]: the role's two waits; [2]: helpers' wait for a batch, stagers' for segments
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(), prof_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    if (wave == 0) {
        // ---------------------------------------------------------------- resolver ---
        for (int c = 0; c < nchunks; ++c) {
            bool ok = WSSDL_ASYNC_WAIT(0, &sh.staged[c % ASYNC_STAGERS], c + 1, 0);
            if (ok && c >= SWEEP_AHEAD + 1) ok = WSSDL_ASYNC_WAIT(0, &sh.hdone[c & 7], NHELPERS, 1);
            if (!ok) break;                                         // (only after an abort: nobody else sets `stop`)
            unsigned long long rem = 0ull;
            if (lane == 0) { rem = sh.ring[c & 7];  sh.ring[c & 7] = 0ull;  sh.hdone[c & 7] = 0; }   // slots reused by word c + 8
            unsigned long long w[SWEEP_AHEAD + 1];
#pragma unroll
            for (int j = 0; j <= SWEEP_AHEAD; ++j) w[j] = sh.rowbuf[c & (ASYNC_ROW_SLOTS - 1)][j][lane];
            rem = readlane_u64(rem, 0);
            const int nv = n - c * 64;
            const unsigned long long valid = (nv >= 64) ? ~0ull : ((1ull << nv) - 1ull);
            const unsigned long long cand = ~rem & valid;
            const unsigned long long t_mine = w[0];
            unsigned long long kept = cand;
            for (;;) {
                const unsigned long long nk = cand & ~__ballot((t_mine & kept) != 0ull);
                if (nk == kept) break;
                kept = nk;
            }
            if ((kept >> lane) & 1ull) {
#pragma unroll
                for (int j = 1; j <= SWEEP_AHEAD; ++j)
                    if (w[j] != 0ull) atomicOr(&sh.ring[(c + j) & 7], w[j]);
            }
            if (lane == 0) {
                SweepAsyncShared::Publish pr;
                pr.kept = kept;  pr.base = count;  pr.count = count + __popcll(kept);
                sh.pub[c & 7] = pr;
            }
            count += __popcll(kept);
            lds_post(&sh.resolved, c + 1);
            if (count >= max_keep) break;
        }
        lds_post(&sh.stop, lds_peek(&sh.resolved));
    } else if (wave == 1 || wave == 2) {
        // ----------------------------------------------------------------- scribes ---
        // turn t (t & 1 == g): column t + 4 to LDS (fetched at turn t - 2), the outputs of chunk t - 3 to memory
        // (fetched at turn t - 2), chunk t - 1 into the kept list (+ its kept bitmask to the mask role), column t + 6
        // requested
        const int g = wave & 1;
        constexpr int CS_PER_LANE = SWEEP_MAX_CHUNKS / 64;
        unsigned long long column[CS_PER_LANE];
        int out_pos = -1, out_idx = 0;
        float out_box[4] = {0.f, 0.f, 0.f, 0.f};
        auto fetch_column = [&](int cb) {
#pragma unroll
            for (int q = 0; q < CS_PER_LANE; ++q) {
                const int rb = lane + 64 * q;
                column[q] = (cb < nchunks && rb <= cb) ? cs[(size_t)cb * ncb + rb] : 0ull;
            }
        };
        auto flush = [&]() {
            if (out_pos >= 0) {
                if (keep) keep[(size_t)img * max_keep + out_pos] = out_idx;
                if (rois_padded) {
                    float *o = rois_padded + ((size_t)img * max_keep + out_pos) * 5;
                    o[0] = (float)img; o[1] = out_box[0]; o[2] = out_box[1]; o[3] = out_box[2]; o[4] = out_box[3];
                }
            }
            out_pos = -1;
        };
#pragma unroll
        for (int q = 0; q < CS_PER_LANE; ++q) column[q] = 0ull;
        // (fused launch: column 5 belongs to segment 0, which the stager of chunk 0 waits for before anything is staged;
        // scribe 1 fetches it behind its first wait for the resolver, i.e. behind that)
        for (int t = g;; t += 2) {
            // chunk t - 1 resolved: it is expanded this turn, and column t (whose slot column t + 4 takes) has been read
            if (t > 0 && !WSSDL_ASYNC_WAIT(1, &sh.resolved, t, 0)) break;
            if (t == 1) fetch_column(5);               // the first column the helpers read (index 2: word 5); waited for here, once
            if (t >= 1) {
#pragma unroll
                for (int q = 0; q < CS_PER_LANE; ++q) sh.colsum[(t + 4) & 3][lane + 64 * q] = column[q];
                lds_post(&sh.colstored[g], t + 5);
            }
            flush();
            if (t > 0) {
                const int chunk = t - 1;
                const unsigned long long kept = sh.pub[chunk & 7].kept;
                const int base = sh.pub[chunk & 7].base;
                if (A.keptpub && lane == 0) {
                    // the mask role computes the far column segments of a resolved row block for its kept rows only
                    // (nms_mask_block_sparse).  Two tagged 8-byte halves, each atomic on its own: no ordering, no wait.
                    const unsigned long long tag = (unsigned long long)(chunk + 1) << 32;
                    unsigned long long *kp = A.keptpub + ((size_t)img * ncb + chunk) * 2;
                    __hip_atomic_store(kp, (kept & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(kp + 1, (kept >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((kept >> lane) & 1ull) {
                    const int row = chunk * 64 + lane;
                    const int pos = base + __popcll(kept & ((1ull << lane) - 1ull));
                    if (pos < max_keep) {
                        kept_rows[pos] = row;
                        out_pos = pos;
                        out_idx = row;
                        if (keep && order) out_idx = order[(size_t)img * order_stride_img + row];
                        if (rois_padded) {
                            const float *bx = boxes + (size_t)img * box_stride_img + (size_t)row * 4;
                            out_box[0] = bx[0]; out_box[1] = bx[1]; out_box[2] = bx[2]; out_box[3] = bx[3];
                        }
                    }
                }
                lds_post(&sh.expanded[g], t);
            }
            if (t >= nchunks) break;                          // the last chunk was this turn's
            // (fused launch: column t + 6 is complete once the stager of chunk t + 2 has waited for its segments)
            if (A.segdone && t + 2 < nchunks && !WSSDL_ASYNC_WAIT(1, &sh.staged[(t + 2) % ASYNC_STAGERS], t + 3, 1)) break;
            fetch_column(t + 6);
        }
        flush();
    } else if (wave >= 3 && wave < 3 + ASYNC_STAGERS) {
        // ----------------------------------------------------------------- stagers ---
        // stager s: chunks k = s, s + 3, ...; T and the words k+1..k+4 of chunk k's rows, loaded three chunks before they
        // are written to the slot k & 3 (free once chunk k - 4 is resolved)
        const int sidx = wave - 3;
        unsigned long long rows[SWEEP_AHEAD + 1];
        int seg_ready = 0;
        auto load_rows = [&](int chunk) {
            if (A.segdone && chunk < nchunks) {
                const int want = min(chunk + SWEEP_AHEAD, nchunks - 1) / MASK_SEG + 1;
                while (seg_ready < want) {
                    sweep_wait_segment(A.segdone, MASK_WAVES * min((seg_ready + 1) * MASK_SEG, A.nrb), img * ncb + seg_ready,
                                       &sh.timed_out, A.wait_ticks);
                    ++seg_ready;
                }
            }
            const int row = chunk * 64 + lane;
#pragma unroll
            for (int j = 0; j <= SWEEP_AHEAD; ++j) rows[j] = 0ull;
            if (row < n) {
                rows[0] = dt[row];
#pragma unroll
                for (int j = 1; j <= SWEEP_AHEAD; ++j)
                    if (chunk + j < nchunks) rows[j] = m[(size_t)row * ncb + chunk + j];
            }
        };
        if (sidx < nchunks) load_rows(sidx);
        for (int k = sidx; k < nchunks; k += ASYNC_STAGERS) {
            if (k >= ASYNC_ROW_SLOTS && !WSSDL_ASYNC_WAIT(1, &sh.resolved, k - ASYNC_ROW_SLOTS + 1, 0)) break;
#pragma unroll
            for (int j = 0; j <= SWEEP_AHEAD; ++j) sh.rowbuf[k & (ASYNC_ROW_SLOTS - 1)][j][lane] = rows[j];
            lds_post(&sh.staged[sidx], k + 1);
            if (k + ASYNC_STAGERS < nchunks) load_rows(k + ASYNC_STAGERS);
        }
    } else {
        // ----------------------------------------------------------------- helpers ---
        // (the batches of gathers live in sixteen fixed registers, v80-v95, that only these asm statements name: see the
        // barrier version above for why; helper waves issue no other vector memory operation)
        constexpr int HELPER_SLOTS = 4;
        constexpr int HELPER_LANES = NHELPERS * 64;
        static_assert(HELPER_SLOTS * HELPER_LANES >= SWEEP_LH * SWEEP_GROUP, "every list position has a slot");
#define WSSDL_TAKE(R0, R1, R2, R3, R4, R5, R6, R7)                                                                      \
    [&]() {                                                                                                             \
        unsigned lo, hi;                                                                                                \
        asm volatile("s_waitcnt vmcnt(4)\n\tv_or3_b32 %0, " R0 ", " R2 ", " R4 "\n\tv_or3_b32 %1, " R1 ", " R3 ", " R5       \
                     "\n\tv_or_b32 %0, %0, " R6 "\n\tv_or_b32 %1, %1, " R7                                              \
                     : "=&v"(lo), "=&v"(hi)                                                                             \
                     :                                                                                                  \
                     : "memory");                                                                                       \
        return ((unsigned long long)hi << 32) | lo;                                                                     \
    }
#define WSSDL_ISSUE(P0, P1, P2, P3, C0, C1, C2, C3, C4, C5, C6, C7)                                                     \
    [&](int j, const unsigned long long *src) {                                                                         \
        if (j == 0) asm volatile("global_load_dwordx2 " P0 ", %0, off" : : "v"(src) : "memory", C0, C1);                \
        else if (j == 1) asm volatile("global_load_dwordx2 " P1 ", %0, off" : : "v"(src) : "memory", C2, C3);           \
        else if (j == 2) asm volatile("global_load_dwordx2 " P2 ", %0, off" : : "v"(src) : "memory", C4, C5);           \
        else asm volatile("global_load_dwordx2 " P3 ", %0, off" : : "v"(src) : "memory", C6, C7);                       \
    }
        auto take_a = WSSDL_TAKE("v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
        auto take_b = WSSDL_TAKE("v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
        auto issue_a = WSSDL_ISSUE("v[80:81]", "v[82:83]", "v[84:85]", "v[86:87]", "v80", "v81", "v82", "v83", "v84",
                                   "v85", "v86", "v87");
        auto issue_b = WSSDL_ISSUE("v[88:89]", "v[90:91]", "v[92:93]", "v[94:95]", "v88", "v89", "v90", "v91", "v92",
                                   "v93", "v94", "v95");
#undef WSSDL_TAKE
#undef WSSDL_ISSUE
        asm volatile("v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\t"
                     "v_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\tv_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\t"
                     "v_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\t"
                     "v_mov_b32 v95, 0" ::: "memory", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
                     "v90", "v91", "v92", "v93", "v94", "v95");
        const int hw = wave - SWEEP_FIRST_HELPER;
        int hposl[HELPER_SLOTS], hpos[HELPER_SLOTS];           // 256 consecutive list positions per wave
#pragma unroll
        for (int j = 0; j < HELPER_SLOTS; ++j) { hposl[j] = hw * (HELPER_SLOTS * 64) + j * 64 + lane;  hpos[j] = min(hposl[j], max_keep); }
        auto helper_turn = [&](auto &take, auto &issue, int c) -> bool {
            // word c + 1: this batch was issued at index c - 2; the one issued at c - 1 may stay in flight
#ifdef WSSDL_SWEEP_PROFILE
            const unsigned long long prof_m0 = __builtin_amdgcn_s_memtime();
#endif
            const unsigned long long acc = take();
#ifdef WSSDL_SWEEP_PROFILE
            prof_site[2] += __builtin_amdgcn_s_memtime() - prof_m0;
#endif
            if (acc != 0ull)
                asm volatile("ds_or_b64 %0, %1" : : "v"((unsigned)(size_t)&sh.ring[(c + 1) & 7]), "v"(acc) : "memory");
            if (c + 1 >= SWEEP_AHEAD + 1 && c + 1 < nchunks) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&sh.hdone[(c + 1) & 7], 1);
            }
            // word c + 3 of every box kept in chunks <= c - 2 for which the column's summary has a bit
            int lim = 0;
            if (c >= 2 && c + 3 < nchunks) {
                if (!WSSDL_ASYNC_WAIT(1, &sh.expanded[(c - 1) & 1], c - 1, 0)) return false;          // chunk c - 2
                if (c >= 3 && !WSSDL_ASYNC_WAIT(1, &sh.expanded[c & 1], c - 2, 0)) return false;      // chunk c - 3 (the other scribe)
                if (!WSSDL_ASYNC_WAIT(1, &sh.colstored[(c - 1) & 1], c + 4, 1)) return false;         // column c + 3
                lim = min(lds_peek(&sh.pub[(c - 2) & 7].count), max_keep);
            }
            const unsigned *colsum_now = reinterpret_cast<const unsigned *>(sh.colsum[(c + 3) & 3]);
            unsigned row[HELPER_SLOTS], cw[HELPER_SLOTS];
#pragma unroll
            for (int j = 0; j < HELPER_SLOTS; ++j) row[j] = (unsigned)kept_rows[hpos[j]];
#pragma unroll
            for (int j = 0; j < HELPER_SLOTS; ++j) row[j] = (hposl[j] < lim) ? row[j] : 0u;
#pragma unroll
            for (int j = 0; j < HELPER_SLOTS; ++j) cw[j] = colsum_now[row[j] >> 5];
            const unsigned long long *src[HELPER_SLOTS];
#pragma unroll
            for (int j = 0; j < HELPER_SLOTS; ++j) {
                const unsigned hit = (unsigned)(hposl[j] < lim) & (cw[j] >> (row[j] & 31u)) & 1u;
                src[j] = hit ? m + (__umul24(row[j], (unsigned)ncb) + (unsigned)(c + 3)) : zero_word;
            }
#pragma unroll
            for (int j = 0; j < HELPER_SLOTS; ++j) issue(j, src[j]);
            return true;
        };
        for (int c = 0; c + 1 < nchunks; ++c) {
            const bool ok = (c & 1) ? helper_turn(take_b, issue_b, c) : helper_turn(take_a, issue_a, c);
            if (!ok) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (batches still in flight)
    }
#ifdef WSSDL_SWEEP_PROFILE
    if (lane == 0 && img < 64) {
        unsigned long long *o = wssdl_sweep_prof[img][wave];
        o[0] = __builtin_amdgcn_s_memtime() - prof_t0;                   // the role's loop, start to end
        o[1] = prof_site[0];  o[4] = prof_site[1];  o[5] = prof_site[2];
        o[2] = (unsigned long long)lds_peek(&sh.resolved);
        o[3] = __builtin_amdgcn_s_memrealtime() - prof_rt0;
        o[6] = 0xA51C;                                                   // marks the layout of this form
    }
#endif
    __syncthreads();
    if (tid == 0) {
        const int resolved = sh.resolved;
        int total = 0;
        if (resolved > 0) total = sh.pub[(resolved - 1) & 7].count;
        num_keep[img] = (sh.timed_out || sh.abort) ? WSSDL_NMS_TIMED_OUT : min(total, max_keep);
        if (done_out) done_out[img] = (total >= max_keep || n_dev[img] <= n_limit) ? 1 : 0;
    }
}

// (amdgpu_num_vgpr(80): the register allocator stays below v80, which the helpers' asm statements own)
union SweepSharedAny {
    SweepShared barrier;
    SweepAsyncShared async;
};

// (one kernel per form: with both blocks inlined behind a run-time branch the scalar registers of the two spilled to
// vector lanes -- 233 against 20 -- inside the resolver's loop)
template <bool ASYNC>
__global__ __launch_bounds__(SWEEP_BLOCK) __attribute__((amdgpu_num_vgpr(80))) void nms_sweep_pipelined_kernel(SweepArgs A) {
    extern __shared__ unsigned long long sweep_dyn[];        // the kept list
    __shared__ SweepSharedAny sh;
    if (ASYNC) nms_sweep_async_block(A, blockIdx.x, reinterpret_cast<int *>(sweep_dyn), sh.async);
    else nms_sweep_pipelined_block(A, blockIdx.x, reinterpret_cast<int *>(sweep_dyn), sh.barrier);
}

// Mask and sweep in ONE launch.  The sweep of an image is a single workgroup walking 64-box chunks
// (~1 us each: 0.2 ms for 188 chunks) on 8 of 256 CUs; the mask kernel fills the chip for 0.16 ms before
// it.  At chunk c the sweep reads column blocks <= c + 7 only -- of ALL rows above the diagonal -- so the
// matrix is handed over COLUMN SEGMENT BY COLUMN SEGMENT (16 column blocks = one 128-byte line of every
// row): workgroups 0 .. n_images-1 are the sweeps, all others compute the mask in the order segment 0,
// 1, 2, ... (inside a segment row block by row block, all images abreast), every wave counting its segment
// up in `segdone` when its words are stored (release); the sweep's stagers wait for the segments a chunk's
// loads reach into (acquire).  Segment s costs (s + 1) / 78 of the mask's work (upper triangle), so the
// sweep starts after ~1 % of it and the mask stays ahead of it from then on.  (The first version handed
// over ROW blocks: complete per-row summaries, but the early row blocks are the long ones, so the sweep
// trailed the mask for the first ~90 chunks.)  Workgroups are dispatched in index order, so the
// sweeps hold n_images workgroup slots while the mask blocks flow through the rest of the chip: every
// wait ends.  No line of the matrix, of diag_t or of the summaries holds words of two segments (row
// pitch and summary pitch are multiples of 16 words, n_max a multiple of 16), so the reader's caches
// never see a line before it is complete.
// A mask workgroup skips the blocks of an image whose sweep has finished (it kept max_keep boxes before the
// last chunk; the block is still counted): in training with a trained RPN that is most of the matrix.
// Control words, all in the int region behind the summaries (zeroed by the launcher):
//   ctl[img * ncb + s]        s < nseg: finished waves of column segment s
//   ctl[img * ncb + ncb - 2]  1 once the image's sweep has finished
// What did NOT help the sweep (it walks a chunk in ~2 us while mask blocks run, 1.1 us alone):
// * keeping the mask workgroups off its CU (a first-generation mask workgroup that found itself on a sweep's
//   CU -- HW_ID / XCC_ID against a word the sweep published -- parked after its own blocks, one wave polling
//   the finished flags between long sleeps): no change, so it is not the shared CU;
// * mask WORKERS that draw blocks from a queue, as many as the chip holds: any loop around the mask block --
//   queue, static stride, even a single trip -- made the whole launch 7x slower (not understood; the same
//   block without the loop runs at full speed).
// What did: the register count, twice.  (1) NOT capping the kernel at 64 VGPRs: the cap bought the mask role 8
// waves per SIMD (two workgroups per CU, 0.20 -> 0.17 ms) and cost the sweep role 5 spilled VGPRs and 55 SGPRs
// spilled to lanes, in the resolver's loop: 1.4 us per chunk even with the mask finished.  Uncapped (78 VGPRs,
// one workgroup per CU) the bench's full-walk steps went 0.32 -> 0.25 ms for the launch.  (2) NOT using all 128:
// one 16-wave workgroup fits a CU from 65 to 128 VGPRs, but at 128 it owns every register of the CU and the next
// mask workgroup cannot start before the last wave of the previous one has finished (its four mask blocks end
// at different times); at 96 a SIMD holds five waves, so the next workgroup's waves move in while the previous
// one drains.  With the helpers' sixteen reserved registers at v112-v127 (128 in all) the 8-image layer took
// 0.303-0.305 ms in the bench's steps, at v80-v95 (96 in all) 0.275-0.277, same box, alternating runs.
constexpr int MASK_MAX_SEGS = SWEEP_MAX_CHUNKS / MASK_SEG;
struct SegTable {
    int start[MASK_MAX_SEGS + 1];       // start[s] = (row block, segment) pairs of the segments before s
};

template <bool ASYNC>
__global__ __launch_bounds__(SWEEP_BLOCK) __attribute__((amdgpu_num_vgpr(80))) void nms_mask_sweep_fused_kernel(MaskArgs M, SweepArgs S, int n_images, int nseg, SegTable table,
                                                                            int *ctl, int fault) {
    extern __shared__ unsigned long long sweep_dyn[];
    __shared__ SweepSharedAny sh;
    const int ncb = M.ncb;
    if ((int)blockIdx.x < n_images) {
        // the latency-bound role wins every issue arbitration against mask waves that share its CU (round 4: -1.5 % on the
        // fused launch, full walk 0.448 -> 0.441 ms and early stop 0.1643 -> 0.1636, same box, alternating runs)
        __builtin_amdgcn_s_setprio(3);
        if (ASYNC) nms_sweep_async_block(S, blockIdx.x, reinterpret_cast<int *>(sweep_dyn), sh.async);
        else nms_sweep_pipelined_block(S, blockIdx.x, reinterpret_cast<int *>(sweep_dyn), sh.barrier);
        if (threadIdx.x == 0)
            __hip_atomic_store(ctl + (size_t)blockIdx.x * ncb + ncb - 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // 16 waves = 4 mask blocks; mask block v = (pair * n_images + img), pairs ordered by segment, then row block
    const int tid = threadIdx.x, lane = tid & 63, pw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long v = ((long long)blockIdx.x - n_images) * (SWEEP_BLOCK / 64 / MASK_WAVES) + (pw / MASK_WAVES);
    if (v >= (long long)table.start[nseg] * n_images) return;
    const int img = (int)(v % n_images);
    const int pair = (int)(v / n_images);
    int seg = 0;
    while (seg + 1 < nseg && pair >= table.start[seg + 1]) ++seg;
    const int rb = pair - table.start[seg];
    float (*cbox)[5][64] = reinterpret_cast<float (*)[5][64]>(sweep_dyn);
    nms_float4v (*cgeo)[64] = reinterpret_cast<nms_float4v (*)[64]>(reinterpret_cast<char *>(sweep_dyn) +
                                                                     sizeof(float) * (SWEEP_BLOCK / 64) * 5 * 64);
    nms_mask_block<true>(M, rb, seg, img, pw % MASK_WAVES, lane, cbox[pw], cgeo[pw]);
    // this wave's words (and its entries of the summary and of diag_t) have been written through: once they
    // are acknowledged, count the segment up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (fault injection, tests only: image 0's segments are never reported, so its sweep must time out)
    if (lane == 0 && !(fault && img == 0))
        __hip_atomic_fetch_add(ctl + (size_t)img * ncb + seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_nms_sweep(const unsigned long long *mask, const unsigned long long *diag_t,
                     const unsigned long long *summ, const int *n_dev, int n_max, int n_images,
                     int max_keep, const int *order, int order_stride_img, int *keep,
                     int *num_keep, const float *boxes, int box_stride_img, float *rois_padded,
                     int *kept_scratch, hipStream_t st, int n_limit, const int *done_in, int *done_out) {
    const int ncb = nms_mask_pitch(n_max);
    if (n_images == 0) return WSSDL_OK;
    size_t lds = ((size_t)max_keep + 64) * sizeof(int);
    if (nms_sweep_is_pipelined(n_max, max_keep, diag_t, summ)) {
        const SweepArgs S = {mask, diag_t, summ, n_dev, n_max, ncb, max_keep, order, order_stride_img, keep, num_keep,
                             boxes, box_stride_img, rois_padded, n_limit, done_in, done_out, nullptr, 0, 0ull, nullptr,
                             tuning().nms_sweep_async != 0 ? 1 : 0};
        if (S.async) hipLaunchKernelGGL(nms_sweep_pipelined_kernel<true>, dim3(n_images), dim3(SWEEP_BLOCK), lds, st, S);
        else hipLaunchKernelGGL(nms_sweep_pipelined_kernel<false>, dim3(n_images), dim3(SWEEP_BLOCK), lds, st, S);
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
                       box_stride_img, rois_padded, kept_scratch, n_limit, done_in, done_out);
    return check_launch();
}

// Greedy NMS only ever consults the rows of KEPT boxes and stops after max_keep of them: with
// 12000 candidates and 2000 to keep the sweep ends around candidate 4700-7000, yet the mask
// kernel (the largest piece of the proposal chain, VALU-bound) tests all 72 M pairs.  Two
// passes: (1) mask + sweep over the first `probe` candidates only; an image is done when that
// kept max_keep boxes or had no more candidates; (2) for the other images the mask kernel adds
// the column blocks beyond the probe (rows below it re-use pass 1's words) and the sweep runs
// again over everything.  Both passes of pass 2 return at once for finished images, so the usual
// case costs (probe / n)^2 of the pair tests plus two empty launches, the worst case one extra sweep.
int nms_probe_size(int n_max, int max_keep) {
    long long p = ((long long)max_keep * 4 + 1023) / 1024 * 1024;      // whole mask segments (16 x 64)
    if (p < 2048) p = 2048;
    // A failed probe costs its sweep (and the launches) on top of the full work; in training (2000 of
    // 12000: probe 8192) later steps of the bench network keep fewer than 2000 boxes and the probe
    // failed every time: 0.60 -> 0.78 ms.  Only probe when it is small against the candidate count
    // (test mode: 300 of 6000 -> probe 2048).
    return (p * 2 <= (long long)n_max) ? (int)p : n_max;
}

// the column summaries [n_images][pitch][pitch] and, behind them, the counters of the fused launch
// [n_images * pitch] ints
size_t nms_summary_alloc_words(int n_images, int n_max) {
    const size_t pitch = (size_t)nms_mask_pitch(n_max);
    // + [n_images * pitch ints of control words, rounded to 16 bytes] + [n_images][pitch][2] words of published kept bitmasks
    return (size_t)n_images * pitch * pitch + (((size_t)n_images * pitch + 3) / 4) * 2 + (size_t)n_images * pitch * 2;
}

// CUs of the current device (asked once per device; 0 when the runtime cannot tell: no fused launch then)
static int device_cu_count() {
    static int count[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (count[dev] == 0) {
        int n = 0;
        count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
    }
    return count[dev] > 0 ? count[dev] : 0;
}

static int launch_nms_fused(const float *boxes, int box_stride_img, const int *n_dev, int n_max, int n_images,
                            double thresh, unsigned long long *mask, unsigned long long *diag_t,
                            unsigned long long *summ, int max_keep, const int *order, int order_stride_img,
                            int *keep, int *num_keep, float *rois_padded, hipStream_t st) {
    const int ncb = nms_mask_pitch(n_max), nrb = cdiv(n_max, 64);
    const int nseg = cdiv(nrb, MASK_SEG);
    const int fault = tuning().nms_fused_fault;
    int *segdone = reinterpret_cast<int *>(summ + (size_t)n_images * ncb * ncb);
    const size_t ctl_words = (((size_t)n_images * ncb + 3) / 4) * 2;
    unsigned long long *keptpub = summ + (size_t)n_images * ncb * ncb + ctl_words;
    // the control words and the published kept bitmasks = 0 (the summaries need no initialisation: every entry that
    // is read is written)
    if (hipMemsetAsync(segdone, 0, sizeof(unsigned long long) * (ctl_words + (size_t)n_images * ncb * 2), st) != hipSuccess)
        return WSSDL_ERR_LAUNCH;
    const bool sparse = tuning().nms_sparse > 0;
    const MaskArgs M = {boxes, box_stride_img, n_dev, n_max, thresh, mask, ncb, diag_t, summ, 0x7fffffff, 0, nullptr,
                        NMS_DENSE_AHEAD, segdone + ncb - 2, ncb, 0, sparse ? keptpub : nullptr, tuning().nms_sparse};
    const SweepArgs S = {mask, diag_t, summ, n_dev, n_max, ncb, max_keep, order, order_stride_img, keep, num_keep,
                         boxes, box_stride_img, rois_padded, 0x7fffffff, nullptr, nullptr, segdone, nrb,
                         // "nms_wait_us" (50 ms: a column segment is ~50 us of mask work, so three orders of magnitude of
                         // slack for a GPU shared with another queue; it was 0.5 s until round 6);
                         // wssdl_set_tuning("nms_fused_fault", microseconds) shortens the wait AND withholds image 0's
                         // segment counts: the test of the time-out path
                         fault > 0 ? (unsigned long long)fault * 100ull
                                   : (unsigned long long)(tuning().nms_wait_us > 0 ? tuning().nms_wait_us : 50000) * 100ull,
                         sparse ? keptpub : nullptr, tuning().nms_sweep_async != 0 ? 1 : 0};
    const size_t lds_mask = (size_t)(SWEEP_BLOCK / 64) * (5 * 64 * sizeof(float) + 64 * sizeof(nms_float4v));
    SegTable table;
    if (nseg > MASK_MAX_SEGS) return WSSDL_ERR_INVALID_ARGUMENT;
    table.start[0] = 0;
    for (int sgm = 0; sgm < MASK_MAX_SEGS; ++sgm)      // segment sgm: the row blocks 0 .. its last column block
        table.start[sgm + 1] = table.start[sgm] + (sgm < nseg ? min((sgm + 1) * MASK_SEG, nrb) : 0);
    const long long vblocks = (long long)table.start[nseg] * n_images;
    const long long blocks = n_images + (vblocks + SWEEP_BLOCK / 64 / MASK_WAVES - 1) / (SWEEP_BLOCK / 64 / MASK_WAVES);
    if (blocks > 0x7fffffffLL) return WSSDL_ERR_INVALID_ARGUMENT;
    const size_t lds_sweep = ((size_t)max_keep + 64) * sizeof(int);
    if (S.async)
        hipLaunchKernelGGL(nms_mask_sweep_fused_kernel<true>, dim3((unsigned)blocks), dim3(SWEEP_BLOCK),
                           lds_sweep > lds_mask ? lds_sweep : lds_mask, st, M, S, n_images, nseg, table, segdone, fault);
    else
        hipLaunchKernelGGL(nms_mask_sweep_fused_kernel<false>, dim3((unsigned)blocks), dim3(SWEEP_BLOCK),
                           lds_sweep > lds_mask ? lds_sweep : lds_mask, st, M, S, n_images, nseg, table, segdone, fault);
    return check_launch();
}

int launch_nms_two_pass(const float *boxes, int box_stride_img, const int *n_dev, int n_max, int n_images,
                        double thresh, unsigned long long *mask, unsigned long long *diag_t,
                        unsigned long long *summ, int max_keep, const int *order, int order_stride_img,
                        int *keep, int *num_keep, float *rois_padded, int *kept_scratch, int *done,
                        hipStream_t st) {
    const int probe = done ? nms_probe_size(n_max, max_keep) : n_max;
    const int NO_LIMIT = 0x7fffffff;
    int rc;
    // (the sweeps hold one workgroup slot each -- one CU each, at 128 VGPRs -- for the whole launch and wait for
    // mask blocks that need the other slots: at most a quarter of the device's CUs, and never more than 64)
    if (probe >= n_max && tuning().nms_fused != 0 && nms_sweep_is_pipelined(n_max, max_keep, diag_t, summ) &&
        n_max >= 2048 && n_max % 16 == 0 && n_images <= 64 && 4 * n_images <= device_cu_count() &&
        ((reinterpret_cast<uintptr_t>(boxes) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(diag_t) |
          reinterpret_cast<uintptr_t>(summ)) & 127) == 0)
        return launch_nms_fused(boxes, box_stride_img, n_dev, n_max, n_images, thresh, mask, diag_t, summ, max_keep,
                                order, order_stride_img, keep, num_keep, rois_padded, st);
    if (probe >= n_max) {
        if ((rc = launch_nms_mask(boxes, box_stride_img, n_dev, n_max, n_images, thresh, mask, diag_t, summ, st,
                                  NO_LIMIT, 0, nullptr, max_keep)))
            return rc;
        return launch_nms_sweep(mask, diag_t, summ, n_dev, n_max, n_images, max_keep, order, order_stride_img, keep,
                                num_keep, boxes, box_stride_img, rois_padded, kept_scratch, st, NO_LIMIT, nullptr,
                                nullptr);
    }
    if ((rc = launch_nms_mask(boxes, box_stride_img, n_dev, n_max, n_images, thresh, mask, diag_t, summ, st, probe, 0,
                              nullptr, max_keep)))
        return rc;
    if ((rc = launch_nms_sweep(mask, diag_t, summ, n_dev, n_max, n_images, max_keep, order, order_stride_img, keep,
                               num_keep, boxes, box_stride_img, rois_padded, kept_scratch, st, probe, nullptr, done)))
        return rc;
    if ((rc = launch_nms_mask(boxes, box_stride_img, n_dev, n_max, n_images, thresh, mask, diag_t, summ, st, NO_LIMIT,
                              probe / 64, done, max_keep)))
        return rc;
    return launch_nms_sweep(mask, diag_t, summ, n_dev, n_max, n_images, max_keep, order, order_stride_img, keep,
                            num_keep, boxes, box_stride_img, rois_padded, kept_scratch, st, NO_LIMIT, done, nullptr);
}

// ------------------------------------------------------- standalone nms entry ---
__global__ void nms_prepare_kernel(const float *__restrict__ dets, int n,
                                   unsigned long long *__restrict__ keys, int *__restrict__ order,
                                   int *__restrict__ n_sorted, int *__restrict__ cand_fill) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = score_key(dets[(size_t)i * 5 + 4], (unsigned)i);
        order[i] = -1;                       // (initialisations that used to be three memset launches)
    }
    if (i == 0) { n_sorted[0] = 0;  cand_fill[0] = 0; }
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
    unsigned long long *keys, *cand, *thresh, *mask, *summ;
    int *order, *n_sorted, *cand_fill, *kept;
    float *boxes;
};

static size_t carve_nms(void *ws, int n, NmsWs *out) {
    Carver c(ws);
    const int ncb = nms_mask_pitch(n);
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
    w.summ = c.take<unsigned long long>(nms_summary_alloc_words(1, n));
    if (out) *out = w;
    return c.off;
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    return carve_nms(nullptr, n, nullptr);
}

static int nms_entry(const float *dets, int n, double thresh, int max_keep, int32_t *keep, int32_t *num_keep,
                     void *workspace, size_t workspace_bytes, wssdl_stream_t stream, int rule) {
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
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dets, n, w.keys, w.order,
                       w.n_sorted, w.cand_fill);
    int rc = check_launch();
    if (rc) return rc;
    const size_t mask_bytes = sizeof(unsigned long long) * (size_t)n * nms_mask_pitch(n);
    if (tuning().topk_sort != 0 && order_sort_supported(n, 1) && order_sort_scratch_bytes(1, n) <= mask_bytes)
        rc = launch_order_sort(w.keys, n, 1, n, w.order, w.n_sorted, w.cand_fill, w.mask, mask_bytes, st);
    else
        rc = launch_rank_topk(w.keys, n, 1, n, w.cand, w.thresh, w.cand_fill, w.order, w.n_sorted, w.mask, mask_bytes, st);
    if (rc) return rc;
    hipLaunchKernelGGL(nms_gather_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dets, w.order,
                       w.n_sorted, n, w.boxes);
    rc = check_launch();
    if (rc) return rc;
    rc = launch_nms_mask(w.boxes, n * 4, w.n_sorted, n, 1, thresh, w.mask, w.cand, w.summ, st, 0x7fffffff, 0, nullptr, max_keep,
                         rule);
    if (rc) return rc;
    return launch_nms_sweep(w.mask, w.cand, w.summ, w.n_sorted, n, 1, max_keep, w.order, n, keep, num_keep,
                            nullptr, 0, nullptr, w.kept, st, 0x7fffffff, nullptr, nullptr);
}

extern "C" int wssdl_nms(const float *dets, int n, double thresh, int max_keep, int32_t *keep,
                         int32_t *num_keep, void *workspace, size_t workspace_bytes,
                         wssdl_stream_t stream) {
    return nms_entry(dets, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes, stream, 0);
}

extern "C" int wssdl_nms_new(const float *dets, int n, double thresh, int max_keep, int32_t *keep,
                             int32_t *num_keep, void *workspace, size_t workspace_bytes,
                             wssdl_stream_t stream) {
    return nms_entry(dets, n, thresh, max_keep, keep, num_keep, workspace, workspace_bytes, stream, 1);
}

#ifdef WSSDL_SWEEP_PROFILE
// profile build only (tools/nms_sweep_profile.py): copy the sweep's per-wave cycle counts to the host and zero them
extern "C" __attribute__((visibility("default"))) int wssdl_debug_sweep_profile_read(unsigned long long *host, size_t bytes) {
    if (bytes > sizeof(wssdl::wssdl_sweep_prof)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (hipDeviceSynchronize() != hipSuccess) return WSSDL_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(wssdl::wssdl_sweep_prof), bytes) != hipSuccess) return WSSDL_ERR_LAUNCH;
    static unsigned long long zeros[sizeof(wssdl::wssdl_sweep_prof) / 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(wssdl::wssdl_sweep_prof), zeros, sizeof(zeros)) == hipSuccess ? WSSDL_OK : WSSDL_ERR_LAUNCH;
}
#endif
