// Internal interface of nms.hip, shared with proposal.hip.
#pragma once
#include "common.hip.h"

namespace wssdl {

// Order-preserving 64-bit sort key: f32 score bits mapped so that unsigned
// compare == float compare, index in the low word as the tie-break (equal
// scores: higher index first).  0 is reserved for "not a candidate".
__host__ __device__ __forceinline__ unsigned long long score_key(float score, unsigned idx) {
    unsigned u = __builtin_bit_cast(unsigned, score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)idx;
}

// keys [n_images, M] (0 = not a candidate); sorted_index [n_images, topn] (pre-filled with -1)
// receives the index (low key word) of the element at each descending-score position < topn;
// n_sorted [n_images] receives min(#valid, topn).  Scratch: cand [n_images, topn] u64,
// thresh [n_images] u64, cand_fill [n_images] i32 (pre-zeroed).
// `scratch` (optional, rank_topk_scratch_bytes; may alias memory that is only written after the
// ranking, e.g. the suppression matrix) enables the bucketed ranking for large topn.
size_t rank_topk_scratch_bytes(int n_images, int topn);
int launch_rank_topk(const unsigned long long *keys, int M, int n_images, int topn,
                     unsigned long long *cand, unsigned long long *thresh, int *cand_fill,
                     int *sorted_index, int *n_sorted, void *scratch, size_t scratch_bytes,
                     hipStream_t st);

// The same result by sorted runs + cross ranks (order_sort.hip): every image's keys are cut into runs of 2048,
// each run sorted by one workgroup (hand-written bitonic network, order_sort.hip.h), and a candidate's position is
// its position in its run plus, for every other run of its image, the number of greater keys (binary searches);
// faster than the ranking above whenever it applies: at most 64 runs per image (M <= 131 072).  `valid` is unused
// (kept for the signature); scratch = order_sort_scratch_bytes (the sorted runs; may alias memory that is only
// written after the ordering, e.g. the suppression matrix).
bool order_sort_supported(int M, int n_images);
size_t order_sort_scratch_bytes(int n_images, int M);
int launch_order_sort(const unsigned long long *keys, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      int *valid, void *scratch, size_t scratch_bytes, hipStream_t st);
// Its two launches apart, for a caller whose own kernel produces the sorted runs (order_sort.hip.h):
// sorted_runs [n_images, order_runs_of(M), 2048] u64; optional gather boxes [n_images, M, 4] -> sorted_boxes
// [n_images, topn, 4] of the ranked candidates.
int order_runs_of(int M);
int launch_order_rank(const unsigned long long *sorted_runs, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      const float *boxes, float *sorted_boxes, hipStream_t st);

typedef float nms_float4v __attribute__((ext_vector_type(4)));
typedef float nms_float2v __attribute__((ext_vector_type(2)));

// cpu_nms.pyx:11-15 define max(a, b) = a if a >= b else b (min likewise).  For finite operands
// v_max_f32 / v_min_f32 return the same value up to the sign of a zero result, and a zero
// result only ever feeds `x - zero` or `zero + 1.0f` here, which do not depend on that sign;
// one instruction instead of compare + hazard nop + select (the kernel is VALU-bound).  NaN
// coordinates (not a valid input) would differ.
// (inline asm: __builtin_fmaxf would add a v_max x, x canonicalisation per operand in IEEE mode)
__device__ __forceinline__ float fmax_ref(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmin_ref(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmax0_ref(float a) {      // max(0.0f, a)
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
    return r;
}

__device__ __forceinline__ float box_area_ref(float x1, float y1, float x2, float y2) {
    float w = x2 - x1;  w = w + 1.0f;       // numpy f32: (x2 - x1 + 1) * (y2 - y1 + 1), cpu_nms.pyx:24
    float h = y2 - y1;  h = h + 1.0f;
    return w * h;
}


// boxes: per image [n_max, 4] f32 in score order (image stride box_stride_img floats);
// mask: per image [n_max, nms_mask_pitch(n_max)] u64, upper triangle written -- every word when the sweep
// for (n_max, max_keep) is the general one; only the non-zero words and the band next to the diagonal
// when it is the pipelined one, which finds the others through summ.
// diag_t (optional): per image [n_max] u64, for box i the boxes of ITS OWN 64-chunk with a lower
// index that suppress it (the transposed diagonal block), consumed by the pipelined sweep.
// summ (optional): per image [pitch column blocks][pitch row blocks] u64, bit r of entry (cb, rb) = mask
// word cb of row 64 rb + r is non-zero; lets the sweep skip the loads of all-zero words.
int nms_mask_pitch(int n_max);       // row pitch of mask in u64 words: ceil(n_max / 64) rounded up to 16
// u64 words to allocate for `summ` of launch_nms_two_pass: the summaries plus the counters of
// the fused mask + sweep launch behind them
size_t nms_summary_alloc_words(int n_images, int n_max);
int launch_nms_mask(const float *boxes, int box_stride_img, const int *n_dev, int n_max,
                    int n_images, double thresh, unsigned long long *mask,
                    unsigned long long *diag_t, unsigned long long *summ, hipStream_t st,
                    int n_limit, int cb_min, const int *done, int max_keep, int rule = 0);

// keep (optional) [n_images, max_keep] i32; rois_padded (optional) [n_images, max_keep, 5];
// kept_scratch [n_images, max_keep + 64] i32: only needed when the kept list does not fit in
// LDS (max_keep > ~15k), may be NULL otherwise.
int launch_nms_sweep(const unsigned long long *mask, const unsigned long long *diag_t,
                     const unsigned long long *summ, const int *n_dev, int n_max, int n_images,
                     int max_keep, const int *order, int order_stride_img, int *keep,
                     int *num_keep, const float *boxes, int box_stride_img, float *rois_padded,
                     int *kept_scratch, hipStream_t st, int n_limit, const int *done_in, int *done_out);

// mask + sweep in two passes (a probe over the first candidates, then the rest for the images that
// need it); `done` [n_images] i32 scratch, NULL = one pass.  See nms.hip.
// Alignment contract of the fused one-pass form (mask and sweep in one launch, the matrix handed over column
// segment by column segment): no 128-byte line may hold words of two segments, so `boxes`, `mask`, `diag_t` and
// `summ` must be 128-byte aligned (every workspace slice of Carver is 256-byte aligned) and n_max % 16 == 0;
// launch_nms_two_pass checks both and runs the two-launch form otherwise.
int nms_probe_size(int n_max, int max_keep);
int launch_nms_two_pass(const float *boxes, int box_stride_img, const int *n_dev, int n_max, int n_images,
                        double thresh, unsigned long long *mask, unsigned long long *diag_t,
                        unsigned long long *summ, int max_keep, const int *order, int order_stride_img,
                        int *keep, int *num_keep, float *rois_padded, int *kept_scratch, int *done,
                        hipStream_t st);

}  // namespace wssdl
