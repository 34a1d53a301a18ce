// Device half of order_sort.hip that other kernels fuse with: one workgroup sorts one run of RUN keys.
//
// Round 4: the run sort is hand-written (rounds 3's rocprim::block_sort is gone, and with it the last library
// primitive inside the hot path).  RUN = 2048 unique 64-bit keys (0 = not a candidate, may repeat), 1024 threads,
// two keys per thread in registers (thread t holds elements 2t and 2t + 1), descending bitonic network:
//   * partner distance 1            the thread's own two registers,
//   * distance 2 .. 64              the partner is a lane of the same wave: two 64-bit __shfl_xor (ds_bpermute through
//                                   the LDS crossbar, no LDS memory, no barrier),
//   * distance 128 .. 1024          across waves: both keys through an LDS image of the run and a barrier pair
// -- 66 compare-exchange stages, 10 of them through LDS (20 barriers) instead of a barrier pair per stage.
#pragma once

#include "nms.hip.h"

namespace wssdl {

// 1024 threads x 2 keys: with the decode fused in front (proposal.hip) 16.7 us for 88 runs against 18.2 (512 x 4)
// and 24.0 (256 x 8) -- the f64 exp of the decode wants the threads, the merge rounds do not mind them.
constexpr int RUN = 2048;
constexpr int SORT_THREADS = 1024, SORT_ITEMS = RUN / SORT_THREADS;
static_assert(SORT_ITEMS == 2 && SORT_THREADS == 1024, "the bitonic network below is written for 1024 threads x 2 keys");

struct RunSortStorage {
    unsigned long long image[RUN];      // the run, element e at image[e], for the stages whose partner is in another wave
};
struct RunSort {
    typedef RunSortStorage storage_type;
};

// one compare-exchange: this thread keeps the larger key when keep_max, else the smaller one
__device__ __forceinline__ unsigned long long run_sort_keep(unsigned long long mine, unsigned long long other, bool keep_max) {
    const bool other_greater = other > mine;
    return (other_greater == keep_max) ? other : mine;
}

// k = the thread's SORT_ITEMS consecutive keys of the run (blocked arrangement, 0 = not a candidate); the run
// goes to `out_run` [RUN] in descending order, zeros last.  Called by all SORT_THREADS threads of the workgroup.
__device__ __forceinline__ void sort_and_store_run(unsigned long long (&k)[SORT_ITEMS], RunSort::storage_type &storage,
                                                   unsigned long long *__restrict__ out_run) {
    const int t = threadIdx.x;
    unsigned long long a = k[0], b = k[1];               // elements e = 2t, 2t + 1
#pragma unroll
    for (int size = 2; size <= RUN; size <<= 1) {
        // blocks of `size` elements alternate direction; the last merge (size == RUN) is descending throughout
        const bool desc = ((2 * t) & size) == 0;
#pragma unroll
        for (int j = size >> 1; j >= 1; j >>= 1) {
            if (j >= 128) {
                // partner element e ^ j lives in another wave: exchange through LDS
                __syncthreads();                          // (the image may still be read by the previous stage)
                storage.image[2 * t] = a;
                storage.image[2 * t + 1] = b;
                __syncthreads();
                const unsigned long long pa = storage.image[(2 * t) ^ j], pb = storage.image[(2 * t + 1) ^ j];
                const bool lower = ((2 * t) & j) == 0;    // this thread holds the lower index of both pairs
                a = run_sort_keep(a, pa, lower == desc);
                b = run_sort_keep(b, pb, lower == desc);
            } else if (j >= 2) {
                // partner thread t ^ (j / 2) is a lane of this wave (j / 2 <= 32)
                const unsigned long long pa = __shfl_xor(a, j >> 1, 64), pb = __shfl_xor(b, j >> 1, 64);
                const bool lower = (t & (j >> 1)) == 0;
                a = run_sort_keep(a, pa, lower == desc);
                b = run_sort_keep(b, pb, lower == desc);
            } else {
                // j == 1: the thread's own pair; element 2t is the lower index
                const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
                a = desc ? hi : lo;
                b = desc ? lo : hi;
            }
        }
    }
    unsigned long long *o = out_run + t * SORT_ITEMS;
    o[0] = a;
    o[1] = b;
}

}  // namespace wssdl
