// Device half of order_sort.hip that other kernels fuse with: one workgroup sorts one run of RUN keys.
#pragma once
#include <rocprim/block/block_sort.hpp>

#include "nms.hip.h"

namespace wssdl {

// 1024 threads x 2 keys: with the decode fused in front (proposal.hip) 16.7 us for 88 runs against 18.2 (512 x 4)
// and 24.0 (256 x 8) -- the f64 exp of the decode wants the threads, the merge rounds do not mind them.
constexpr int RUN = 2048;
constexpr int SORT_THREADS = 1024, SORT_ITEMS = RUN / SORT_THREADS;

struct KeyGreater {
    __device__ __forceinline__ bool operator()(const unsigned long long &a, const unsigned long long &b) const { return a > b; }
};

using RunSort = rocprim::block_sort<unsigned long long, SORT_THREADS, SORT_ITEMS>;

// k = the thread's SORT_ITEMS consecutive keys of the run (blocked arrangement, 0 = not a candidate); the run
// goes to `out_run` [RUN] in descending order, zeros last.  Called by all SORT_THREADS threads of the workgroup.
__device__ __forceinline__ void sort_and_store_run(unsigned long long (&k)[SORT_ITEMS], typename RunSort::storage_type &storage,
                                                   unsigned long long *__restrict__ out_run) {
    RunSort().sort(k, storage, KeyGreater());
    unsigned long long *o = out_run + threadIdx.x * SORT_ITEMS;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) o[i] = k[i];
}

}  // namespace wssdl
