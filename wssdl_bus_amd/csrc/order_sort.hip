// Descending order of the candidates of every image, chip-wide (the ranking of nms.hip runs one workgroup per
// image for three of its four phases: 0.10 ms for 8 x 21 546 keys, 0.20 ms for 1 x 56 700, nearly all latency):
// sorted runs + cross ranks, two launches.  Nothing is allocated or synchronised.
//
// (A first form sorted ALL keys of ALL images with one device-wide library sort, rocprim::radix_sort_keys_desc on
// composite keys (n_images - 1 - image) << 48 | score bits << 16 | anchor index, between a pack and a finish
// kernel.  At these sizes the library runs its merge sort -- a block sort and EIGHT merge passes of 6.5 us each:
// 0.074 ms for 8 x 21 546 keys against 0.034 here; it needed M <= 65 535 and a stub for the library's own
// environment look-up.  Measured in profiles/r03_order_ab.log, removed.)
#include "order_sort.hip.h"

namespace wssdl {

typedef float float4v __attribute__((ext_vector_type(4)));

// A candidate's final position is
//   its position in its own sorted run  +  for every other run of its image, the number of keys greater than it,
// and the second term needs no merge passes: `order_runs_kernel` sorts runs of 2048 keys (one workgroup each,
// rocprim::block_sort in LDS), `order_rank_kernel` gives every run a workgroup of four 256-lane groups that
// stage the other runs through LDS (group q takes runs q, q + 4, ...; the next run's keys are in flight while one
// is searched) and count by binary search -- a thread's 8 own keys are consecutive in its run: two full
// searches (first and last key, interleaved) bracket the other six, which are searched inside the bracket
// (a run is a contiguous slice of anchors whose scores may all fall into one gap of another run, so a linear
// scan from the previous key's position would not be bounded; this is, by log2 of the gap).  Keys are the
// plain score_key values (no composite, no 16-bit index limit); zero keys sort to the end of their run and are
// neither ranked nor written.
constexpr int RUN_THREADS = 256, RUN_ITEMS = 8;      // the rank kernel's view of a run: 8 keys per thread of a 256-lane group
constexpr int RANK_GROUPS = 4;
constexpr int MAX_RUNS = 64;

int order_runs_of(int M) { return cdiv(M, RUN); }
static int runs_of(int M) { return order_runs_of(M); }

__global__ __launch_bounds__(SORT_THREADS) void order_runs_kernel(const unsigned long long *__restrict__ keys, int M, int runs,
                                                                 unsigned long long *__restrict__ sorted_runs) {
    __shared__ typename RunSort::storage_type storage;
    const int img = blockIdx.x / runs, r = blockIdx.x - img * runs;
    const int base = r * RUN + threadIdx.x * SORT_ITEMS;
    unsigned long long k[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) k[i] = (base + i < M) ? keys[(size_t)img * M + base + i] : 0ull;
    sort_and_store_run(k, storage, sorted_runs + (size_t)blockIdx.x * RUN);
}

// OWN = own keys per thread: a workgroup ranks RUN_THREADS * OWN keys of its run (a 1/(8/OWN) part of it), so that
// short launches spread over more CUs -- the rounds are bound by LDS reads at random addresses, per CU.
template <int OWN>
__global__ __launch_bounds__(RUN_THREADS *RANK_GROUPS) void order_rank_kernel(const unsigned long long *__restrict__ sorted_runs,
                                                                              int runs, int topn, int *__restrict__ sorted_index,
                                                                              int *__restrict__ n_sorted, const float *__restrict__ boxes,
                                                                              int M, float *__restrict__ sorted_boxes) {
    __shared__ unsigned long long stage[RANK_GROUPS][RUN];
    constexpr int PARTS = RUN_ITEMS / OWN, PART = RUN / PARTS;
    __shared__ int rank[PART];
    __shared__ int s_candidates;
    const int part = blockIdx.x % PARTS, run_id = blockIdx.x / PARTS;
    const int img = run_id / runs, r = run_id - img * runs;
    const int q = threadIdx.x / RUN_THREADS, t = threadIdx.x - q * RUN_THREADS;
    const unsigned long long *image_runs = sorted_runs + (size_t)img * runs * RUN;
    unsigned long long own[OWN], nxt[RUN_ITEMS];
    int greater[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        own[i] = image_runs[(size_t)r * RUN + part * PART + t * OWN + i];
        greater[i] = 0;
    }
    int n_own = 0;                            // the thread's non-zero keys: a prefix (zeros are the tail of the run)
#pragma unroll
    for (int i = 0; i < OWN; ++i) n_own += own[i] != 0ull;
    for (int i = threadIdx.x; i < PART; i += RUN_THREADS * RANK_GROUPS) rank[i] = 0;
    if (threadIdx.x == 0) s_candidates = 0;
    auto fetch = [&](unsigned long long (&dst)[RUN_ITEMS], int s) {
#pragma unroll
        for (int i = 0; i < RUN_ITEMS; ++i) dst[i] = (s < runs) ? image_runs[(size_t)s * RUN + i * RUN_THREADS + t] : 0ull;
    };
    fetch(nxt, q);
    int candidates = 0;                       // non-zero keys this thread staged (every run is staged exactly once)
    // one round: the fetched keys of run s0 + q go to LDS, the buffer is refilled for the next round, then the
    // searches (both barriers are reached by every thread: the early return sits behind them).  A round costs
    // ~5.8 us and it is the LDS: 16 waves x ~56 reads at random addresses (4-5 lanes per bank); padding the
    // stage against power-of-two strides and a second round of runs in flight changed nothing (measured).
    auto round = [&](unsigned long long (&buf)[RUN_ITEMS], int s0) {
        const int s = s0 + q;
        __syncthreads();                      // the searches of the previous round are done with `stage`
#pragma unroll
        for (int i = 0; i < RUN_ITEMS; ++i) {
            stage[q][i * RUN_THREADS + t] = buf[i];
            candidates += buf[i] != 0ull;
        }
        fetch(buf, s + RANK_GROUPS);
        __syncthreads();
        if (s >= runs || s == r || n_own == 0) return;
        const unsigned long long *st = stage[q];
        // keys of run s greater than a key k: the first position p with st[p] <= k.  own[0] and the thread's
        // last non-zero key by the branch-free search (RUN is a power of two: 12 reads, both chains in flight
        // together), the keys between them inside that bracket (own keys descend, so their counts ascend).
        const unsigned long long k_first = own[0];
        unsigned long long k_last = own[0];   // (selects, not own[n_own - 1]: the keys stay in registers)
#pragma unroll
        for (int i = 1; i < OWN; ++i) k_last = (i < n_own) ? own[i] : k_last;
        int p0 = 0, p1 = 0;
#pragma unroll
        for (int half = RUN / 2; half >= 1; half >>= 1) {
            const unsigned long long a = st[p0 + half - 1], b = st[p1 + half - 1];
            p0 += (a > k_first) ? half : 0;
            p1 += (b > k_last) ? half : 0;
        }
        p0 += st[p0] > k_first;
        p1 += st[p1] > k_last;
        greater[0] += p0;
#pragma unroll
        for (int i = 1; i < OWN; ++i) greater[i] += (i == n_own - 1) ? p1 : 0;
        int lo[OWN], hi[OWN];
#pragma unroll
        for (int i = 1; i < OWN - 1; ++i) { lo[i] = p0;  hi[i] = (i < n_own - 1) ? p1 : p0; }
        bool open = OWN > 2;
        while (open) {
            open = false;
#pragma unroll
            for (int i = 1; i < OWN - 1; ++i) {
                if (lo[i] < hi[i]) {
                    const int mid = (lo[i] + hi[i]) >> 1;
                    if (st[mid] > own[i]) lo[i] = mid + 1;
                    else hi[i] = mid;
                    open = true;
                }
            }
        }
#pragma unroll
        for (int i = 1; i < OWN - 1; ++i)
            if (i < n_own - 1) greater[i] += lo[i];
    };
    for (int s0 = 0; s0 < runs; s0 += RANK_GROUPS) round(nxt, s0);
#pragma unroll
    for (int i = 0; i < OWN; ++i)
        if (greater[i]) atomicAdd(&rank[t * OWN + i], greater[i]);
    for (int off = 32; off; off >>= 1) candidates += __shfl_down(candidates, off, 64);
    if ((threadIdx.x & 63) == 0 && candidates) atomicAdd(&s_candidates, candidates);
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            const int pos = part * PART + t * OWN + i + rank[t * OWN + i];
            if (own[i] != 0ull && pos < topn) {
                const int idx = (int)(own[i] & 0xffffffffull);
                sorted_index[(size_t)img * topn + pos] = idx;
                if (sorted_boxes)         // the caller's gather, fused: boxes [n_images, M, 4] -> [n_images, topn, 4]
                    *reinterpret_cast<float4v *>(sorted_boxes + ((size_t)img * topn + pos) * 4) =
                        *reinterpret_cast<const float4v *>(boxes + ((size_t)img * M + idx) * 4);
            }
        }
    }
    if (r == 0 && part == 0 && threadIdx.x == 0) n_sorted[img] = min(s_candidates, topn);
}

bool order_sort_supported(int M, int n_images) { return M >= 1 && runs_of(M) <= MAX_RUNS && n_images >= 1 && n_images <= 32768; }

size_t order_sort_scratch_bytes(int n_images, int M) {
    return (size_t)n_images * runs_of(M) * RUN * sizeof(unsigned long long);
}

// Same contract as launch_rank_topk: sorted_index [n_images, topn] pre-filled with -1, n_sorted written
// (`valid` is not used any more; kept in the signature for the callers' workspace layout).
int launch_order_sort(const unsigned long long *keys, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      int *valid, void *scratch, size_t scratch_bytes, hipStream_t st) {
    (void)valid;
    if (!order_sort_supported(M, n_images) || scratch_bytes < order_sort_scratch_bytes(n_images, M))
        return WSSDL_ERR_WORKSPACE;
    const int runs = runs_of(M);
    unsigned long long *sorted_runs = static_cast<unsigned long long *>(scratch);
    hipLaunchKernelGGL(order_runs_kernel, dim3(n_images * runs), dim3(SORT_THREADS), 0, st, keys, M, runs, sorted_runs);
    int rc = check_launch();
    if (rc) return rc;
    return launch_order_rank(sorted_runs, M, n_images, topn, sorted_index, n_sorted, nullptr, nullptr, st);
}

// The second launch alone, for callers whose own first kernel writes the sorted runs (sort_and_store_run):
// sorted_runs [n_images, order_runs_of(M), RUN].  boxes / sorted_boxes (optional): the gather of the ranked
// candidates' boxes, [n_images, M, 4] -> [n_images, topn, 4].
int launch_order_rank(const unsigned long long *sorted_runs, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      const float *boxes, float *sorted_boxes, hipStream_t st) {
    if (!order_sort_supported(M, n_images)) return WSSDL_ERR_INVALID_ARGUMENT;
    const int runs = runs_of(M);
    // as many parts per run as keep the launch within one workgroup per CU
    const dim3 block(RUN_THREADS * RANK_GROUPS);
    const int run_count = n_images * runs;
    if (run_count * 4 <= 256)
        hipLaunchKernelGGL(order_rank_kernel<2>, dim3(run_count * 4), block, 0, st, sorted_runs, runs, topn, sorted_index, n_sorted,
                           boxes, M, sorted_boxes);
    else if (run_count * 2 <= 256)
        hipLaunchKernelGGL(order_rank_kernel<4>, dim3(run_count * 2), block, 0, st, sorted_runs, runs, topn, sorted_index, n_sorted,
                           boxes, M, sorted_boxes);
    else
        hipLaunchKernelGGL(order_rank_kernel<8>, dim3(run_count), block, 0, st, sorted_runs, runs, topn, sorted_index, n_sorted,
                           boxes, M, sorted_boxes);
    return check_launch();
}

}  // namespace wssdl
