// Descending order of the candidates of every image, chip-wide (the ranking of nms.hip runs one workgroup per
// image for three of its four phases: 0.10 ms for 8 x 21 546 keys, 0.20 ms for 1 x 56 700, nearly all latency).
//
// Form 1 (tuning topk_sort = 2): ONE device-wide sort of all keys of all images by rocPRIM
// (rocprim::radix_sort_keys_desc, a plain primitive like a GEMM) on composite keys
//   (n_images - 1 - image) << 48 | order-preserving score bits << 16 | anchor index
// so that one descending sort yields, image after image, the candidates by descending score with ties
// broken towards the higher index -- exactly the total order of the 64-bit keys of nms.hip.h (score_key),
// hence the same sorted_index as launch_rank_topk.  Non-candidates (key 0) sink to the end of their image.
// Needs M <= 65 535 anchors per image.  0.074 ms for 8 x 21 546 keys.
// Form 2 (topk_sort = 1, the default): sorted runs + cross ranks, below.
// Nothing is allocated or synchronised in either.
#include <cstdlib>
#include <cstring>

#include "order_sort.hip.h"

// The C ABI promises that this library never reads the environment (include/wssdl_bus_hip.h); rocPRIM consults
// one variable of its own (ROCPRIM_USE_ATOMIC_BLOCK_ID, device/detail/ordered_block_id.hpp).  Inside this
// translation unit its std::getenv is a function that knows no variables, so the library keeps its default.
namespace std {
inline char *wssdl_no_environment(const char *) { return nullptr; }
}  // namespace std
#define getenv wssdl_no_environment
#include <rocprim/device/device_radix_sort.hpp>
#undef getenv

namespace wssdl {

typedef float float4v __attribute__((ext_vector_type(4)));

static bool device_sort_supported(int M, int n_images) { return M >= 1 && M <= 65535 && n_images >= 1 && n_images <= 32768; }

static int image_bits(int n_images) {
    int b = 0;
    while ((1 << b) < n_images) ++b;
    return b;
}

static size_t sort_temp_bytes(size_t n, int end_bit) {
    size_t bytes = 0;
    unsigned long long *p = nullptr;
    if (rocprim::radix_sort_keys_desc(nullptr, bytes, p, p, n, 0, end_bit, (hipStream_t)0) != hipSuccess) return 0;
    return (bytes + 255) & ~size_t(255);
}

static size_t device_sort_scratch_bytes(int n_images, int M) {
    if (!device_sort_supported(M, n_images)) return 0;
    const size_t n = (size_t)n_images * M;
    const size_t arr = (n * sizeof(unsigned long long) + 255) & ~size_t(255);
    return 2 * arr + sort_temp_bytes(n, 48 + image_bits(n_images));
}

// keys (score_key format, 0 = not a candidate) -> composite keys
__global__ __launch_bounds__(256) void order_pack_kernel(const unsigned long long *__restrict__ keys, int M, int n_images,
                                                         unsigned long long *__restrict__ packed) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)n_images * M) return;
    const int img = (int)(g / M);
    const unsigned long long k = keys[g];
    packed[g] = ((unsigned long long)(n_images - 1 - img) << 48) | (k != 0ull ? (((k >> 32) << 16) | (k & 0xffffull)) : 0ull);
}

// The candidates of an image are the sorted keys with a non-zero score field (a candidate's order-preserving
// score bits have their top bit set or are the complement of a negative float's: never 0): their count is the
// position of the first key whose low 48 bits are zero -- a binary search per workgroup (counting them with
// atomics in the pack kernel cost 33 us: 2700 waves on 8 counters).
__global__ __launch_bounds__(256) void order_finish_kernel(const unsigned long long *__restrict__ sorted, int M, int topn,
                                                           int *__restrict__ sorted_index, int *__restrict__ n_sorted) {
    const int img = blockIdx.y;
    const unsigned long long *seg = sorted + (size_t)img * M;
    int lo = 0, hi = M;                        // first position with an empty score field, in [0, M]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((seg[mid] & 0xffffffffffffull) != 0ull) lo = mid + 1;
        else hi = mid;
    }
    const int n = min(lo, topn);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) n_sorted[img] = n;
    if (p < n) sorted_index[(size_t)img * topn + p] = (int)(seg[p] & 0xffffull);
}

static int launch_device_sort(const unsigned long long *keys, int M, int n_images, int topn, int *sorted_index,
                              int *n_sorted, void *scratch, hipStream_t st) {
    const size_t n = (size_t)n_images * M;
    const size_t arr = (n * sizeof(unsigned long long) + 255) & ~size_t(255);
    const int end_bit = 48 + image_bits(n_images);
    unsigned long long *a = static_cast<unsigned long long *>(scratch);
    unsigned long long *b = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + arr);
    void *temp = static_cast<char *>(scratch) + 2 * arr;
    size_t temp_bytes = sort_temp_bytes(n, end_bit);
    hipLaunchKernelGGL(order_pack_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, st, keys, M, n_images, a);
    int rc = check_launch();
    if (rc) return rc;
    const hipError_t e = rocprim::radix_sort_keys_desc(temp, temp_bytes, a, b, n, 0, end_bit, st);
    if (e != hipSuccess) { set_last_error(e);  return WSSDL_ERR_LAUNCH; }
    hipLaunchKernelGGL(order_finish_kernel, dim3(cdiv(topn, 256), n_images), dim3(256), 0, st, b, M, topn, sorted_index,
                       n_sorted);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Form 2 (the default): sorted runs + cross ranks, two launches.
//
// The device-wide sort above is, at these sizes (8 x 21 546 keys), the library's merge sort: one block-sort
// launch and EIGHT merge passes of 6.5 us each -- latency again.  A candidate's final position is
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
    const size_t runs_bytes = (size_t)n_images * runs_of(M) * RUN * sizeof(unsigned long long);
    return runs_bytes > device_sort_scratch_bytes(n_images, M) ? runs_bytes : device_sort_scratch_bytes(n_images, M);
}

// Same contract as launch_rank_topk: sorted_index [n_images, topn] pre-filled with -1, n_sorted written
// (`valid` is not used any more; kept in the signature for the callers' workspace layout).
int launch_order_sort(const unsigned long long *keys, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      int *valid, void *scratch, size_t scratch_bytes, hipStream_t st) {
    (void)valid;
    if (!order_sort_supported(M, n_images) || scratch_bytes < order_sort_scratch_bytes(n_images, M))
        return WSSDL_ERR_WORKSPACE;
    if (tuning().topk_sort == 2 && device_sort_supported(M, n_images))
        return launch_device_sort(keys, M, n_images, topn, sorted_index, n_sorted, scratch, st);
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
