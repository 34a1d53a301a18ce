// Descending order of the candidates of every image by ONE device-wide radix sort (rocPRIM).
//
// The hand-written ranking of nms.hip (radix select of the topn-th key, compaction, sample sort) runs as one
// workgroup per image for three of its four phases: 0.10 ms for 8 x 21 546 keys, 0.20 ms for 1 x 56 700 (test
// mode), nearly all of it latency of single-workgroup passes.  Sorting ALL keys of ALL images in one
// device-wide LSD radix sort uses the whole chip instead: 0.065 / 0.045 ms for the same inputs
// (tools/probes/rocprim_sort_probe.py).  The sort is the library's (rocprim::radix_sort_keys_desc, a plain
// primitive like a GEMM); the keys, what is sorted and what comes out are defined here:
//   composite key = (n_images - 1 - image) << 48 | order-preserving score bits << 16 | anchor index
// so that one descending sort yields, image after image, the candidates by descending score with ties
// broken towards the higher index -- exactly the total order of the 64-bit keys of nms.hip.h (score_key),
// hence the same sorted_index as launch_rank_topk.  Non-candidates (key 0) sink to the end of their image.
// Needs M <= 65 535 anchors per image and n_images <= 32 768; nothing is allocated or synchronised.
#include <cstdlib>
#include <cstring>

#include "nms.hip.h"

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

bool order_sort_supported(int M, int n_images) { return M >= 1 && M <= 65535 && n_images >= 1 && n_images <= 32768; }

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

size_t order_sort_scratch_bytes(int n_images, int M) {
    const size_t n = (size_t)n_images * M;
    const size_t arr = (n * sizeof(unsigned long long) + 255) & ~size_t(255);
    return 2 * arr + sort_temp_bytes(n, 48 + image_bits(n_images));
}

// keys (score_key format, 0 = not a candidate) -> composite keys; candidates counted per image
__global__ __launch_bounds__(256) void order_pack_kernel(const unsigned long long *__restrict__ keys, int M, int n_images,
                                                         unsigned long long *__restrict__ packed, int *__restrict__ valid) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n_images * M;
    const int img = g < total ? (int)(g / M) : -1;
    unsigned long long k = g < total ? keys[g] : 0ull;
    const bool cand = k != 0ull;
    if (g < total)
        packed[g] = ((unsigned long long)(n_images - 1 - img) << 48) |
                    (cand ? (((k >> 32) << 16) | (k & 0xffffull)) : 0ull);
    // one atomic per wave and image (a wave spans at most two images when M >= 64; general loop)
    unsigned long long todo = __ballot(cand);
    const int lane = threadIdx.x & 63;
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int i0 = __builtin_amdgcn_readlane(img, leader);
        const unsigned long long same = __ballot(cand && img == i0);
        if (lane == leader) atomicAdd(&valid[i0], __popcll(same));
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void order_finish_kernel(const unsigned long long *__restrict__ sorted, int M, int topn,
                                                           const int *__restrict__ valid, int *__restrict__ sorted_index,
                                                           int *__restrict__ n_sorted) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(valid[img], topn);
    if (p == 0) n_sorted[img] = n;
    if (p < n) sorted_index[(size_t)img * topn + p] = (int)(sorted[(size_t)img * M + p] & 0xffffull);
}

// Same contract as launch_rank_topk: sorted_index [n_images, topn] pre-filled with -1, n_sorted written;
// `valid` [n_images] pre-zeroed scratch counters.
int launch_order_sort(const unsigned long long *keys, int M, int n_images, int topn, int *sorted_index, int *n_sorted,
                      int *valid, void *scratch, size_t scratch_bytes, hipStream_t st) {
    if (!order_sort_supported(M, n_images) || scratch_bytes < order_sort_scratch_bytes(n_images, M))
        return WSSDL_ERR_WORKSPACE;
    const size_t n = (size_t)n_images * M;
    const size_t arr = (n * sizeof(unsigned long long) + 255) & ~size_t(255);
    const int end_bit = 48 + image_bits(n_images);
    unsigned long long *a = static_cast<unsigned long long *>(scratch);
    unsigned long long *b = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + arr);
    void *temp = static_cast<char *>(scratch) + 2 * arr;
    size_t temp_bytes = sort_temp_bytes(n, end_bit);
    hipLaunchKernelGGL(order_pack_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, st, keys, M, n_images, a, valid);
    int rc = check_launch();
    if (rc) return rc;
    const hipError_t e = rocprim::radix_sort_keys_desc(temp, temp_bytes, a, b, n, 0, end_bit, st);
    if (e != hipSuccess) { set_last_error(e);  return WSSDL_ERR_LAUNCH; }
    hipLaunchKernelGGL(order_finish_kernel, dim3(cdiv(topn, 256), n_images), dim3(256), 0, st, b, M, topn, valid,
                       sorted_index, n_sorted);
    return check_launch();
}

}  // namespace wssdl
