#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
extern "C" size_t sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    unsigned long long *p = nullptr;
    rocprim::radix_sort_keys_desc(nullptr, bytes, p, p, n, 0, 52, (hipStream_t)0);
    return bytes;
}
extern "C" int sort_keys(void *temp, size_t bytes, const unsigned long long *in, unsigned long long *out, size_t n, hipStream_t st) {
    return (int)rocprim::radix_sort_keys_desc(temp, bytes, in, out, n, 0, 52, st);
}
