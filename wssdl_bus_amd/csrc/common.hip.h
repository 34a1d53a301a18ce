// Shared helpers for the gfx950 kernels of libwssdl_bus_hip.so.
// Built with -ffp-contract=off: parity with the reference depends on every
// multiply and add being rounded separately (no FMA contraction).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wssdl_bus_hip.h"

#define WSSDL_WAVE 64

namespace wssdl {

// thread-local record of the last HIP error, surfaced by wssdl_last_error()
void set_last_error(hipError_t e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(e);
        return WSSDL_ERR_LAUNCH;
    }
    return WSSDL_OK;
}

inline hipStream_t as_stream(wssdl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// base anchors travel to kernels by value (<= 32 x 4 doubles = 1 KiB of kernarg)
struct BaseAnchors {
    double v[WSSDL_MAX_ANCHORS][4];
};

inline int load_base_anchors(const double *host, int A, BaseAnchors *out) {
    if (!host || A < 1 || A > WSSDL_MAX_ANCHORS) return WSSDL_ERR_INVALID_ARGUMENT;
    for (int a = 0; a < A; ++a)
        for (int j = 0; j < 4; ++j) out->v[a][j] = host[a * 4 + j];
    return WSSDL_OK;
}

// Tuning knobs (wssdl_set_tuning): plain ints read at call time; nothing in the library reads the
// environment.  Defaults = automatic choices.
struct Tuning {
    int roi_bwd_plan = -1;      // plan id of the list-driven RoI-pool backward (-1: by launch size)
    int roi_bwd_owner = -1;     // owner plan id of the bin-owner backward that wssdl_roi_pool_backward_owner_plan suggests (-1: its rule)
    int roi_bwd_owner_segments = 0; // segments of the bin-owner backward that wssdl_roi_pool_backward_owner_segments suggests (0: its rule)
    int roi_fwd_one_bin = 7;    // small forward launches: waves per bin row (7 = one bin each, 4), 0 = the sliced kernel; + 100: any launch
    int roi_fwd_variant = 0;    // shape of the compact RoI-pool forward (0: automatic)
    int roi_fwd_blocks = -1;    // block-table forward (roi_pool_blocks.hip): -1 by launch shape, 0 never, 1 wherever supported
    int roi_fwd_blocks_parts = 0; // its pooling kernel: waves per bin row (1, 2, 4 or 7; anything else = 2, the measured best)
    int roi_fwd_blocks_sort = 1; // 0: its bin rows in RoI order instead of (image, first window row) order
    int roi_bwdc_variant = 0;   // shape of the tile-owner fallback backward
    int roi_bwd_cg = 0;         // channels per workgroup of the fallback backwards (0: automatic)
    int nms_one_pass = 0;       // 1: the proposal layer never uses the probe pass
    int topk_sort = 1;          // order of the proposal candidates: 1 sorted runs + cross ranks (order_sort.hip),
                                //    0 the select + sample sort of nms.hip
    int nms_fused = 1;          // 0: mask and sweep of a one-pass NMS as two launches instead of the fused one
    int nms_sparse = 16;        // fused launch: far column segments of a row block the sweep has resolved are computed for its kept rows
                                //    only when it kept at most this many of its 64 boxes (0: never)
    int nms_sweep_async = 0;    // 1: the NMS sweep's roles run without the per-chunk workgroup barrier (nms_sweep_async_block; measured slower: EXPERIMENTS.md), 0: the barrier version
    int nms_wait_us = 50000;    // fused launch: how long a sweep waits for a column segment of the mask before it reports WSSDL_NMS_TIMED_OUT
    int nms_fused_fault = 0;    // fault injection (tests): > 0 = the fused launch withholds image 0's segment counts
                                //    and its sweep gives up after this many microseconds -> roi count -1
};
Tuning &tuning();

// log(sum_k exp(s_k)) - max_k s_k as log1p of the sum WITHOUT the maximum's own term (which is exactly 1): for a
// confident prediction the sum is 1 + tiny and logf(1 + tiny) loses the low bits of tiny once the sum is rounded to
// f32 -- a cross-entropy of 0.003 came out 1.5e-5 (relative) away from its f64 value, outside north_star's 1e-5
// (tools/mil_fuzz.py found it; round 4).  *m_out = the maximum.
__device__ __forceinline__ float lse_minus_max(const float *__restrict__ s, int K, float *m_out) {
    float m = s[0];
    int km = 0;
    for (int k = 1; k < K; ++k)
        if (s[k] > m) { m = s[k];  km = k; }
    float z1 = 0.0f;
    for (int k = 0; k < K; ++k)
        if (k != km) z1 += expf(s[k] - m);
    *m_out = m;
    return log1pf(z1);
}

// workspace carving: 256-byte aligned slices
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base(static_cast<char *>(p)), off(0) {}
    template <typename T>
    T *take(size_t n) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += (n * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};

}  // namespace wssdl
