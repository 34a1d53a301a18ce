// Workgroup-wide k-th element of a set of unique 64-bit keys (MSB-first radix select, 8 bits per
// pass), shared by the top-K threshold of the proposal layer and the device fg/bg samplers.
//
// The naive version (8 sweeps over all items, one LDS atomic per item) is slow for two reasons
// measured on MI355X: (1) float-score keys concentrate in 2-3 values of the top digit, so the
// 64 lanes of a wave hit the same LDS word and the atomics serialise; (2) when the key is a hash
// it is recomputed in every sweep.  Here
//   * each sweep first merges equal digits inside the wave (up to 3 ballot rounds, the lane
//     that owns the digit adds the whole count), leftovers use a plain atomic;
//   * as soon as the bucket that holds the k-th key has <= LIST members, one more sweep gathers
//     them into LDS and the remaining digits are resolved on that list.
#pragma once
#include "common.hip.h"

namespace wssdl {

constexpr int SELECT_BATCH = 8;

template <int LIST>
struct SelectScratch {
    int hist[256];
    int sel[4];
    int wsum[4];
    int fill;
    unsigned long long result;
    unsigned long long list[LIST];
};

__device__ __forceinline__ void select_hist_add(int *hist, bool valid, int digit) {
    // wave-aggregated histogram update
    unsigned long long active = __ballot(valid);
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int round = 0; round < 3 && active != 0ull; ++round) {
        const int leader = __ffsll((long long)active) - 1;
        const int d0 = __builtin_amdgcn_readlane(digit, leader);
        const unsigned long long same = __ballot(valid && digit == d0);
        if (lane == leader) atomicAdd(&hist[d0], __popcll(same));
        active &= ~same;
        valid = valid && digit != d0;
    }
    if (valid) atomicAdd(&hist[digit], 1);
}

// digit bucket that holds the want-th key (1-based) counting from the top (LARGEST) or bottom:
// a 256-wide scan by the first four waves (a single thread walking the histogram costs ~12 us
// per pass in LDS round trips -- that, not the sweeps, dominated the naive kernels).
// Called by every thread of the workgroup (contains barriers); BLOCK >= 256.
template <bool LARGEST>
__device__ __forceinline__ void select_pick(const int *hist, int want, int *sel, int *wsum) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int h = 0, idx = 0, inc = 0;
    if (t < 256) {
        idx = LARGEST ? 255 - t : t;
        h = hist[idx];
        inc = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
    }
    __syncthreads();
    if (t < 256) {
        int base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) base += (w < wave) ? wsum[w] : 0;
        inc += base;
        const int exc = inc - h;
        if (exc < want && want <= inc) {
            sel[0] = idx;
            sel[1] = want - exc;
            sel[2] = h;
        }
    }
    __syncthreads();
}

// key_at(i, key) -> bool: item i of [0, n_items) is a member and `key` its (unique) key.
// Returns the want-th largest / smallest member key; requires 1 <= want <= #members.
// All BLOCK threads must call it; `sc` is LDS.  The first BLOCK * IPT items are evaluated once
// (all loads in flight together) and kept in registers for every sweep; items beyond that are
// re-evaluated per sweep.
// want_of(members) -> the rank wanted given the member count (counted here, in the same sweep
// that fills the register cache); <= 0 or > members: nothing to select, returns 0.
template <int BLOCK, int LIST, bool LARGEST, int IPT = 24, typename KeyFn, typename WantFn>
__device__ unsigned long long block_radix_select(KeyFn key_at, int n_items, WantFn want_of,
                                                 SelectScratch<LIST> &sc, int *members_out = nullptr) {
    const int t = threadIdx.x;
    unsigned long long cv[IPT];
    bool cok[IPT];
    __syncthreads();
    if (t == 0) sc.fill = 0;
    __syncthreads();
    int mc = 0;
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = u * BLOCK + t;
        cv[u] = 0ull;
        cok[u] = i < n_items && key_at(i, cv[u]);
        mc += cok[u] ? 1 : 0;
    }
    for (int i0 = BLOCK * IPT; i0 < n_items; i0 += BLOCK) {
        unsigned long long v = 0ull;
        mc += (i0 + t < n_items && key_at(i0 + t, v)) ? 1 : 0;
    }
    {
        const unsigned long long any = __ballot(mc != 0);
        if (any != 0ull) {
            // wave sum, one atomic per wave
            int ws = mc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ws += __shfl_xor(ws, off, 64);
            if ((t & 63) == 0) atomicAdd(&sc.fill, ws);
        }
    }
    __syncthreads();
    const int members = sc.fill;
    if (members_out) *members_out = members;
    int want = want_of(members);
    __syncthreads();
    if (want <= 0 || want > members) return 0ull;
    unsigned long long prefix = 0ull, pmask = 0ull;
    int bucket = 0x7fffffff;
    int shift = 56;
    for (; shift >= 0 && bucket > LIST; shift -= 8) {
        __syncthreads();
        if (t < 256) sc.hist[t] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IPT; ++u)
            if (u * BLOCK < n_items)
                select_hist_add(sc.hist, cok[u] && (cv[u] & pmask) == prefix, (int)((cv[u] >> shift) & 0xff));
        for (int i0 = BLOCK * IPT; i0 < n_items; i0 += BLOCK) {
            const int i = i0 + t;
            unsigned long long v = 0ull;
            const bool ok = i < n_items && key_at(i, v) && (v & pmask) == prefix;
            select_hist_add(sc.hist, ok, (int)((v >> shift) & 0xff));
        }
        __syncthreads();
        select_pick<LARGEST>(sc.hist, want, sc.sel, sc.wsum);
        prefix |= (unsigned long long)sc.sel[0] << shift;
        pmask |= 0xffull << shift;
        want = sc.sel[1];
        bucket = sc.sel[2];
    }
    if (shift < 0) return prefix;
    // gather the bucket (<= LIST keys) into LDS; the answer is the key with exactly want-1
    // members before it, found by counting
    __syncthreads();
    if (t == 0) sc.fill = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IPT; ++u)
        if (cok[u] && (cv[u] & pmask) == prefix) sc.list[atomicAdd(&sc.fill, 1)] = cv[u];
    for (int i0 = BLOCK * IPT; i0 < n_items; i0 += BLOCK) {
        const int i = i0 + t;
        unsigned long long v = 0ull;
        if (i < n_items && key_at(i, v) && (v & pmask) == prefix) sc.list[atomicAdd(&sc.fill, 1)] = v;
    }
    __syncthreads();
    const int nl = sc.fill;
    for (int i = t; i < nl; i += BLOCK) {
        const unsigned long long mine = sc.list[i];
        int before = 0;
        for (int j = 0; j < nl; ++j) {
            const unsigned long long o = sc.list[j];
            before += (LARGEST ? (o > mine) : (o < mine)) ? 1 : 0;
        }
        if (before == want - 1) sc.result = mine;
    }
    __syncthreads();
    return sc.result;
}

}  // namespace wssdl
