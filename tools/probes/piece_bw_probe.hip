// Probe: HBM read rate as a function of the size of a contiguous piece, for the access pattern of the RoI-pool
// backward walk: every wave reads a stream of pieces at addresses that are random at DRAM-page scale inside a
// 1.7 GB buffer (top_diff at R = 8512, C = 1024), DEPTH x 8 pieces in flight, 64 B (one code load) ... 1024 B.
//   piece 256 B  = one dword per lane        (a 64-channel wave of the walk)
//   piece 512 B  = one dwordx2 per lane      (the 128-channel wave: the default)
//   piece 1024 B = one dwordx4 per lane
// hipcc --offload-arch=gfx950 -O3 -w tools/probes/piece_bw_probe.hip -o /tmp/piece_bw_probe && /tmp/piece_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int DW>   // dwords per lane: 1, 2, 4
__global__ __launch_bounds__(64) void probe(const float *buf, const unsigned *offs /* piece index per (wave, step) */,
                                            int steps, float *out) {
    const int lane = threadIdx.x;
    const unsigned *mine = offs + (size_t)blockIdx.x * steps;
    float acc = 0.f;
    for (int s = 0; s < steps; s += 8) {
        float v[8][DW];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned o = __builtin_amdgcn_readfirstlane(mine[s + j]);          // scalar piece index
            const float *p = buf + (size_t)o * (64 * DW) + lane * DW;
            if (DW == 1) v[j][0] = *p;
            if (DW == 2) { const float2v t = *reinterpret_cast<const float2v *>(p);  v[j][0] = t.x;  v[j][DW - 1] = t.y; }
            if (DW == 4) { const float4v t = *reinterpret_cast<const float4v *>(p);  v[j][0] = t.x;  v[j][1 % DW] = t.y;  v[j][2 % DW] = t.z;  v[j][3 % DW] = t.w; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < DW; ++k) acc += v[j][k];
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int DW>
static void run(const float *buf, size_t buf_bytes, int waves, int steps) {
    const size_t pieces = buf_bytes / (256 * DW);
    unsigned *h = (unsigned *)malloc(sizeof(unsigned) * (size_t)waves * steps);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < (size_t)waves * steps; ++i) {
        x ^= x << 13;  x ^= x >> 7;  x ^= x << 17;
        h[i] = (unsigned)(x % pieces);
    }
    unsigned *d;
    float *o;
    hipMalloc(&d, sizeof(unsigned) * (size_t)waves * steps);
    hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(unsigned) * (size_t)waves * steps, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<DW>, dim3(waves), dim3(64), 0, 0, buf, d, steps, o);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)waves * steps * 256.0 * DW;
    printf("piece %4d B, %6d waves x %5d pieces: %.3f ms  %.2f TB/s  (%.2f GB)\n", 256 * DW, waves, steps, best,
           bytes / best / 1e9, bytes / 1e9);
    hipFree(d);
    hipFree(o);
    free(h);
}

int main() {
    const size_t buf_bytes = (size_t)1708 << 20;          // ~ top_diff at R = 8512, C = 1024
    float *buf;
    if (hipMalloc(&buf, buf_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, buf_bytes);
    // the same number of bytes (~2.1 GB) per run, 16 / 8 waves per CU
    for (int wpc : {8, 16}) {
        const int waves = 256 * wpc * 4;
        run<1>(buf, buf_bytes, waves, 2048 * 8 / wpc / 4 * 4);
        run<2>(buf, buf_bytes, waves, 1024 * 8 / wpc / 4 * 4);
        run<4>(buf, buf_bytes, waves, 512 * 8 / wpc / 4 * 4);
    }
    hipFree(buf);
    return 0;
}
