// Probe: what the RoI-pool forward's STORES cost on their own.  The rows kernel (roi_pool_fwd_rows_kernel<4,4,7>)
// writes, per wave = (4 RoIs) x (256 channels), 4 x 49 bins x (1 KB of f32 top_data + 256 B of arg-max codes) =
// R x 49 x C x 5 B = 2.135 GB per launch at R = 8512, C = 1024 -- 96 % of the bytes the kernel has to move.  This
// kernel issues exactly those stores (same addresses, same order: RoI, bin row, bin) and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/fwd_store_probe.hip -o /tmp/fwd_store_probe && /tmp/fwd_store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float float4v __attribute__((ext_vector_type(4)));

template <bool CODES, int WPB>
__global__ __launch_bounds__(64 * WPB) void store_probe(float *top, unsigned *codes, int R, int C) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    const int slices = C / 256;
    const int r0 = (wave / slices) * 4, c0 = (wave % slices) * 256 + lane * 4;
    if (r0 >= R) return;
    for (int r = r0; r < r0 + 4 && r < R; ++r)
        for (int ph = 0; ph < 7; ++ph) {
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) {
                const size_t bin = ((size_t)r * 7 + ph) * 7 + pw;
                const float f = (float)(bin & 1023) + lane;
                const float4v v = {f, f + 1.f, f + 2.f, f + 3.f};
                *reinterpret_cast<float4v *>(top + bin * C + c0) = v;
                if (CODES) codes[(bin * C + c0) >> 2] = (unsigned)bin * 0x01010101u + lane;
            }
        }
}

template <bool CODES, int WPB>
static void run(float *top, unsigned *codes, int R, int C) {
    const int waves = (R + 3) / 4 * (C / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int reps = 8;
    for (int rep = 0; rep < reps + 1; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((store_probe<CODES, WPB>), dim3((waves + WPB - 1) / WPB), dim3(64 * WPB), 0, 0, top, codes, R, C);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) { sum += ms;  if (ms < best) best = ms; }
    }
    const double bytes = (double)R * 49 * C * (CODES ? 5.0 : 4.0);
    printf("%s, %d waves per workgroup: best %.3f ms  mean %.3f ms  %.2f TB/s at the mean  (%.3f GB)\n",
           CODES ? "f32 + 1-byte codes" : "f32 only          ", WPB, best, sum / reps, bytes / (sum / reps) / 1e9, bytes / 1e9);
}

int main() {
    const int R = 8512, C = 1024;
    float *top;
    unsigned *codes;
    if (hipMalloc(&top, (size_t)R * 49 * C * 4) != hipSuccess || hipMalloc(&codes, (size_t)R * 49 * C) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    run<true, 1>(top, codes, R, C);
    run<true, 4>(top, codes, R, C);
    run<false, 1>(top, codes, R, C);
    run<false, 4>(top, codes, R, C);
    // a plain memset of the same bytes for comparison
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipMemsetAsync(top, 0, (size_t)R * 49 * C * 4, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) sum += ms;
    }
    printf("hipMemsetAsync of the f32 tensor: mean %.3f ms  %.2f TB/s\n", sum / 4, (double)R * 49 * C * 4 / (sum / 4) / 1e9);
    return 0;
}
