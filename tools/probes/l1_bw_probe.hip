// Probe: vector-L1 hit bandwidth per CU for 16-byte-per-lane buffer loads, and the shader clock
// under that load.  Every wave re-reads the same 16 KiB (L1-resident) block.
// hipcc --offload-arch=gfx950 -O3 tools/probes/l1_bw_probe.hip -o /tmp/l1_bw_probe && /tmp/l1_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float float4v __attribute__((ext_vector_type(4)));

template <int VALU_PER_LOAD>
__global__ __launch_bounds__(256) void probe(const float *p, int iters, float *out, long long *clk) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 1 << 14, 0x00020000);
    const int voff = (threadIdx.x & 63) * 16;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int i = 0; i < iters; i += 4) {
        float4v v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, ((i + j) & 15) << 10, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < VALU_PER_LOAD; ++k) acc[k & 3] = fmaxf(acc[k & 3], v[j][k & 3] + (float)k);
        }
    }
    const long long t1 = clock64();
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int V>
static void run(const float *d, float *o, long long *c, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(wgs), dim3(256), 0, 0, d, 64, o, c);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<V>, dim3(wgs), dim3(256), 0, 0, d, iters, o, c);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cycles;
    hipMemcpy(&cycles, c, sizeof(cycles), hipMemcpyDeviceToHost);
    const double bytes = (double)wgs * 4 * iters * 1024.0;
    printf("valu/load %2d  wgs %5d: %.3f ms  %.1f TB/s  = %.1f B/clk/CU at 2.4 GHz; wave 0 of block 0: %lld clock64 ticks for %d loads "
           "(%.1f per load)\n", V, wgs, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, cycles, iters, (double)cycles / iters);
}

int main() {
    float *d, *o;
    long long *c;
    hipMalloc(&d, 1 << 14);
    hipMemset(d, 0, 1 << 14);
    hipMalloc(&o, 16);
    hipMalloc(&c, 16);
    run<1>(d, o, c, 256 * 8, 8192);
    run<4>(d, o, c, 256 * 8, 8192);
    run<12>(d, o, c, 256 * 8, 8192);
    run<1>(d, o, c, 256 * 2, 8192);
    return 0;
}
