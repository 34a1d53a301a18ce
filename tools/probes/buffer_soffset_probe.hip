// Probe: does the range check of a raw buffer load (stride 0) include the scalar offset?
// Every access stays inside the 16 KiB allocation whatever the answer.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_soffset_probe.hip -o /tmp/soffset_probe && /tmp/soffset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const float *p, int num_bytes, int soff, int voff, float *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, num_bytes, 0x00020000);
    out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * (int)threadIdx.x, soff, 0));
}

int main() {
    const int n = 4096;                 // floats allocated and filled with 1 + index
    float *d, *o, h[64], src[n];
    for (int i = 0; i < n; ++i) src[i] = 1.0f + i;
    hipMalloc(&d, sizeof(src));
    hipMalloc(&o, sizeof(h));
    hipMemcpy(d, src, sizeof(src), hipMemcpyHostToDevice);
    const int rec = 1024 * 4;           // the descriptor covers the first 1024 floats only
    struct { int soff, voff; const char *what; } cases[] = {
        {0, 0, "in range"},
        {rec, 0, "soffset = num_records, voffset in range"},
        {0, rec, "voffset = num_records"},
        {rec - 128, 0, "soffset + voffset straddles the end (lanes 32..63 beyond)"},
    };
    for (auto &c : cases) {
        hipMemset(o, 0xff, sizeof(h));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rec, c.soff, c.voff, o);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: FAULT\n", c.what); return 1; }
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-60s lane0 %.1f lane31 %.1f lane32 %.1f lane63 %.1f\n", c.what, h[0], h[31], h[32], h[63]);
    }
    return 0;
}
