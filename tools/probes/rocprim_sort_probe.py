#!/usr/bin/env python3
"""How long does rocPRIM's device radix sort take on the proposal layer's keys?  (A candidate replacement for
the single-workgroup-per-image top-K + ranking: 0.10 ms at 8 x 21546 keys, 0.20 ms at 1 x 56700.)
Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared tools/probes/rocprim_sort_probe.hip -o tools/probes/rocprim_sort_probe.so"""
import ctypes
import os
import torch

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocprim_sort_probe.so"))
L.sort_temp_bytes.restype = ctypes.c_size_t
L.sort_temp_bytes.argtypes = [ctypes.c_size_t]
L.sort_keys.restype = ctypes.c_int
L.sort_keys.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
for n_img, M in ((8, 21546), (2, 21546), (1, 56700), (8, 56700)):
    n = n_img * M
    scores = torch.rand((n,), device="cuda")
    keys = (scores.view(torch.int32).to(torch.int64) << 16) | (torch.arange(n, device="cuda") % M)
    keys = keys | (torch.arange(n, device="cuda") // M << 48)
    out = torch.empty_like(keys)
    tb = L.sort_temp_bytes(n)
    temp = torch.empty((tb,), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: L.sort_keys(ctypes.c_void_p(temp.data_ptr()), tb, ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, st)
    assert run() == 0
    torch.cuda.synchronize()
    assert torch.equal(out, torch.sort(keys, descending=True).values)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        run()
    s.record()
    for _ in range(50):
        run()
    e.record()
    torch.cuda.synchronize()
    print("images %d x %d keys: %.4f ms per sort (temp %d KB)" % (n_img, M, s.elapsed_time(e) / 50, tb // 1024))
