#!/usr/bin/env python3
"""Where an iteration of the pipelined NMS sweep goes: per wave (= per role) the cycles between barriers (work) and
inside them (wait), from a PROFILE BUILD of the library (-DWSSDL_SWEEP_PROFILE: two s_memtime stamps per iteration
and wave; the product build carries none of it).

    python3 tools/nms_sweep_profile.py --build          # here (hipcc cross-compiles): tools/probes/libwssdl_sweep_profile.so
    python3 tools/nms_sweep_profile.py [--images 8]     # on the GPU box

The role whose waves wait least at the barrier sets the iteration time.  Two regimes of the 8 x 12000 -> 2000 layer:
early stop (2000 kept by chunk ~60) and full walk (all 188 chunks, ~140 kept), fused launch and two launches."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILE_LIB = os.path.join(ROOT, "tools", "probes", "libwssdl_sweep_profile.so")

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--define", action="append", default=[], help="with --build / at run time: extra -D macro; names the library variant")
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bench", action="store_true",
                help="run bench.py's default workload for a few steps on the profile build and report the LAST step's walks")
args = ap.parse_args()
if args.define:
    PROFILE_LIB = PROFILE_LIB[:-3] + "_" + "_".join(d.lower() for d in args.define) + ".so"

if args.build:
    sys.path.insert(0, ROOT)
    from wssdl_bus_amd import build as b
    cmd = [b.hipcc()] + b.FLAGS + ["-DWSSDL_SWEEP_PROFILE"] + ["-D" + d for d in args.define] + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", PROFILE_LIB]
    print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the helpers' reserved registers (v80-v95) must hold in THIS build too: the stamps add register pressure, and a
    # build whose allocator went above v79 corrupts in-flight addresses (round 6: such a build ended in a GPU memory
    # access fault) -- refuse it here, like wssdl_bus_amd.build does for the product
    from wssdl_bus_amd import isa_check
    try:
        isa_check.check_library(PROFILE_LIB)
    except isa_check.IsaCheckError as e:
        os.remove(PROFILE_LIB)
        sys.exit("profile build REFUSED: " + str(e))
    sys.exit(0)

os.environ["WSSDL_BUS_HIP_LIB"] = PROFILE_LIB          # read by wssdl_bus_amd._lib at import
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kernel_bench import synth_rpn, timeit  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded  # noqa: E402

L = _lib.lib()
raw = ctypes.CDLL(PROFILE_LIB)
raw.wssdl_debug_sweep_profile_read.restype = ctypes.c_int
raw.wssdl_debug_sweep_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ROLES = ["resolver", "scribe", "scribe", "stager", "stager", "spare"] + ["helper"] * 10


def read_profile():
    buf = np.zeros((64, 16, 32), dtype=np.uint64)
    rc = raw.wssdl_debug_sweep_profile_read(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf


def report_async(p, img, head):
    """the form without the barrier: per wave the cycles of its loop and, of those, the cycles inside each of its waits"""
    chunks = int(p[img, 0, 2])
    rt_us = float(p[img, 0, 3]) / 100.0
    head.update(chunks=chunks, resolver_loop_us=round(rt_us, 1), us_per_chunk=round(rt_us / max(chunks, 1), 3), form="no barrier")
    print(json.dumps(head))
    roles = ["resolver", "scribe", "scribe", "stager", "stager", "stager"] + ["helper"] * 10
    sites = {"resolver": ("staged rows", "helpers' words", "-"), "scribe": ("chunk resolved", "rows staged (fused)", "-"),
             "stager": ("slot free", "-", "-"), "helper": ("chunks expanded", "column stored", "batch landed")}
    for w in range(16):
        tot = float(p[img, w, 0])
        a, b, c = float(p[img, w, 1]), float(p[img, w, 4]), float(p[img, w, 5])
        n = max(chunks, 1)
        print("    wave %2d %-8s loop %6.0f cycles per chunk: waits %5.0f (%s) + %5.0f (%s) + %5.0f (%s) = %3.0f %% of the loop; own work %5.0f" % (
            w, roles[w], tot / n, a / n, sites[roles[w]][0], b / n, sites[roles[w]][1], c / n, sites[roles[w]][2],
            100.0 * (a + b + c) / max(tot, 1.0), (tot - a - b - c) / n))


def report(p, img, head):
    if int(p[img, 0, 6]) == 0xA51C:
        return report_async(p, img, head)
    iters = int(p[img, 0, 2])
    rt_us = float(p[img, 0, 3]) / 100.0
    cyc = float(p[img, 0, 0] + p[img, 0, 1])
    head.update(chunks=iters, walk_us=round(rt_us, 1), us_per_chunk=round(rt_us / max(iters, 1), 3),
                cycles_per_chunk=round(cyc / max(iters, 1)), mhz=round(cyc / max(rt_us, 1e-9)))
    print(json.dumps(head))
    nb = (iters + 15) // 16
    for w in (0, 1, 3, 6, 15):
        print("    work per chunk by 16-chunk bucket, wave %2d %-8s: " % (w, ROLES[w]) + " ".join(
            "%5d" % (int(p[img, w, 8 + k]) // min(16, iters - 16 * k)) for k in range(nb)))
    for w in range(16):
        work, wait, mid = float(p[img, w, 0]), float(p[img, w, 1]), float(p[img, w, 4])
        turns = iters if ROLES[w] == "helper" else max(iters / 2.0, 1)
        print("    wave %2d %-8s work %6.0f  wait %6.0f cycles per chunk (%.0f %% waiting)   own loads %4.0f per turn; last at the barrier %3.0f %% of the chunks, longest turn %5d (chunk %3d)" % (
            w, ROLES[w], work / max(iters, 1), wait / max(iters, 1), 100.0 * wait / max(work + wait, 1.0), mid / turns,
            100.0 * float(p[img, w, 5]) / max(iters, 1), int(p[img, w, 6]), int(p[img, w, 7])))
        if ROLES[w] == "helper" and float(p[img, w, 28:32].sum()) > 0:
            h = [float(v) / max(iters, 1) for v in p[img, w, 28:32]]
            print("            helper turn by section, cycles per chunk: take + OR %4.0f, list entries %4.0f, summary words %4.0f, addresses + issue %4.0f" % tuple(h))


if args.bench:
    import bench  # noqa: E402
    sys.argv = ["bench.py", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--roofline-iters", "2"]
    bench.main()
    p = read_profile()
    for img in range(8):
        if img in (0, 7):
            report(p, img, dict(case="bench step (last)", image=img))
        else:
            print(json.dumps(dict(image=img, chunks=int(p[img, 0, 2]), walk_us=float(p[img, 0, 3]) / 100.0)))
    sys.exit(0)

N = args.images
info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
prob, pred0 = synth_rpn(N, 38, 63, 9, 3)
for name, scale, thresh in (("early stop", 1.0, 0.7), ("full walk", 0.3, 0.3)):
    cfg.TRAIN.RPN_NMS_THRESH = thresh
    pred = pred0 * scale
    for fused, asyn in ((1, 0), (1, 1), (0, 0), (0, 1)):
        with _lib.tuned(nms_fused=fused, nms_sweep_async=asyn):
            out = proposal_layer_padded(prob, pred, info, True)
            ms = timeit(lambda: proposal_layer_padded(prob, pred, info, True), args.iters, warmup=3)
            read_profile()                                  # (zeroes the counters)
            proposal_layer_padded(prob, pred, info, True)   # ONE call: the counters hold that walk
            p = read_profile()
        report(p, 0, dict(case=name, nms_fused=fused, nms_sweep_async=asyn, layer_ms=round(ms, 4), kept=int(out[1][0]), defines=args.define))
