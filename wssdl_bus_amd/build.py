"""Build libwssdl_bus_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m wssdl_bus_amd.build [--force]

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off   no FMA contraction: IoU / NMS / decode arithmetic must round
                      every multiply and add separately, like the reference's C.
The output lives next to the sources (git-ignored; it travels to the GPU box).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("WSSDL_BUS_HIP_LIB") or os.path.join(HERE, "libwssdl_bus_hip.so")
SOURCES = ["api_common.hip", "bbox_overlaps.hip", "nms.hip", "proposal.hip", "anchor_target.hip",
           "roi_targets.hip", "roi_pool.hip", "roi_pool_compact.hip", "roi_pool_blocks.hip", "roi_pool_walk.hip", "mil.hip", "image.hip", "loss.hip",
           "post_detect.hip", "order_sort.hip"]
HEADERS = ["common.hip.h", "nms.hip.h", "select.hip.h", "order_sort.hip.h", "roi_pool.hip.h", os.path.join("..", "..", "include", "wssdl_bus_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "isa_check.py")]
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for q in paths:
        with open(q, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _compile_one(args):
    src, obj, flags, verbose = args
    tmp = "%s.%d.tmp" % (obj, os.getpid())          # (several ranks may build at import time: no shared temporary names)
    cmd = [hipcc()] + flags + ["-c", src, "-o", tmp]
    if verbose:
        print("[wssdl_bus_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, obj)
    return obj


def compile_objects(sources, headers, flags, verbose=True, jobs=None):
    """One object per source under csrc/_obj/ (git- and gpurun-ignored), named by a digest of the source, the
    headers and the flags: only what changed is recompiled, and the files compile in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in flags if f != "-shared"]
    hdrs = [os.path.join(CSRC, h) for h in headers]
    todo, objs = [], []
    for s in sources:
        src = os.path.join(CSRC, s)
        tag = _digest([src] + hdrs, " ".join(cflags))
        obj = os.path.join(objdir, "%s.%s.o" % (os.path.basename(s), tag))
        objs.append(obj)
        if not os.path.exists(obj):
            for old in os.listdir(objdir):
                if old.startswith(os.path.basename(s) + "."):
                    os.remove(os.path.join(objdir, old))
            todo.append((src, obj, cflags, verbose))
    jobs = jobs or int(os.environ.get("WSSDL_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(_compile_one, todo))
    return objs


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    if force:
        import shutil
        shutil.rmtree(os.path.join(CSRC, "_obj"), ignore_errors=True)
    objs = compile_objects(SOURCES, HEADERS, FLAGS, verbose)
    tmp = "%s.%d.tmp" % (OUT, os.getpid())
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden"] + objs + ["-o", tmp]
    if verbose:
        print("[wssdl_bus_amd] " + " ".join(cmd))
    subprocess.check_call(cmd)
    # the NMS sweep's reserved registers (v80-v95): disassemble what was just linked; a library that violates the
    # rule -- or that cannot be checked -- is not installed (isa_check.py)
    from . import isa_check
    try:
        isa_check.check_library(tmp)
    except Exception:
        os.remove(tmp)
        raise
    os.replace(tmp, OUT)
    return OUT


# Second, separate library: kernels for the plumbing AROUND the hot path (fused row batch-norm
# of the per-RoI head).  Not part of the drop-in C ABI, so it does not share its header.
PLUMB_OUT = os.path.join(HERE, "libwssdl_plumbing_hip.so")
PLUMB_SOURCES = [os.path.join("plumbing", "rowbn.hip"), os.path.join("plumbing", "im2col.hip")]
PLUMB_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wall"]


def build_plumbing(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in PLUMB_SOURCES]
    if not force and os.path.exists(PLUMB_OUT) and \
            all(os.path.getmtime(f) <= os.path.getmtime(PLUMB_OUT) for f in srcs + [os.path.abspath(__file__)]):
        return PLUMB_OUT
    tmp = "%s.%d.tmp" % (PLUMB_OUT, os.getpid())
    cmd = [hipcc()] + PLUMB_FLAGS + srcs + ["-o", tmp]
    if verbose:
        print("[wssdl_bus_amd] " + " ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, PLUMB_OUT)
    return PLUMB_OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_plumbing(force="--force" in sys.argv)
    print(OUT)
    print(PLUMB_OUT)
