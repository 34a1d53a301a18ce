#!/usr/bin/env python3
"""Build recipe for ``oracle/_ref/`` -- the reference's own Cython kernels, compiled
from the sources where they lie under ``/root/reference`` (TEST INFRASTRUCTURE ONLY).

What is built (all outputs land in ``oracle/_ref/``, which is git-ignored but
travels to the GPU box; intermediate ``.c`` files go to a temp dir and are
deleted, so no reference source text is ever kept in this repo):

  cython_bbox.so     <- code/lib/utils/bbox.pyx     (bbox_overlaps,    unmodified)
  cython_bbox_ui.so  <- code/lib/utils/bbox_ui.pyx  (bbox_overlaps_ui, unmodified)
  cpu_nms.so         <- code/lib/nms/cpu_nms.pyx    (cpu_nms; dtype-alias patch only)
  cython_nms.so      <- code/lib/utils/nms.pyx      (nms, nms_new: the file fast_rcnn/test_bus.py:10 imports for
                                                     the per-class NMS of the test path; same alias patch)

``cpu_nms.pyx`` and ``utils/nms.pyx`` name NumPy aliases that NumPy 2 no longer has.  The recipe pipes
the file through three textual alias substitutions on its way to Cython (never
written back, never stored here): ``np.int_t -> np.intp_t``, ``dtype=np.int ->
dtype=np.intp`` and the argument annotation ``np.float thresh -> thresh`` (an
untyped Python object, which is what the vendored ``cpu_nms.c`` has: it
type-tests ``PyFloat_Type`` and compares through ``PyObject_RichCompare``,
``code/lib/nms/cpu_nms.c:1715,2495``).  No arithmetic line is touched.

``bbox.pyx``/``bbox_ui.pyx`` evaluate ``DTYPE = np.float`` at import; the importer
(``oracle/ref_kernels.py``) sets ``numpy.float = float`` first.

NOT buildable here (stated in DESIGN.md): ``roi_pooling_op.cc`` /
``roi_pooling_op_gpu.cu.cc`` need TensorFlow core headers, Eigen and
``libtensorflow_framework`` -- absent, and we do not write stand-ins.

The recipe is a no-op (keeps prebuilt files) when ``/root/reference`` is absent,
which is the situation on the GPU box.
"""
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("WSSDL_REFERENCE", "/root/reference")
LIB = os.path.join(REF, "code", "lib")

TARGETS = [
    # (module name, pyx path relative to code/lib, needs alias patch)
    ("cython_bbox", "utils/bbox.pyx", False),
    ("cython_bbox_ui", "utils/bbox_ui.pyx", False),
    ("cpu_nms", "nms/cpu_nms.pyx", True),
    ("cython_nms", "utils/nms.pyx", True),
]

_ALIAS_SUBS = [
    (r"np\.int_t", "np.intp_t"),
    (r"dtype=np\.int\b", "dtype=np.intp"),
    (r"np\.float thresh", "thresh"),
]


def _so_path(mod):
    return os.path.join(OUT, mod + ".so")


def have_ref_build():
    return all(os.path.exists(_so_path(m)) for m, _, _ in TARGETS)


def build(force=False, verbose=True):
    if not os.path.isdir(LIB):
        if verbose:
            print("[oracle/_ref] %s absent: keeping prebuilt files (%s)"
                  % (REF, "complete" if have_ref_build() else "none"))
        return have_ref_build()
    os.makedirs(OUT, exist_ok=True)
    import numpy as np
    inc_py = sysconfig.get_paths()["include"]
    inc_np = np.get_include()
    for mod, rel, patch in TARGETS:
        src = os.path.join(LIB, rel)
        dst = _so_path(mod)
        if (not force and os.path.exists(dst)
                and os.path.getmtime(dst) >= os.path.getmtime(src)
                and os.path.getmtime(dst) >= os.path.getmtime(__file__)):
            continue
        tmp = tempfile.mkdtemp(prefix="wssdl_ref_")
        try:
            pyx = os.path.join(tmp, mod + ".pyx")
            with open(src) as f:
                text = f.read()
            if patch:
                for pat, rep in _ALIAS_SUBS:
                    text = re.sub(pat, rep, text)
            with open(pyx, "w") as f:
                f.write(text)
            c = os.path.join(tmp, mod + ".c")
            subprocess.check_call(
                # language level 2: the .pyx files are Python-2-era sources
                [sys.executable, "-m", "cython", "-2",
                 "--module-name", mod, pyx, "-o", c],
                stdout=subprocess.DEVNULL if not verbose else None,
                stderr=subprocess.DEVNULL)
            subprocess.check_call(
                ["gcc", "-O2", "-fPIC", "-shared", "-fno-strict-aliasing",
                 "-Wno-cpp", "-Wno-unused-function",
                 "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                 "-I", inc_py, "-I", inc_np, c, "-o", dst])
            if verbose:
                print("[oracle/_ref] built", dst)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    sys.exit(0 if ok else 1)
