"""Loader for ``oracle/_ref/*.so`` -- the reference's own Cython kernels compiled by
``oracle/build_ref.py`` (TEST INFRASTRUCTURE ONLY: imported by tests/, smoke() and
bench.py's cpu_baseline leg; never by the product path).

Exposes ``bbox_overlaps`` (code/lib/utils/bbox.pyx:15), ``bbox_overlaps_ui``
(code/lib/utils/bbox_ui.pyx:12), ``cpu_nms`` (code/lib/nms/cpu_nms.pyx:17) and the test path's
``nms`` / ``nms_new`` (code/lib/utils/nms.pyx:17,70), or ``None`` for each when the build is absent.
"""
import importlib.util
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _load(mod):
    path = os.path.join(_DIR, mod + ".so")
    if not os.path.exists(path):
        return None
    # bbox.pyx:12 / bbox_ui.pyx:9 evaluate ``DTYPE = np.float`` at import time.
    if not hasattr(np, "float"):
        np.float = float  # harness-side alias, see SURVEY.md section 8c
    spec = importlib.util.spec_from_file_location(mod, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_bbox = _load("cython_bbox")
_bbox_ui = _load("cython_bbox_ui")
_nms = _load("cpu_nms")
_cython_nms = _load("cython_nms")

bbox_overlaps = getattr(_bbox, "bbox_overlaps", None)
bbox_overlaps_ui = getattr(_bbox_ui, "bbox_overlaps_ui", None)
cpu_nms = getattr(_nms, "cpu_nms", None)
nms = getattr(_cython_nms, "nms", None)                # utils/nms.pyx:17
nms_new = getattr(_cython_nms, "nms_new", None)        # utils/nms.pyx:70


def available():
    return bbox_overlaps is not None and bbox_overlaps_ui is not None and cpu_nms is not None
