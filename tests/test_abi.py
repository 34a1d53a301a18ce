"""C-ABI checks that run without a GPU: the library builds/loads, exports exactly the
symbols include/wssdl_bus_hip.h declares, the ctypes table covers them, and the
host-side entry points work.  No kernel is launched here."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wssdl_bus_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    from wssdl_bus_amd import build
    return build.build(verbose=False)


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"WSSDL_API[^;(]*?\b(wssdl_\w+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for s in ("wssdl_roi_pool_forward", "wssdl_roi_pool_backward", "wssdl_nms",
              "wssdl_bbox_overlaps", "wssdl_bbox_overlaps_ui", "wssdl_proposal_layer",
              "wssdl_anchor_labels", "wssdl_anchor_targets", "wssdl_roi_gt_assign",
              "wssdl_roi_targets", "wssdl_generate_anchors_host", "wssdl_shifted_anchors"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = sorted(set(re.findall(r" T (wssdl_\w+)", out)))
    assert exported == declared_symbols()


def test_ctypes_table_matches_header(lib_path):
    from wssdl_bus_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_symbols()
    L = _lib.lib()                       # resolves every symbol or raises
    assert L.wssdl_version().decode().startswith("wssdl_bus_hip")
    assert L.wssdl_last_error() == b""


def test_argument_counts_match_header(lib_path):
    from wssdl_bus_amd import _lib
    text = open(HEADER).read()
    for name, (_, argtypes) in _lib.SYMBOLS.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args == "void" else len(re.sub(r"/\*.*?\*/", "", args, flags=re.S).split(","))
        assert n == len(argtypes), (name, n, len(argtypes))


def test_generate_anchors_host_matches_golden(lib_path):
    from conftest import load_golden
    from wssdl_bus_amd.rpn_msr.generate_anchors import generate_anchors
    g = load_golden("anchors")
    assert np.array_equal(generate_anchors(scales=np.array([8, 16, 32])), g["a_8_16_32"])
    assert np.array_equal(generate_anchors(scales=np.array([4, 8, 16, 32])), g["a_4_8_16_32"])
    assert np.array_equal(generate_anchors(), g["a_default"])


def test_workspace_queries_are_pure_host(lib_path):
    from wssdl_bus_amd import _lib
    L = _lib.lib()
    assert L.wssdl_nms_workspace_bytes(0) > 0
    n = 12000
    assert L.wssdl_nms_workspace_bytes(n) >= n * ((n + 63) // 64) * 8
    assert L.wssdl_proposal_workspace_bytes(8, 38, 63, 9, 12000) >= 8 * 12000 * 188 * 8
    assert L.wssdl_anchor_workspace_bytes(4) >= 4 * 64 * 8


def test_invalid_arguments_return_status_not_crash(lib_path):
    from wssdl_bus_amd import _lib
    L = _lib.lib()
    # R == 0 is a valid no-op; bad shapes give WSSDL_ERR_INVALID_ARGUMENT.  No launch happens.
    assert L.wssdl_roi_pool_forward(None, 1, 4, 4, 4, None, 0, 7, 7, 0.0625, 0, None, None, None) == 0
    assert L.wssdl_roi_pool_forward(None, 1, 0, 4, 4, None, 1, 7, 7, 0.0625, 0, None, None, None) == 1
    assert L.wssdl_roi_pool_forward(None, 1, 4, 4, 4, None, 1, 7, 7, 0.0625, 5, None, None, None) == 1
    assert L.wssdl_bbox_overlaps(None, 0, 4, None, 3, 4, None, None) == 0
    assert L.wssdl_bbox_overlaps(None, 5, 3, None, 3, 4, None, None) == 1
    with pytest.raises(_lib.HipCallError):
        _lib.check(1, "x")


def test_product_package_does_not_import_oracle():
    # the product path must never route through the oracle (or any CPU fallback)
    pkg = os.path.join(ROOT, "wssdl_bus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text, f


def test_library_does_not_read_the_environment(lib_path):
    """Tuning goes through wssdl_set_tuning; the default build has no getenv import and no
    lab-build (ablation / trace) switches in its sources."""
    out = subprocess.check_output(["nm", "-D", "--undefined-only", lib_path]).decode()
    assert not re.search(r"\b(secure_)?getenv\b", out)
    csrc = os.path.join(ROOT, "wssdl_bus_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(csrc, f)).read()
            assert "_ABLATE" not in text and "_TRACE" not in text, f
            assert "getenv" not in text, f


def test_tuning_knobs_are_plain_ints(lib_path):
    from wssdl_bus_amd import _lib
    L = _lib.lib()
    assert _lib.get_tuning("roi_bwd_plan") == -1
    with _lib.tuned(roi_bwd_plan=11, nms_one_pass=1):
        assert _lib.get_tuning("roi_bwd_plan") == 11 and _lib.get_tuning("nms_one_pass") == 1
    assert _lib.get_tuning("roi_bwd_plan") == -1 and _lib.get_tuning("nms_one_pass") == 0
    assert L.wssdl_set_tuning(b"no_such_knob", 1) == _lib.ERR_INVALID_ARGUMENT
    assert L.wssdl_roi_pool_backward_plan_count() >= 26
    off = L.wssdl_roi_pool_backward_status_offset(8512, 8, 38, 63, 7, 7)
    assert off % 256 == 0 and 0 < off < L.wssdl_roi_pool_backward_workspace_bytes(8512, 8, 38, 63, 7, 7)
