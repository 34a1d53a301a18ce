"""The NMS sweep's helper waves keep two batches of in-flight gather loads in sixteen FIXED registers
(v80-v95, wssdl_bus_amd/csrc/nms.hip) that only their inline-asm statements may name.  The check itself lives in
wssdl_bus_amd/isa_check.py and is part of build() since round 5 (a library that violates it is never installed; missing
LLVM tools are an error); these tests run it on the built library and on the one the GPU process has loaded, and show
that build() really refuses."""
import os

import pytest

from wssdl_bus_amd import isa_check


def test_only_the_helpers_asm_names_the_reserved_registers():
    from wssdl_bus_amd import build
    assert isa_check.check_library(build.build(verbose=False))


def test_build_refuses_without_the_disassembler(monkeypatch, tmp_path):
    """fail, not skip: without llvm-objdump & co. the check raises, and build() runs it before installing."""
    from wssdl_bus_amd import build
    monkeypatch.setattr(isa_check, "_llvm_dirs", lambda: [str(tmp_path)])
    monkeypatch.setattr(isa_check.shutil, "which", lambda name: None)
    with pytest.raises(isa_check.IsaCheckError, match="not found"):
        isa_check.check_library(build.OUT)
    import inspect
    src = inspect.getsource(build.build)
    assert src.index("isa_check.check_library") < src.index("os.replace(tmp, OUT")


def test_a_violation_is_caught(monkeypatch):
    """the checker rejects a listing that touches v80-v95 outside the three allowed instruction shapes"""
    good = ["\tglobal_load_dwordx2 v[80:81], v[2:3], off"] * 8 + ["\tv_mov_b32_e32 v80, 0"] * 16 + \
           ["\tv_or3_b32 v4, v80, v82, v84"] * 8
    def both(lines):       # each kernel in its two instances (the barrier form and the form without it)
        return {"%s<%s>" % (k, b): list(lines) for k in isa_check.KERNELS for b in ("false", "true")}
    monkeypatch.setattr(isa_check, "kernel_listings", lambda lib, tmp: both(good))
    assert isa_check.check_library("unused")
    monkeypatch.setattr(isa_check, "kernel_listings", lambda lib, tmp: both(good + good))       # the turn duplicated whole
    assert isa_check.check_library("unused")
    bad = good + ["\tv_add_f32_e32 v85, v1, v2"]
    monkeypatch.setattr(isa_check, "kernel_listings", lambda lib, tmp: both(bad))
    with pytest.raises(isa_check.IsaCheckError, match="outside the helpers"):
        isa_check.check_library("unused")
    monkeypatch.setattr(isa_check, "kernel_listings",
                        lambda lib, tmp: both(good + ["\tbuffer_load_dwordx4 v[78:81], v2, s[4:7], 0 offen"]))   # a tuple across the boundary
    with pytest.raises(isa_check.IsaCheckError, match="reaches into"):
        isa_check.check_library("unused")
    monkeypatch.setattr(isa_check, "kernel_listings", lambda lib, tmp: both(good[1:]))          # a batch load went missing
    with pytest.raises(isa_check.IsaCheckError, match="multiples"):
        isa_check.check_library("unused")
    monkeypatch.setattr(isa_check, "kernel_listings",
                        lambda lib, tmp: {k: list(good) for k in isa_check.KERNELS})            # one instance each only
    with pytest.raises(isa_check.IsaCheckError, match="two instances"):
        isa_check.check_library("unused")


@pytest.mark.gpu
def test_loaded_library_keeps_the_reserved_registers_to_the_helpers():
    """The same check in the GPU suite, on the library file this process has actually loaded: the
    driver's round-end run then disassembles the very code object the GPU box executes."""
    from wssdl_bus_amd import _lib
    _lib.lib()
    assert isa_check.check_library(_lib.LIB_PATH)
