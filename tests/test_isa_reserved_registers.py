"""The NMS sweep's helper waves keep two batches of in-flight gather loads in sixteen FIXED registers
(v80-v95, wssdl_bus_amd/csrc/nms.hip) that only their inline-asm statements may name; both kernels carry
amdgpu_num_vgpr(80) so that the register allocator stays below them.  That attribute is a request, not a
guarantee: this test disassembles the two kernels out of the BUILT library and checks that every instruction
that mentions v80-v95 is one of those asm statements (a load into a register pair, the zeroing of the registers
before the first turn, or the OR that takes a landed batch out of them).

(The request is only honoured in a range: with the registers moved to v48-v63 / v64-v79 and amdgpu_num_vgpr(48) /
(64) the allocator of ROCm 7.2 ignored it and used them -- which is how this test earned its keep in round 4.)"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KERNELS = ("nms_sweep_pipelined_kernel", "nms_mask_sweep_fused_kernel")
RESERVED = re.compile(r"\bv(8[0-9]|9[0-5])\b|\bv\[(8[0-9]|9[0-5]):(8[0-9]|9[0-5])\]")


def _tool(name):
    path = os.path.join(LLVM, name)
    if not os.path.exists(path):
        pytest.skip("%s not available" % path)
    return path


def _kernel_listings(lib_path, tmp):
    """{kernel name: [instruction lines]} from the gfx950 code objects embedded in the library."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = {}
    for i, o in enumerate(starts):
        piece = os.path.join(tmp, "bundle%d.bin" % i)
        with open(piece, "wb") as f:
            f.write(data[o:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = os.path.join(tmp, "dev%d.co" % i)
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        dis = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True,
                             text=True).stdout
        for part in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = part.split("\n", 1)[0]
            for k in KERNELS:
                if k in head:
                    out[k] = part.split("\n")[1:]
    return out


def test_only_the_helpers_asm_names_the_reserved_registers():
    from wssdl_bus_amd import build
    _check(build.build(verbose=False))


@pytest.mark.gpu
def test_loaded_library_keeps_the_reserved_registers_to_the_helpers():
    """The same check in the GPU suite, on the library file this process has actually loaded: the
    driver's round-end run then disassembles the very code object the GPU box executes."""
    from wssdl_bus_amd import _lib
    _lib.lib()
    _check(_lib.LIB_PATH)


def _check(lib_path):
    tmp = tempfile.mkdtemp(prefix="wssdl_isa_")
    try:
        listings = _kernel_listings(lib_path, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert set(listings) == set(KERNELS), sorted(listings)
    for k, lines in listings.items():
        loads = zeroed = ors = 0
        for line in lines:
            if not RESERVED.search(line):
                continue
            words = line.replace(",", " ").split()
            op, args = words[0], words[1:]
            if op == "global_load_dwordx2" and RESERVED.fullmatch(args[0]) and not any(RESERVED.search(a) for a in args[1:]):
                loads += 1                      # a batch load: the reserved pair is the destination only
            elif op.startswith("v_mov_b32") and RESERVED.fullmatch(args[0]) and args[1] == "0":
                zeroed += 1                     # before the first turn: never-issued batches read as zero words
            elif op.startswith(("v_or3_b32", "v_or_b32")) and not RESERVED.search(args[0]) and \
                    any(RESERVED.fullmatch(a) for a in args[1:]):
                ors += 1                        # the consume step: the landed words are sources only
            else:
                raise AssertionError("%s: reserved register used outside the helpers' asm: %s" % (k, line.strip()))
        assert loads == 8 and zeroed == 16 and ors == 8, (k, loads, zeroed, ors)
