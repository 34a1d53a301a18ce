"""Build-time check of the NMS sweep's reserved registers (round 5: part of build(), not only a test).

The sweep's helper waves keep two batches of in-flight gather loads in sixteen FIXED registers (v80-v95,
csrc/nms.hip) that only their inline-asm statements may name; both kernels carry amdgpu_num_vgpr(80) so that the
register allocator stays below them.  That attribute is a request, not a guarantee (with the registers at v48-v63 /
v64-v79 the allocator of ROCm 7.2 ignored it), and a violation corrupts keep lists silently.  So the build disassembles
the two kernels out of the library it has just linked and REFUSES to install it unless every instruction that
mentions v80-v95 is one of those asm statements -- a load into a register pair, the zeroing before the first turn,
or the OR that takes a landed batch out.  Missing LLVM tools are an error, not a skip.
"""
import os
import re
import shutil
import subprocess
import tempfile

def _llvm_dirs():
    """Where the LLVM binutils of the ROCm install that builds the library live: next to the hipcc in use, under
    ROCM_PATH / HIP_PATH, then the default install."""
    out = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if hipcc and os.path.isabs(hipcc):
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        out += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    for var in ("ROCM_PATH", "HIP_PATH"):
        if os.environ.get(var):
            out += [os.path.join(os.environ[var], "lib", "llvm", "bin"), os.path.join(os.environ[var], "llvm", "bin")]
    out.append("/opt/rocm/lib/llvm/bin")
    return out


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KERNELS = ("nms_sweep_pipelined_kernel", "nms_mask_sweep_fused_kernel")
RESERVED = re.compile(r"\bv(8[0-9]|9[0-5])\b|\bv\[(8[0-9]|9[0-5]):(8[0-9]|9[0-5])\]")
ANY_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")


def _straddles(line):
    """a register range that reaches into v80-v95 from outside (v[78:81]: a tuple the allocator placed across the
    boundary) -- RESERVED only sees ranges that lie inside"""
    for a, b in ANY_RANGE.findall(line):
        a, b = int(a), int(b)
        if b >= 80 and a <= 95 and not (80 <= a <= 95 and 80 <= b <= 95):
            return True
    return False


class IsaCheckError(RuntimeError):
    pass


def _tool(name):
    for d in _llvm_dirs():
        path = os.path.join(d, name)
        if os.path.exists(path):
            return path
    found = shutil.which(name)
    if found:
        return found
    raise IsaCheckError("%s not found (looked in %s and on PATH): the reserved-register check of the NMS sweep cannot "
                        "run, and a library that has not passed it must not be installed" % (name, _llvm_dirs()))


def kernel_listings(lib_path, tmp):
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
                    # (each kernel exists in two instances since round 6: the barrier form and the form without it)
                    m = re.search(r"<([^>]+)>", head)
                    out[m.group(1) if m else head] = part.split("\n")[1:]
    return out


def check_library(lib_path):
    """Raises IsaCheckError unless every instance of the sweep kernels keeps v80-v95 to the helpers' asm statements."""
    tmp = tempfile.mkdtemp(prefix="wssdl_isa_")
    try:
        listings = kernel_listings(lib_path, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k in KERNELS:
        if sum(1 for name in listings if k in name) < 2:
            raise IsaCheckError("kernel %s: fewer than its two instances found in %s: have %s" % (k, lib_path, sorted(listings)))
    for k, lines in listings.items():
        loads = zeroed = ors = 0
        for line in lines:
            if _straddles(line):
                raise IsaCheckError("%s: a register range reaches into the reserved v80-v95: %s" % (k, line.strip()))
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
                raise IsaCheckError("%s: reserved register used outside the helpers' asm: %s" % (k, line.strip()))
        # per copy of the helpers' code the compiler emits: 8 batch loads, 16 zeroing moves, 8 ORs (it may duplicate
        # the turn, e.g. by unrolling: whole multiples are fine, anything else means a statement was split or dropped)
        if not (loads >= 8 and loads % 8 == 0 and zeroed >= 16 and zeroed % 16 == 0 and ors >= 8 and ors % 8 == 0):
            raise IsaCheckError("%s: expected multiples of 8 batch loads, 16 zeroing moves and 8 ORs on v80-v95, found "
                                "%d / %d / %d" % (k, loads, zeroed, ors))
    return True
