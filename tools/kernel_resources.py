"""Per-kernel resources of the built objects: scratch bytes, VGPRs, spills, SGPRs, LDS -- read from the gfx950 code
objects' metadata notes (llvm-objcopy --dump-section .hip_fatbin, clang-offload-bundler --unbundle, llvm-readelf --notes).
    python tools/kernel_resources.py [blp_amd/csrc/build | one .o] [substring filter]
tests/test_abi.py uses kernels_of() to assert that no kernel of the default routes spills."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    """The gfx950 code objects bundled in a host shared library / object (clang-offload-bundler)."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        target = "hipv4-amdgcn-amd-amdhsa--gfx950"
        fat, dst = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", path], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return out  # no device code in this object (api.cpp)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={target}",
                            f"--input={fat}", f"--output={dst}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(dst) and os.path.getsize(dst) > 0:
            out.append(open(dst, "rb").read())
    return out


def kernels_of(path):
    """{demangled-ish kernel name: {field: int}} from every .o under a build directory, or one file."""
    files = [path]
    if os.path.isdir(path):
        files = [os.path.join(path, f) for f in sorted(os.listdir(path)) if f.endswith(".o")]
    result = {}
    for f in files:
        for blob in code_objects(f):
            with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
                tmp.write(blob)
                tmp.flush()
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", tmp.name], capture_output=True, text=True).stdout
            for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, flags=re.S):
                body = m.group(0)
                name = re.search(r"\.name:\s+(\S+)", body)
                if not name:
                    continue
                fields = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_count|"
                                                           r"sgpr_spill_count|group_segment_fixed_size|agpr_count):\s+(\d+)", body)}
                result[name.group(1)] = fields
    return result


def disassembly(path):
    """llvm-objdump -d of the gfx950 code objects of one .o / .so: a list of instruction lines (labels included)."""
    lines = []
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(blob)
            tmp.flush()
            out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True).stdout
        lines += [l.strip() for l in out.splitlines() if l.strip()]
    return lines


def sgprs_of(text):
    """The scalar registers an instruction's operand text names."""
    regs = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(r) for r in re.findall(r"\bs(\d+)\b", text))
    return regs


def early_uses_of_scalar_loads(path):
    """Hand-issued scalar loads (rank_common.h: sload16 + sdrain, rank_stream.hip: sload16_pinned) are two asm statements:
    the request, and later the wait.  Between them the destination registers hold nothing yet -- and the compiler, which
    takes the request's result for available, is free to copy or spill them there.  Returns every instruction between an
    s_load_dwordx8 / x16 and the next full lgkmcnt wait (straight-line code only) that names one of its destination registers."""
    bad, lines = [], disassembly(path)
    for i, line in enumerate(lines):
        m = re.match(r"s_load_dwordx(?:8|16) s\[(\d+):(\d+)\]", line)
        if not m:
            continue
        dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
        for later in lines[i + 1:i + 400]:
            if later.endswith(":") or later.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if later.startswith("s_waitcnt") and ("lgkmcnt(0)" in later or later.strip() == "s_waitcnt 0"):
                break
            if sgprs_of(later.split(" ", 1)[1] if " " in later else "") & dest:
                bad.append((line, later))
                break
    return bad


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "blp_amd", "csrc", "build")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels_of(path)
    names = list(ks)
    for name, nice in zip(names, demangle(names)):
        if flt in nice:
            f = ks[name]
            print(f"{nice[:120]:120s} scratch {f.get('private_segment_fixed_size', 0):5d}  vgpr {f.get('vgpr_count', 0):3d} "
                  f"spill {f.get('vgpr_spill_count', 0):3d}  sgpr {f.get('sgpr_count', 0):3d}  lds {f.get('group_segment_fixed_size', 0)}")
