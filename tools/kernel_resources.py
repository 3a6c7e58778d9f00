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


def valu_sgpr_hazards(path, min_gap=2):
    """gfx950: a VALU instruction that reads an SGPR (pair) written by a VALU instruction needs `min_gap` wait states in
    between (any instruction is one; s_nop N is N + 1).  The compiler pads its own code; the hand-scheduled K-steps of
    rank_gemm.hip (inline asm: v_cmp -> SGPR pair -> v_addc_co) must space themselves.  Returns (writer, reader) pairs that
    are closer, straight-line code only."""
    bad, lines = [], disassembly(path)

    def split(line):
        body = line.split("//")[0].strip()
        mnem, _, ops = body.partition(" ")
        return mnem, [o.strip() for o in ops.split(",")] if ops else []

    def written(mnem, ops):  # SGPRs a VALU instruction writes
        if not mnem.startswith("v_") or not ops:
            return set()
        if mnem.startswith("v_cmp") and mnem.endswith("_e64"):
            return sgprs_of(ops[0])
        if mnem.startswith(("v_addc_co", "v_add_co", "v_subb_co", "v_sub_co", "v_subrev_co", "v_subbrev_co")) and mnem.endswith("_e64") and len(ops) > 1:
            return sgprs_of(ops[1])
        if mnem.startswith(("v_readlane", "v_readfirstlane")):
            return sgprs_of(ops[0])
        return set()

    for i, line in enumerate(lines):
        mnem, ops = split(line)
        w = written(mnem, ops)
        if not w:
            continue
        gap = 0
        for later in lines[i + 1:i + 1 + min_gap + 2]:
            lm, lo = split(later)
            if later.endswith(":") or lm.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) or gap >= min_gap:
                break
            if lm.startswith("v_"):
                dests = written(lm, lo)
                reads = set()
                dest_index = 0 if lm.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) else (1 if dests else -1)
                for k, o in enumerate(lo):
                    if k == dest_index:
                        continue  # a destination (an add-with-carry names its carry pair again, further on, as carry-in)
                    reads |= sgprs_of(o)
                if reads & w:
                    bad.append((line.split("//")[0].strip(), later.split("//")[0].strip(), gap))
                    break
            gap += int(lo[0], 0) + 1 if lm == "s_nop" and lo else 1
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
