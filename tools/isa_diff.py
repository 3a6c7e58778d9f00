"""Instruction-for-instruction comparison of one kernel between two builds of a .hip file (compiled to assembly with the
library's flags): `python tools/isa_diff.py old.hip new.hip <substring of the mangled kernel name>`.  Used in round 5 to settle
whether the bilinear pre-pass's 909 -> 947 us between profiles/r03 and profiles/r04 came from the code (it did not: identical
instruction streams; the profiles were taken on different boxes)."""
import difflib, os, re, subprocess, sys, tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fno-gpu-rdc",
         "-fno-slp-vectorize", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only"]


def kernel(asm, needle):
    m = re.search(r"^(\S*%s\S*):" % re.escape(needle), asm, re.M)
    if not m:
        raise SystemExit(f"no kernel matching {needle}")
    body = asm[m.end():asm.index(".Lfunc_end", m.end())]
    lines = [re.sub(r";.*$", "", l).strip() for l in body.splitlines()]
    lines = [re.sub(r"\.LBB\d+_", ".LBB_", l) for l in lines]  # (block labels carry the function's ordinal in the file)
    return m.group(1), [l for l in lines if l and not l.startswith(".")]


def main():
    old, new, needle = sys.argv[1:4]
    out = []
    for src in (old, new):
        with tempfile.NamedTemporaryFile(suffix=".s") as f:
            subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", f.name, src])
            out.append(kernel(open(f.name).read(), needle))
    (na, a), (nb, b) = out
    diff = list(difflib.unified_diff(a, b, lineterm="", n=0))
    print(f"{na}\n  {old}: {len(a)} instructions\n  {new}: {len(b)} instructions\n  identical: {a == b}")
    print("\n".join(diff[:60]))


if __name__ == "__main__":
    main()
