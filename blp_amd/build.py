"""Build libblp_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build(); also runnable
as ``python -m blp_amd.build``.  hipcc cross-compiles without a GPU.  The .so stays next to this
file (git-ignored, but it travels with the tree to the GPU box).

Two libraries come out of the same sources: libblp_hip.so, the product (no test hooks, no mutable process-wide state),
and libblp_hip.hooks.so (-DBLP_TEST_HOOKS: the same kernels + blp_debug_set_knob / blp_debug_gemm_dump), which only
tests/ and tools/ load (blp_amd._lib.set_knob)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libblp_hip.so")
HOOKS_VARIANT, HOOKS_FLAGS = "hooks", ("-DBLP_TEST_HOOKS",)
HOOKS_LIB = os.path.join(HERE, f"libblp_hip.{HOOKS_VARIANT}.so")
SOURCES = ["rank_all.hip", "rank_small.hip", "rank_stream.hip", "rank_stream16.hip", "rank_gemm.hip", "rank_sad.hip", "rank_sad_wide.hip", "rank_dense.hip", "score.hip", "inbatch_loss.hip", "project.hip", "bow.hip", "dkrl.hip", "queries.hip", "api.cpp"]
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "blp_hip.h")
# -ffp-contract=off: the kernels restate the reference's per-operation rounding; an FMA would change
# the last bit of a score and with it a rank.  No fast-math for the same reason.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


GLUE_SRC = os.path.join(CSRC, "torch_glue.cpp")
GLUE_LIB = os.path.join(HERE, "_torch_glue.so")


def build_glue(force=False, verbose=False):
    """Compile blp_amd/_torch_glue.so: the C++ autograd plumbing around the C-ABI's in-batch loss (torch_glue.cpp: no
    kernels, no HIP calls -- host compiler only, against the installed torch).  In-tree, so it travels with the snapshot."""
    import sysconfig

    import pybind11
    import torch
    from torch.utils import cpp_extension as ce
    if not force and os.path.exists(GLUE_LIB) and os.path.getmtime(GLUE_LIB) >= max(os.path.getmtime(GLUE_SRC), os.path.getmtime(__file__)):
        return GLUE_LIB
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler (g++) for blp_amd/_torch_glue.so")
    inc = [*ce.include_paths(), sysconfig.get_paths()["include"], pybind11.get_include()]
    libdir = ce.library_paths()[0]
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DTORCH_EXTENSION_NAME=_torch_glue",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *[f"-I{i}" for i in inc], GLUE_SRC, "-o", GLUE_LIB, f"-L{libdir}", "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python",
           f"-Wl,-rpath,{libdir}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return GLUE_LIB


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC); blp_amd needs the ROCm toolchain to build libblp_hip.so")


def _newest_header():
    # every header under csrc/ (a changed constant must rebuild its users) + the public C-ABI header
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [PUBLIC_HEADER]
    return max(os.path.getmtime(h) for h in headers)


def build(force=False, verbose=False, variant=None, variant_flags=()):
    """Compile and link libblp_hip.so.  ``variant`` (a name) + ``variant_flags`` build an experiment library
    libblp_hip.<variant>.so next to it from the same sources (own object directory; tools/ only)."""
    global_obj, global_lib = OBJ, LIB
    if variant:
        global_obj = os.path.join(HERE, "csrc", "build", variant)
        global_lib = os.path.join(HERE, f"libblp_hip.{variant}.so")
    return _build(force, verbose, global_obj, global_lib, list(variant_flags))


def build_hooks(force=False, verbose=False):
    """The test build: libblp_hip.hooks.so with the knobs and the dump hook compiled in."""
    return build(force, verbose, variant=HOOKS_VARIANT, variant_flags=HOOKS_FLAGS)


def _build(force, verbose, OBJ, LIB, variant_flags):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    dep_time = max(_newest_header(), os.path.getmtime(__file__))
    jobs = []
    for src in SOURCES:
        src_path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src_path), dep_time):
            extra = os.environ.get("BLP_EXTRA_HIPCC_FLAGS", "").split()  # experiments only
            cmd = [cc, *FLAGS, *extra, *variant_flags, "-x", "hip", "-c", src_path, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(4, min(8, os.cpu_count() or 4))) as pool:
            list(pool.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        run(cmd)
    return LIB


if __name__ == "__main__":
    build_glue(verbose=True)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_hooks(force="--force" in sys.argv, verbose=True))
