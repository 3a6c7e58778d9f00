"""A/B timing of the bilinear pre-pass on the FB15k-237 block (105 740 queries x 14 541 candidates):
    python tools/gemm_ab.py [path/to/libblp_hip.<variant>.so] [models...]
Loads the given build of the library (default: the product library), checks the counts of one call against the
exact f32 kernels, then times the ranking pass (HIP events recorded by the library around pre-pass + refinement)
and the whole call.  Variant libraries: python -c "from blp_amd import build; build.build(variant='noasm',
variant_flags=['-DBLP_GEMM_ASM_STAGE=0'])"; a -DBLP_TIMING build also prints where a wave's cycles go."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blp_amd import _lib  # noqa: E402

args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(args.pop(0))
from blp_amd import ops  # noqa: E402
import bench  # noqa: E402

models = args or ["distmult", "complex"]
dev = torch.device("cuda", 0)
events = bench.HipEvents()
L = _lib.lib()
print("library:", _lib.LIB_PATH)
for model in models:
    cfg = bench.WORKLOADS[f"fb15k237-{model}"]
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    T = heads.shape[0]
    got = ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row)
    _lib.set_knob("rank_kernel", 1)
    want = ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row)
    _lib.reset_knobs()
    same = torch.equal(got, want)
    out = torch.empty_like(got)
    for _ in range(3):
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
    pairs = []
    for _ in range(20):
        a, b = events.pair()
        _lib.check(L.blp_profile_next_rank_kernel(a, b), "hook")
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
        pairs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(events.elapsed_ms(a, b) for a, b in pairs)
    t0 = time.perf_counter()
    for _ in range(20):
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
    torch.cuda.synchronize()
    call = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{model:9s} rank pass median {ms[len(ms) // 2]:.3f} ms (min {ms[0]:.3f}, max {ms[-1]:.3f}); whole call {call:.3f} ms; "
          f"counts == exact kernels: {same}", flush=True)
    if hasattr(L, "blp_debug_read_timing"):
        buf = (ctypes.c_ulonglong * 8)()
        L.blp_debug_read_timing(buf)
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
        torch.cuda.synchronize()
        L.blp_debug_read_timing(buf)
        names = ["outside stages (prologue, first tile)", "stage: MFMAs of t+1 + decision of t", "stage: settle (entries / flags / counters)",
                 "stage: wait for the LDS-DMA of t+2", "stage: barrier", "epilogue (flush counters, entries)"]
        total = sum(buf[i] for i in range(6))
        print(f"  {buf[7]} waves, {total / max(buf[7], 1):.0f} ticks per wave")
        for i, n in enumerate(names):
            print(f"    {n:46s} {buf[i] / max(buf[7], 1):10.0f} ticks/wave  {100.0 * buf[i] / max(total, 1):5.1f} %")
