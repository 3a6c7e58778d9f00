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

models = [a for a in args if a != "--sad"] or ["distmult", "complex"]
if "--sad" in args:  # the TransE pre-pass: which clock does it run at? (a -DBLP_TIMING build)
    models = ["transe"]
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
    if model == "transe" and hasattr(L, "blp_debug_read_sad_timing"):
        buf = (ctypes.c_ulonglong * 4)()
        L.blp_debug_read_sad_timing(buf)
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
        torch.cuda.synchronize()
        L.blp_debug_read_sad_timing(buf)
        print(f"  rank_sad_kernel: {buf[2]} waves, {buf[0] / max(buf[2], 1):.0f} shader-clock ticks and {buf[1] / max(buf[2], 1) / 100.0:.1f} us "
              f"per wave: {buf[0] / max(buf[1], 1) * 0.1:.2f} GHz while the kernel runs")
    if model != "transe" and hasattr(L, "blp_debug_read_timing"):
        buf = (ctypes.c_ulonglong * 8)()
        L.blp_debug_read_timing(buf)
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
        torch.cuda.synchronize()
        L.blp_debug_read_timing(buf)
        names = ["outside stages (prologue, first tile)", "stage: MFMAs of t+1 + decision of t", "stage: settle (entries / flags / counters)",
                 "stage: wait for the LDS-DMA of t+2", "stage: barrier", "epilogue (flush counters, entries)"]
        total = sum(buf[i] for i in range(6))
        wall = buf[6] / max(buf[7], 1) / 100.0  # us per wave (s_memrealtime: 100 MHz)
        print(f"  {buf[7]} waves, {total / max(buf[7], 1):.0f} shader-clock ticks and {wall:.1f} us per wave: "
              f"{total / max(buf[6], 1) * 0.1:.2f} GHz while the kernel runs")
        for i, n in enumerate(names):
            print(f"    {n:46s} {buf[i] / max(buf[7], 1):10.0f} ticks/wave  {100.0 * buf[i] / max(total, 1):5.1f} %")
        if hasattr(L, "blp_debug_read_trace"):  # the workgroups' lifetimes: how many run at a time, and where
            nb = min(int(buf[7]) // 4, 4096)
            tr = (ctypes.c_ulonglong * (3 * nb))()
            L.blp_debug_read_trace(tr, nb)
            import numpy as np
            a = np.frombuffer(tr, dtype=np.uint64).reshape(nb, 3)
            start, end = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
            hw, xcc = (a[:, 2] & 0xffffffff).astype(np.int64), (a[:, 2] >> 32).astype(np.int64) & 0xf
            t0, t1 = start.min(), end.max()
            span = (t1 - t0) / 100.0
            life = (end - start) / 100.0
            cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
            where = xcc * 1000 + se * 100 + sh * 10 + cu  # a label per CU
            print(f"  {nb} workgroups over {span:.1f} us; lifetime mean {life.mean():.1f} us (min {life.min():.1f}, max {life.max():.1f}); "
                  f"sum of lifetimes / span = {life.sum() / span:.0f} workgroups at a time on {len(np.unique(where))} distinct CU labels")
            grid = np.linspace(t0, t1, 41)
            conc = [int(((start <= g) & (end > g)).sum()) for g in grid[:-1]]
            print("  running at 40 instants:", conc)
            per_cu = {}
            for w, st, en in zip(where, start, end):
                per_cu.setdefault(int(w), []).append((int(st), int(en)))
            busy = np.array([sum(e - s_ for s_, e in v) / 100.0 for v in per_cu.values()])
            cnt = np.array([len(v) for v in per_cu.values()])
            print(f"  per CU: workgroups {cnt.min()} .. {cnt.max()} (mean {cnt.mean():.1f}); sum of lifetimes {busy.min():.0f} .. {busy.max():.0f} us (mean {busy.mean():.0f}); "
                  f"per XCC workgroups {np.bincount(xcc).tolist()}")
