# Per-call time of blp_rank_all on small query blocks (the reference's own eval batches: 64 triples = 128
# queries against the FB15k-237 table) under the exact f32 kernels and the fixed-point pre-pass.
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops
import bench
cfg = bench.WORKLOADS["fb15k237-transe"]
dev = torch.device("cuda", 0)
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
T = heads.shape[0]
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for t in (4, 16, 32, 64, 96, 128, 256, 512):
    qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
    tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
    out = torch.empty((2 * t, 4), dtype=torch.int32, device=dev)
    row = []
    for env in ({"rank_kernel": 1}, {"sad_min_queries": 1}, {}):
        _lib.reset_knobs()
        for k, v in env.items(): _lib.set_knob(k, v)
        row.append(timeit(lambda: ops.rank_all("transe", table, qf, qr, t, true_row=tr, out=out)))
    print(f"{2 * t:5d} queries: exact f32 {row[0]:7.1f} us   pre-pass {row[1]:7.1f} us   default {row[2]:7.1f} us   "
          f"(whole test set at this block size: {row[2] * T / t / 1e3:7.1f} ms)")
cfg = bench.WORKLOADS["fb15k237-distmult"]
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
_lib.reset_knobs()
for t in (4, 16, 32, 64, 128, 512):
    qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
    tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
    out = torch.empty((2 * t, 4), dtype=torch.int32, device=dev)
    us = timeit(lambda: ops.rank_all("distmult", table, qf, qr, t, true_row=tr, out=out))
    print(f"distmult {2 * t:5d} queries: {us:7.1f} us   (whole test set at this block size: {us * T / t / 1e3:7.1f} ms)")
