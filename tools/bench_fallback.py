# The fixed-point pre-pass's fallback (non-finite values or a degenerate range: every tile goes to the exact
# sweep) at the FB15k-237 block size: how slow is "still exact"?
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops
import bench
dev = torch.device("cuda", 0)
cfg = bench.WORKLOADS["fb15k237-transe"]
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
T = heads.shape[0]
def run(tab):
    ops.rank_all("transe", tab, q_fixed, q_rel, T, true_row=true_row); torch.cuda.synchronize()
    t0 = time.perf_counter(); c = ops.rank_all("transe", tab, q_fixed, q_rel, T, true_row=true_row); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, c
ms, base = run(table)
print(f"finite table: {ms:.2f} ms")
bad = table.clone(); bad[7, 3] = float("inf")
ms, c = run(bad)
print(f"one inf in the table (pre-pass off, every tile swept exactly): {ms:.2f} ms")
_lib.set_knob("rank_kernel", 1)
ms, c2 = run(bad)
print(f"same through the exact f32 kernel: {ms:.2f} ms; identical counts: {torch.equal(c, c2)}")
