# A few hundred queries against the Wikidata5M-scale table (one candidate slab, thousands of flag words per
# query): per-call time and the kernel timeline, to keep the flag sweep honest in that corner.
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops
import bench
dev = torch.device("cuda", 0)
cfg = bench.WORKLOADS["wikidata5m-transe-block"]
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
T = heads.shape[0]
for t in (64, 128, 512):
    qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
    tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
    for _ in range(2): ops.rank_all("transe", table, qf, qr, t, true_row=tr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.rank_all("transe", table, qf, qr, t, true_row=tr)
    torch.cuda.synchronize()
    print(f"{2 * t} queries x 4.6 M candidates: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call")
