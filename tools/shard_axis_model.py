# Per-rank time of the two sharding axes on ONE GPU (no comm): candidate shard = all Q x N/W rows,
# query shard = Q/W queries x all N rows.  FB15k-237 shapes.
import sys, time, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops
import bench
for model in ("transe", "distmult"):
    cfg = bench.WORKLOADS["fb15k237-" + model]
    dev = torch.device("cuda", 0)
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    T = heads.shape[0]; N = table.shape[0]
    q_true = table[true_row].contiguous()
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    base = timeit(lambda: ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row))
    print(f"{model}: 1 GPU {base:.3f} ms")
    for W in (2, 4, 8):
        per = (N + W - 1) // W
        shard = table[:per]
        tc = timeit(lambda: ops.rank_all(model, shard, q_fixed, q_rel, T, q_true=q_true))
        t = (T + W - 1) // W
        qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])); qr = torch.cat((q_rel[:t], q_rel[T:T + t])); tr = torch.cat((true_row[:t], true_row[T:T + t]))
        tq = timeit(lambda: ops.rank_all(model, table, qf, qr, t, true_row=tr))
        print(f"  W={W}: candidate-shard {tc:.3f} ms ({base/tc:.2f}x)   query-shard {tq:.3f} ms ({base/tq:.2f}x)")
