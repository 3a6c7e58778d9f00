# One 8-way query shard's share of the FB15k-237 evaluation on ONE GPU: eager launches vs hipGraph replay.
import sys, time, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops
import bench
for model in ("transe", "distmult"):
    cfg = bench.WORKLOADS["fb15k237-" + model]
    dev = torch.device("cuda", 0)
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    T = heads.shape[0]
    for W in (1, 8):
        t = (T + W - 1) // W
        qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
        tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
        out = torch.empty((2 * t, 4), dtype=torch.int32, device=dev)
        pick = torch.tensor([0, 2, 3, 4], device=dev)
        def work():
            ops.rank_all(model, table, qf, qr, t, true_row=tr, out=out)
            return ops.rank_metric_sums(out).index_select(0, pick)
        def timeit(fn, n=50):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        eager = timeit(work)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sums = work()
        replay = timeit(g.replay)
        # host time of the eager step alone (no GPU wait): how close the launch loop is to the GPU time
        t0 = time.perf_counter()
        for _ in range(50): work()
        host = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        print(f"{model} W={W}: eager {eager:.3f} ms  graph {replay:.3f} ms  host-side launch loop {host:.3f} ms")
