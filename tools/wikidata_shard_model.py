# Per-rank time of candidate-axis shards of the Wikidata5M-scale table on ONE GPU (no exchange): the
# reference's eval batch (2 triples = 4 queries per table pass, 64 passes per step) and the whole test set
# as one block, against 1/W of the rows.  What the exchange may cost on top: one all-gather of (Q, 4) int32
# per step.
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops
import bench
dev = torch.device("cuda", 0)
for name in ("wikidata5m-transe", "wikidata5m-transe-block"):
    cfg = bench.WORKLOADS[name]
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    Tall, N, blk = heads.shape[0], table.shape[0], min(cfg["block"], heads.shape[0])
    passes = (Tall + blk - 1) // blk
    q_true_all = table[true_row].contiguous()
    blocks = []
    for i in range(passes):  # (q_fixed, q_rel, q_true) of block i: its head-replacing queries, then its tail-replacing ones
        sl = torch.cat((torch.arange(i * blk, min((i + 1) * blk, Tall)), Tall + torch.arange(i * blk, min((i + 1) * blk, Tall)))).to(dev)
        blocks.append((q_fixed[sl].contiguous(), q_rel[sl].contiguous(), q_true_all[sl].contiguous()))
    T = blk
    out = torch.empty((passes, 2 * T, 4), dtype=torch.int32, device=dev)
    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    base = None
    for W in (1, 2, 4, 8):
        shard = table[: (N + W - 1) // W]
        def step():
            for i, (qf, qr, qt) in enumerate(blocks):
                ops.rank_all("transe", shard, qf, qr, qf.shape[0] // 2, q_true=qt, out=out[i, : qf.shape[0]])
        ms = timeit(step)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        ms_graph = timeit(graph.replay)
        base = base or (ms, ms_graph)
        print(f"{name} W={W}: eager {ms:8.3f} ms per step ({ms / passes * 1e3:7.1f} us per table pass) {base[0] / ms:.2f}x   "
              f"hipGraph replay {ms_graph:8.3f} ms ({ms_graph / passes * 1e3:7.1f} us per pass) {base[1] / ms_graph:.2f}x", flush=True)
        del graph
    del table, q_true_all, blocks
    torch.cuda.empty_cache()
