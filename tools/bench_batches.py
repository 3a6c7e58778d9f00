"""What keeping the reference's batching costs, and what blp_rank_all_batches makes of it: the FB15k-237-shaped test set
(52 870 triples, eval_batch_size = 64 -> 827 batches; bench.py's synthetic workload) ranked raw + filtered
  (a) one blp_rank_all_idx call per batch  -- the loop of INTEGRATION.md 2, as in the reference's train.py:128-171;
  (b) all batches in one blp_rank_all_batches call (the same layout in, the same layout out);
  (c) the whole set as one block (blp_amd.ranking.rank_triples: what eval_link_prediction does).
    python tools/bench_batches.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from blp_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("fb15k237-transe", "fb15k237-distmult"):
    job = bench.Job(name, dev)
    model, table, rel_w, T, batch = job.cfg["model"], job.table, job.model.rel_emb.weight.detach(), job.T, 64
    qb = ops.build_queries(job.triples, job.ent2idx, table, rel_w, batch, index=job.index, gather=False)

    def per_batch():
        out = torch.empty((2 * T, 4), dtype=torch.int32, device=dev)
        for start in range(0, T, batch):
            b = min(batch, T - start)
            sl = slice(2 * start, 2 * (start + b))
            seg = qb.filter._replace(seg_lo=qb.filter.seg_lo[sl], seg_hi=qb.filter.seg_hi[sl], exclude=qb.filter.exclude[sl])
            ops.rank_all_idx(model, table, qb.fixed_row[sl], rel_w, qb.rel_ids[sl], b, qb.true_row[sl], filter=seg, out=out[sl])
        return out

    def batches():
        return ops.rank_all_batches(model, table, qb.fixed_row, rel_w, qb.rel_ids, qb.true_row, T, batch, filter=qb.filter)

    def one_block():
        return job.step(True)[1]

    def ms(fn, n):
        for _ in range(3):  # (one warm-up call left the first timed loop of a process 25 % slow: round 4, tools/per_batch_probe.py)
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out

    a, ca = ms(per_batch, 3)
    b, cb = ms(batches, 20)
    c, _ = ms(one_block, 20)
    assert torch.equal(ca, cb)
    print(f"{name}: {T} triples in batches of {batch}: one call per batch {a:.2f} ms | blp_rank_all_batches {b:.3f} ms | "
          f"rank_triples (one block, incl. its prelude and metric sums) {c:.3f} ms; counts identical")
