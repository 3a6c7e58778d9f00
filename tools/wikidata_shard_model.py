"""Per-rank time of candidate-axis shards of the Wikidata5M-scale table on ONE GPU (no exchange), through the calls
blp_amd.ranking.rank_triples makes on that axis (ops.gather_triple_vectors, ops.build_queries with the shard's row_base,
ops.rank_all_batches with one ranking pass per block, ops.rank_metric_sums): the reference's eval batch (2 triples = 4 queries per table pass,
64 passes per step), TransE and ComplEx, and the whole test set as one block -- against 1/W of the rows.  What a real run
adds on top: one all-reduce of the (2T, D) query vectors and one all-gather of (2T, 4) int32 counts per step
(bench.py --gpus N reports them as exchange_ms).  W = 1 is the unsharded evaluation (bench.Job.step).
    python tools/wikidata_shard_model.py [workload ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from blp_amd import ops, ranking  # noqa: E402

dev = torch.device("cuda", 0)
if os.environ.get("BLP_STREAM_KERNEL"):  # A/B: force a streaming kernel (hooks build; 3: workgroup tile, 4: ring)
    from blp_amd import _lib
    _lib.set_knob("stream_kernel", int(os.environ["BLP_STREAM_KERNEL"]))


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name in (sys.argv[1:] or ["wikidata5m-transe", "wikidata5m-complex", "wikidata5m-transe-block"]):
    job = bench.Job(name, dev)
    cfg, T, N, blk = job.cfg, job.T, job.N, min(job.cfg["block"], job.T)
    passes = (T + blk - 1) // blk
    rel_w = job.model.rel_emb.weight.detach()
    unsharded = timeit(lambda: job.step(True))
    print(f"{name}: unsharded evaluation (bench.Job.step) {unsharded:8.3f} ms per step ({unsharded / passes * 1e3:7.1f} us per table pass)", flush=True)
    base = None
    for W in (1, 2, 4, 8):
        lo, hi = ranking.shard_bounds(N, W, 0)
        shard = job.table[lo:hi]

        def step():  # rank 0's share of ranking.rank_triples(axis="candidate"), collectives left out
            source = ops.gather_triple_vectors(job.triples, job.ent2idx, shard, row_base=lo) if W > 1 else job.table
            if W > 1:  # (what the all-reduce leaves: here the vectors of the rows this shard does not own stay zero)
                source = torch.cat((job.table[job.triples[:, 0]], job.table[job.triples[:, 1]])) if step.exact else source
            qb = ops.build_queries(job.triples, job.ent2idx, source, rel_w, blk, index=job.index, gather=False, row_base=lo,
                                   by_position=W > 1, num_rows=N)
            counts = ops.rank_all_batches(cfg["model"], shard, qb.fixed_row, rel_w, qb.rel_ids, qb.true_row, T, blk, filter=qb.filter,
                                          source=source, block_triples=blk)
            return ops.rank_metric_sums(counts)

        # the vectors as the all-reduce leaves them (an owner-filled array with zero rows elsewhere would send every row of
        # a bilinear model to the exact routine: a zero query vector gives a zero band and a zero true key)
        step.exact = True
        ms = timeit(step)
        base = base or ms
        print(f"{name} W={W}: {ms:8.3f} ms per step ({ms / passes * 1e3:7.1f} us per table pass)  {base / ms:.2f}x of W=1,  "
              f"{unsharded / ms:.2f}x of the unsharded evaluation", flush=True)
    del job
    torch.cuda.empty_cache()
