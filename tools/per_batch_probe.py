import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from blp_amd import ops
dev = torch.device("cuda", 0)
job = bench.Job("fb15k237-transe", dev)
model, table, rel_w, T, batch = job.cfg["model"], job.table, job.model.rel_emb.weight.detach(), job.T, 64
qb = ops.build_queries(job.triples, job.ent2idx, table, rel_w, batch, index=job.index, gather=False)
def per_batch(filt=True):
    out = torch.empty((2 * T, 4), dtype=torch.int32, device=dev)
    for start in range(0, T, batch):
        b = min(batch, T - start)
        sl = slice(2 * start, 2 * (start + b))
        seg = qb.filter._replace(seg_lo=qb.filter.seg_lo[sl], seg_hi=qb.filter.seg_hi[sl], exclude=qb.filter.exclude[sl]) if filt else None
        ops.rank_all_idx(model, table, qb.fixed_row[sl], rel_w, qb.rel_ids[sl], b, qb.true_row[sl], filter=seg, out=out[sl])
    return out
def ms(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
orig = ops._workspace
fresh = lambda dev_, stream, n: torch.empty(max(n, 1), dtype=torch.uint8, device=dev_)
for rnd in range(3):
    for name, fn in (("cached", orig), ("fresh", fresh)):
        ops._workspace = fn
        print(f"round {rnd} {name} workspace, filter:    issued %.2f ms, completed %.2f ms" % ms(per_batch), flush=True)
ops._workspace = orig
print("cached workspace, no filter: issued %.2f ms, completed %.2f ms" % ms(lambda: per_batch(False)))
