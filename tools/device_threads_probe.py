"""What the one-process / one-thread-per-device evaluation (blp_amd.multidevice) costs around the kernels, on ONE GPU: the
Wikidata5M-scale reference-batched evaluation and the FB15k-237 evaluation ranked (a) by the calling thread, (b) as 2 and
4 candidate-axis shards by device threads that all sit on this GPU (peer-copy exchange; with distinct GPUs the exchange is
RCCL's group launch and the shards run at the same time).  On one GPU the shards' kernels share the device, so (b) - (a) is
what threads, barriers and the two exchanges add.    python tools/device_threads_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from blp_amd import models, multidevice, ranking  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("wikidata5m-transe", "fb15k237-transe", "fb15k237-distmult"):
    cfg = bench.WORKLOADS[name]
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev, sort=False)
    model = models.LinkPrediction(cfg["D"], cfg["model"], "margin", cfg["R"], 0)
    model.rel_emb.weight.data = rel_w.cpu()
    model = model.to(dev)
    triples = torch.stack((heads, tails, rels), dim=1).contiguous()
    ent2idx = torch.arange(cfg["N"], device=dev)
    index = bench.make_filter_index(cfg, heads, tails, rels)
    N, block = cfg["N"], cfg["block"]

    def single():
        return ranking.rank_triples(model, table, triples, ent2idx, index, block_size=block)[1]

    def threads(world):
        group = multidevice.DeviceGroup([0] * world)
        shards = [table[slice(*ranking.shard_bounds(N, world, r))].contiguous() for r in range(world)]

        def work(m):
            return ranking.rank_triples(model, shards[m.rank], triples, ent2idx, index, num_entities=N, group=m, world=world,
                                        rank=m.rank, axis="candidate", block_size=block)[1]

        return lambda: group.run(work)[0]

    def ms(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out

    base, want = ms(single)
    line = {"workload": name, "one thread ms": round(base, 3)}
    for world in (2, 4):
        t, got = ms(threads(world))
        line[f"{world} device threads on this GPU ms"] = round(t, 3)
        line[f"{world} threads: counts equal"] = bool(torch.equal(got, want))
    print(line, flush=True)
    del table, index
    torch.cuda.empty_cache()
