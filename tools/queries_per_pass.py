"""Pass time of a reference-batched call against the queries per pass (1 / 2 / 4 triples per batch), 4.6 M x 128 table:
    python tools/queries_per_pass.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
N = 4600000
table = torch.nn.functional.normalize(torch.randn((N, 128), device=dev, generator=g), dim=-1)
rel = torch.randn((9, 128), device=dev, generator=g) * 0.1
for batch in (1, 2, 4):
    T = 32 * batch
    fixed = torch.randint(0, N, (2 * T,), device=dev, generator=g)
    true = torch.randint(0, N, (2 * T,), device=dev, generator=g)
    rid = torch.randint(0, 9, (2 * T,), device=dev, generator=g)
    for model in ("transe", "complex", "distmult"):
        def step():
            return ops.rank_all_batches(model, table, fixed, rel, rid, true, T, batch, block_triples=batch)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); step(); step(); torch.cuda.synchronize()
        print(f"{model} batch={batch} ({2*batch} queries per pass): {(time.perf_counter() - t0) / 3 / 32 * 1e6:.1f} us per pass", flush=True)
