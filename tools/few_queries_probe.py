# A handful of queries (the reference's Wikidata5M eval batch) against tables of 7 k .. 1 M rows: the streaming kernels
# (rank_stream.hip) against the small-block kernels (rank_small.hip, knob small_kernel=1): where the routing rule switches.
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for model in ("transe", "distmult", "complex"):
    print(f"{model}: us per call, rows x queries: default | small_kernel=1")
    for N in (7400, 14541, 50000, 120000, 250000, 500000, 1000000):
        table = torch.nn.functional.normalize(torch.randn((N, 128), device=dev, generator=g), dim=-1)
        rel = torch.randn((16, 128), device=dev, generator=g) * 0.1
        row = []
        for Q in (4, 8):
            fixed = torch.randint(0, N, (Q,), device=dev, generator=g); true = torch.randint(0, N, (Q,), device=dev, generator=g)
            qf, qr = table[fixed].contiguous(), rel[:Q].contiguous()
            out = torch.empty((Q, 4), dtype=torch.int32, device=dev)
            res = []
            for knob in (0, 1):
                _lib.reset_knobs(); _lib.set_knob("small_kernel", knob)
                def step():
                    for _ in range(100): ops.rank_all(model, table, qf, qr, Q // 2, true_row=true, out=out)
                step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
                res.append(((time.perf_counter() - t0) / 100 * 1e6, out.clone()))
            assert torch.equal(res[0][1], res[1][1])
            row.append(f"{Q} q: {res[0][0]:6.1f} | {res[1][0]:6.1f}")
        print(f"  {N:8d} rows   " + "    ".join(row), flush=True)
_lib.reset_knobs()
