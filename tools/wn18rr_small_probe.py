import sys, time, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops, _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
N = 40943
table = torch.nn.functional.normalize(torch.randn((N, 128), device=dev, generator=g), dim=-1)
rel = torch.randn((11, 128), device=dev, generator=g) * 0.1
for model in ("transe", "distmult"):
    for Q in (32, 64, 128, 256):
        fixed = torch.randint(0, N, (Q,), device=dev, generator=g); true = torch.randint(0, N, (Q,), device=dev, generator=g)
        r = torch.randint(0, 11, (Q,), device=dev, generator=g)
        qf, qr = table[fixed].contiguous(), rel[r].contiguous()
        out = torch.empty((Q, 4), dtype=torch.int32, device=dev)
        row = []
        for name, kn in (("default", {}), ("small", {"small_kernel": 1}), ("tiles", {"rank_kernel": 1}), ("prepass", {"small_kernel": 2, "sad_min_queries": 16})):
            _lib.reset_knobs()
            for k, v in kn.items(): _lib.set_knob(k, v)
            def step():
                for _ in range(100): ops.rank_all(model, table, qf, qr, Q // 2, true_row=true, out=out)
            step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
            row.append(f"{name} {(time.perf_counter() - t0) / 100 * 1e6:6.1f}")
        print(f"WN18RR-sized table ({N} rows) {model} {Q:4d} queries: " + "   ".join(row), flush=True)
_lib.reset_knobs()
