import sys, time, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops, _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for model, D in (("transe", 64), ("transe", 128), ("transe", 256), ("distmult", 128), ("complex", 64), ("complex", 128), ("simple", 128)):
    for N in (1, 63, 64, 65, 257, 5000, 70001, 575000):
        table = torch.nn.functional.normalize(torch.randn((N, D), device=dev, generator=g), dim=-1)
        if N > 300: table[17] = table[5]  # ties
        rel = torch.randn((9, D), device=dev, generator=g) * 0.1
        for qh, qt in ((2, 2), (4, 4), (0, 3), (1, 0), (3, 1)):
            Q = qh + qt
            fixed = torch.randint(0, N, (Q,), device=dev, generator=g)
            true = torch.randint(0, N, (Q,), device=dev, generator=g)
            if N > 300: true[0] = 5
            r = torch.randint(0, 9, (Q,), device=dev, generator=g)
            qf, qr = table[fixed].contiguous(), rel[r].contiguous()
            _lib.reset_knobs(); _lib.set_knob("small_kernel", 2)
            a = ops.rank_all(model, table, qf, qr, qh, true_row=true)
            _lib.set_knob("stream_kernel", 2)
            b = ops.rank_all(model, table, qf, qr, qh, true_row=true)
            if not torch.equal(a, b):
                bad += 1; print("MISMATCH", model, D, N, qh, qt, a.tolist(), b.tolist())
print("mismatches", bad)
_lib.reset_knobs()
for N in (575000, 1150000, 4600000):
    table = torch.nn.functional.normalize(torch.randn((N, 128), device=dev, generator=g), dim=-1)
    rel = torch.randn((9, 128), device=dev, generator=g) * 0.1
    fixed = torch.randint(0, N, (4,), device=dev, generator=g); true = torch.randint(0, N, (4,), device=dev, generator=g)
    qf, qr, qt = table[fixed].contiguous(), rel[:4].contiguous(), table[true].contiguous()
    out = torch.empty((4, 4), dtype=torch.int32, device=dev)
    for model in ("transe", "distmult", "complex", "simple"):
        for knob in ((0, 3, 4, 2) if model == "transe" else (0, 3, 4, 5, 2)):
            _lib.set_knob("stream_kernel", knob)
            def step():
                for _ in range(64): ops.rank_all(model, table, qf, qr, 2, q_true=qt, out=out)
            step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); step(); torch.cuda.synchronize()
            print(f"{model} N={N} stream_kernel={knob}: {(time.perf_counter() - t0) / 128 * 1e6:.1f} us per pass")

# the same passes as ONE call (blp_rank_all_batches, a pass per batch of two triples: the ring kernels take them in one launch)
print("reference-batched calls: 64 passes of 2 triples per call, us per pass")
for N in (575000, 2300000, 4600000):
    table = torch.nn.functional.normalize(torch.randn((N, 128), device=dev, generator=g), dim=-1)
    rel = torch.randn((9, 128), device=dev, generator=g) * 0.1
    T, batch = 128, 2
    fixed = torch.randint(0, N, (2 * T,), device=dev, generator=g)
    true = torch.randint(0, N, (2 * T,), device=dev, generator=g)
    rid = torch.randint(0, 9, (2 * T,), device=dev, generator=g)
    for model in ("transe", "distmult", "complex", "simple"):
        for knob in (0, 3, 4):
            _lib.reset_knobs()
            if knob: _lib.set_knob("stream_kernel", knob)
            def step():
                return ops.rank_all_batches(model, table, fixed, rel, rid, true, T, batch, block_triples=batch)
            a = step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); step(); step(); torch.cuda.synchronize()
            print(f"{model} N={N} stream_kernel={knob}: {(time.perf_counter() - t0) / 3 / (T // batch) * 1e6:.1f} us per pass")
_lib.reset_knobs()
