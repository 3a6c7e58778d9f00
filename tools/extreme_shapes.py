"""Parity at the corners of the shape space the random soak does not reach: a million-plus queries against
a handful of candidates, one candidate, one query per side, a table exactly at tile / slab boundaries.
HIP ranking through the C-ABI against the CPU oracle (test infrastructure)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from blp_amd import ops
from oracle import oracle
from test_gpu_parity import oracle_counts

CASES = [  # model, D, N, q_head, q_tail
    ("transe", 64, 70, 700_000, 800_001), ("distmult", 64, 70, 600_000, 500_003), ("complex", 64, 33, 400_000, 1),
    ("transe", 128, 1, 300, 300), ("distmult", 128, 1, 300, 300), ("simple", 128, 2, 64, 0),
    ("transe", 128, 64, 128, 128), ("transe", 128, 65, 129, 127), ("transe", 256, 4096, 64, 64),
    ("distmult", 128, 32, 32, 32), ("distmult", 128, 33, 2049, 2047), ("complex", 128, 16384, 16, 48),
    ("transe", 300, 5, 300, 0), ("transe", 768, 129, 0, 257), ("transe", 36, 1000, 256, 1),
    # a handful of queries against long tables (rank_stream.hip): many tiles per wave, a ragged last tile, one side only
    ("transe", 128, 1_500_001, 2, 2), ("transe", 256, 300_007, 4, 4), ("transe", 64, 2_000_003, 0, 3),
    ("distmult", 128, 1_200_005, 2, 2), ("complex", 128, 700_001, 4, 4), ("simple", 64, 1_000_001, 1, 0),
    ("complex", 256, 200_003, 2, 2),
]
bad = 0
for model, D, N, qh, qt in CASES:
    g = torch.Generator().manual_seed(N * 7 + qh)
    table = torch.randn(N, D, generator=g) * 0.3
    Q = qh + qt
    q_fixed = table[torch.randint(0, N, (Q,), generator=g)].clone()
    q_rel = torch.randn(237, D, generator=g)[torch.randint(0, 237, (Q,), generator=g)] * 0.1
    true_row = torch.randint(0, N, (Q,), generator=g)
    t0 = time.time()
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), qh, true_row=true_row.cuda()).cpu().numpy()
    t1 = time.time()
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, qh, true_row=true_row)
    ok = np.array_equal(got, want)
    bad += not ok
    print(f"{model:9s} D={D:4d} N={N:6d} q={qh}+{qt}: {'identical' if ok else 'MISMATCH'}  (hip {t1 - t0:.2f} s, oracle {time.time() - t1:.1f} s)", flush=True)
sys.exit(1 if bad else 0)
