"""Randomised parity soak: random model / width / sizes / data shapes / filter / true-entity form / slab
knobs, HIP ranking through the C-ABI against the CPU oracle (test infrastructure), for a wall-clock budget.
    python tools/fuzz_parity.py [seconds] [seed] [--shipped]
Prints every mismatch with the seed that reproduces it; exit code 1 if there was one.
--shipped: no knobs at all -- the product library at the dispatch it ships with, and block shapes drawn around its
thresholds (tests/test_gpu_dispatch.py runs a seeded, bounded slice of this mode inside `pytest -m gpu`)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from blp_amd import _lib, ops  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_parity import oracle_counts, random_csr  # noqa: E402


def shipped_shape(rng, model):
    """Block shapes for the shipped dispatch: most draws land near one of its thresholds (rank_common.h: 6 M / 4 M pairs
    for TransE, 0.4 M for the bilinear models, 32 / 64-query floors, <= 4 + 4 queries, 256 / 1 024 tiles)."""
    u = rng.random()
    if u < 0.25:   # pairs around the small-block / pre-pass switch
        N = int(rng.integers(3000, 30000))
        pairs = rng.uniform(3e6, 9e6) if model == "transe" else rng.uniform(2e5, 1.6e6)
        Q = max(2, int(pairs / N))
    elif u < 0.4:  # the query floors of the pre-pass paths on a table that would otherwise qualify
        N = int(rng.integers(60000, 120000)) if model == "transe" else int(rng.integers(13000, 40000))
        Q = int(rng.integers(56, 72)) if model == "transe" else int(rng.integers(26, 40))
    elif u < 0.55:  # <= 4 + 4 queries and just above, short and long tables (streaming kernels / small block / rank_tiles)
        N = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(16000, 17000)), int(rng.integers(60000, 200000))]))
        Q = int(rng.integers(1, 11))
    elif u < 0.65:  # tile-count limits of the small-block kernels
        N = int(rng.choice([16384, 65536])) + int(rng.integers(-70, 71))
        Q = int(rng.integers(5, 60))
    else:
        N = int(rng.integers(1, 2500))
        Q = int(rng.integers(1, 1000))
    q_head = int(rng.integers(0, Q + 1))
    if Q <= 10 and rng.random() < 0.5:
        q_head = min(Q, int(rng.integers(0, 6)))
    return N, q_head, Q - q_head


def make_case(rng, shipped=False):
    model = rng.choice(["transe", "distmult", "complex", "simple"])
    if model == "transe":
        D = int(rng.choice([64, 128, 256, 300, 100, 768, 36]))
    else:
        D = int(rng.choice([64, 128, 256]))
    N = int(rng.integers(1, 2500)) if rng.random() < 0.9 else int(rng.integers(2500, 12000))
    q_head, q_tail = int(rng.integers(0, 500)), int(rng.integers(0, 500))
    if rng.random() < 0.1:
        q_head, q_tail = int(rng.integers(0, 5)), int(rng.integers(0, 5))  # the few-queries (HBM-streaming) mode
    elif rng.random() < 0.1:  # many query tiles per side (chunking of the pair lists), few candidates
        q_head, q_tail = int(rng.integers(0, 4000)), int(rng.integers(0, 4000))
        N = int(rng.integers(1, 600))
    if rng.random() < 0.06:  # a table of more than 256 tiles (tiles sharing slots in the small-block kernels; the streaming kernels)
        N = int(rng.integers(16385, 80000))
        q_head, q_tail = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        if rng.random() < 0.4:
            q_head, q_tail = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    heavy_ties = False
    if rng.random() < 0.04:  # blocks of >= 2 048 queries against a table of few distinct rows: the pair lists overflow, flags become
        heavy_ties = True     # entries of the spill region (rank_gemm.hip: flags_to_entries_kernel) or stay flags when that is full too
        q_head, q_tail = int(rng.integers(900, 3200)), int(rng.integers(1200, 3200))
        N = int(rng.integers(1500, 9000))
        if model == "transe":
            D = int(rng.choice([64, 128]))
    if shipped:
        N, q_head, q_tail = shipped_shape(rng, model)
        if N * D > 6_000_000 and D not in (64, 128):  # keep the host-side table generation of a case short
            D = int(rng.choice([64, 128]))
    if q_head + q_tail == 0:
        q_tail = 1
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    kind = rng.choice(["normal", "normalized", "dyadic", "ties", "outlier", "tiny", "mixed_scale", "nan", "inf", "zero_rows",
                       "constant"])
    table = torch.randn(N, D, generator=g)
    if heavy_ties and not shipped:
        kind = "ties"
        table = table[torch.randint(0, max(1, N // int(rng.choice([20, 60, 200]))), (N,), generator=g)]
    elif kind == "normalized":
        table = torch.nn.functional.normalize(table, dim=-1)
    elif kind == "dyadic":
        table = torch.randint(-8, 9, (N, D), generator=g).float() / 8
    elif kind == "ties" and not heavy_ties:
        table = table[torch.randint(0, max(1, N // 5), (N,), generator=g)]
    elif kind == "outlier":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float(rng.choice([1e4, -1e6, 1e30]))
    elif kind == "tiny":
        table = table * 1e-25
    elif kind == "mixed_scale":
        table = table * torch.exp(torch.randn(N, 1, generator=g) * 3)
    elif kind == "nan":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float("nan")
    elif kind == "inf":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float(rng.choice([float("inf"), -float("inf")]))
    elif kind == "zero_rows":
        table[torch.rand(N, generator=g) < 0.3] = 0.0
    elif kind == "constant":
        table[:] = float(rng.choice([0.0, 0.25, -3.0]))
    Q = q_head + q_tail
    fixed_idx = torch.randint(0, N, (Q,), generator=g)
    q_fixed = table[fixed_idx].clone()
    q_rel = torch.randn(Q, D, generator=g) * float(rng.choice([0.0, 0.1, 1.0]))
    true_row = torch.randint(0, N, (Q,), generator=g)
    rel_ids, rel_table = None, None
    if rng.random() < 0.3:  # queries that are (table row, relation row) pairs: also ranked through blp_rank_all_idx
        rel_ids = torch.randint(0, 7, (Q,), generator=g)
        if rng.random() < 0.5:
            rel_ids[:q_head] = torch.sort(rel_ids[:q_head]).values
        rel_table = torch.randn(7, D, generator=g) * 0.2
        q_rel = rel_table[rel_ids]
    csr = random_csr(Q, N, true_row.numpy(), seed=int(rng.integers(0, 2 ** 31))) if rng.random() < 0.5 else None
    by_vector = rng.random() < 0.3
    if rng.random() < 0.2:  # a strided view (ld > D)
        wide = torch.zeros(N, D + 4 * int(rng.integers(1, 5)))
        wide[:, :D] = table
        table = wide[:, :D]
    env = {}  # test knobs of the library (blp_debug_set_knob)
    if rng.random() < 0.3:
        env["sad_pass_groups"] = int(rng.integers(1, 4))
        env["gemm_pass_words"] = int(rng.integers(1, 3))
    if rng.random() < 0.15:
        env["gemm_kernel"] = 1
    if rng.random() < 0.15:
        env["rank_kernel"] = 1
    if rng.random() < 0.6:
        env["sad_min_queries"] = 64  # the tables here are small: without it TransE blocks take the exact kernels
    u = rng.random()
    if u < 0.35:
        env["small_kernel"] = 2      # ... and every small block the small-block kernel
    elif u < 0.6:
        env["small_kernel"] = int(rng.choice([1, 3]))  # (3: TransE on the register-tile variant) wherever it can run (D = 64 / 128, up to 4 096 queries), whatever the block size
        if rng.random() < 0.5:
            env["exact_query_chunk"] = int(rng.integers(1, 200))
    if rng.random() < 0.45:
        env["stream_kernel"] = int(rng.choice([2, 2, 3, 4]))  # <= 4 + 4 queries on rank_tiles<STATIC> (2) / TransE's workgroup-tile (3) or ring (4) kernel
    if shipped:
        env = {}
    return model, D, N, q_head, q_tail, kind, table, q_fixed, q_rel, true_row, csr, by_vector, env, rel_ids, (fixed_idx, rel_table)


def run(budget=60.0, seed0=0, max_cases=None, shipped=False, out=sys.stdout):
    """Cases seed0, seed0 + 1, ... until the budget (seconds) or max_cases is used up.  Returns (cases, mismatches)."""
    t0, n, bad = time.time(), 0, 0
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or n < max_cases):
        seed = seed0 + n
        rng = np.random.default_rng(seed)
        model, D, N, q_head, q_tail, kind, table, q_fixed, q_rel, true_row, csr, by_vector, env, rel_ids, (fixed_idx, rel_table) = make_case(rng, shipped)
        if os.environ.get("BLP_FUZZ_TRACE"):  # the last line names the case a crash happened in
            print(f"seed={seed} model={model} D={D} N={N} q=({q_head},{q_tail}) data={kind} csr={csr is not None} "
                  f"by_vector={by_vector} env={env}", file=sys.stderr, flush=True)
        kw = dict(true_row=true_row) if not by_vector else dict(q_true=table[true_row])
        want = oracle_counts(oracle, model, table.contiguous(), q_fixed, q_rel, q_head, csr=csr, **kw)
        for k, v in env.items():
            _lib.set_knob(k, v)
        try:
            gkw = {k: v.cuda() for k, v in kw.items()}
            if csr is not None:
                gkw.update(filt_rowptr=torch.from_numpy(csr[0]).cuda(), filt_col=torch.from_numpy(csr[1]).cuda())
            if rel_ids is not None:
                gkw["rel_ids"] = rel_ids.cuda()
            dev_table = table.cuda() if table.is_contiguous() else table._base.cuda()[:, :D]
            if rel_ids is not None and model != "transe":
                gkw.pop("rel_ids")  # the hint is TransE's
            got = ops.rank_all(model, dev_table, q_fixed.cuda(), q_rel.cuda(), q_head, **gkw).cpu().numpy()
            if rel_table is not None and not by_vector and ops.rank_all_supported(model, D, q_head, q_tail):
                filt = None
                if csr is not None:  # the CSR as a segment filter
                    rp = torch.from_numpy(csr[0]).cuda()
                    filt = ops.SegmentFilter(rp[:-1].contiguous(), rp[1:].contiguous(), torch.from_numpy(csr[1]).cuda(), None, None, 0)
                got_idx = ops.rank_all_idx(model, dev_table, fixed_idx.cuda(), rel_table.cuda(), rel_ids.cuda(), q_head,
                                           true_row.cuda(), filter=filt).cpu().numpy()
                if not np.array_equal(got_idx, want):
                    got = got_idx  # reported below as a mismatch
        finally:
            _lib.reset_knobs()
        if not np.array_equal(got, want):
            bad += 1
            rows = np.nonzero((got != want).any(axis=1))[0]
            print(f"MISMATCH seed={seed} model={model} D={D} N={N} q=({q_head},{q_tail}) data={kind} csr={csr is not None} "
                  f"by_vector={by_vector} env={env}: {len(rows)} queries differ, first {rows[:3].tolist()} "
                  f"got {got[rows[0]].tolist()} want {want[rows[0]].tolist()}", file=out, flush=True)
        n += 1
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} mismatches (seeds {seed0}..{seed0 + n - 1}, "
          f"{'shipped dispatch' if shipped else 'random knobs'})", file=out)
    return n, bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if len(args) > 0 else 60.0
    seed0 = int(args[1]) if len(args) > 1 else 0
    n, bad = run(budget, seed0, shipped="--shipped" in sys.argv)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
