"""Randomised soak of the 16-bit candidate-table entry (blp_rank_all_batches): random model / storage type / width / table
length (short: the widened copy; long: the ring kernels of rank_stream16.hip) / batch and block sizes / row stride / data
shapes (ties with the true entity, zero rows, outliers, NaN, Inf, tiny and mixed magnitudes) / filter, against the CPU oracle
on the table widened to float32 (test infrastructure), for a wall-clock budget.
    python tools/fuzz_table16.py [seconds] [seed]
Prints every mismatch with the seed that reproduces it; exit code 1 if there was one."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blp_amd import ops, utils  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_shard import _oracle_counts, _problem  # noqa: E402

MODELS = ("transe", "distmult", "complex", "simple")


def one_case(seed):
    rng = np.random.default_rng(seed)
    model = MODELS[int(rng.integers(0, 4))]
    dtype = (torch.float16, torch.bfloat16)[int(rng.integers(0, 2))]
    D = int(rng.choice([128, 128, 128, 256, 64] + ([300] if model == "transe" else [])))
    N = int(rng.choice([rng.integers(1, 400), rng.integers(400, 6000), rng.integers(16000, 21000), rng.integers(64000, 150000)]))
    batch = int(rng.integers(1, 7))
    T = int(rng.integers(1, 24))
    block = int(rng.choice([batch, batch, 0]))
    R = 4
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=seed)
    kind = rng.choice(["plain", "ties", "zero_rows", "outlier", "nan", "inf", "tiny", "mixed_scale", "big"])
    rows = ent2idx[triples[:, 0]]
    g = torch.Generator().manual_seed(seed)
    if kind == "ties":
        dup = torch.randint(0, N, (min(N, 8),), generator=g)
        table[dup] = table[rows[torch.randint(0, T, (dup.shape[0],), generator=g)]]
    elif kind == "zero_rows":
        table[torch.rand(N, generator=g) < 0.2] = 0.0
    elif kind == "outlier":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float(rng.choice([1e3, -6e4]))
    elif kind == "nan":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float("nan")
    elif kind == "inf":
        table[int(rng.integers(0, N)), int(rng.integers(0, D))] = float(rng.choice([float("inf"), -float("inf")]))
    elif kind == "tiny":
        table = table * 1e-4          # (float16 subnormals)
    elif kind == "mixed_scale":
        table = table * torch.exp(torch.randn(N, 1, generator=g))
    elif kind == "big":
        table = table * 50.0
    pad = int(rng.choice([0, 0, 8, 24]))
    backing = torch.zeros(N, D + pad, dtype=dtype, device="cuda")
    backing[:, :D] = table.to(dtype).cuda()
    table16 = backing[:, :D]
    wide = table16.float().contiguous()
    filtered = rng.random() < 0.7
    index = utils.FilterIndex(edges, num_relations=R)
    dev_rel, dev_e2i, dev_triples = rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    source = ops.gather_triple_vectors(dev_triples, dev_e2i, table16)
    qb = ops.build_queries(dev_triples, dev_e2i, source, dev_rel, batch, index=index, gather=False, by_position=True, num_rows=N)
    got = ops.rank_all_batches(model, table16, qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, T, batch,
                               filter=qb.filter if filtered else None, source=source, block_triples=block).cpu().numpy()
    want = _oracle_counts(oracle, model, wide.cpu(), rel_w, ent2idx, triples, index)
    idx = torch.arange(T)
    first = idx // batch * batch
    head_pos = 2 * first + (idx - first)
    tail_pos = head_pos + torch.clamp(T - first, max=batch)
    got = np.concatenate((got[head_pos.numpy()], got[tail_pos.numpy()]))
    if not filtered:
        want = np.concatenate((want[:, :2], want[:, :2]), axis=1)
    ok = np.array_equal(got, want)
    desc = f"seed={seed} model={model} dtype={dtype} D={D} N={N} T={T} batch={batch} block={block} pad={pad} data={kind} filtered={filtered}"
    return ok, desc


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if args else 60.0
    seed0 = int(args[1]) if len(args) > 1 else 0
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        ok, desc = one_case(seed0 + n)
        if os.environ.get("BLP_FUZZ_TRACE"):
            print(desc, file=sys.stderr, flush=True)
        if not ok:
            bad += 1
            print("MISMATCH " + desc, flush=True)
        n += 1
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} mismatches (seeds {seed0}..{seed0 + n - 1}, 16-bit candidate tables)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
