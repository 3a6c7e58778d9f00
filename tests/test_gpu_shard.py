"""GPU: the candidate-axis shard form of the ranking (blp_rank_all_shard, blp_gather_triple_vectors, blp_queries.by_position)
-- the north_star's multi-GPU layout -- against the CPU oracle, and blp_amd.ranking.rank_triples on that axis as two gloo
ranks sharing this one GPU: the fused device path (no torch prelude kernels per block), counts equal to the unsharded run.
The reference evaluates on one device (train.py:79-80); what has to match is its result (train.py:132-171)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REL_MODELS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(model, N, D, T, R, seed):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(N, D, generator=g)
    table = torch.nn.functional.normalize(table, dim=-1) if model == "transe" else table * 0.1
    rel_w = (torch.rand(R, D, generator=g) - 0.5) * 0.25
    ids = torch.randperm(3 * N, generator=g)[:N]            # entity ids are NOT table rows: ent2idx maps them
    ent2idx = torch.full((3 * N,), -1, dtype=torch.long)
    ent2idx[ids] = torch.arange(N)
    triples = torch.stack((ids[torch.randint(0, N, (T,), generator=g)], ids[torch.randint(0, N, (T,), generator=g)],
                           torch.randint(0, R, (T,), generator=g)), dim=1)
    extra = torch.stack((ids[torch.randint(0, N, (6 * T,), generator=g)], ids[torch.randint(0, N, (6 * T,), generator=g)],
                         torch.randint(0, R, (6 * T,), generator=g)), dim=1)
    extra[: 3 * T, 1] = triples[:, 1].repeat(3)  # edges that share (tail, rel) / (head, rel) with test triples: non-empty filters
    extra[: 3 * T, 2] = triples[:, 2].repeat(3)
    extra[3 * T:, 0] = triples[:, 0].repeat(3)
    extra[3 * T:, 2] = triples[:, 2].repeat(3)
    return table, rel_w, ent2idx, triples, torch.cat((triples, extra))


def _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index):
    h, t, r = ent2idx[triples[:, 0]], ent2idx[triples[:, 1]], triples[:, 2]
    rowptr, col = index.csr(triples, ent2idx)
    T = triples.shape[0]
    tab, rel = table.numpy(), rel_w[r].numpy()
    head = oracle.rank_counts(model, oracle.SIDE_HEAD, tab, tab[t.numpy()], rel, true_row=h.numpy(),
                              filt_rowptr=rowptr[:T + 1].numpy(), filt_col=col[:rowptr[T]].numpy())
    tail = oracle.rank_counts(model, oracle.SIDE_TAIL, tab, tab[h.numpy()], rel, true_row=t.numpy(),
                              filt_rowptr=(rowptr[T:] - rowptr[T]).numpy(), filt_col=col[rowptr[T]:].numpy())
    return np.concatenate((head, tail))


@pytest.mark.parametrize("model", REL_MODELS)
@pytest.mark.parametrize("world,form", [(2, "table"), (3, "vectors"), (5, "vectors")])
def test_shards_add_up_to_the_oracle(oracle, model, world, form):
    """W candidate shards ranked one after the other on this GPU exactly as W ranks would (blp_amd.ranking.rank_triples'
    calls): the queries index the replicated table or the all-reduced vector array, the segment filter carries row_base;
    the sum of the shards' counts equals the oracle's unsharded counts, raw and filtered."""
    from blp_amd import ops, ranking, utils
    N, D, T, R = 1500, 128, 230, 7
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=world + len(model))
    index = utils.FilterIndex(edges, num_relations=R)
    want = _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index)
    dev_table, dev_rel, dev_e2i, dev_triples = table.cuda(), rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    bounds = [ranking.shard_bounds(N, world, r) for r in range(world)]
    if form == "vectors":  # what the all-reduce leaves on every rank: the sum of the owner-filled arrays
        parts = [ops.gather_triple_vectors(dev_triples, dev_e2i, dev_table[lo:hi], row_base=lo) for lo, hi in bounds]
        owners = sum((p != 0).any(dim=1).int() for p in parts)
        assert int(owners.max()) <= 1  # every vector has one owner; the others contribute exact zeros
        source = torch.stack(parts).sum(dim=0)
        h, t = dev_e2i[dev_triples[:, 0]], dev_e2i[dev_triples[:, 1]]
        assert torch.equal(source, torch.cat((dev_table[h], dev_table[t])))
    else:
        source = dev_table
    total = torch.zeros((2 * T, 4), dtype=torch.int32, device="cuda")
    for lo, hi in bounds:
        qb = ops.build_queries(dev_triples, dev_e2i, source, dev_rel, 65536, index=index, gather=False, row_base=lo,
                               by_position=form == "vectors", num_rows=N)
        assert int(qb.ids_min) == 0
        total += ops.rank_all_shard(model, dev_table[lo:hi].contiguous(), source, qb.fixed_row, dev_rel, qb.rel_ids, T,
                                    qb.true_row, filter=qb.filter)
    assert np.array_equal(total.cpu().numpy(), want)


@pytest.mark.parametrize("model", REL_MODELS)
def test_reference_batched_shards_of_a_long_table_add_up_to_the_oracle(oracle, model):
    """The Wikidata5M evaluation as the north_star shards it: the reference's batches of two triples (4 queries per table
    pass, scripts/blp-*-wikidata5m.sh:18) against candidate shards long enough for the streaming kernels (the bilinear
    models: approximate keys + band + exact re-scoring, rank_stream.hip), the queries indexing the all-reduced vector
    array, all passes of a shard in one library call.  The shards' counts add up to the oracle's, raw and filtered."""
    from blp_amd import ops, ranking, utils
    N, D, T, R, W, batch = 240017, 128, 9, 5, 2, 2
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=11 + len(model))
    table[N // 3] = table[int(ent2idx[triples[0, 0]])]  # a tie with a true entity, in the first shard or not
    table[N - 5, 7] = float("nan")
    index = utils.FilterIndex(edges, num_relations=R)
    want = _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index)
    dev_table, dev_rel, dev_e2i, dev_triples = table.cuda(), rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    bounds = [ranking.shard_bounds(N, W, r) for r in range(W)]
    source = torch.stack([ops.gather_triple_vectors(dev_triples, dev_e2i, dev_table[lo:hi], row_base=lo) for lo, hi in bounds]).sum(dim=0)
    total = torch.zeros((2 * T, 4), dtype=torch.int32, device="cuda")
    for lo, hi in bounds:
        qb = ops.build_queries(dev_triples, dev_e2i, source, dev_rel, batch, index=index, gather=False, row_base=lo,
                               by_position=True, num_rows=N)
        total += ops.rank_all_batches(model, dev_table[lo:hi].contiguous(), qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, T, batch,
                                      filter=qb.filter, source=source, block_triples=batch)
    idx = torch.arange(T)
    first = idx // batch * batch
    nb = torch.clamp(T - first, max=batch)
    head_pos = 2 * first + (idx - first)
    got = total.cpu().numpy()
    assert np.array_equal(got[head_pos], want[:T])
    assert np.array_equal(got[head_pos + nb], want[T:])


@pytest.mark.parametrize("model,T,block", [("transe", 2600, 65536), ("distmult", 500, 65536), ("complex", 37, 2), ("transe", 9, 2)])
def test_rank_triples_candidate_axis_single_process_equals_unsharded(oracle, model, T, block):
    """world == 1: rank_triples is the same code with source == table -- its counts are the oracle's."""
    from blp_amd import models, ranking, utils
    N, D, R = 2100, 128, 5
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=T)
    index = utils.FilterIndex(edges, num_relations=R)
    m = models.LinkPrediction(D, model, "margin", R, 0)
    m.rel_emb.weight.data = rel_w.clone()
    m = m.cuda()
    _, counts, ok = ranking.rank_triples(m, table.cuda(), triples.cuda(), ent2idx.cuda(), index, block_size=block)
    assert bool(ok)
    assert np.array_equal(counts.cpu().numpy(), _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index))


@pytest.mark.default_routing
@pytest.mark.parametrize("model,N,T,block", [("transe", 3000, 900, 65536), ("complex", 40000, 64, 2), ("distmult", 3000, 2600, 1024)])
def test_candidate_axis_two_gloo_ranks_take_the_fused_path(model, N, T, block):
    """blp_amd.ranking.rank_triples on the candidate axis as TWO ranks (gloo, sharing this GPU; tests/shard_worker.py):
    * counts of every rank == the single-process counts (hence the oracle's, test above);
    * inside the per-block loop NO torch kernel runs (TorchDispatchMode sees only views / allocations): the torch
      prelude of round 2 is gone from this axis;
    * the kernel trace of one block holds only the library's own chain: <= 3 launches for a reference-sized batch or a
      small block, <= 6 for the bilinear pre-pass, <= 8 for TransE's (true keys, range x 2, quantise, pre-pass, pair
      refinement, flag sweep, filter + finalize)."""
    port = 29600 + (N + T) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py"), model, str(N), str(T), str(block)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2
    for r in lines:
        assert r["counts_equal_single_process"] and r["ids_ok"]
        assert r["source"] == ("table" if N <= 2 * T else "vectors")
        assert r["torch_compute_ops_in_block_loop"] == [], r["torch_compute_ops_in_block_loop"]
        assert r["library_calls_for_all_blocks"] == 1  # blp_rank_all_batches: the blocks are issued by the library
        # the collectives actually issued == the plan bench.py / the CPU tests compute without a GPU (the vectors of
        # the queries replicated once, ONE all-gather of the (2T, 4) int32 counts)
        from blp_amd import ranking
        plan = ranking.exchange_plan(N, 128, T, 2, "candidate")
        assert r["collectives"] == [[e["op"], e["bytes_per_rank"]] for e in plan], (r["collectives"], plan)
        assert plan[-1]["bytes_total"] == 2 * (2 * T) * 16
        limit = 3 if 2 * min(T, block) * (N // 2) < 400000 or min(T, block) <= 4 else (8 if model == "transe" else 6)
        if r["kernels_per_block"] is not None:  # the profiler saw the device activity of this process
            assert 1 <= r["kernels_per_block"] <= limit, (r["kernels_per_block"], r["kernel_names"])
            assert not [k for k in r["kernel_names"] if "at::" in k or "elementwise" in k], r["kernel_names"]


@pytest.mark.parametrize("model,N,T,batch", [("transe", 2100, 1000, 64), ("distmult", 2100, 333, 50), ("complex", 40000, 7, 2),
                                             ("simple", 900, 64, 64), ("transe", 300, 70001, 64), ("transe", 70000, 9, 4),
                                             ("distmult", 20000, 31, 3), ("simple", 17000, 5, 1), ("complex", 110000, 9, 2),
                                             ("distmult", 120000, 7, 4), ("simple", 110017, 5, 3), ("complex", 130001, 301, 2),
                                             ("transe", 130001, 203, 3)])
def test_rank_all_batches_equals_per_batch_calls_and_oracle(oracle, model, N, T, batch):
    """blp_rank_all_batches: the reference's loop layout (train.py:128-157: eval_batch_size triples per batch, each batch
    [head queries | tail queries]) handed over in one call == one blp_rank_all_idx call per batch == the oracle, raw and
    filtered; 70 001 triples cross the 65 536-triple block the library ranks at a time."""
    from blp_amd import ops, utils
    D, R = 128, 5
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=T % 97)
    index = utils.FilterIndex(edges, num_relations=R)
    dev_table, dev_rel, dev_e2i, dev_triples = table.cuda(), rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    qb = ops.build_queries(dev_triples, dev_e2i, dev_table, dev_rel, batch, index=index, gather=False)
    got = ops.rank_all_batches(model, dev_table, qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, T, batch, filter=qb.filter)
    if T // batch <= 1200:  # a ranking pass per batch, issued by the library (<= 4 + 4 queries per pass against a long table:
        for filt in (qb.filter, None):  # one preparation and one finalisation launch for all the passes)
            per_pass = ops.rank_all_batches(model, dev_table, qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, T, batch,
                                            filter=filt, block_triples=batch)
            assert torch.equal(per_pass, got) if filt is not None else torch.equal(per_pass[:, :2], got[:, :2])
            assert filt is not None or torch.equal(per_pass[:, 2:], per_pass[:, :2])
    # the loop a maintainer's patch of train.py:128-171 would run, first and last batches
    for start in list(range(0, T, batch))[:3] + list(range(0, T, batch))[-2:]:
        b = min(batch, T - start)
        sl = slice(2 * start, 2 * (start + b))
        seg = qb.filter._replace(seg_lo=qb.filter.seg_lo[sl], seg_hi=qb.filter.seg_hi[sl], exclude=qb.filter.exclude[sl])
        one = ops.rank_all_idx(model, dev_table, qb.fixed_row[sl], dev_rel, qb.rel_ids[sl], b, qb.true_row[sl], filter=seg)
        assert torch.equal(got[sl], one), start
    if T <= 2000:  # the oracle's order is [all heads | all tails]
        want = _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index)
        idx = torch.arange(T)
        first = idx // batch * batch
        nb = torch.clamp(T - first, max=batch)
        head_pos = 2 * first + (idx - first)
        assert np.array_equal(got[head_pos].cpu().numpy(), want[:T])
        assert np.array_equal(got[head_pos + nb].cpu().numpy(), want[T:])


def test_exchange_helpers_run_through_rccl_on_one_rank():
    """The collectives this package issues (blp_amd.ranking: one all-reduce of the (2T, D) query vectors, one all-gather of
    the (2T, 4) int32 counts, the row all-gather of the table build) through backend "nccl" = RCCL, world size 1, on this
    GPU (tests/rccl_worker.py): the process group comes up on the device and the helpers' tensors go through unchanged.
    Not a measurement of anything between GPUs -- the multi-rank behaviour is tests/test_sharded_gloo.py and the two-rank
    gloo runs above."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), "29613"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    out = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"backend": "nccl", "world": 1, "all_reduce_exact": True, "all_gather_exact": True, "all_gather_rows_exact": True}
