"""GPU: the candidate table in a 16-bit storage type (blp_rank_all_batches / blp_gather_triple_vectors; include/blp_hip.h,
blp_amd/csrc/table_elem.h, rank_stream16.hip) -- the half-precision copy the table build can emit.  The
reference has no such table (train.py:96-97 keeps `ent_emb` in float32); what has to hold is that a 16-bit table is ranked as
the reference ranks the SAME VALUES in float32: every test compares with the CPU oracle on the table widened to float32
(exact) and with the library's own float32 path on that widened table -- counts identical, raw and filtered."""
import numpy as np
import pytest
import torch

from conftest import REL_MODELS
from test_gpu_shard import _oracle_counts, _problem

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def _loop_positions(T, batch):
    idx = torch.arange(T)
    first = idx // batch * batch
    nb = torch.clamp(T - first, max=batch)
    head_pos = 2 * first + (idx - first)
    return head_pos, head_pos + nb


def _rank16(model, table16, rel_w, ent2idx, triples, index, batch, block_triples, filtered=True):
    from blp_amd import ops
    dev_rel, dev_e2i, dev_triples = rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    T = triples.shape[0]
    source = ops.gather_triple_vectors(dev_triples, dev_e2i, table16)  # float32, widened exactly
    qb = ops.build_queries(dev_triples, dev_e2i, source, dev_rel, batch, index=index, gather=False, by_position=True,
                           num_rows=table16.shape[0])
    return ops.rank_all_batches(model, table16, qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, T, batch,
                                filter=qb.filter if filtered else None, source=source, block_triples=block_triples)


def _rank32(model, table32, rel_w, ent2idx, triples, index, batch, block_triples):
    from blp_amd import ops
    dev_rel, dev_e2i, dev_triples = rel_w.cuda(), ent2idx.cuda(), triples.cuda()
    qb = ops.build_queries(dev_triples, dev_e2i, table32, dev_rel, batch, index=index, gather=False)
    return ops.rank_all_batches(model, table32, qb.fixed_row, dev_rel, qb.rel_ids, qb.true_row, triples.shape[0], batch,
                                filter=qb.filter, block_triples=block_triples)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("model,N,T,batch,D", [("transe", 70000, 9, 2, 128), ("transe", 130001, 41, 4, 128), ("transe", 66000, 7, 3, 256),
                                               ("transe", 75000, 5, 1, 128), ("distmult", 120000, 7, 4, 128), ("complex", 110000, 9, 2, 128),
                                               ("simple", 110017, 5, 3, 128), ("complex", 70001, 6, 2, 256), ("distmult", 66000, 5, 1, 256),
                                               ("simple", 80000, 1, 2, 128)])
def test_reference_batched_passes_read_the_16_bit_table(oracle, model, N, T, batch, D, dtype):
    """eval_batch_size <= 4 against a long table (scripts/blp-*-wikidata5m.sh:18; train.py:128-171): the ring kernels read the
    16-bit table as it is, every pass in one launch (blp_rank_all_batches_passes_per_launch says so); counts == the oracle's
    on the widened table == the float32 path's on the widened table, raw and filtered; T = 1 / a short last batch included."""
    from blp_amd import _lib, utils
    R = 5
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=T + N % 89)
    table16 = table.to(dtype).cuda()
    wide = table16.float()  # exact
    index = utils.FilterIndex(edges, num_relations=R)
    L, mid = _lib.lib(), _lib.MODEL_IDS[model]
    assert L.blp_rank_all_batches_passes_per_launch(mid, {torch.float16: 1, torch.bfloat16: 2}[dtype], N, D, D, T, batch, batch) == -(-T // batch)
    got = _rank16(model, table16, rel_w, ent2idx, triples, index, batch, batch)
    ref32 = _rank32(model, wide, rel_w, ent2idx, triples, index, batch, batch)
    assert torch.equal(got, ref32)
    raw = _rank16(model, table16, rel_w, ent2idx, triples, index, batch, batch, filtered=False)
    assert torch.equal(raw[:, :2], got[:, :2]) and torch.equal(raw[:, 2:], raw[:, :2])
    want = _oracle_counts(oracle, model, wide.cpu(), rel_w, ent2idx, triples, index)
    head_pos, tail_pos = _loop_positions(T, batch)
    assert np.array_equal(got[head_pos].cpu().numpy(), want[:T])
    assert np.array_equal(got[tail_pos].cpu().numpy(), want[T:])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("model,N,T,batch,block,D", [("transe", 2100, 700, 64, 0, 128), ("distmult", 2100, 333, 50, 0, 128),
                                                     ("complex", 900, 64, 64, 0, 64), ("simple", 20000, 31, 3, 0, 128),
                                                     ("transe", 1500, 300, 16, 0, 300), ("transe", 300, 40, 2, 2, 128),
                                                     ("distmult", 3000, 9, 2, 2, 64), ("transe", 70000, 9, 2, 2, 64),
                                                     ("transe", 16500, 64, 64, 64, 128)])
def test_other_shapes_rank_a_widened_copy(oracle, model, N, T, batch, block, D, dtype):
    """Blocks of many queries (the pre-pass paths), short tables (the small-block kernels), widths the ring does not take
    (64; the bag-of-words 300): the call widens the table into its workspace and ranks that -- the same counts again."""
    from blp_amd import utils
    R = 5
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=N % 53 + T)
    table16 = table.to(dtype).cuda()
    wide = table16.float()
    index = utils.FilterIndex(edges, num_relations=R)
    got = _rank16(model, table16, rel_w, ent2idx, triples, index, batch, block)
    assert torch.equal(got, _rank32(model, wide, rel_w, ent2idx, triples, index, batch, block))
    want = _oracle_counts(oracle, model, wide.cpu(), rel_w, ent2idx, triples, index)
    head_pos, tail_pos = _loop_positions(T, batch)
    assert np.array_equal(got[head_pos].cpu().numpy(), want[:T])
    assert np.array_equal(got[tail_pos].cpu().numpy(), want[T:])


@pytest.mark.parametrize("dtype", DTYPES)
def test_padded_rows_special_values_and_ties(oracle, dtype):
    """A 16-bit table with a row stride above D (ld % 8 == 0), duplicate rows of the true entity (ties: gt vs ge), an
    all-zero row, infinities and a NaN in candidate rows (widened exactly; the oracle sees the same values)."""
    from blp_amd import utils
    N, D, T, R, batch = 66000, 128, 6, 3, 2
    for model in REL_MODELS:
        table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=11)
        rows = ent2idx[triples[:, 0]]
        table[5] = table[rows[0]]            # a tie with a true entity
        table[6] = 0.0
        table[7, 3] = float("inf")
        table[8, 100] = float("-inf")
        table[9, 64] = float("nan")
        backing = torch.zeros(N, D + 24, dtype=dtype, device="cuda")
        backing[:, :D] = table.to(dtype).cuda()
        table16 = backing[:, :D]             # stride D + 24 elements
        assert table16.stride(0) == D + 24
        wide = table16.float().contiguous()
        index = utils.FilterIndex(edges, num_relations=R)
        got = _rank16(model, table16, rel_w, ent2idx, triples, index, batch, batch)
        assert torch.equal(got, _rank32(model, wide, rel_w, ent2idx, triples, index, batch, batch)), model
        want = _oracle_counts(oracle, model, wide.cpu(), rel_w, ent2idx, triples, index)
        head_pos, tail_pos = _loop_positions(T, batch)
        assert np.array_equal(got[head_pos].cpu().numpy(), want[:T]), model
        assert np.array_equal(got[tail_pos].cpu().numpy(), want[T:]), model


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("model", ["transe", "complex"])
def test_rank_triples_takes_a_16_bit_table(oracle, model, dtype):
    """blp_amd.ranking.rank_triples (the evaluation loop body, train.py:128-171) with the 16-bit copy of the table, at the
    reference's Wikidata5M batching (block_size = 2) and as one block: counts == the oracle's on the widened table."""
    from blp_amd import models, ranking, utils
    N, D, T, R = 70000, 128, 11, 5
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=3)
    m = models.LinkPrediction(D, model, "margin", R, 0).cuda()
    with torch.no_grad():
        m.rel_emb.weight.copy_(rel_w.cuda())
    table16 = table.to(dtype).cuda()
    index = utils.FilterIndex(edges, num_relations=R)
    want = _oracle_counts(oracle, model, table16.float().cpu(), rel_w, ent2idx, triples, index)
    for block_size in (2, 65536):
        _, counts, ids_ok = ranking.rank_triples(m, table16, triples.cuda(), ent2idx.cuda(), index, block_size=block_size)
        assert bool(ids_ok)
        assert np.array_equal(counts.cpu().numpy(), want), block_size


def test_bad_arguments_of_the_typed_entries():
    """Unknown dtype, a 16-bit row stride that is not a multiple of 8, a short workspace: status codes, nothing launched."""
    import ctypes
    from blp_amd import _lib
    L = _lib.lib()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    args = lambda dt, ld, ws: (0, p, dt, 100, 128, ld, p, 100, 128, p, p, 5, p, p, 4, 2, 2, None, p, p, ws, 0, None)
    assert L.blp_rank_all_batches(*args(7, 128, 1 << 16)) == -1
    assert L.blp_rank_all_batches(*args(1, 132, 1 << 16)) == -1          # ld % 8 != 0
    assert L.blp_rank_all_batches(*args(1, 128, 16)) == -4                # workspace
    assert b"workspace" in L.blp_last_error()
    assert L.blp_gather_triple_vectors(p, 4, None, 0, p, 9, 100, 128, 128, 0, p, 0, None) == -1
    assert L.blp_rank_all_batches_workspace_bytes(0, 1, 100, 128, 128, 4, 2, 2) > 0


@pytest.mark.parametrize("seed", range(24))
def test_seeded_random_cases(oracle, seed):
    """A seeded slice of random shapes around the routes of a 16-bit call: the ring (long tables, <= 4 triples per pass), the
    widened copy (short tables, blocks, D = 64), row strides above D, with and without a filter."""
    from blp_amd import utils
    rng = np.random.default_rng(1000 + seed)
    model = REL_MODELS[seed % 4]
    dtype = DTYPES[(seed // 4) % 2]
    D = int(rng.choice([128, 128, 256, 64]))
    N = int(rng.choice([rng.integers(200, 3000), rng.integers(16000, 20000), rng.integers(65000, 140000)]))
    batch = int(rng.integers(1, 5))
    T = int(rng.integers(1, 40))
    block = int(rng.choice([batch, batch, 0]))
    R = 4
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=seed)
    pad = int(rng.choice([0, 8, 40]))
    backing = torch.zeros(N, D + pad, dtype=dtype, device="cuda")
    backing[:, :D] = table.to(dtype).cuda()
    table16 = backing[:, :D]
    wide = table16.float().contiguous()
    index = utils.FilterIndex(edges, num_relations=R)
    got = _rank16(model, table16, rel_w, ent2idx, triples, index, batch, block)
    assert torch.equal(got, _rank32(model, wide, rel_w, ent2idx, triples, index, batch, block)), (model, dtype, D, N, T, batch, block, pad)
    want = _oracle_counts(oracle, model, wide.cpu(), rel_w, ent2idx, triples, index)
    head_pos, tail_pos = _loop_positions(T, batch)
    assert np.array_equal(got[head_pos].cpu().numpy(), want[:T]) and np.array_equal(got[tail_pos].cpu().numpy(), want[T:])


def test_integration_md_16_bit_stub_is_runnable(oracle):
    """The second ctypes stub of INTEGRATION.md section 2 (blp_rank_all_batches + blp_gather_triple_vectors on a 16-bit copy
    of the table), executed as printed on top of the first one: counts == the oracle's on the widened table, in the loop's
    layout, at the reference's Wikidata5M batching (a pass per 2 triples: the ring kernels) and at a batch of 5 (a widened copy)."""
    import os
    import re
    import types
    from blp_amd import _lib, utils
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    first = re.search(r"```python\n(# blp_hip\.py.*?)```", text, flags=re.S).group(1)
    second = re.search(r"```python\n(_L\.blp_gather_triple_vectors\.argtypes.*?)```", text, flags=re.S).group(1)
    stub = types.ModuleType("blp_hip_stub16")
    exec(first.replace('"libblp_hip.so"', repr(_lib.LIB_PATH)) + "\n" + second, stub.__dict__)
    N, D, R, T = 70000, 128, 5, 9
    for model, dtype, batch in (("transe", torch.float16, 2), ("complex", torch.bfloat16, 2), ("distmult", torch.float16, 5)):
        table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=7)
        ent16 = table.to(dtype).cuda()
        rows = torch.stack((ent2idx[triples[:, 0]], ent2idx[triples[:, 1]], triples[:, 2]), dim=1).cuda()
        head_pos, tail_pos = _loop_positions(T, batch)
        rel_id = torch.empty(2 * T, dtype=torch.int64)
        rel_id[head_pos] = rel_id[tail_pos] = triples[:, 2]
        got = stub.rank_all_batches_16(model, ent16, rows, rel_w.cuda(), rel_id.cuda(), T, batch)
        index = utils.FilterIndex(edges, num_relations=R)
        want = _oracle_counts(oracle, model, ent16.float().cpu(), rel_w, ent2idx, triples, index)
        assert np.array_equal(got[head_pos][:, :2].cpu().numpy(), want[:T, :2]), model   # (no filter handed over: raw counts)
        assert np.array_equal(got[tail_pos][:, :2].cpu().numpy(), want[T:, :2]), model
        assert torch.equal(got[:, 2:], got[:, :2])


@pytest.mark.parametrize("rel_model", ["transe", "complex"])
def test_rank_table_dtype_is_honoured_only_where_the_passes_read_16_bits(rel_model, monkeypatch, caplog):
    """ranking.eval_link_prediction(..., rank_table_dtype=float16): at the reference's Wikidata5M batching (block_size = 2 triples
    per table pass, dim 128, a long table) the evaluation ranks the 16-bit copy -- metrics = the float32 ranking of the ROUNDED
    table; at the default block size the library would only widen such a copy back to float32 (more memory and time, other
    metrics, for nothing), so the float32 table is ranked, the metrics are the float32 ones and a log line says why."""
    import logging
    from blp_amd import models, ranking
    from test_host_golden import _Run, _Triples
    N, D, R, T = 20000, 128, 5, 40
    table, rel_w, ent2idx, triples, _ = _problem(rel_model, N, D, T, R, seed=3)
    triples = torch.stack((ent2idx[triples[:, 0]], ent2idx[triples[:, 1]], triples[:, 2]), dim=1)  # (transductive: entity id = table row)
    model = models.TransductiveLinkPrediction(D, rel_model, "margin", N, R, 0)
    model.rel_emb.weight.data = rel_w.clone()
    model = model.cuda()
    dev_table = table.cuda()
    monkeypatch.setattr(ranking, "build_entity_table", lambda *a, **k: dev_table.clone())
    seen = []
    real = ranking.rank_triples
    monkeypatch.setattr(ranking, "rank_triples", lambda m, t, *a, **k: (seen.append(t.dtype), real(m, t, *a, **k))[1])
    loader = torch.utils.data.DataLoader(_Triples(triples, torch.zeros(R, dtype=torch.long)), batch_size=2)
    log = logging.getLogger("table16")

    def run(**kw):
        r = _Run()
        ranking.eval_link_prediction(model, loader, None, None, 0, 512, r, log, prefix="test", **kw)
        return r.scalars

    f32 = run(block_size=2)
    rounded = dev_table.half().float()
    monkeypatch.setattr(ranking, "build_entity_table", lambda *a, **k: rounded.clone())
    f32_of_rounded = run(block_size=2)
    monkeypatch.setattr(ranking, "build_entity_table", lambda *a, **k: dev_table.clone())
    seen.clear()
    with caplog.at_level(logging.INFO, logger="table16"):
        half_passes = run(block_size=2, rank_table_dtype=torch.float16)
        assert seen == [torch.float16] and "not used" not in caplog.text
        half_block = run(rank_table_dtype=torch.float16)  # default block: 65 536 triples per ranking call
        assert seen == [torch.float16, torch.float32] and "rank_table_dtype=torch.float16: not used" in caplog.text
    assert half_passes == f32_of_rounded  # the same arithmetic on the rounded rows
    assert half_block == f32
    assert set(f32) >= {"test_mrr", "test_hits@1", "test_hits@10"}
