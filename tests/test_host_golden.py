"""Host-side layer (blp_amd.models / utils / data / ranking / sacred_shim) against golden vectors from
the imported reference.  CPU only: CPU tensors take the torch-expression route by design."""
import logging
import types

import numpy as np
import pytest
import torch

from conftest import REL_MODELS, golden, golden_names
from blp_amd import data, models, ranking, utils
from blp_amd.sacred_shim import Experiment


def test_make_ent2idx_docstring_and_golden():
    g = golden("neg_sampling")
    assert utils.make_ent2idx(torch.tensor([4, 5, 0]), 5).tolist() == [2, -1, -1, -1, 0, 1]  # utils.py:36-38
    assert np.array_equal(utils.make_ent2idx(torch.tensor([4, 5, 0]), 5).numpy(), g["ent2idx_doc"])
    assert np.array_equal(utils.make_ent2idx(torch.from_numpy(g["ent2idx_ents"]), 14).numpy(), g["ent2idx_out"])


def test_negative_sampling_matches_reference_rng_stream():
    g = golden("neg_sampling")
    for i in range(4):
        b, k, rep, seed = (int(x) for x in g[f"case{i}_args"])
        torch.manual_seed(seed)
        got = data.get_negative_sampling_indices(b, k, repeats=rep)
        assert np.array_equal(got.numpy(), g[f"case{i}_neg_idx"])


@pytest.mark.parametrize("b,k,rep", [(2, 1, 1), (5, 7, 1), (6, 3, 2), (64, 64, 1)])
def test_negative_sampling_properties(b, k, rep):
    """data.py:36-56: shape (B*rep, K, 2); exactly one slot of the pair replaced, by a slot of
    another row of the same device-local batch."""
    torch.manual_seed(b * 100 + k)
    idx = data.get_negative_sampling_indices(b, k, repeats=rep)
    assert idx.shape == (b * rep, k, 2) and idx.dtype == torch.int64
    assert int(idx.min()) >= 0 and int(idx.max()) < 2 * b
    own = torch.arange(2 * b).reshape(b, 2).repeat(rep, 1).unsqueeze(1).expand(-1, k, -1)
    kept = idx == own
    assert torch.all(kept.sum(-1) == 1)                      # one kept, one replaced
    replaced = idx[~kept].reshape(b * rep, k)
    assert torch.all(replaced // 2 != (torch.arange(b * rep) % b).unsqueeze(1))  # never the own row


def test_sampler_index_construction_equals_the_reference_on_its_own_goldens():
    """The device sampler = integer draws + a deterministic index construction (data.negative_indices_from_draws).  The
    construction is held against the REFERENCE's sampler: every golden index tensor of the imported reference
    (neg_sampling.npz, repeats = 1 cases: device-local indices) is inverted into (draw, which), and the construction maps the
    draws back onto exactly those indices; the plain-loop restatement (oracle/ref_port.py: the checker used on the GPU box)
    agrees with both."""
    from oracle import ref_port
    g = golden("neg_sampling")
    checked = 0
    for i in range(4):
        b, k, rep, _ = (int(x) for x in g[f"case{i}_args"])
        want = torch.from_numpy(g[f"case{i}_neg_idx"])
        for r in range(rep):  # (a slice per device: indices local to the slice, data.py:289-298)
            part = want[r * b:(r + 1) * b]
            draw, which = data.draws_from_negative_indices(part)
            assert int(draw.min()) >= 0 and int(draw.max()) < 2 * b - 2 and set(which.unique().tolist()) <= {0, 1}
            assert torch.equal(data.negative_indices_from_draws(draw, which), part)
            assert torch.equal(ref_port.neg_idx_from_draws(draw.tolist(), which.tolist()), part)
            checked += part.numel()
    assert checked > 8000
    # and every possible draw of a small batch: the construction is a bijection onto the valid index pairs
    b = 4
    draw = torch.arange(2 * b - 2).repeat(b, 2)
    which = torch.cat((torch.zeros(b, 2 * b - 2, dtype=torch.long), torch.ones(b, 2 * b - 2, dtype=torch.long)), dim=1)
    idx = data.negative_indices_from_draws(draw, which)
    assert torch.equal(idx, ref_port.neg_idx_from_draws(draw.tolist(), which.tolist()))
    assert len({tuple(x) for x in idx[1].tolist()}) == 2 * (2 * b - 2)
    d2, w2 = data.draws_from_negative_indices(idx)
    assert torch.equal(d2, draw) and torch.equal(w2, which)


def test_device_sampler_has_the_reference_law():
    """The on-device sampler keeps one slot, replaces the other from another row, uniformly."""
    g = torch.Generator().manual_seed(3)
    b, k = 6, 4000
    idx = data.get_negative_sampling_indices_on_device(b, k, "cpu", generator=g)
    assert idx.shape == (b, k, 2) and idx.dtype == torch.int64
    own = torch.arange(2 * b).reshape(b, 1, 2).expand(-1, k, -1)
    kept = idx == own
    assert torch.all(kept.sum(-1) == 1)
    replaced = idx[~kept].reshape(b, k)
    assert torch.all(replaced // 2 != torch.arange(b).unsqueeze(1))
    # uniform over the 2B - 2 foreign slots and over the two columns (chi-square-ish bounds)
    for row in range(b):
        hist = torch.bincount(replaced[row], minlength=2 * b).float()
        assert hist[2 * row] == 0 and hist[2 * row + 1] == 0
        expected = k / (2 * b - 2)
        assert (hist[hist > 0] - expected).abs().max() < 6 * expected ** 0.5
    assert abs(float((~kept)[..., 0].float().mean()) - 0.5) < 0.02


def test_filter_index_matches_reference_masks():
    g = golden("filters_toy")
    triples = torch.from_numpy(g["triples"])
    index = utils.FilterIndex(torch.from_numpy(g["graph_edges"]))
    ent2idx = torch.from_numpy(g["ent2idx"])
    n = g["entities"].shape[0]
    hf, tf = index.masks(triples, n, ent2idx)
    assert np.array_equal(hf.numpy(), g["heads_filter"])
    assert np.array_equal(tf.numpy(), g["tails_filter"])
    hf2, tf2 = utils.get_triple_filters(triples, index, n, ent2idx)
    assert torch.equal(hf, hf2) and torch.equal(tf, tf2)
    rowptr, cols = index.csr(triples, ent2idx)
    assert rowptr.shape[0] == 2 * triples.shape[0] + 1 and rowptr[-1] == cols.shape[0]
    assert int(rowptr[-1]) == int(g["heads_filter"].sum() + g["tails_filter"].sum())  # no duplicates
    # the true entity is never filtered (utils.py:71,78)
    true_rows = torch.cat((ent2idx[triples[:, 0]], ent2idx[triples[:, 1]]))
    for q in range(2 * triples.shape[0]):
        assert int(true_rows[q]) not in cols[rowptr[q]:rowptr[q + 1]].tolist()
    # networkx graphs are accepted too
    import networkx as nx
    graph = nx.MultiDiGraph()
    graph.add_weighted_edges_from(g["graph_edges"].tolist())
    hf3, tf3 = utils.get_triple_filters(triples, graph, n, ent2idx)
    assert torch.equal(hf, hf3) and torch.equal(tf, tf3)


def expand_segments(seg, num_rows):
    """What the kernels' filter_row (blp_amd/csrc/rank_common.h) makes of a SegmentFilter: per query the table
    rows removed, as a dense (Q, num_rows) mask."""
    mask = torch.zeros((seg.seg_lo.shape[0], num_rows), dtype=torch.bool)
    for q in range(seg.seg_lo.shape[0]):
        for v in seg.values[int(seg.seg_lo[q]):int(seg.seg_hi[q])].tolist():
            if v == int(seg.exclude[q]):
                continue
            row = int(seg.ent2idx[v]) if 0 <= v < seg.ent2idx.shape[0] else -1
            row -= seg.row_base
            if 0 <= row < num_rows:
                assert not mask[q, row], "a row listed twice in one segment"
                mask[q, row] = True
    return mask


def test_filter_segments_match_reference_masks():
    """FilterIndex.segments (the device-side form the HIP ranking takes) removes exactly the rows of the
    reference's masks (utils.py:46-83), also seen from a candidate shard (row_base)."""
    g = golden("filters_toy")
    triples = torch.from_numpy(g["triples"])
    index = utils.FilterIndex(torch.from_numpy(g["graph_edges"]))
    ent2idx = torch.from_numpy(g["ent2idx"])
    n = g["entities"].shape[0]
    want = np.concatenate((g["heads_filter"], g["tails_filter"]))
    seg = index.segments(triples, ent2idx, "cpu")
    assert np.array_equal(expand_segments(seg, n).numpy(), want)
    lo = n // 3
    shard = expand_segments(seg._replace(row_base=lo), n - lo)
    assert np.array_equal(shard.numpy(), want[:, lo:])


def test_filter_index_ignores_relations_it_never_saw():
    """Keys pack (entity, relation) as entity * R + relation with R taken from the graph: a query with a relation
    id >= R must match nothing (it used to alias onto (entity + 1, relation - R))."""
    edges = torch.tensor([[0, 1, 0], [0, 2, 1], [1, 2, 1], [3, 2, 0]])  # (head, tail, rel), R = 2
    index = utils.FilterIndex(edges)
    ent2idx = torch.arange(4)
    triples = torch.tensor([[0, 3, 2], [2, 1, 3], [0, 3, -1]])  # relation 2 would alias onto (1, rel 0) / (4, rel 0)
    rowptr, cols = index.csr(triples, ent2idx)
    assert rowptr.tolist() == [0] * 7 and cols.numel() == 0
    assert not expand_segments(index.segments(triples, ent2idx, "cpu"), 4).any()
    known = torch.tensor([[0, 3, 1]])  # (0, rel 1) -> tail 2 is a known edge
    assert expand_segments(index.segments(known, ent2idx, "cpu"), 4)[1].tolist() == [False, False, True, False]
    wide = utils.FilterIndex(edges, num_relations=5)  # the dataset's count, as train.py passes it
    assert expand_segments(wide.segments(known, ent2idx, "cpu"), 4)[1].tolist() == [False, False, True, False]


def test_get_metrics_matches_golden():
    g = golden("scores_transe_ties_d128")
    pred = torch.from_numpy(np.concatenate((g["head_pred"], g["tail_pred"])))
    true = torch.from_numpy(np.concatenate((g["heads"], g["tails"])))
    rr, hits = utils.get_metrics(pred, true, torch.tensor([[1, 3, 10]]))
    assert np.array_equal(rr.numpy(), g["rr"]) and np.array_equal(hits.numpy(), g["hits"])


class _Run:
    _id = None

    def __init__(self):
        self.scalars = {}

    def log_scalar(self, name, value, step=None):
        self.scalars[name] = float(value)


class _Triples(torch.utils.data.Dataset):
    def __init__(self, triples, cats):
        self.triples, self.rel_categories, self.has_rel_categories = triples, cats, True

    def __getitem__(self, i):
        return self.triples[i]

    def __len__(self):
        return self.triples.shape[0]


def toy_eval_setup(g, rel_model, device="cpu"):
    model = models.BOW(rel_model, "margin", g["rel_w"].shape[0], 0.0, embeddings=torch.from_numpy(g["word_emb"]).clone())
    model.rel_emb.weight.data = torch.from_numpy(g["rel_w"]).clone()
    model = model.to(device)
    text = types.SimpleNamespace(text_data=torch.from_numpy(g["text_data"]))
    text.get_entity_description = types.MethodType(data.TextGraphDataset.get_entity_description, text)
    triples = torch.from_numpy(g["triples"])
    loader = torch.utils.data.DataLoader(_Triples(triples, torch.from_numpy(g["rel_categories"])),
                                         int(g["eval_batch_size"]))
    index = utils.FilterIndex(torch.from_numpy(g["graph_edges"]))
    return model, text, loader, index, torch.from_numpy(g["entities"]), set(g["new_entities"].tolist())


@pytest.mark.parametrize("rel_model", REL_MODELS)
def test_eval_link_prediction_reproduces_reference_scalars(rel_model):
    """The full train.eval_link_prediction output of the reference on a toy graph (raw + filtered MRR /
    Hits@k, by-new-position split, returned embeddings).  MRR within 1e-6 (the tolerance of the task
    is 1e-5), Hits@k exact."""
    g = golden(f"eval_toy_{rel_model}")
    model, text, loader, index, entities, new_ents = toy_eval_setup(g, rel_model)
    run, log = _Run(), logging.getLogger("test")
    mrr, ent_emb = ranking.eval_link_prediction(model, loader, text, entities, 3, int(g["emb_batch_size"]), run, log,
                                                prefix="test", filtering_graph=index, new_entities=new_ents,
                                                return_embeddings=True)
    want = dict(zip(g["scalar_names"].tolist(), g["scalar_values"].tolist()))
    assert set(run.scalars) == set(want)
    for name, value in want.items():
        tol = 0.0 if "hits" in name else 1e-6
        assert run.scalars[name] == pytest.approx(value, abs=tol), name
    assert mrr == pytest.approx(float(g["returned_mrr"]), abs=1e-6)
    assert ent_emb.shape == (1,) + g["ent_emb"].shape
    np.testing.assert_allclose(ent_emb[0].numpy(), g["ent_emb"], rtol=0, atol=0)
    # raw-only evaluation limited to max_num_batches (the per-epoch validation call)
    run2 = _Run()
    ranking.eval_link_prediction(model, loader, text, entities, 3, int(g["emb_batch_size"]), run2, log,
                                 prefix="valid", max_num_batches=5)
    want2 = dict(zip(g["raw_names"].tolist(), g["raw_values"].tolist()))
    assert set(run2.scalars) == set(want2)
    for name, value in want2.items():
        assert run2.scalars[name] == pytest.approx(value, abs=0.0 if "hits" in name else 1e-6), name


@pytest.mark.parametrize("name", golden_names("loss_"))
def test_link_prediction_compute_loss_cpu_route(name):
    g = golden(name)
    _, rel_model, loss_fn, _ = name.split("_")
    nrel, d = g["rel_w"].shape
    model = models.LinkPrediction(d, rel_model, loss_fn, nrel, float(g["regularizer"]))
    model.rel_emb.weight.data = torch.from_numpy(g["rel_w"]).clone()
    ent = torch.from_numpy(g["ent_embs"]).requires_grad_(True)
    loss = model.compute_loss(ent, torch.from_numpy(g["rels"]), torch.from_numpy(g["neg_idx"]))
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(ent.grad.numpy(), g["grad_ent"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(model.rel_emb.weight.grad.numpy(), g["grad_rel_w"], rtol=1e-5, atol=1e-7)


def test_loader_triples_fast_path_equals_iteration():
    """ranking._loader_triples reads a sequential default-collate loader over a dataset that stores its
    triples as one tensor as a slice; it must see exactly what `for i, triples in enumerate(loader)`
    sees (train.py:128-131), including max_num_batches, drop_last, and must fall back to iteration for
    shuffled loaders, custom collates and datasets without a `triples` tensor."""
    from torch.utils.data import DataLoader, Dataset, TensorDataset

    class Triples(Dataset):
        def __init__(self, t):
            self.triples = t

        def __getitem__(self, i):
            return self.triples[i]

        def __len__(self):
            return self.triples.shape[0]

    g = torch.Generator().manual_seed(0)
    t = torch.randint(0, 50, (203, 3), generator=g)

    def iterate(loader, max_num_batches):
        out = []
        for i, batch in enumerate(loader):
            if max_num_batches is not None and i == max_num_batches:
                break
            out.append(batch)
        return torch.cat(out) if out else torch.zeros((0, 3), dtype=torch.long)

    for kw in (dict(batch_size=64), dict(batch_size=64, drop_last=True), dict(batch_size=7)):
        for mb in (None, 0, 2, 100):
            loader = DataLoader(Triples(t), **kw)
            assert torch.equal(ranking._loader_triples(loader, mb), iterate(loader, mb)), (kw, mb)
    # not the fast path: still the same triples as the plain loop
    torch.manual_seed(1)
    shuffled = DataLoader(Triples(t), batch_size=64, shuffle=True)
    assert sorted(map(tuple, ranking._loader_triples(shuffled, None).tolist())) == sorted(map(tuple, t.tolist()))
    plain = DataLoader(TensorDataset(t), batch_size=64, collate_fn=lambda rows: torch.stack([r[0] for r in rows]))
    assert torch.equal(ranking._loader_triples(plain, 2), t[:128])
    assert torch.equal(ranking._loader_triples([t[:5], t[5:9]], None), t[:9])  # any iterable of batches


def test_model_interface_and_errors():
    with pytest.raises(ValueError, match="Unknown relational model"):
        models.LinkPrediction(8, "rotate", "margin", 3, 0)
    with pytest.raises(ValueError, match="Unkown loss function"):
        models.LinkPrediction(8, "transe", "hinge", 3, 0)
    with pytest.raises(ValueError, match="Unkown model"):
        utils.get_model("gpt", 8, "transe", "margin", 5, 3, None, 0)
    with pytest.raises(ValueError, match="Must provided one of"):
        models.BOW("transe", "margin", 3, 0)
    m = models.LinkPrediction(8, "transe", "margin", 3, 0)
    assert m.normalize_embs and m.score_fn is models.transe_score and m.loss_fn is models.margin_loss
    assert not models.LinkPrediction(8, "complex", "nll", 3, 1e-3).normalize_embs
    assert list(m.state_dict()) == ["rel_emb.weight"]
    t = utils.get_model("transductive", 16, "distmult", "nll", 11, 4, None, 0)
    assert sorted(t.state_dict()) == ["ent_emb.weight", "rel_emb.weight"]
    bow = models.BOW("transe", "margin", 3, 1e-2, embeddings=torch.randn(20, 12))
    assert bow.dim == 12 and sorted(bow.state_dict()) == ["embeddings.weight", "rel_emb.weight"]
    dkrl = models.DKRL(16, "transe", "margin", 3, 1e-2, embeddings=torch.randn(20, 12))
    assert {"conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"} <= set(dkrl.state_dict())
    tok = torch.randint(1, 20, (5, 2, 6))
    mask = torch.ones(5, 2, 6)
    assert bow(tok, mask).shape == (10, 12) and dkrl(tok, mask).shape == (10, 16)
    loss = dkrl(tok, mask, torch.randint(0, 3, (5, 1)), data.get_negative_sampling_indices(5, 4))
    assert loss.dim() == 0 and torch.isfinite(loss)
    blp = models.BertEmbeddingsLP(16, "transe", "margin", 3, dict(hidden_size=32, num_hidden_layers=1,
                                  num_attention_heads=2, intermediate_size=64, vocab_size=50), 0)
    assert blp.enc_linear.bias is None and blp(tok, mask).shape == (10, 16)
    assert any(k.startswith("encoder.") for k in blp.state_dict()) and "enc_linear.weight" in blp.state_dict()
    # encode_into (the table build's per-chunk call): on CPU tensors it is encode + a row copy for every model
    for m in (bow, dkrl, blp.eval()):
        rows = torch.full((7, m.dim), 3.0)
        with torch.no_grad():
            m.encode_into(rows[1:6], tok[:, 0], mask[:, 0])
            assert torch.equal(rows[1:6], m.encode(tok[:, 0], mask[:, 0])) and (rows[0] == 3).all() and (rows[6] == 3).all()


def test_sacred_shim_cli_semantics():
    ex = Experiment()

    @ex.config
    def config():
        dataset = 'umls'
        dim = 128
        lr = 2e-5
        checkpoint = None
        use_scheduler = True

    @ex.capture
    def helper(x, dim, _run, _log, prefix=''):
        return x, dim, _run._id, prefix

    @ex.command
    def link_prediction(dataset, dim, lr, checkpoint, use_scheduler, _run, _log):
        _run.log_scalar("m", 1.5, 0)
        return dataset, dim, lr, checkpoint, use_scheduler, helper(7, prefix="p")

    run = ex.run_commandline(["train.py", "link_prediction", "with", "dataset=FB15k-237", "dim=64", "lr=1e-3",
                              "checkpoint=None", "use_scheduler=False"])
    assert run.result == ("FB15k-237", 64, 1e-3, None, False, (7, 64, None, "p"))
    assert run.scalars["m"] == [(0, 1.5)]
    with pytest.raises(KeyError):
        ex.run_commandline(["train.py", "link_prediction", "with", "nope=1"])
    with pytest.raises(KeyError):
        ex.run_commandline(["train.py", "rerank"])


def _golden_encoders(g, rel_model, device="cpu"):
    word_emb = torch.from_numpy(g["word_emb"]).clone()
    dkrl = models.DKRL(128, rel_model, "margin", 3, 0.0, embeddings=word_emb.clone())
    with torch.no_grad():
        dkrl.conv1.weight.copy_(torch.from_numpy(g["conv1_w"])); dkrl.conv1.bias.copy_(torch.from_numpy(g["conv1_b"]))
        dkrl.conv2.weight.copy_(torch.from_numpy(g["conv2_w"])); dkrl.conv2.bias.copy_(torch.from_numpy(g["conv2_b"]))
    bow = models.BOW(rel_model, "margin", 3, 0.0, embeddings=word_emb.clone())
    return dkrl.to(device), bow.to(device)


@pytest.mark.parametrize("name", golden_names("encoders_"))
def test_description_encoders_reproduce_the_reference(name):
    """models.BOW / models.DKRL `encode` (the stock-module expressions a CPU model and every training step run; the fused
    table-build kernels are held against the same fixtures on the GPU) == the reference's encoders on the same weights, tokens
    and masks (models.py:140-204, normalised for TransE: models.py:38-43)."""
    g = golden(name)
    tok, mask = torch.from_numpy(g["tok"]), torch.from_numpy(g["mask"])
    for rel_model in ("transe", "distmult"):
        dkrl, bow = _golden_encoders(g, rel_model)
        with torch.no_grad():
            assert torch.allclose(dkrl.encode(tok, mask), torch.from_numpy(g[f"dkrl_{rel_model}"]), rtol=1e-6, atol=1e-7)
            assert torch.allclose(bow.encode(tok, mask), torch.from_numpy(g[f"bow_{rel_model}"]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rank_triples_widens_a_16_bit_table_on_the_dense_routes(dtype):
    """A 16-bit copy of the table on a CPU tensor (the reference's dense route, no HIP library): ranked as the table widened to
    float32 -- the rule the 16-bit entry of the library follows on the GPU (blp_rank_all_batches; tests/test_gpu_table16.py)."""
    g = torch.Generator().manual_seed(4)
    N, D, T, R = 90, 16, 23, 3
    table = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1).to(dtype)
    model = models.LinkPrediction(D, "transe", "margin", R, 0)
    triples = torch.stack((torch.randint(0, N, (T,), generator=g), torch.randint(0, N, (T,), generator=g),
                           torch.randint(0, R, (T,), generator=g)), dim=1)
    ent2idx = torch.arange(N)
    index = utils.FilterIndex(torch.cat((triples, triples.flip(0)[:, [1, 0, 2]])), num_relations=R)
    _, got, _ = ranking.rank_triples(model, table, triples, ent2idx, index, block_size=8)
    _, want, _ = ranking.rank_triples(model, table.float(), triples, ent2idx, index, block_size=8)
    assert torch.equal(got, want)
