"""The oracle (oracle/blp_oracle.c, oracle/ref_port.py) pinned against golden vectors produced by the
imported reference (tests/golden/make_golden.py).  Bit-exact for scores / counts; CPU only."""
import numpy as np
import pytest
import torch

from conftest import REL_MODELS, golden, golden_names
from oracle import ref_port


def _queries(g):
    table, rel_w = g["table"], g["rel_w"]
    heads, tails, rels = g["heads"][:, 0], g["tails"][:, 0], g["rels"][:, 0]
    return table, table[heads], table[tails], rel_w[rels], heads, tails


@pytest.mark.parametrize("name", golden_names("scores_"))
def test_c_oracle_scores_bit_exact(oracle, name):
    g = golden(name)
    model = name.split("_")[1]
    table, h, t, r, heads, tails = _queries(g)
    head_pred = oracle.score_all(model, oracle.SIDE_HEAD, table, t, r)
    tail_pred = oracle.score_all(model, oracle.SIDE_TAIL, table, h, r)
    assert np.array_equal(head_pred.view(np.uint32), g["head_pred"].view(np.uint32))
    assert np.array_equal(tail_pred.view(np.uint32), g["tail_pred"].view(np.uint32))


@pytest.mark.parametrize("name", golden_names("scores_"))
def test_c_oracle_counts_and_metrics(oracle, name):
    g = golden(name)
    model = name.split("_")[1]
    table, h, t, r, heads, tails = _queries(g)
    ch = oracle.rank_counts(model, oracle.SIDE_HEAD, table, t, r, true_row=heads)
    ct = oracle.rank_counts(model, oracle.SIDE_TAIL, table, h, r, true_row=tails)
    counts = np.concatenate((ch, ct))
    assert np.array_equal(counts[:, 0], g["gt"])
    assert np.array_equal(counts[:, 1], g["ge"])
    # no filter given -> filtered counts equal raw counts
    assert np.array_equal(counts[:, 2:], counts[:, :2])
    rr, hits = oracle.metrics_from_counts(counts[:, 0], counts[:, 1])
    assert np.array_equal(rr.view(np.uint32), g["rr"][:, 0].view(np.uint32))
    assert np.array_equal(hits, g["hits"])
    # replicated true-entity vector (sharded form) gives the same counts
    ch2 = oracle.rank_counts(model, oracle.SIDE_HEAD, table, t, r, q_true=table[heads])
    assert np.array_equal(ch, ch2)


@pytest.mark.parametrize("name", golden_names("scores_"))
def test_torch_port_scores_bit_exact(name):
    g = golden(name)
    model = name.split("_")[1]
    table = torch.from_numpy(g["table"])
    rel_w = torch.from_numpy(g["rel_w"])
    heads, tails, rels = (torch.from_numpy(g[k]) for k in ("heads", "tails", "rels"))
    fn = ref_port.SCORE_FNS[model]
    ent = table.unsqueeze(0)
    head_pred = fn(ent, table[tails], rel_w[rels]).numpy()
    tail_pred = fn(table[heads], ent, rel_w[rels]).numpy()
    assert np.array_equal(head_pred.view(np.uint32), g["head_pred"].view(np.uint32))
    assert np.array_equal(tail_pred.view(np.uint32), g["tail_pred"].view(np.uint32))
    out = ref_port.eval_batch(model, table, heads[:, 0], tails[:, 0], rel_w[rels[:, 0]])
    assert np.array_equal(out["gt"].numpy(), g["gt"])
    assert np.array_equal(out["ge"].numpy(), g["ge"])
    assert np.array_equal(out["rr"].numpy(), g["rr"][:, 0])


def test_sum_order_all_widths(oracle):
    g = golden("sum_order")
    for n in (8, 32, 64, 100, 128, 200, 300, 512, 768, 1024, 2080):
        x = g[f"x_{n}"]
        got = np.array([oracle.torch_inner_sum(row) for row in x], np.float32)
        assert np.array_equal(got.view(np.uint32), g[f"sum_{n}"].view(np.uint32)), n
        # torch.norm(p=1) = strict left-to-right sum: TransE with h = x, r = t = 0
        z = np.zeros_like(x)
        l1 = -oracle.score_pairs("transe", x, z, z)
        assert np.array_equal(l1.view(np.uint32), g[f"l1_{n}"].view(np.uint32)), n


def test_score_pairs_wide_and_training_shapes(oracle):
    g = golden("score_pairs")
    for d in (300, 768):
        got = oracle.score_pairs("transe", g[f"h_{d}"], g[f"t_{d}"], g[f"r_{d}"])
        assert np.array_equal(got.view(np.uint32), g[f"transe_{d}"].view(np.uint32))
    for model in REL_MODELS:
        h, t, r = g[f"train_h_{model}"], g[f"train_t_{model}"], g[f"train_r_{model}"]
        b, k, d = h.shape
        rb = np.broadcast_to(r, h.shape).reshape(-1, d)
        got = oracle.score_pairs(model, h.reshape(-1, d), t.reshape(-1, d), rb).reshape(b, k)
        assert np.array_equal(got.view(np.uint32), g[f"train_{model}"].view(np.uint32))


@pytest.mark.parametrize("name", golden_names("loss_"))
def test_torch_port_loss_and_grads(name):
    g = golden(name)
    _, model, loss_fn, _ = name.split("_")
    ent = torch.from_numpy(g["ent_embs"]).requires_grad_(True)
    rel_w = torch.from_numpy(g["rel_w"]).requires_grad_(True)
    rels = torch.from_numpy(g["rels"])
    neg_idx = torch.from_numpy(g["neg_idx"])
    loss = ref_port.compute_loss(model, loss_fn, ent, rel_w[rels], neg_idx, float(g["regularizer"]))
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(ent.grad.numpy(), g["grad_ent"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rel_w.grad.numpy(), g["grad_rel_w"], rtol=1e-5, atol=1e-7)
