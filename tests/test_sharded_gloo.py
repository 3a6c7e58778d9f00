"""Candidate-axis sharded ranking on CPU with the gloo backend, world_size 2 and 3: per-shard counts
combined by one all-gather equal the unsharded counts (and the oracle's), and the sharded
eval_link_prediction reproduces the reference's scalars.  The collective code path is the one the
GPU run uses with RCCL (backend 'nccl'); only the per-shard scoring differs (CPU route here)."""
import logging
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rel_model, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blp_amd import models, ranking, utils
        from conftest import golden
        from test_host_golden import _Run, toy_eval_setup
        torch.manual_seed(0)
        n, d, q_head, q_tail = 203, 128, 9, 12
        model = models.LinkPrediction(d, rel_model, "margin", 5, 0)
        table = torch.randn(n, d)
        table = torch.nn.functional.normalize(table, dim=-1) if rel_model == "transe" else table * 0.1
        fixed_rows = torch.randint(0, n, (q_head + q_tail,))
        true_rows = torch.randint(0, n, (q_head + q_tail,))
        q_rel = model.rel_emb(torch.randint(0, 5, (q_head + q_tail,))).detach()
        rowptr = torch.arange(0, 3 * (q_head + q_tail) + 1, 3)
        cols = torch.stack([torch.tensor([(int(t) + 1 + j * 17) % n for j in range(3)]) for t in true_rows]).reshape(-1)
        whole = ranking.rank_block(model, table, table[fixed_rows], q_rel, q_head, true_row=true_rows,
                                   filt_rowptr=rowptr, filt_col=cols)
        lo, hi = ranking.shard_bounds(n, world, rank)
        ranker = ranking.ShardedRanker(model, table[lo:hi].clone(), n)
        vectors = ranker.gather_rows(torch.cat((fixed_rows, true_rows)))
        assert torch.equal(vectors, table[torch.cat((fixed_rows, true_rows))])  # exact replication
        q = q_head + q_tail
        ranker.rank_block(vectors[:q], q_rel, vectors[q:], q_head, rowptr, cols)
        ranker.rank_block(vectors[:q_head], q_rel[:q_head], vectors[q:q + q_head], q_head)  # a second block
        total = ranker.finish()
        assert torch.equal(total[:q], whole), "sharded counts differ from unsharded"
        assert torch.equal(total[q:, :2], whole[:q_head, :2])
        # the full evaluation loop, sharded, against the reference's scalars
        g = golden(f"eval_toy_{rel_model}")
        emodel, text, loader, index, entities, new_ents = toy_eval_setup(g, rel_model)
        run = _Run()
        mrr, ent_emb = ranking.eval_link_prediction(emodel, loader, text, entities, 3, int(g["emb_batch_size"]), run,
                                                    logging.getLogger("t"), prefix="test", filtering_graph=index,
                                                    new_entities=new_ents, return_embeddings=True, block_size=16)
        want = dict(zip(g["scalar_names"].tolist(), g["scalar_values"].tolist()))
        for name, value in want.items():
            assert abs(run.scalars[name] - value) <= (0.0 if "hits" in name else 1e-6), (name, run.scalars[name], value)
        assert np.array_equal(ent_emb[0].numpy(), g["ent_emb"])
        # the other axis: table replicated by one all-gather, the test triples sharded across ranks
        run_q = _Run()
        mrr_q, ent_emb_q = ranking.eval_link_prediction(emodel, loader, text, entities, 3, int(g["emb_batch_size"]),
                                                        run_q, logging.getLogger("t"), prefix="test",
                                                        filtering_graph=index, new_entities=new_ents,
                                                        return_embeddings=True, block_size=5, shard_axis="query")
        assert run_q.scalars == run.scalars and mrr_q == mrr
        assert np.array_equal(ent_emb_q[0].numpy(), g["ent_emb"])
        assert ranking.choose_shard_axis(14541, 128, 105740, 8) == "query"        # FB15k-237 test set
        assert ranking.choose_shard_axis(4_600_000, 128, 10266, 8) == "candidate"  # Wikidata5M
        assert ranking.choose_shard_axis(14541, 128, 128, 8) == "candidate"        # one reference batch
        np.save(os.path.join(out_dir, f"counts_{rank}.npy"), total.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("rel_model", ["transe", "complex"])
def test_sharded_counts_and_eval(tmp_path, world, rel_model):
    mp.spawn(_worker, args=(world, _free_port(), rel_model, str(tmp_path)), nprocs=world, join=True)
    counts = [np.load(tmp_path / f"counts_{r}.npy") for r in range(world)]
    for c in counts[1:]:
        assert np.array_equal(c, counts[0])  # every rank ends with the same global counts


def test_sharded_counts_match_oracle(oracle):
    """Unsharded CPU route vs the C oracle on the same inputs as the workers (bit-exact counts)."""
    from blp_amd import models, ranking
    torch.manual_seed(0)
    n, d, q_head, q_tail = 203, 128, 9, 12
    model = models.LinkPrediction(d, "transe", "margin", 5, 0)
    table = torch.nn.functional.normalize(torch.randn(n, d), dim=-1)
    fixed_rows = torch.randint(0, n, (q_head + q_tail,))
    true_rows = torch.randint(0, n, (q_head + q_tail,))
    q_rel = model.rel_emb(torch.randint(0, 5, (q_head + q_tail,))).detach()
    got = ranking.rank_block(model, table, table[fixed_rows], q_rel, q_head, true_row=true_rows).numpy()
    t, f, r = table.numpy(), table[fixed_rows].numpy(), q_rel.numpy()
    want = np.concatenate((oracle.rank_counts("transe", 0, t, f[:q_head], r[:q_head], true_row=true_rows[:q_head].numpy()),
                           oracle.rank_counts("transe", 1, t, f[q_head:], r[q_head:], true_row=true_rows[q_head:].numpy())))
    assert np.array_equal(got, want)
