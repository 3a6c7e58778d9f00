"""One process, one thread per device (blp_amd.multidevice) -- the reference's own process model for several GPUs
(nn.DataParallel, train.py:329-330,344) applied to the evaluation -- on CPU "devices": the threads, the two collectives
of an evaluation, failure propagation, and eval_link_prediction(devices=[...]) against the reference's
scalars.  The GPU side (two shards on one GPU through the HIP path) is tests/test_gpu_multidevice.py."""
import logging
import threading

import numpy as np
import pytest
import torch

from blp_amd import multidevice, ranking
from conftest import golden
from test_host_golden import _Run, toy_eval_setup


def test_collectives_between_device_threads():
    group = multidevice.DeviceGroup(["cpu", "cpu", "cpu"])
    assert group.exchange == "copy" and group.world == 3

    def work(m):
        assert threading.current_thread().name == ("MainThread" if m.rank == 0 else f"blp-device-{m.rank}")
        x = torch.full((5,), float(m.rank + 1))
        m.all_reduce(x)
        part = torch.arange(4, dtype=torch.int32) + 10 * m.rank
        full = torch.empty(12, dtype=torch.int32)
        m.all_gather_into(full, part)
        m.barrier()
        return x, full

    out = group.run(work)
    for x, full in out:
        assert torch.equal(x, torch.full((5,), 6.0))
        assert full.tolist() == [0, 1, 2, 3, 10, 11, 12, 13, 20, 21, 22, 23]
    assert group.issued == [("all_reduce", 20), ("all_gather", 16)]
    # the helpers blp_amd.ranking issues its exchanges through take a member as `group`
    rows = group.run(lambda m: ranking.all_gather_rows(torch.full((ranking.shard_bounds(7, 3, m.rank)[1] - ranking.shard_bounds(7, 3, m.rank)[0], 2),
                                                                  float(m.rank)), 7, 3, m))
    for r in rows:
        assert r[:, 0].tolist() == [0, 0, 0, 1, 1, 1, 2]


def test_a_failing_thread_does_not_leave_the_others_waiting():
    group = multidevice.DeviceGroup(["cpu", "cpu"])

    def work(m):
        if m.rank == 1:
            raise KeyError("shard 1 broke")
        m.all_reduce(torch.zeros(1))  # would wait for rank 1 forever

    with pytest.raises(KeyError, match="shard 1 broke"):
        group.run(work)
    # and the group is usable again
    assert [int(x) for x in group.run(lambda m: m.rank)] == [0, 1]


def test_a_failed_rccl_group_launch_falls_back_to_peer_copies_and_says_so():
    """RCCL's single-process group launch (torch.cuda.nccl) has never run between two GPUs of this package's: if it raises on a
    box, the evaluation must not die in the reference's default launch -- the group says so in a warning (and in
    `nccl_error`), exchanges by peer copies from then on, and the numbers are the same (here: the launch is MADE to fail by
    handing it host tensors)."""
    group = multidevice.DeviceGroup(["cpu", "cpu"])
    group.exchange = "nccl"  # (what "auto" picks for distinct HIP devices)

    def work(m):
        x = torch.full((3,), float(m.rank + 1))
        m.all_reduce(x)
        full = torch.empty(4, dtype=torch.int32)
        m.all_gather_into(full, torch.tensor([m.rank, 7], dtype=torch.int32))
        return x, full

    with pytest.warns(UserWarning, match="RCCL group launch failed"):
        out = group.run(work)
    assert group.exchange == "copy" and group.nccl_error
    for x, full in out:
        assert x.tolist() == [3.0, 3.0, 3.0] and full.tolist() == [0, 7, 1, 7]


def test_device_list_checks():
    with pytest.raises(ValueError):
        multidevice.DeviceGroup([])
    with pytest.raises(ValueError, match="distinct"):
        multidevice.DeviceGroup(["cpu", "cpu"], exchange="nccl")
    with pytest.raises(ValueError, match="not both"):
        g = golden("eval_toy_transe")
        model, text, loader, index, entities, _ = toy_eval_setup(g, "transe")
        ranking.eval_link_prediction(model, loader, text, entities, 0, 8, _Run(), logging.getLogger("t"), devices=["cpu"], group=object())


@pytest.mark.parametrize("rel_model", ["transe", "complex", "simple"])
@pytest.mark.parametrize("devices,axis", [(["cpu", "cpu"], "auto"), (["cpu"] * 3, "candidate"), (["cpu"] * 3, "query"), (["cpu"], "auto")])
def test_eval_on_device_threads_reproduces_reference_scalars(rel_model, devices, axis):
    """eval_link_prediction(devices=[...]): every thread encodes its rows of the table with its replica and ranks them, the
    counts are combined by one all-gather + sum -- the reference's scalars (golden, generated from the imported reference)
    and its returned embeddings, whatever the number of shards and the axis."""
    g = golden(f"eval_toy_{rel_model}")
    model, text, loader, index, entities, new_ents = toy_eval_setup(g, rel_model)
    run = _Run()
    mrr, ent_emb = ranking.eval_link_prediction(model, loader, text, entities, 3, int(g["emb_batch_size"]), run, logging.getLogger("t"),
                                                prefix="test", filtering_graph=index, new_entities=new_ents, return_embeddings=True,
                                                block_size=16, devices=devices, shard_axis=axis)
    want = dict(zip(g["scalar_names"].tolist(), g["scalar_values"].tolist()))
    assert set(run.scalars) == set(want)
    for name, value in want.items():
        assert run.scalars[name] == pytest.approx(value, abs=0.0 if "hits" in name else 1e-6), name
    assert mrr == pytest.approx(float(g["returned_mrr"]), abs=1e-6)
    assert np.array_equal(ent_emb[0].numpy(), g["ent_emb"])


def test_counts_of_device_threads_equal_unsharded_counts():
    """rank_triples with a thread member as `group`: every member ends with the unsharded counts, and the exchanges issued
    are the plan's (one replication of the queries' vectors, ONE all-gather of the (2T, 4) int32 counts)."""
    from blp_amd import models, utils
    torch.manual_seed(3)
    N, D, T, R, world = 301, 128, 40, 5, 3
    model = models.LinkPrediction(D, "distmult", "margin", R, 0)
    table = torch.randn(N, D) * 0.1
    triples = torch.stack((torch.randint(0, N, (T,)), torch.randint(0, N, (T,)), torch.randint(0, R, (T,))), dim=1)
    ent2idx = torch.arange(N)
    index = utils.FilterIndex(torch.cat((triples, torch.stack((triples[:, 0], torch.randint(0, N, (T,)), triples[:, 2]), dim=1))), num_relations=R)
    _, whole, _ = ranking.rank_triples(model, table, triples, ent2idx, index, block_size=16)
    group = multidevice.DeviceGroup(["cpu"] * world)

    def shard(m):
        lo, hi = ranking.shard_bounds(N, world, m.rank)
        return ranking.rank_triples(model, table[lo:hi].clone(), triples, ent2idx, index, num_entities=N, group=m, world=world,
                                    rank=m.rank, axis="candidate", block_size=16)[1]

    for counts in group.run(shard):
        assert torch.equal(counts, whole)
    assert [op for op, _ in group.issued] == ["all_reduce", "all_gather"]
    assert group.issued[-1] == ("all_gather", 2 * T * 16)


def test_table16_decision_is_the_same_on_every_rank():
    """a per-rank decision whether the 16-bit copy of the table is ranked could differ between ranks (the last
    candidate shard is shorter; the library's routing has a row threshold) -- mixed inputs in the summed counts and
    mismatched collectives.  table16_everywhere evaluates every rank's shard, so all ranks agree by construction."""
    from blp_amd import ops
    D, T, block = 128, 6, 2
    # find a world size whose last shard falls on the other side of the library's threshold than the others
    straddle = None
    for n in range(49100, 49200):  # (rank_small takes <= 16 384 rows at <= 4 triples per pass: 3 shards of ~16 384 rows)
        for world in (3,):
            per = [ranking.shard_bounds(n, world, r) for r in range(world)]
            each = [ops.table16_is_read_directly("transe", torch.float16, hi - lo, D, T, block) for lo, hi in per]
            if any(each) and not all(each):
                straddle = (n, world, each)
                break
        if straddle:
            break
    assert straddle is not None, "no shape straddles the threshold: the routing changed; revisit this test"
    n, world, each = straddle
    assert ranking.table16_everywhere("transe", torch.float16, n, D, T, block, world, "candidate") is False
    assert ranking.table16_everywhere("transe", torch.float16, 4_600_000, D, T, block, 8, "candidate") is True
    assert ranking.table16_everywhere("transe", torch.float16, 4_600_000, D, T, 65536, 8, "candidate") is False  # big blocks: not native
