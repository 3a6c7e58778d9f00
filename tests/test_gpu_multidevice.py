"""compute_loss (models.py:51-70) the way the reference's training wrapper runs it: nn.DataParallel replicas, one Python
thread per replica (train.py:329-330, 344), each on its device-local batch with device-local negative indices
(data.py:289-298) -- and the same split as two processes.  On a box with one GPU both replicas sit on device 0 (the
threads, the scatter / replicate / gather plumbing and the per-call `device` / `stream` arguments are exercised all the
same); with two or more GPUs they sit on devices 0 and 1."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REL_MODELS, golden, golden_names

pytestmark = [pytest.mark.gpu, pytest.mark.default_routing]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_halves(rel_model, loss_fn, ent, rel_w, rels, local_negs, reg):
    """The reference's expressions (oracle/ref_port.py, CPU) on every device-local batch: mean loss over the replicas, and
    the gradients of that mean."""
    from oracle import ref_port
    e = ent.clone().requires_grad_(True)
    w = rel_w.clone().requires_grad_(True)
    per = ent.shape[0] // len(local_negs)
    losses = [ref_port.compute_loss(rel_model, loss_fn, e[i * per:(i + 1) * per], w[rels[i * per:(i + 1) * per]].unsqueeze(1), neg, reg)
              for i, neg in enumerate(local_negs)]
    total = torch.stack(losses).mean()
    total.backward()
    return total.item(), e.grad, w.grad


@pytest.mark.parametrize("rel_model", REL_MODELS)
@pytest.mark.parametrize("loss_fn", ["margin", "nll"])
def test_compute_loss_under_data_parallel(rel_model, loss_fn):
    """TransductiveLinkPrediction under nn.DataParallel, two replicas: `net(pairs, rels, neg_idx).mean()` and its gradients
    on ent_emb.weight / rel_emb.weight equal the reference's expressions evaluated per device-local batch on the CPU."""
    from blp_amd import models
    torch.manual_seed(5)
    n_dev = 2
    B, K, D, E, R = 32, 16, 128, 50, 7
    reg = 1e-3 if rel_model == "complex" else 0.0
    net = models.TransductiveLinkPrediction(D, rel_model, loss_fn, E, R, reg)
    pairs = torch.randint(0, E, (B, 2))
    rels = torch.randint(0, R, (B, 1))
    per = B // n_dev
    local_negs = [torch.randint(0, 2 * per, (per, K, 2)) for _ in range(n_dev)]
    neg_idx = torch.cat(local_negs)  # scattered along dim 0: replica i gets local_negs[i], indices local to ITS 2 * per rows
    ent_w, rel_w = net.ent_emb.weight.detach().clone(), net.rel_emb.weight.detach().clone()
    ent = ent_w[pairs]
    if rel_model == "transe":
        ent = torch.nn.functional.normalize(ent, dim=-1)
    # reference: through the embedding lookup (+ normalisation) so that the gradient lands on the tables
    from oracle import ref_port
    ew, rw = ent_w.clone().requires_grad_(True), rel_w.clone().requires_grad_(True)
    losses = []
    for i, neg in enumerate(local_negs):
        e = ew[pairs[i * per:(i + 1) * per]]
        e = torch.nn.functional.normalize(e, dim=-1) if rel_model == "transe" else e
        losses.append(ref_port.compute_loss(rel_model, loss_fn, e, rw[rels[i * per:(i + 1) * per, 0]].unsqueeze(1), neg, reg))
    want = torch.stack(losses).mean()
    want.backward()

    ids = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    dp = torch.nn.DataParallel(net.cuda(), device_ids=ids)
    got = dp(pairs.cuda(), rels.cuda(), neg_idx.cuda())
    assert got.shape == (n_dev,)  # one loss per replica, gathered on device 0 (train.py:344 takes their mean)
    loss = got.mean()
    loss.backward()
    assert loss.item() == pytest.approx(want.item(), rel=2e-6, abs=1e-7)
    np.testing.assert_allclose(net.ent_emb.weight.grad.cpu().numpy(), ew.grad.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(net.rel_emb.weight.grad.cpu().numpy(), rw.grad.numpy(), rtol=2e-5, atol=2e-7)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices")
def test_compute_loss_on_the_second_device_matches_the_first():
    """The same batch on cuda:0 and on cuda:1 (the C-ABI's explicit `device` argument, DeviceGuard): bit-equal loss and
    gradients -- and the current device of the calling thread is left alone."""
    from blp_amd import ops
    torch.manual_seed(9)
    B, K, D = 64, 64, 128
    ent, rel = torch.randn(B, 2, D) * 0.4, torch.randn(B, 1, D) * 0.3
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    out = []
    for d in (0, 1):
        dev = torch.device("cuda", d)
        e, r = ent.to(dev).requires_grad_(True), rel.to(dev).requires_grad_(True)
        loss = ops.inbatch_loss("complex", "margin", e, r, neg_idx.to(dev), 1e-3)  # (current device stays 0 throughout)
        loss.backward()
        out.append((loss.cpu(), e.grad.cpu(), r.grad.cpu()))
        assert torch.cuda.current_device() == 0
    assert all(torch.equal(a, b) for a, b in zip(*out))


@pytest.mark.parametrize("name", [n for n in golden_names("loss_") if n.endswith("_reg1e-3")])
def test_compute_loss_on_two_gloo_ranks_equals_the_reference(name, tmp_path):
    """Two processes (gloo; sharing this GPU unless there are two), each with its half of a golden batch and negatives local
    to that half: the all-reduced loss and gradients equal the reference's expressions on the two halves; and the WHOLE
    golden batch, run by rank 1 while rank 0 runs it too, gives the reference's own loss and gradients (tests/golden)."""
    out = tmp_path / "pieces.pt"
    port = 29700 + (hash(name) % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "loss_worker.py"), name, str(out)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    got = torch.load(out)
    g = golden(name)
    _, rel_model, loss_fn, _ = name.split("_")
    ent, rel_w = torch.from_numpy(g["ent_embs"]), torch.from_numpy(g["rel_w"])
    rels, neg_idx = torch.from_numpy(g["rels"]), torch.from_numpy(g["neg_idx"])
    per = ent.shape[0] // 2
    local_negs = [torch.randint(0, 2 * per, (per, neg_idx.shape[1], 2), generator=torch.Generator().manual_seed(100 + r)) for r in range(2)]
    want_loss, want_e, want_w = _reference_halves(rel_model, loss_fn, ent, rel_w, rels.reshape(-1), local_negs, float(g["regularizer"]))
    assert got["loss"].item() / 2 == pytest.approx(want_loss, rel=2e-6, abs=1e-7)
    np.testing.assert_allclose(got["grad_ent"].numpy() / 2, want_e.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(got["grad_rel_w"].numpy() / 2, want_w.numpy(), rtol=2e-5, atol=2e-7)
    assert got["full_loss"].item() == pytest.approx(float(g["loss"]), rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(got["full_grad_ent"].numpy(), g["grad_ent"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["full_grad_rel_w"].numpy(), g["grad_rel_w"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------------------------------------
# The EVALUATION on several devices from one process: one thread per device (blp_amd.multidevice), each with its replica, its
# rows of the entity table, its stream; counts combined by one all-gather + sum.  One GPU here: two (three)
# shards on device 0 -- the threads, the per-call device / stream arguments, the shard form of the kernels and the exchange
# are all exercised; with >= 2 GPUs the devices are distinct and the exchange is RCCL's group launch.
def _device_list(n):
    have = torch.cuda.device_count()
    return list(range(n)) if have >= n else [0] * n


@pytest.mark.parametrize("rel_model", REL_MODELS)
@pytest.mark.parametrize("n_dev,axis", [(2, "auto"), (3, "candidate"), (2, "query")])
def test_eval_link_prediction_on_device_threads_matches_reference(rel_model, n_dev, axis):
    """eval_link_prediction(devices=[...]) through the HIP path: the reference's own scalar dict (golden, generated by the
    imported reference) and its returned embeddings."""
    import logging
    from blp_amd import ops, ranking
    from test_host_golden import _Run, toy_eval_setup
    g = golden(f"eval_toy_{rel_model}")
    model, text, loader, index, entities, new_ents = toy_eval_setup(g, rel_model, device="cuda")
    run = _Run()
    mrr, ent_emb = ranking.eval_link_prediction(model, loader, text, entities, 3, int(g["emb_batch_size"]), run, logging.getLogger("t"),
                                                prefix="test", filtering_graph=index, new_entities=new_ents, return_embeddings=True,
                                                block_size=16, devices=_device_list(n_dev), shard_axis=axis)
    want = dict(zip(g["scalar_names"].tolist(), g["scalar_values"].tolist()))
    assert set(run.scalars) == set(want)
    for name, value in want.items():
        assert run.scalars[name] == pytest.approx(value, abs=1e-6), name   # (the tolerance: the GPU-built table, test_gpu_eval.py)
    assert ent_emb.device == torch.device("cuda", 0)
    np.testing.assert_allclose(ent_emb[0].cpu().numpy(), g["ent_emb"], rtol=1e-6, atol=1e-7)
    assert not ops._workspaces


@pytest.mark.parametrize("model,N,T,block", [("transe", 3000, 900, 65536), ("complex", 40000, 24, 2), ("distmult", 3000, 2600, 1024),
                                             ("simple", 50000, 7, 2)])
def test_device_threads_counts_equal_the_oracle(oracle, model, N, T, block):
    """ranking.rank_triples on the candidate axis from device threads (big blocks: the pre-pass kernels on a shard; the
    reference's Wikidata5M batching, 2 triples per pass: the streaming ring kernels on a shard): every thread ends with the
    ORACLE's counts, raw and filtered; the exchanges issued are the plan's."""
    from blp_amd import models, multidevice, ranking, utils
    from test_gpu_shard import _oracle_counts, _problem
    D, R, world = 128, 5, 3
    table, rel_w, ent2idx, triples, edges = _problem(model, N, D, T, R, seed=N + T)
    index = utils.FilterIndex(edges, num_relations=R)
    want = _oracle_counts(oracle, model, table, rel_w, ent2idx, triples, index)
    net = models.LinkPrediction(D, model, "margin", R, 0)
    net.rel_emb.weight.data = rel_w.clone()
    group = multidevice.DeviceGroup(_device_list(world))
    replicas = [net.cuda() if d == group.devices[0] else None for d in group.devices]
    import copy
    replicas = [r if r is not None else copy.deepcopy(net).to(d) for r, d in zip(replicas, group.devices)]

    def shard(m):
        lo, hi = ranking.shard_bounds(N, world, m.rank)
        _, counts, ok = ranking.rank_triples(replicas[m.rank], table[lo:hi].to(m.device), triples.to(m.device), ent2idx.to(m.device), index,
                                             num_entities=N, group=m, world=world, rank=m.rank, axis="candidate", block_size=block)
        return counts.cpu().numpy(), bool(ok)

    for counts, ok in group.run(shard):
        assert ok and np.array_equal(counts, want)
    plan = ranking.exchange_plan(N, D, T, world, "candidate")
    assert [(op, nbytes) for op, nbytes in group.issued] == [(p["op"], p["bytes_per_rank"]) for p in plan]


@pytest.mark.default_routing
def test_link_prediction_cli_evaluates_on_device_threads(tmp_path):
    """python train.py link_prediction ... eval_devices=[0,0]: the reference's launch (one process), every evaluation sharded
    over the listed devices by threads -- and the same numbers as the run that evaluates on one device."""
    import json
    from blp_amd.data import write_synthetic_dataset
    write_synthetic_dataset(str(tmp_path / "data"), "umls-synth", num_entities=135, num_relations=46,
                            num_train=1280, num_valid=160, num_test=160, vocab_size=500, emb_dim=128, seed=0)
    base = [sys.executable, os.path.join(ROOT, "train.py"), "link_prediction", "with", "dataset=umls-synth", "inductive=False",
            "model=glove-bow", "rel_model=transe", "loss_fn=margin", "regularizer=1e-2", "max_len=32", "num_negatives=16", "lr=1e-3",
            "use_scheduler=False", "batch_size=64", "emb_batch_size=512", "eval_batch_size=64", "max_epochs=1",
            f"data_root={tmp_path / 'data'}", "seed=1"]
    scalars = {}
    for name, extra in (("threads", ["eval_devices=[0,0]"] if torch.cuda.device_count() < 2 else []), ("single", ["eval_devices=[0]"])):
        cwd = tmp_path / name
        cwd.mkdir()
        # (on a multi-GPU box both runs train alike -- DataParallel over all devices --; "threads" then evaluates on all of them
        #  by default, "single" on one)
        proc = subprocess.run(base + extra, cwd=cwd, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900)
        assert proc.returncode == 0, proc.stderr[-3000:]
        log = proc.stderr + proc.stdout
        n_threads = max(torch.cuda.device_count(), 2) if name == "threads" else 1
        assert f"Evaluating on {n_threads} device thread(s)" in log, log[-2000:]
        scalars[name] = json.load(open(cwd / "output" / "scalars-None.json"))
    for key, value in scalars["single"].items():
        if key.startswith(("valid_", "test_", "train_mrr", "train_hits")):
            assert scalars["threads"][key] == pytest.approx(value, abs=1e-6), key


@pytest.mark.default_routing
def test_link_prediction_cli_under_torchrun_on_the_gpu(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 train.py ... on this box: with two GPUs the ranks take cuda:0 / cuda:1
    and exchange through RCCL; with one they share it and exchange through gloo (RCCL refuses two ranks on a device) --
    either way every rank trains on its slice of the batch (fused in-batch loss), evaluates its candidate shard through the
    HIP ranking, and rank 0's scalars equal a single process's on the checkpoint the job saved."""
    import json
    import socket
    from blp_amd.data import write_synthetic_dataset
    write_synthetic_dataset(str(tmp_path / "data"), "umls-synth", num_entities=135, num_relations=46,
                            num_train=1280, num_valid=160, num_test=160, vocab_size=500, emb_dim=128, seed=0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    args = ["link_prediction", "with", "dataset=umls-synth", "inductive=False", "model=glove-bow", "rel_model=transe", "loss_fn=margin",
            "regularizer=1e-2", "max_len=32", "num_negatives=16", "lr=1e-3", "use_scheduler=False", "batch_size=64", "emb_batch_size=512",
            "eval_batch_size=64", f"data_root={tmp_path / 'data'}", "seed=1"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = tmp_path / "two"
    two.mkdir()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "train.py"), *args, "max_epochs=1"]
    proc = subprocess.run(cmd, cwd=two, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    log = proc.stderr + proc.stdout
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert f"2 processes, backend {backend}" in log, log[-2000:]
    sharded = json.load(open(two / "output" / "scalars-None.json"))
    one = tmp_path / "one"
    one.mkdir()
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), *args, "max_epochs=0", f"checkpoint={two / 'output' / 'model-None.pt'}",
           "eval_devices=[0]"]
    proc = subprocess.run(cmd, cwd=one, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    single = json.load(open(one / "output" / "scalars-None.json"))
    compared = [k for k in single if k.startswith(("valid_", "test_"))]
    assert len(compared) == 22
    for key in compared:
        assert sharded[key] == pytest.approx(single[key], abs=1e-6), key
