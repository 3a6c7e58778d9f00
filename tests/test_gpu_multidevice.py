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
