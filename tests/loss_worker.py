"""Worker of tests/test_gpu_multidevice.py::test_compute_loss_on_two_gloo_ranks_equals_the_golden_gradients: one process
per rank (torch.distributed.run, gloo; the ranks share GPU 0 unless there is a device per rank).  Each rank takes ITS slice
of a golden batch -- the reference's per-device batch with device-local negative indices (data.py:289-298) --, runs
LinkPrediction.compute_loss through the fused HIP kernels and all-reduces loss and gradients the way DDP would.
    python -m torch.distributed.run --nproc-per-node 2 ... tests/loss_worker.py GOLDEN_NAME OUT.pt"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    from conftest import golden
    from blp_amd import models
    g = golden(name)
    _, rel_model, loss_fn, _ = name.split("_")
    ent, rel_w = torch.from_numpy(g["ent_embs"]), torch.from_numpy(g["rel_w"])
    rels, neg_idx = torch.from_numpy(g["rels"]), torch.from_numpy(g["neg_idx"])
    B, D = ent.shape[0], ent.shape[2]
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    model = models.LinkPrediction(D, rel_model, loss_fn, rel_w.shape[0], float(g["regularizer"]))
    model.rel_emb.weight.data = rel_w.clone()
    model = model.to(device)
    # this rank's device-local batch: its own pairs, negatives drawn among ITS 2 * per rows (local indices)
    gen = torch.Generator().manual_seed(100 + rank)
    local_neg = torch.randint(0, 2 * per, (per, neg_idx.shape[1], 2), generator=gen)
    e = ent[lo:hi].to(device).requires_grad_(True)
    loss = model.compute_loss(e, rels[lo:hi].reshape(per, 1).to(device), local_neg.to(device))
    loss.backward()
    pieces = {"loss": loss.detach().cpu().reshape(1), "grad_ent": torch.zeros_like(ent), "grad_rel_w": model.rel_emb.weight.grad.cpu()}
    pieces["grad_ent"][lo:hi] = e.grad.cpu()
    for t in pieces.values():  # what DDP's gradient averaging amounts to (sum here; the test divides)
        dist.all_reduce(t)
    # the whole golden batch with the reference's own indices, on every rank at once (two processes, one GPU); rank 1 reports
    model.zero_grad()
    e = ent.to(device).requires_grad_(True)
    full = model.compute_loss(e, rels.to(device), neg_idx.to(device))
    full.backward()
    mine = {"full_loss": full.detach().cpu().reshape(1), "full_grad_ent": e.grad.cpu(), "full_grad_rel_w": model.rel_emb.weight.grad.cpu()}
    for key, t in mine.items():
        t = t if rank == world - 1 else torch.zeros_like(t)
        dist.all_reduce(t)
        pieces[key] = t
    if rank == 0:
        torch.save(pieces, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
