"""Micro-benchmark of the in-batch-negatives loss step (forward + backward) on one GPU:
fused HIP kernels (blp_amd.ops.inbatch_loss) vs the reference's torch expressions (oracle/ref_port.py,
the same ops the unmodified reference launches through PyTorch-ROCm).  Shapes from the reference's
scripts: B = 64 (FB15k-237) and B = 128 (Wikidata5M per GPU), K = 64, D = 128.
    python tools/bench_inbatch.py
This path moves ~100 KB per step and is launch/latency-bound: the figure of merit is microseconds
per step and launches per step, not a roofline fraction."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops  # noqa: E402
from oracle import ref_port  # noqa: E402  (baseline leg: the thing timed)


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    torch.manual_seed(0)
    print(f"{'model':9s} {'loss':7s} {'B':>4s} | fused fwd+bwd us | torch fwd+bwd us | speedup")
    for B in (64, 128):
        K, D = 64, 128
        for model in ("transe", "distmult", "complex", "simple"):
            for loss_fn in ("margin", "nll"):
                ent = torch.randn(B, 2, D, device="cuda") * 0.4
                rel = torch.randn(B, 1, D, device="cuda") * 0.3
                neg_idx = torch.randint(0, 2 * B, (B, K, 2), device="cuda")
                reg = 1e-3 if model == "complex" else 0.0

                def fused():
                    e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
                    ops.inbatch_loss(model, loss_fn, e, r, neg_idx, reg).backward()

                def stock():
                    e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
                    ref_port.compute_loss(model, loss_fn, e, r, neg_idx, reg).backward()

                a, b = timeit(fused), timeit(stock)
                print(f"{model:9s} {loss_fn:7s} {B:4d} | {a:16.1f} | {b:16.1f} | {b / a:5.2f}x")
    # BASELINE config 5: ComplEx, margin, 128 triples per GPU, half-precision embeddings (autocast mix:
    # f16 / bf16 encoder output, f32 relation rows); torch leg = the reference expressions on the same tensors
    B, K, D, model, loss_fn, reg = 128, 64, 128, "complex", "margin", 1e-3
    for dtype in (torch.float16, torch.bfloat16):
        ent = (torch.randn(B, 2, D, device="cuda") * 0.4).to(dtype)
        rel = torch.randn(B, 1, D, device="cuda") * 0.3
        neg_idx = torch.randint(0, 2 * B, (B, K, 2), device="cuda")

        def fused():
            e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
            ops.inbatch_loss(model, loss_fn, e, r, neg_idx, reg).backward()

        def stock():
            e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
            ref_port.compute_loss(model, loss_fn, e, r, neg_idx, reg).backward()

        a, b = timeit(fused), timeit(stock)
        print(f"{model:9s} {loss_fn:7s} {B:4d} | {a:16.1f} | {b:16.1f} | {b / a:5.2f}x   ({str(dtype)[6:]} embeddings)")


if __name__ == "__main__":
    main()
