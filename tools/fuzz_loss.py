"""Randomised soak of the fused in-batch loss (forward + backward) and of score_fn against the torch port
(test infrastructure), random shapes / models / losses / storage types.  python tools/fuzz_loss.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops  # noqa: E402
from oracle import ref_port  # noqa: E402

TOL = {torch.float32: 3e-5, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        seed = seed0 + n
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        model = str(rng.choice(["transe", "distmult", "complex", "simple"]))
        loss_fn = str(rng.choice(["margin", "nll"]))
        D = int(rng.choice([64, 128, 256, 320])) if model != "transe" else int(rng.choice([64, 128, 300, 768, 36]))
        B, K = int(rng.integers(2, 160)), int(rng.integers(2, 80))
        dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
        rel_f32 = dtype == torch.float32 or rng.random() < 0.5
        reg = float(rng.choice([0.0, 1e-3, 1e-2]))
        ent = (torch.randn(B, 2, D) * float(rng.choice([0.1, 0.4, 1.0]))).to(dtype)
        rel = torch.randn(B, 1, D) * 0.3
        rel = rel if rel_f32 else rel.to(dtype)
        neg_idx = torch.randint(0, 2 * B, (B, K, 2))
        e_ref, r_ref = ent.float().clone().requires_grad_(True), rel.float().clone().requires_grad_(True)
        ref = ref_port.compute_loss(model, loss_fn, e_ref, r_ref, neg_idx, reg)
        ref.backward()
        e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
        loss = ops.inbatch_loss(model, loss_fn, e, r, neg_idx.cuda(), reg)
        loss.backward()
        ok = abs(loss.item() - ref.item()) <= 3e-6 * max(1.0, abs(ref.item()))
        # a gradient element is a sum of up to B (K + 1) terms of either sign: the summation order moves it by a
        # few ulps of the LARGEST partial sum, so the absolute tolerance scales with max |grad| (seed 604)
        for got_g, want_g, tol in ((e.grad, e_ref.grad, TOL[dtype]), (r.grad, r_ref.grad, TOL[rel.dtype])):
            want_np = want_g.numpy()
            ok &= np.allclose(got_g.float().cpu().numpy(), want_np, rtol=tol, atol=max(2e-7, 2e-6 * float(np.abs(want_np).max())))
        # score_fn forward on the training broadcast (B, K, D) x (B, 1, D): bit-identical in f32
        if dtype == torch.float32 and D % 32 == 0 or model == "transe" and dtype == torch.float32:
            h, t = ent[:, :1].float(), torch.randn(B, K, D)
            want = ref_port.SCORE_FNS[model](h, t, rel.float())
            got = ops.score(model, h.cuda(), t.cuda(), rel.float().cuda()).cpu()
            ok &= bool(torch.equal(got, want))
        # score_fn on a random broadcast (each leading dim present or 1 per operand): forward bit-identical where
        # the reduction order is pinned, backward against autograd of the reference expressions
        if dtype == torch.float32 and (D % 32 == 0 or model == "transe"):
            lead = [int(rng.integers(1, 7)) for _ in range(int(rng.integers(1, 4)))]
            shapes = [[n if rng.random() < 0.6 else 1 for n in lead] + [D] for _ in range(3)]
            cpu_in = [(torch.randn(*sh) * 0.5).requires_grad_(True) for sh in shapes]
            gpu_in = [x.detach().cuda().requires_grad_(True) for x in cpu_in]
            want = ref_port.SCORE_FNS[model](*cpu_in)
            got = ops.score(model, *gpu_in)
            ok_b = bool(torch.equal(got.cpu(), want.detach()))
            w = torch.randn(want.shape)
            (want * w).sum().backward()
            (got * w.cuda()).sum().backward()
            for a, b in zip(gpu_in, cpu_in):
                bn = b.grad.numpy()
                ok_b &= np.allclose(a.grad.cpu().numpy(), bn, rtol=3e-5, atol=max(2e-7, 2e-6 * float(np.abs(bn).max())))
            if not ok_b:
                print(f"  (score broadcast shapes {shapes})", flush=True)
            ok &= ok_b
        if not ok:
            bad += 1
            print(f"MISMATCH seed={seed} {model} {loss_fn} B={B} K={K} D={D} {dtype} rel_f32={rel_f32} reg={reg}: "
                  f"loss {loss.item()} vs {ref.item()}", flush=True)
        n += 1
    print(f"{n} cases in {time.time() - t0:.0f} s, {bad} mismatches (seeds {seed0}..{seed0 + n - 1})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
