"""What the stock DKRL table build costs (models.py:165-204: lookup, conv1, mask, max-pool, tanh, conv2, masked mean, tanh),
per chunk of emb_batch_size entities, next to the ranking it feeds.
    python tools/dkrl_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, models  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for E, V, n, L in ((768, 28996, 512, 32), (300, 400001, 512, 32), (768, 28996, 12288, 64)):
        emb_path = f"/tmp/emb_{E}.pt"
        torch.save(torch.randn(V, E) * 0.05, emb_path)
        m = models.DKRL(128, "transe", "margin", 237, 1e-2, embeddings=emb_path).to(dev)
        g = torch.Generator().manual_seed(0)
        tok = torch.randint(1, V, (n, L), generator=g).to(dev)
        lens = torch.randint(4, L + 1, (n,), generator=g)
        mask = (torch.arange(L)[None, :] < lens[:, None]).float().to(dev)
        with torch.no_grad():
            for _ in range(3):
                out = m.encode(tok, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20 if n <= 512 else 3
            for _ in range(reps):
                out = m.encode(tok, mask)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            out2 = torch.empty(n, 128, device=dev)
            for _ in range(3):
                m.encode_into(out2, tok, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                m.encode_into(out2, tok, mask)
            torch.cuda.synchronize()
            fused = (time.perf_counter() - t0) / reps * 1e6
            err = (out2 - out).abs().max().item()
            by_split = {}
            for split in (1, 2, 4):  # waves per M-tile, forced (hooks build)
                _lib.set_knob("dkrl_split", split)
                m.encode_into(out2, tok, mask)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    m.encode_into(out2, tok, mask)
                torch.cuda.synchronize()
                by_split[split] = round((time.perf_counter() - t0) / reps * 1e6, 1)
            _lib.set_knob("dkrl_split", 0)
        flops = 2.0 * n * L * 2 * E * 128 + 2.0 * n * (L // 4) * 2 * 128 * 128
        print(f"DKRL E={E} chunk {n} x {L}: stock {us:9.1f} us per chunk ({flops / us / 1e6:6.2f} TF/s of conv arithmetic), fused "
              f"{fused:9.1f} us ({flops / fused / 1e6:6.2f} TF/s; {us / fused:4.1f}x; max |diff| {err:.1e}); FB15k-237 table "
              f"(14 541 entities): {us * 14541 / n / 1e3:7.2f} -> {fused * 14541 / n / 1e3:7.2f} ms; waves per M-tile forced: {by_split}", flush=True)


if __name__ == "__main__":
    main()
