"""The bag-of-words table build (models.py:143-155 + F.normalize + the row assignment, train.py:109-113): the stock
PyTorch-ROCm modules against the fused kernel (blp_bow_rows), per emb_batch_size chunk of the scripts:
    python tools/bow_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import models, ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
# (name, V, E, n, L): bert-bow (BERT word embeddings 28 996 x 768), glove-bow (400 001 x 300); emb_batch_size 512 / 12 288, max_len 32 / 64
for name, V, E, n, L in (("bert-bow fb15k237 chunk", 28996, 768, 512, 32), ("bert-bow wikidata5m chunk", 28996, 768, 12288, 64),
                         ("glove-bow fb15k237 chunk", 400001, 300, 512, 32), ("glove-bow wikidata5m chunk", 400001, 300, 12288, 64)):
    model = models.BOW("transe", "margin", 5, 0, embeddings=torch.randn(V, E, generator=g) * 0.1).to(dev)
    tok = torch.randint(1, V, (n, L), generator=g).to(dev)
    lengths = torch.randint(L // 2, L + 1, (n,), generator=g)
    mask = (torch.arange(L).unsqueeze(0) < lengths.unsqueeze(1)).float().to(dev)
    out = torch.empty(n, E, device=dev)
    def stock():
        with torch.no_grad():
            out.copy_(model.encode(tok, mask))
    def fused():
        with torch.no_grad():
            model.encode_into(out, tok, mask)
    res = {}
    for label, fn in (("stock", stock), ("fused", fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize()
        res[label] = (time.perf_counter() - t0) / 30 * 1e6
    gathered = n * L * E * 4
    print(f"{name:28s} n={n:6d} L={L:3d} E={E:4d}: stock {res['stock']:9.1f} us | fused {res['fused']:8.1f} us ({gathered / res['fused'] / 1e3:7.0f} GB/s of "
          f"gathered rows, {gathered / 1e6:7.1f} MB) | {res['stock'] / res['fused']:.1f}x", flush=True)
