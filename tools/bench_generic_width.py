"""All-entity ranking at the embedding widths of the reference's BOW / DKRL encoders (300 = GloVe,
768 = BERT word embeddings; scripts/{glove,bert}-{bow,dkrl}-*.sh, all TransE), FB15k-237 shape:
the any-width fixed-point pre-pass (rank_sad_wide.hip) through blp_amd.ranking.rank_block."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import models, ranking

def main():
    N, T = 14541, 52870
    g = torch.Generator().manual_seed(0)
    for D in (300, 768):
        model = models.LinkPrediction(D, "transe", "margin", 237, 0).cuda()
        table = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1).cuda()
        heads, tails = torch.randint(0, N, (T,), generator=g).cuda(), torch.randint(0, N, (T,), generator=g).cuda()
        rel = model.rel_emb(torch.randint(0, 237, (T,), generator=g).cuda()).detach()
        q_fixed, q_rel, true_row = torch.cat((table[tails], table[heads])), torch.cat((rel, rel)), torch.cat((heads, tails))
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            counts = ranking.rank_block(model, table, q_fixed, q_rel, T, true_row=true_row)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"D={D}: {dt * 1e3:.1f} ms for {2 * T} queries x {N} candidates ({2 * T * N / dt / 1e9:.1f} G scored triples/s)")

if __name__ == "__main__":
    main()
