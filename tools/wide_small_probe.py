"""TransE at the widths of the BOW / DKRL encoders (300, 768) with the reference's own eval batches (scripts/*-bow-*.sh:
eval_batch_size 16 / 32 -> 32 / 64 queries; *-dkrl-*.sh: 128 -> 256 queries) against the FB15k-237-sized table:
us per ranking.rank_block call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import models, ranking
N = 14541
g = torch.Generator().manual_seed(0)
for D in (300, 768):
    model = models.LinkPrediction(D, "transe", "margin", 237, 0).cuda()
    table = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1).cuda()
    for Q in (32, 64, 128, 256, 512):
        t = Q // 2
        heads, tails = torch.randint(0, N, (t,), generator=g).cuda(), torch.randint(0, N, (t,), generator=g).cuda()
        rel = model.rel_emb(torch.randint(0, 237, (t,), generator=g).cuda()).detach()
        q_fixed, q_rel, true_row = torch.cat((table[tails], table[heads])), torch.cat((rel, rel)), torch.cat((heads, tails))
        def step():
            for _ in range(20): ranking.rank_block(model, table, q_fixed, q_rel, t, true_row=true_row)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        print(f"D={D} {Q:4d} queries: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us per call", flush=True)
