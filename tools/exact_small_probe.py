# Ranking-pass time (HIP-event bracket around the ranking kernel) of small blocks against the FB15k-237 table:
# the exact tile kernel (rank_tiles, forced query chunk) against the small-block kernel (rank_small).
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops
import bench
dev = torch.device("cuda", 0)
events = bench.HipEvents()
MODES = [("tiles", {"rank_kernel": 1}), ("small", {"small_kernel": 1}), ("small/32", {"small_kernel": 1, "exact_query_chunk": 32}),
         ("small/64", {"small_kernel": 1, "exact_query_chunk": 64}), ("small/128", {"small_kernel": 1, "exact_query_chunk": 128}),
         ("prepass", {"small_kernel": 2, "sad_min_queries": 16})]
for wl in sys.argv[1:] or ("fb15k237-transe", "fb15k237-distmult", "fb15k237-complex"):
    cfg = bench.WORKLOADS[wl]
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    T = heads.shape[0]
    print(f"{wl}: kernel us | call us for " + ", ".join(m for m, _ in MODES))
    for t in (8, 32, 64, 128, 256, 512):
        qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
        tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
        out = torch.empty((2 * t, 4), dtype=torch.int32, device=dev)
        row = []
        ref = None
        for name, kn in MODES:
            _lib.reset_knobs()
            for k, v in kn.items(): _lib.set_knob(k, v)
            for _ in range(5): ops.rank_all(cfg["model"], table, qf, qr, t, true_row=tr, out=out)
            got = out.clone()
            ref = got if ref is None else ref
            assert torch.equal(got, ref), (wl, t, name)
            pairs = []
            for _ in range(50):
                a, b = events.pair()
                _lib.check(_lib.lib().blp_profile_next_rank_kernel(a, b), "profile")
                ops.rank_all(cfg["model"], table, qf, qr, t, true_row=tr, out=out)
                pairs.append((a, b))
            torch.cuda.synchronize()
            ms = sorted(events.elapsed_ms(a, b) for a, b in pairs)
            import time
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): ops.rank_all(cfg["model"], table, qf, qr, t, true_row=tr, out=out)
            torch.cuda.synchronize(); call = (time.perf_counter() - t0) / 200 * 1e6
            row.append((ms[len(ms) // 2] * 1e3, call))
        print(f"  {2 * t:4d} queries: " + "  ".join(f"{k:6.1f}|{c:6.1f}" for k, c in row), flush=True)
_lib.reset_knobs()
