"""The bilinear pre-pass against query tiles per workgroup (knob gemm_tiles_per_chunk; hooks build): ranking pass ms on the
FB15k-237 block.  python tools/gemm_chunk_sweep.py [model]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops
import bench

model = sys.argv[1] if len(sys.argv) > 1 else "distmult"
dev = torch.device("cuda", 0)
events = bench.HipEvents()
cfg = bench.WORKLOADS[f"fb15k237-{model}"]
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
T = heads.shape[0]
want = ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row)
out = torch.empty_like(want)
for per in (0, 32, 48, 56, 60, 64, 68, 72, 76, 80, 88, 96):
    _lib.reset_knobs()
    _lib.set_knob("gemm_tiles_per_chunk", per)
    L = _lib.lib()
    for _ in range(3):
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
    same = torch.equal(out, want)
    pairs = []
    for _ in range(15):
        a, b = events.pair()
        _lib.check(L.blp_profile_next_rank_kernel(a, b), "hook")
        ops.rank_all(model, table, q_fixed, q_rel, T, true_row=true_row, out=out)
        pairs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(events.elapsed_ms(a, b) for a, b in pairs)
    print(f"{model} tiles_per_chunk {per:3d}: rank pass median {ms[len(ms) // 2]:.3f} ms (min {ms[0]:.3f}); counts equal: {same}", flush=True)
_lib.reset_knobs()
