"""Where a wave of rank_gemm_bf16_kernel spends its cycles (library built with -DBLP_TIMING):
python -c "from blp_amd import build" is NOT enough -- build with
    BLP_EXTRA_HIPCC_FLAGS=-DBLP_TIMING python -c "from blp_amd import build; build.build(force=True)"
then run this script; rebuild without the flag afterwards."""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops, _lib
import bench
cfg = bench.WORKLOADS["fb15k237-distmult"]
dev = torch.device("cuda", 0)
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
L = _lib.lib()
out = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    ops.rank_all("distmult", table, q_fixed, q_rel, heads.shape[0], true_row=true_row)
torch.cuda.synchronize()
L.blp_debug_read_timing(out)  # reset
ops.rank_all("distmult", table, q_fixed, q_rel, heads.shape[0], true_row=true_row)
torch.cuda.synchronize()
L.blp_debug_read_timing(out)
waves = out[7]
names = ["outside stages (prologue, first tile)", "stage: MFMAs of t+1 + decision of t", "stage: settle (pairs / flags / counters)",
         "stage: wait for the LDS-DMA of t+2", "stage: barrier", "epilogue (flush counters, pairs)"]
total = sum(out[i] for i in range(6))
print(f"{waves} waves, {total / waves:.0f} ticks per wave")
for i, n in enumerate(names):
    print(f"  {n:46s} {out[i] / waves:10.0f} ticks/wave  {100.0 * out[i] / total:5.1f} %")
