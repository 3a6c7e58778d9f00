"""Where a wave of rank_small_kernel spends its time (needs a -DBLP_TIMING build of the library):
    python -c "from blp_amd import build; build.build(variant='timing', variant_flags=['-DBLP_TIMING'])"
    python tools/small_timing.py blp_amd/libblp_hip.timing.so
Phases: 0 = tile loads + first coefficient batch into LDS, 1 = barrier + tile rows out of LDS, 2 = scoring (and the later
coefficient batches).  Average nanoseconds per wave (100 MHz wall clock)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blp_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from blp_amd import ops
import bench
dev = torch.device("cuda", 0)
L = _lib.lib()
for wl in sys.argv[2:] or ("fb15k237-transe",):
    cfg = bench.WORKLOADS[wl]
    table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
    q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
    T = heads.shape[0]
    for t, chunk in ((8, 0), (32, 0), (64, 0), (64, 32), (128, 0), (256, 0), (512, 0)):
        qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])).contiguous(); qr = torch.cat((q_rel[:t], q_rel[T:T + t])).contiguous()
        tr = torch.cat((true_row[:t], true_row[T:T + t])).contiguous()
        _lib.reset_knobs(); _lib.set_knob("small_kernel", 1); _lib.set_knob("exact_query_chunk", chunk)
        for _ in range(20): ops.rank_all(cfg["model"], table, qf, qr, t, true_row=tr)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        L.blp_debug_read_small_timing(buf)
        for _ in range(20): ops.rank_all(cfg["model"], table, qf, qr, t, true_row=tr)
        torch.cuda.synchronize()
        L.blp_debug_read_small_timing(buf)
        n = max(buf[7], 1)
        print(f"{wl} {2 * t:5d} queries chunk {chunk or 'auto':>4}: waves/call {n // 20:6d}  ns per wave: " +
              "  ".join(f"p{i} {buf[i] * 10 / n:8.0f}" for i in range(4)), flush=True)
_lib.reset_knobs()
