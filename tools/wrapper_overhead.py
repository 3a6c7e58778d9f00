"""Host-side cost of one ops.rank_all call (Python checks, workspace allocation, ctypes): a problem so small that the GPU
is idle most of the time, timed without synchronising, and profiled."""
import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops
dev = torch.device("cuda", 0)
table = torch.randn(640, 128, device=dev); qf = torch.randn(8, 128, device=dev); qr = torch.randn(8, 128, device=dev)
true = torch.zeros(8, dtype=torch.int64, device=dev); out = torch.empty((8, 4), dtype=torch.int32, device=dev)
for _ in range(200): ops.rank_all("transe", table, qf, qr, 4, true_row=true, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): ops.rank_all("transe", table, qf, qr, 4, true_row=true, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e6 * (t1 - t0) / 2000:.1f} us per call issued, {1e6 * (t2 - t0) / 2000:.1f} us per call completed")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ops.rank_all("transe", table, qf, qr, 4, true_row=true, out=out)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
