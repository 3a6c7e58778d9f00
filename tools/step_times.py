"""Per-step wall times of a bench workload (each step synchronised), to spot one-off costs:
    python tools/step_times.py fb15k237-complex [n_steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "fb15k237-complex"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
for prev in ("fb15k237-transe",):  # what bench.py runs first
    j0 = bench.Job(prev, dev); j0.step(); torch.cuda.synchronize(); del j0; torch.cuda.empty_cache()
job = bench.Job(name, dev)
for filtered in (True, False, True):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); job.step(filtered); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "filtered" if filtered else "raw", " ".join(f"{t:.2f}" for t in ts), f"reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)
