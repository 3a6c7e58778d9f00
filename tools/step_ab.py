"""A/B of whole evaluation steps between builds of the library, interleaved on one box:
    python tools/step_ab.py workload libA.so libB.so [rounds]
Each round times 100 steps of bench.Job.step with each library (fresh process per measurement)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time, torch
    sys.path.insert(0, ROOT)
    from blp_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
    import bench
    job = bench.Job(sys.argv[2], torch.device("cuda", 0))
    for _ in range(20): job.step(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): job.step(True)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) * 10:.4f}")
    sys.exit(0)
workload, libs = sys.argv[1], sys.argv[2:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for r in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, __file__, "--child", workload, lib], capture_output=True, text=True)
        print(f"round {r} {os.path.basename(lib):28s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]} ms per step", flush=True)
