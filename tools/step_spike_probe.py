import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
for model, clusters, noise in (("complex", 2000, 0.0), ("distmult", 500, 0.0)):
    cfg = dict(bench.WORKLOADS[f"fb15k237-{model}"]); cfg.update(clusters=clusters, noise=noise, top=145)
    bench.WORKLOADS["x"] = cfg
    job = bench.Job("x", dev)
    for i in range(12):
        s0 = torch.cuda.memory_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        job.step(True)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        s1 = torch.cuda.memory_stats()
        print(model, clusters, i, f"issue {1e3*(t1-t0):.3f} ms, done {1e3*(t2-t0):.3f} ms, device mallocs {s1['num_device_alloc']-s0['num_device_alloc']}, frees {s1['num_device_free']-s0['num_device_free']}, retries {s1['num_alloc_retries']-s0['num_alloc_retries']}")
    del job; torch.cuda.empty_cache()
