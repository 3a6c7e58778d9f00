"""How far the FB15k-237 evaluation degrades away from i.i.d. random tables: clustered rows (near-duplicate
descriptions; noise 0 = exact duplicates) and a trained model's triples (true tail among the top-scoring entities), for a sweep
of cluster counts.  Per configuration: ms per evaluation step (whole evaluation, raw + filtered), against the random-data step,
what the pre-pass left to the exact path (blp_rank_all_prepass_stats) and a parity spot check against the CPU oracle.
    python tools/clustered_sweep.py [model ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda", 0)
only = None
if "--only" in sys.argv:  # --only CLUSTERS,NOISE : one configuration (under a profiler)
    i = sys.argv.index("--only")
    c, nz = sys.argv[i + 1].split(",")
    only = (int(c), float(nz))
    del sys.argv[i:i + 2]
models = sys.argv[1:] or ["transe", "distmult", "complex"]
for model in models:
    base_ms = None
    for clusters, noise in ((0, 0.0), (2000, 0.0), (500, 0.0), (500, 1e-3), (100, 0.0), (20, 0.0), (20, 1e-2)):
        if only is not None and (clusters, noise) != only:
            continue
        name = f"sweep-{model}-{clusters}-{noise}"
        cfg = dict(bench.WORKLOADS[f"fb15k237-{model}"])
        if clusters:
            cfg.update(clusters=clusters, noise=noise, top=145)
        bench.WORKLOADS[name] = cfg
        job = bench.Job(name, dev)
        for _ in range(2):
            job.step(True)
        ms = float("inf")
        for _ in range(3):  # the best of three 5-step loops: a one-off host stall of tens of ms after the previous configuration's
            torch.cuda.synchronize()  # teardown (torch.cuda.empty_cache) once read as 7 - 9 x on configurations that take 1.3 ms
            t0 = time.perf_counter()  # (tools/step_spike_probe.py: no such step when the job is timed step by step)
            for _ in range(5):
                triples, counts, sums = job.step(True)
            torch.cuda.synchronize()
            ms = min(ms, (time.perf_counter() - t0) / 5 * 1e3)
        base_ms = base_ms or ms
        st = job.prepass_stats()
        parity = bench.parity_spot_check(job, triples, counts, n=16)
        print(f"{model:9s} clusters {clusters:5d} noise {noise:7.0e}: {ms:8.3f} ms per step = {ms / base_ms:5.2f} x random; "
              f"decided {st['decided_frac']:.5f} (listed {st['listed']:,} + flagged rows {st['flagged_rows']:,}; {st['path']}); "
              f"MRR {sums[0].item() / (2 * job.T):.4f}; {parity}", flush=True)
        del job
        torch.cuda.empty_cache()
