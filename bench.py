"""bench.py -- scored triples/sec of the all-entities ranking hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full evaluation pass of the workload's query set against the entity table: every
query is scored against every candidate and ranked (raw counts -> reciprocal rank, Hits@{1,3,10}).
One scored triple = one (query, candidate) score evaluated and ranked.

Workloads (synthetic data of the published shapes, seeded; BASELINE.json configs):
  fb15k237-transe   (default; configs[1]) 14 541 x 128 table, 52 870 test triples -> 105 740 queries
  fb15k237-distmult / -complex / -simple   (configs[2]) same shapes, un-normalised table
  wikidata5m-transe (configs[3]) 4.6 M x 128 table (2.36 GB), reference batching: 2 triples =
                    4 queries per table pass; a step = 64 passes
With N > 1 the candidate axis is sharded across ranks (rows [lo, hi) per rank), the queries and the
true entities' vectors are replicated, and the per-shard (Q, 4) int32 counts are combined by one RCCL
all-gather per step.  Total work is fixed as N grows -> "scaling": "strong".

Only the cpu_baseline leg and the parity spot-check touch oracle/ (as the checker / the thing timed
on the host); the measured GPU path goes through libblp_hip.so only.
"""
import argparse
import ctypes
import json
import os
import re
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_PEAK_TFLOPS = 157.3         # f32 MFMA peak == f32 vector (FMA) peak; plain add/sub ops reach half
BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md); the bilinear pre-pass spends 3 bf16
BF16X3_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 3   # products per f32-equivalent product, so its roof is a third of it

WORKLOADS = {
    "fb15k237-transe": dict(model="transe", N=14541, D=128, R=237, triples=52870, passes=1),
    "fb15k237-distmult": dict(model="distmult", N=14541, D=128, R=237, triples=52870, passes=1),
    "fb15k237-complex": dict(model="complex", N=14541, D=128, R=237, triples=52870, passes=1),
    "fb15k237-simple": dict(model="simple", N=14541, D=128, R=237, triples=52870, passes=1),
    # the BERT-BOW / BERT-DKRL width (models.py:118-135, 165-172): TransE at the word-embedding width 768
    "fb15k237-transe-d768": dict(model="transe", N=14541, D=768, R=237, triples=52870, passes=1),
    "wikidata5m-transe": dict(model="transe", N=4_600_000, D=128, R=822, triples=2, passes=64),
    # the same table with the whole Wikidata5M test set (6 894 triples) as ONE query block
    "wikidata5m-transe-block": dict(model="transe", N=4_600_000, D=128, R=822, triples=6894, passes=1),
    "wikidata5m-complex-block": dict(model="complex", N=4_600_000, D=128, R=822, triples=6894, passes=1),
}
# Algorithmic f32 operations per (candidate, query, element), head-side / tail-side:
# TransE: one subtract and one |.|-accumulate once the query-only part is hoisted (h + r | t - r per
# query); bilinear models: the all-entities
# score is a (Q x D) . (D x N) GEMM = 2 flops per (pair, element).
OPS_PER_ELEM = {"transe": (2, 2), "distmult": (2, 2), "complex": (2, 2), "simple": (2, 2)}
DOMINANT_KERNEL = {"transe": "rank_sad_kernel<128> + sad_refine_* (u16 fixed-point v_sad_u16 pre-pass + band + exact f32 refinement; prep kernels included)",
                   "distmult": "rank_gemm_bf16_kernel<DISTMULT,128> + refine_pairs/refine (bf16 x 3 split MFMA GEMM + band + exact f32 refinement)",
                   "complex": "rank_gemm_bf16_kernel<COMPLEX,128> + refine_pairs/refine",
                   "simple": "rank_gemm_bf16_kernel<SIMPLE,128> + refine_pairs/refine"}


def make_data(cfg, device, seed=1):
    """Synthetic FB15k-237 / Wikidata5M-shaped inputs: table rows L2-normalised for
    TransE (models.py:40-41), 0.1 * randn otherwise; rel_emb Xavier-uniform (models.py:28-29)."""
    g = torch.Generator(device=device).manual_seed(seed)
    N, D, R, T = cfg["N"], cfg["D"], cfg["R"], cfg["triples"]
    table = torch.randn(N, D, device=device, generator=g)
    table = torch.nn.functional.normalize(table, dim=-1) if cfg["model"] == "transe" else table * 0.1
    bound = (6.0 / (R + D)) ** 0.5
    rel_w = (torch.rand(R, D, device=device, generator=g) * 2 - 1) * bound
    heads = torch.randint(0, N, (T,), device=device, generator=g)
    tails = torch.randint(0, N, (T,), device=device, generator=g)
    rels = torch.randint(0, R, (T,), device=device, generator=g)
    order = torch.argsort(rels, stable=True)  # evaluation order is free; blp_amd.ranking groups by relation too
    return table, rel_w, heads[order], tails[order], rels[order]


def build_queries(table, rel_w, heads, tails, rels):
    """train.py:141-150: head-replacing queries first (fixed = tail emb), then tail-replacing."""
    q_fixed = torch.cat((table[tails], table[heads]))
    q_rel = torch.cat((rel_w[rels], rel_w[rels]))
    true_row = torch.cat((heads, tails))
    return q_fixed.contiguous(), q_rel.contiguous(), true_row.contiguous()


class HipEvents:
    """Raw hipEvent_t pairs on the HIP runtime torch already loaded (for the C-ABI timing hook)."""

    def __init__(self):
        maps = open("/proc/self/maps").read()
        paths = sorted(set(re.findall(r"/\S*libamdhip64\S*", maps)))
        self.hip = ctypes.CDLL(paths[0])
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]

    def pair(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
        return a, b

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def cpu_baseline(cfg, table, rel_w, heads, tails, rels, budget_s=12.0):
    """The reference's CPU path restated (oracle/ref_port.py: same torch CPU kernels), timed on this
    box's host cores on a bounded sample of reference batches (eval_batch_size = 64 triples for
    FB15k-237, 2 for Wikidata5M: scripts/blp-transe-{fb15k237,wikidata5m}.sh:18)."""
    from oracle import ref_port
    B = 64 if cfg["N"] < 1_000_000 else 2
    tab, rw = table.cpu(), rel_w.cpu()
    h, t, r = heads.cpu(), tails.cpu(), rels.cpu()
    T = h.shape[0]

    def batch(i):
        sl = slice((i * B) % max(T - B + 1, 1), (i * B) % max(T - B + 1, 1) + B)
        return ref_port.eval_batch(cfg["model"], tab, h[sl], t[sl], rw[r[sl]])

    batch(0)  # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        batch(done + 1)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64:
            break
    value = 2.0 * B * cfg["N"] * done / el
    return {"value": value, "unit": "scored triples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} reference batches of {B} triples ({2 * B} queries) x {cfg['N']} candidates, "
                      f"{el:.1f} s of torch-CPU work ({os.cpu_count()} logical CPUs)"}


def torch_gpu_baseline(cfg, table, rel_w, heads, tails, rels, budget_s=3.0):
    """The reference's own expressions (torch broadcasting score_fn + get_metrics, restated in
    oracle/ref_port.py) on THIS GPU through stock PyTorch-ROCm, reference batch size -- what the
    unmodified reference does on a MI355X.  Bounded sample; a reported baseline, not the target."""
    from oracle import ref_port
    if cfg["N"] > 1_000_000:
        return None
    B = 64
    T = heads.shape[0]

    def batch(i):
        lo = (i * B) % max(T - B + 1, 1)
        sl = slice(lo, lo + B)
        out = ref_port.eval_batch(cfg["model"], table, heads[sl], tails[sl], rel_w[rels[sl]])
        return out["rr"].sum().item()  # the reference syncs per batch (train.py:154)

    batch(0)
    torch.cuda.synchronize()
    done, t0 = 0, time.perf_counter()
    while True:
        batch(done + 1)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 400:
            break
    return {"value": 2.0 * B * cfg["N"] * done / el, "unit": "scored triples/s", "kind": "reference expressions on "
            "PyTorch-ROCm (same GPU)", "sample": f"{done} reference batches of {B} triples, {el:.2f} s"}


def parity_spot_check(cfg, table, q_fixed, q_rel, true_row, q_head, counts, n=32):
    """First n head- and first n tail-queries of the measured run against the CPU oracle."""
    import numpy as np
    from oracle import oracle as orc
    if cfg["N"] > 100_000:
        return "skipped (table too large for the CPU oracle inside bench; see tests)"
    tab = table.cpu().numpy()
    idx_h = torch.arange(0, min(n, q_head))
    idx_t = torch.arange(q_head, min(q_head + n, q_fixed.shape[0]))
    ok = True
    for side, idx in ((orc.SIDE_HEAD, idx_h), (orc.SIDE_TAIL, idx_t)):
        if idx.numel() == 0:
            continue
        want = orc.rank_counts(cfg["model"], side, tab, q_fixed[idx].cpu().numpy(), q_rel[idx].cpu().numpy(),
                               true_row=true_row[idx].cpu().numpy())
        ok &= bool(np.array_equal(want, counts[idx].cpu().numpy()))
    return f"{idx_h.numel() + idx_t.numel()} queries vs CPU oracle: " + ("identical counts" if ok else "MISMATCH")


def load_pmc(workload, key="hbm_bytes_per_launch"):
    """A figure of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json):
    HBM bytes per launch (FETCH_SIZE) or the matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES); else None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get(workload, {}).get(key)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fb15k237-transe", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", default="off", choices=["auto", "on", "off"],
                    help="replay a step's launches from a captured hipGraph (auto: fall back to eager launches if capture "
                         "fails).  Off by default: measured equal to eager launches (tools/graph_shard_ab.py), the host "
                         "launch loop is 10-70x ahead of the GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true")
    ap.add_argument("--shard-axis", default="auto", choices=["auto", "candidate", "query"],
                    help="N > 1: which axis to shard (default: what blp_amd.ranking.choose_shard_axis picks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the measured path)")
    # BLP_BENCH_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than ranks
    # (ranks share devices, counts are exchanged through host memory).  Timings are then meaningless.
    backend = os.environ.get("BLP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from blp_amd import _lib, ops, ranking

    cfg = WORKLOADS[args.workload]
    model, N, D = cfg["model"], cfg["N"], cfg["D"]
    table, rel_w, heads, tails, rels = make_data(cfg, device)
    q_fixed, q_rel, true_row = build_queries(table, rel_w, heads, tails, rels)
    q_head = heads.shape[0]
    Q = q_fixed.shape[0]
    rel_ids = torch.cat((rels, rels)).contiguous()
    passes = cfg["passes"]
    # N > 1: shard along the axis blp_amd.ranking would pick for this shape (see choose_shard_axis).
    #   "candidate": rank r ranks every query against table rows [lo, hi); ONE all-gather of the int32
    #                counts + a sum per pass.  Right when the table pass dominates (Wikidata5M).
    #   "query":     the table is replicated (inputs resident: like the table build itself, its one
    #                all-gather per evaluation happens before the hot path), rank r ranks its slice of
    #                the test triples against the whole table; the only collective is the all-reduce
    #                of four metric sums.  Right when per-query work dominates (FB15k-237).
    axis = ranking.choose_shard_axis(N, D, Q, world)
    if args.shard_axis != "auto" and world > 1:
        axis = args.shard_axis
    lo, hi = ranking.shard_bounds(N, world, rank) if axis == "candidate" else (0, N)
    shard = table[lo:hi]
    if world > 1 and axis == "query":
        t_lo, t_hi = ranking.shard_bounds(q_head, world, rank)
        pick = torch.cat((torch.arange(t_lo, t_hi), torch.arange(q_head + t_lo, q_head + t_hi))).to(device)
        q_fixed, q_rel, true_row, rel_ids = q_fixed[pick].contiguous(), q_rel[pick].contiguous(), true_row[pick], rel_ids[pick]
        Q_global, q_head = Q, t_hi - t_lo
        Q = q_fixed.shape[0]
    else:
        Q_global = Q
    q_true = table[true_row].contiguous() if (world > 1 and axis == "candidate") else None
    # one step = `passes` table passes (reference batches); their counts are exchanged ONCE per step
    local_all = torch.empty((passes, Q, 4), dtype=torch.int32, device=device)
    gathered = torch.empty((world, passes * Q, 4), dtype=torch.int32, device=device) if q_true is not None else None

    sum_pick = torch.tensor([0, 2, 3, 4], device=device)  # raw: sum of reciprocal ranks, hits@1/3/10

    def local_work():
        """This rank's launches of one step: `passes` table passes, and (unless per-shard counts have to be
        added up first) the metric sums.  Everything is asynchronous on the current stream, writes fixed
        buffers and needs no host decision, so the whole step can be replayed from a captured hipGraph."""
        for i in range(passes):
            if q_true is not None:
                ops.rank_all(model, shard, q_fixed, q_rel, q_head, q_true=q_true, rel_ids=rel_ids, out=local_all[i])
            else:
                ops.rank_all(model, shard, q_fixed, q_rel, q_head, true_row=true_row, rel_ids=rel_ids, out=local_all[i])
        if q_true is None:
            return ops.rank_metric_sums(local_all.view(passes * Q, 4)).index_select(0, sum_pick)
        return None

    pending = []  # outstanding collectives (query-axis shards)
    graph, graph_note, static_sums = None, "eager launches", None
    if args.graph != "off" and backend != "gloo":
        try:
            local_work()  # lazy initialisation (module load, allocator) happens outside the capture
            torch.cuda.synchronize()
            captured = torch.cuda.CUDAGraph()
            with torch.cuda.graph(captured):
                static_sums = local_work()
            graph, graph_note = captured, "hipGraph replay"
        except Exception as exc:  # noqa: BLE001 -- the eager path below is the same work
            if args.graph == "on":
                raise
            graph, graph_note = None, f"eager launches (graph capture failed: {type(exc).__name__})"
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
            sums = static_sums
        else:
            sums = local_work()
        counts = local_all.view(passes * Q, 4)
        if q_true is not None:  # candidate shards: per-shard counts add up
            if backend == "gloo":
                parts = [torch.empty(counts.shape, dtype=counts.dtype) for _ in range(world)]
                dist.all_gather(parts, counts.cpu())
                gathered.copy_(torch.stack(parts))
            else:
                dist.all_gather_into_tensor(gathered.view(-1), counts.reshape(-1))
            counts = gathered.sum(dim=0, dtype=torch.int32)
            sums = ops.rank_metric_sums(counts).index_select(0, sum_pick)
        elif world > 1:  # query shards: only the metric sums travel
            if backend == "gloo":
                host = sums.cpu()
                dist.all_reduce(host)
                sums = host.to(device)
            else:  # asynchronous on RCCL's stream: the next step's kernels do not wait for this one's exchange
                sums = sums.clone()
                pending.append(dist.all_reduce(sums, async_op=True))
        return counts[-Q:], sums  # sums are valid once fence() has waited for the exchange

    def fence():
        for work in pending:
            work.wait()
        pending.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts, sums = step()
    fence()
    elapsed = time.perf_counter() - t0
    mrr, hits = sums[0] / (Q_global * passes), sums[1:] / (Q_global * passes)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # dominant-kernel duration, measured with HIP events around the rank_tiles launch (C-ABI hook),
    # on the stream the kernel runs on, over a second run of the same steps
    events = HipEvents()
    pairs = []
    for _ in range(max(1, min(args.steps, 10)) * passes):
        a, b = events.pair()
        _lib.check(_lib.lib().blp_profile_next_rank_kernel(a, b), "blp_profile_next_rank_kernel")
        if q_true is not None:
            ops.rank_all(model, shard, q_fixed, q_rel, q_head, q_true=q_true, rel_ids=rel_ids)
        else:
            ops.rank_all(model, shard, q_fixed, q_rel, q_head, true_row=true_row, rel_ids=rel_ids)
        pairs.append((a, b))
    torch.cuda.synchronize()
    kernel_ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / len(pairs)

    if rank == 0:
        n_local = hi - lo
        scored = float(Q_global) * N * passes
        ops_h, ops_t = OPS_PER_ELEM[model]
        alg_flops = n_local * D * (q_head * ops_h + (Q - q_head) * ops_t)
        alg_bytes = n_local * D * 4 + Q * (2 * D * 4 + 24)
        t_k = kernel_ms * 1e-3
        sad_path = model == "transe"
        peak_tf = F32_PEAK_TFLOPS if sad_path else BF16X3_PEAK_TFLOPS
        hbm_time, cmp_time = alg_bytes / (HBM_PEAK_GBPS * 1e9), alg_flops / (peak_tf * 1e12)
        if hbm_time >= cmp_time:
            roofline = {"bound": "hbm", "achieved": alg_bytes / t_k / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
        else:
            roofline = {"bound": "mfma", "achieved": alg_flops / t_k / 1e12, "peak": peak_tf,
                        "unit": "TFLOP/s",
                        "note": ("VALU roof, not MFMA (an L1 norm has no matrix-core form): 157.3 Tops/s is the f32 vector-FMA "
                                 "peak and also what v_sad_u16 delivers at its 4 cycles/instruction (2 elements x "
                                 "(subtract + |.|-accumulate) x 64 lanes); the exact f32 add/sub kernel tops out at half "
                                 "of it. achieved = 2 ops x D x Q x N / time of the whole rank pass (range + quantise + "
                                 "SAD pre-pass + exact refinement of the undecided pairs)."
                                 if sad_path else
                                 "achieved = 2 flops x D x Q x N (the f32 GEMM the reference's scores amount to) / time of "
                                 "GEMM pre-pass + exact refinement.  Every f32 product is three bf16 MFMA products "
                                 "(hi*hi + hi*lo + lo*hi), so the roof is the dense bf16 MFMA peak / 3 = 833 TF "
                                 "(5.3x the f32 MFMA peak of 157.3 TF, which the f32-chain variant BLP_GEMM_KERNEL=f32 "
                                 "is bounded by).")}
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        roofline["traffic"] = load_pmc(args.workload)
        if not sad_path and roofline["bound"] == "mfma":
            roofline["mfma_busy"] = load_pmc(args.workload, "mfma_busy_frac_at_2.4GHz")
        roofline["kernel"] = (DOMINANT_KERNEL[model] if Q >= 64 else "rank_tiles_kernel<STATIC> (lane-per-candidate VALU)")
        if model == "transe" and D not in (64, 128, 256) and Q >= 256:
            roofline["kernel"] = "wide_rank_sad_kernel + wide_refine_* (any-width u16 v_sad_u16 pre-pass, 128 elements at a time)"
        roofline["kernel_ms"] = kernel_ms
        roofline["algorithmic_bytes_per_launch"] = alg_bytes
        roofline["algorithmic_flops_per_launch"] = alg_flops
        result = {
            "metric": "scored triples/sec, all-entity eval (MRR + Hits@k ranked)",
            "value": scored * args.steps / elapsed,
            "unit": "scored triples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded randn table, Xavier rel_emb, uniform random test triples)",
            "config": {"workload": args.workload, "rel_model": model, "entities": N, "dim": D,
                       "queries_per_step": Q_global * passes, "table_passes_per_step": passes,
                       "parallelism": f"{axis}-axis shards x{world}" if world > 1 else "single GPU",
                       "launch": graph_note},
            "mrr": mrr.item(),
            "hits@1,3,10": [x.item() for x in hits],
            "roofline": roofline,
        }
        result["parity_check"] = parity_spot_check(cfg, table, q_fixed, q_rel, true_row, q_head, counts)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, table, rel_w, heads, tails, rels)
            result["torch_gpu_baseline"] = torch_gpu_baseline(cfg, table, rel_w, heads, tails, rels)
        if world == 1 and not args.no_hbm_probe and not args.workload.startswith("wikidata5m"):
            result["hbm_probe"] = hbm_probe(device, events)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def hbm_probe(device, events, reps=10):
    """The same rank_tiles kernel at its HBM-bound operating point (BASELINE config 4 at 1 GPU):
    4.6 M x 128 f32 table = 2.355 GB, reference batching 2 triples = 4 queries per table pass."""
    from blp_amd import _lib, ops
    cfg = WORKLOADS["wikidata5m-transe"]
    table, rel_w, heads, tails, rels = make_data(cfg, device, seed=5)
    q_fixed, q_rel, true_row = build_queries(table, rel_w, heads, tails, rels)
    for _ in range(2):
        ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
    pairs = []
    for _ in range(reps):
        a, b = events.pair()
        _lib.check(_lib.lib().blp_profile_next_rank_kernel(a, b), "blp_profile_next_rank_kernel")
        ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
        pairs.append((a, b))
    torch.cuda.synchronize()
    ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / len(pairs)
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.rank_metrics(ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row))
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    alg_bytes = cfg["N"] * cfg["D"] * 4 + 4 * (2 * cfg["D"] * 4 + 24)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"workload": "wikidata5m-transe, 4 queries per table pass", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "kernel_ms": ms,
            "whole_call_ms": call_ms, "scored_triples_per_s": 4.0 * cfg["N"] / (call_ms * 1e-3),
            "traffic": load_pmc("wikidata5m-transe")}


if __name__ == "__main__":
    main()
