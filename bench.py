"""bench.py -- scored triples/sec of the all-entities evaluation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started WITHOUT a launcher (no WORLD_SIZE in the environment), `python bench.py --gpus N`
starts its own N ranks (torch.distributed.run on 127.0.0.1 and a free port), passes rank 0's single JSON line through and
exits with the launcher's status (non-zero if any rank died).

A "step" is ONE WHOLE EVALUATION of the workload's test triples against the entity table, exactly what
blp_amd.ranking.eval_link_prediction runs after the table build (it is the same function,
ranking.rank_triples): id -> row lookups, query gathers (train.py:132-150), the filter
of every query from the sorted index of the filtering graph (utils.py:46-83), the ranking pass -- every
query scored against every candidate and counted, raw AND filtered (train.py:146-171) -- and the metric
sums (utils.py:86-111, train.py:152-157).  Inputs resident in HBM when the clock starts: the entity table,
rel_emb, the test triples, the id -> row map and the sorted filter index.  One scored triple = one
(query, candidate) score evaluated and ranked.

Workloads (synthetic data of the published shapes, seeded; BASELINE.json configs):
  fb15k237-transe   (default; configs[1]) 14 541 x 128 table, 52 870 test triples -> 105 740 queries,
                    filtering graph of 310 116 edges (the test triples + Zipf(0.8)-popular random edges)
  fb15k237-distmult / -complex / -simple   (configs[2]) same shapes, un-normalised table
  wikidata5m-transe (configs[3]) 4.6 M x 128 table (2.36 GB), reference batching: 2 triples = 4 queries per
                    table pass (scripts/blp-transe-wikidata5m.sh:18); a step = 64 passes (one launch of a streaming
                    kernel walks them all: roofline.passes_per_launch, kernel_ms = one pass's share)
  wikidata5m-complex   the same batching with ComplEx (BASELINE config 5's model; scripts/blp-complex-wikidata5m.sh:16-18)
  wikidata5m-transe-block / -complex-block   the same table, the 6 894 test triples as ONE query block
  wikidata5m-transe-full / -complex-full     the WHOLE Wikidata5M test evaluation as the reference batches it: 6 894 triples,
                    2 per table pass = 3 447 passes of the 2.36 GB table per step (~1.2 s: the sustained HBM figure next to
                    the 64-pass burst above; scripts/blp-transe-wikidata5m.sh:18)
  wikidata5m-protocol   the reference's OWN Wikidata5M candidate set (train.py:312-314: only the 7 475 entities of the test
                    split), 6 894 triples; also timed in the reference loop's layout (eval_batch_size = 2: 3 447 batches
                    through ONE blp_rank_all_batches call, and with a ranking pass per batch)
The default run reports fb15k237-distmult, fb15k237-complex, wikidata5m-transe (reference batching, with its filter),
wikidata5m-complex, wikidata5m-transe-block, wikidata5m-complex-block, wikidata5m-transe-full, wikidata5m-complex-full and
wikidata5m-protocol as `sub_results` (each with its own roofline), the training-side step as `inbatch_loss` and the
HBM-bound operating point as `hbm_probe` -- THE SAME NAMES for every N, so that the N = 1 line of a scaling run can be held
against the single-GPU line field by field.  With N > 1 the FB15k-237 workloads run on the axis
ranking.choose_shard_axis picks (query), everything at Wikidata5M scale on the north_star's CANDIDATE axis, and
`fb15k237-transe@candidate` is added; each sub-result carries `exchange_ms` (device events around the collectives, per
rank), `kernel_ms_per_rank` and the ranks the process group reports.

With N > 1 the evaluation is sharded along one axis (blp_amd.ranking.choose_shard_axis, --shard-axis):
  "candidate" (north_star; the default for the Wikidata5M-scale table): rank r ranks every query against table
              rows [lo, hi) through the same fused device path as one GPU (blp_rank_all_shard); the vectors the
              queries are made of are replicated once (all-gather of the table when it is small, else one all-reduce
              of the vectors of the entities in the test triples), ONE RCCL all-gather of the int32 counts + a sum per
              evaluation;
  "query"     (small table, many triples -- FB15k-237): the table is replicated, rank r ranks its slice of the
              triples, ONE all-gather of the per-triple counts.
Total work is fixed as N grows -> "scaling": "strong".

Only the cpu_baseline leg and the parity spot-check touch oracle/ (as the thing timed on the host / the
checker); the measured GPU path goes through libblp_hip.so only.
"""
import argparse
import ctypes
import json
import os
import re
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_PEAK_TFLOPS = 157.3         # f32 vector (FMA) peak == f32 MFMA peak; plain add/sub ops reach half
BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md); the bilinear pre-pass spends 3 bf16
BF16X3_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 3   # products per f32-equivalent product, so its roof is a third of it
HBM_MEASURED_GBPS = 6290.0      # the guide's measured float4-copy rate (79 % of spec): `peak_measured` beside the spec `peak`
# The TransE pre-pass issues v_sad_u16 (2 elements x (subtract + |.|-accumulate) per lane and instruction).  `peak` is the
# PUBLISHED rate (MI355X_MICROARCH.md: one VALU wave-instruction per 4 clocks and SIMD, 1 024 SIMDs, 2.4 GHz): 2 elements x 2 ops
# x 64 lanes x 1 024 SIMDs x 2.4 GHz / 4 = 157.3 Tops/s -- numerically the guide's f32 vector peak.  The issue rate measured with 4
# waves per SIMD (tools/sad_ubench.hip -> profiles/r04/sad_ubench.log) is 4.23 - 4.48 cycles per instruction and SIMD, taken as
# 4.3 -> 146.3 Tops/s: reported BESIDE the published figure as `peak_measured` / `frac_measured`, not instead of it.
SAD_CYCLES_PER_INST = 4.3
SAD_PUBLISHED_TOPS = 2 * 2 * 64 * 1024 * 2.4e9 / 4.0 / 1e12
SAD_PEAK_TOPS = 2 * 2 * 64 * 1024 * 2.4e9 / SAD_CYCLES_PER_INST / 1e12

WORKLOADS = {
    "fb15k237-transe": dict(model="transe", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116),
    "fb15k237-distmult": dict(model="distmult", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116),
    "fb15k237-complex": dict(model="complex", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116),
    "fb15k237-simple": dict(model="simple", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116),
    # Away from the friendly operating point: near-duplicate descriptions -> clustered rows (`clusters`
    # centres, every row = its centre + `noise` x randn, then normalised / scaled as usual) and a trained model's triples (the
    # true tail drawn from the `top` best-scoring entities of (h, r, ?)): many candidates tie or nearly tie with the true
    # entity, the pre-pass decides fewer pairs and the exact refinement gets more (`decided_frac`, `vs_random_step`).
    # noise = 0: exact duplicates (entities with the same description get the same vector), 29 copies of each row.
    "fb15k237-transe-clustered": dict(model="transe", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116,
                                      clusters=500, noise=0.0, top=145, random_twin="fb15k237-transe"),
    "fb15k237-distmult-clustered": dict(model="distmult", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116,
                                        clusters=500, noise=0.0, top=145, random_twin="fb15k237-distmult"),
    # ... and far away from it: 20 clusters of duplicates = 5 % of the table ties with every query's true entity.  The pre-pass's
    # lists run full and the exact kernel re-ranks the block (csrc/rank_common.h: Gate): the bounded worst case.  Not sub-results
    # (tools/clustered_sweep.py, tools/step_ab.py, tests/test_gpu_fullsize.py use them).
    "fb15k237-transe-ties5pct": dict(model="transe", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116,
                                     clusters=20, noise=0.0, top=145, random_twin="fb15k237-transe"),
    "fb15k237-distmult-ties5pct": dict(model="distmult", N=14541, D=128, R=237, triples=52870, block=65536, edges=310116,
                                       clusters=20, noise=0.0, top=145, random_twin="fb15k237-distmult"),
    # the BERT-BOW / BERT-DKRL width (models.py:118-135, 165-172): TransE at the word-embedding width 768
    "fb15k237-transe-d768": dict(model="transe", N=14541, D=768, R=237, triples=52870, block=65536, edges=310116),
    # reference batching: eval_batch_size = 2 triples per table pass, 64 passes per step; the reference's
    # Wikidata5M filtering graph is the evaluated split's own triples (train.py:381-393)
    "wikidata5m-transe": dict(model="transe", N=4_600_000, D=128, R=822, triples=128, block=2, edges=0),
    "wikidata5m-complex": dict(model="complex", N=4_600_000, D=128, R=822, triples=128, block=2, edges=0),
    # the same batching against the 16-bit COPY of the table (include/blp_hip.h: blp_rank_all_batches; SURVEY 8f row 2: "emit
    # fp16 copy"): the candidates are the table's rows rounded to IEEE half, scored in f32 in the reference's order -- a
    # different INPUT, the same arithmetic; the pass reads half the bytes
    "wikidata5m-transe-f16": dict(model="transe", N=4_600_000, D=128, R=822, triples=128, block=2, edges=0, table_dtype="float16"),
    "wikidata5m-complex-f16": dict(model="complex", N=4_600_000, D=128, R=822, triples=128, block=2, edges=0, table_dtype="float16"),
    # the same table with the whole Wikidata5M test set (6 894 triples) as ONE query block
    "wikidata5m-transe-block": dict(model="transe", N=4_600_000, D=128, R=822, triples=6894, block=65536, edges=0),
    "wikidata5m-complex-block": dict(model="complex", N=4_600_000, D=128, R=822, triples=6894, block=65536, edges=0),
    # the WHOLE test evaluation in the reference's batching: 6 894 triples, 2 per table pass -> 3 447 passes, ~1.2 s per step
    # (sustained; `heavy`: few timed steps inside the default run)
    "wikidata5m-transe-full": dict(model="transe", N=4_600_000, D=128, R=822, triples=6894, block=2, edges=0, heavy=True),
    "wikidata5m-complex-full": dict(model="complex", N=4_600_000, D=128, R=822, triples=6894, block=2, edges=0, heavy=True),
    # the reference's own Wikidata5M protocol: candidates = the entities of the evaluated split only (train.py:312-314)
    "wikidata5m-protocol": dict(model="transe", N=7475, D=128, R=822, triples=6894, block=65536, edges=0, loop_batch=2),
}
SUB_RESULTS = ("fb15k237-distmult", "fb15k237-complex", "fb15k237-transe-clustered", "fb15k237-distmult-clustered", "wikidata5m-transe", "wikidata5m-complex", "wikidata5m-transe-f16",
               "wikidata5m-complex-f16", "wikidata5m-transe-block",
               "wikidata5m-complex-block", "wikidata5m-transe-full", "wikidata5m-complex-full", "wikidata5m-protocol")
# N > 1: the same names on the axis a sharded evaluation of that shape takes ("auto": ranking.choose_shard_axis -> query for the
# FB15k-237-sized tables; the Wikidata5M-scale table always along the north_star's candidate axis), plus the headline
# workload on the candidate axis
SUB_RESULTS_EXTRA_SHARDED = (("fb15k237-transe", "candidate"),)
N1_REFERENCE_FILE = os.path.join("profiles", "n1_reference.json")  # committed one-GPU figures per workload: `vs_1gpu` of an N > 1 line


def headline_workload(world):
    """The workload on the TOP LEVEL of the line.  One GPU: BASELINE.json's metric configuration (FB15k-237 BLP-TransE).  Several:
    the quantity the north_star's scaling target is defined on -- the Wikidata5M-scale TransE ranking in the reference's batching
    (2 triples = 4 queries per table pass) sharded along the CANDIDATE axis, ONE all-gather of the int32 counts per evaluation;
    the FB15k-237 evaluations (query axis, and the candidate axis) are then sub-results."""
    return "fb15k237-transe" if world <= 1 else "wikidata5m-transe"


def n1_reference(workload):
    """{"value", "ms_per_step", "source"} of the committed one-GPU run of `workload` (profiles/n1_reference.json), or None."""
    try:
        with open(os.path.join(ROOT, N1_REFERENCE_FILE)) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


TABLE_CHUNK_ROWS = 1 << 16      # a Wikidata5M-scale table is generated in chunks of this many rows, each from its own seed


def make_table_rows(cfg, device, lo, hi, seed=1):
    """Rows [lo, hi) of a Wikidata5M-scale synthetic table, the same values whoever generates them: chunk c (rows
    [c, c + 1) x TABLE_CHUNK_ROWS) comes from its own seeded generator, so a rank of a candidate-axis shard makes ITS rows
    and nothing else (north_star: the table never moves; round 3 materialised 2.36 GB on every rank and sliced)."""
    N, D = cfg["N"], cfg["D"]
    out = torch.empty((hi - lo, D), dtype=torch.float32, device=device)
    for c in range(lo // TABLE_CHUNK_ROWS, (hi + TABLE_CHUNK_ROWS - 1) // TABLE_CHUNK_ROWS):
        r0, r1 = c * TABLE_CHUNK_ROWS, min((c + 1) * TABLE_CHUNK_ROWS, N)
        g = torch.Generator(device=device).manual_seed(seed * 1_000_003 + c + 1)
        rows = torch.randn(r1 - r0, D, device=device, generator=g)
        rows = torch.nn.functional.normalize(rows, dim=-1) if cfg["model"] == "transe" else rows * 0.1
        a, b = max(r0, lo), min(r1, hi)
        out[a - lo:b - lo] = rows[a - r0:b - r0]
    return out


def make_data(cfg, device, seed=1, sort=True, rows=None):
    """Synthetic FB15k-237 / Wikidata5M-shaped inputs: table rows L2-normalised for
    TransE (models.py:40-41), 0.1 * randn otherwise; rel_emb Xavier-uniform (models.py:28-29).  ``sort``:
    triples grouped by relation (tools that call ops.rank_all directly); the bench step takes them as they come.
    ``rows`` = (lo, hi): only these table rows are generated (tables of more than a million rows come in seeded chunks,
    make_table_rows; smaller ones are generated whole and sliced)."""
    g = torch.Generator(device=device).manual_seed(seed)
    N, D, R, T = cfg["N"], cfg["D"], cfg["R"], cfg["triples"]
    lo, hi = rows if rows is not None else (0, N)
    if N > 1_000_000:
        table = make_table_rows(cfg, device, lo, hi, seed)
    else:
        table = torch.randn(N, D, device=device, generator=g)
        table = torch.nn.functional.normalize(table, dim=-1) if cfg["model"] == "transe" else table * 0.1
        if rows is not None:
            table = table[lo:hi].contiguous()
    if cfg.get("clusters"):  # (small tables only: made whole)
        centres = torch.randn(cfg["clusters"], D, device=device, generator=g)
        member = torch.randint(0, cfg["clusters"], (N,), device=device, generator=g)
        table = centres[member] + cfg["noise"] * torch.randn(N, D, device=device, generator=g)
        table = torch.nn.functional.normalize(table, dim=-1) if cfg["model"] == "transe" else table * 0.1
    bound = (6.0 / (R + D)) ** 0.5
    rel_w = (torch.rand(R, D, device=device, generator=g) * 2 - 1) * bound
    heads = torch.randint(0, N, (T,), device=device, generator=g)
    tails = torch.randint(0, N, (T,), device=device, generator=g)
    rels = torch.randint(0, R, (T,), device=device, generator=g)
    if cfg.get("top"):
        tails = trained_tails(cfg, table, rel_w, heads, rels, g)
    if sort:
        order = torch.argsort(rels, stable=True)
        heads, tails, rels = heads[order], tails[order], rels[order]
    return table, rel_w, heads, tails, rels


def trained_tails(cfg, table, rel_w, heads, rels, g, chunk=2048):
    """A true tail per (h, r) the way a trained model ranks it: uniformly one of the cfg['top'] best-scoring entities of the
    tail query (h, r, ?) -- plain torch, set-up only (not timed): TransE -||h + r - t||_1 (models.py:222-223), DistMult
    sum(h * r * t) (models.py:226-227)."""
    out = torch.empty_like(heads)
    for lo in range(0, heads.shape[0], chunk):
        h, r = table[heads[lo:lo + chunk]], rel_w[rels[lo:lo + chunk]]
        scores = -torch.cdist(h + r, table, p=1) if cfg["model"] == "transe" else (h * r) @ table.T
        best = scores.topk(cfg["top"], dim=1).indices
        pick = torch.randint(0, cfg["top"], (best.shape[0], 1), device=heads.device, generator=g)
        out[lo:lo + chunk] = best.gather(1, pick)[:, 0]
    return out


def build_queries(table, rel_w, heads, tails, rels):
    """train.py:141-150: head-replacing queries first (fixed = tail emb), then tail-replacing."""
    q_fixed = torch.cat((table[tails], table[heads]))
    q_rel = torch.cat((rel_w[rels], rel_w[rels]))
    true_row = torch.cat((heads, tails))
    return q_fixed.contiguous(), q_rel.contiguous(), true_row.contiguous()


def make_filter_index(cfg, heads, tails, rels, seed=3):
    """The filtering graph (train.py:298-302: every known triple): the test triples themselves plus random edges
    up to cfg['edges'] in total, node popularity Zipf(0.8) over a random permutation of the entities, uniform
    relations (310 116 edges, seed 3).  Wikidata5M-style: the split's own triples only."""
    from blp_amd import utils
    test = torch.stack((heads, tails, rels), dim=1).cpu()
    extra = max(cfg["edges"] - test.shape[0], 0)
    if extra:
        g = torch.Generator().manual_seed(seed)
        weight = torch.arange(1, cfg["N"] + 1, dtype=torch.float64).pow(-0.8)[torch.randperm(cfg["N"], generator=g)]
        nodes = torch.multinomial(weight, 2 * extra, replacement=True, generator=g).reshape(2, extra)
        rel = torch.randint(0, cfg["R"], (extra,), generator=g)
        test = torch.cat((test, torch.stack((nodes[0], nodes[1], rel), dim=1)))
    # (timed: the one-off the evaluation's set-up pays -- the sort + unique of the graph's edges on the device, H2D of the edge
    #  list included; generating the synthetic edges above is not part of it.  Built twice: the first build of a process also
    #  loads the sort's code objects (100 - 200 ms once per process), the second is what the build itself costs)
    def build():
        if heads.is_cuda:
            torch.cuda.synchronize(heads.device)
        t0 = time.perf_counter()
        index = utils.FilterIndex(test, num_relations=cfg["R"], device=heads.device)
        if heads.is_cuda:
            torch.cuda.synchronize(heads.device)
        return index, (time.perf_counter() - t0) * 1e3

    _, first_ms = build()
    index, index.build_ms = build()
    index.first_build_ms = first_ms
    return index


def workload_axis(cfg, world, shard_axis="auto"):
    """The axis an evaluation of this workload is sharded along on `world` ranks: blp_amd.ranking.choose_shard_axis (query for
    FB15k-237-sized tables with many triples), the north_star's CANDIDATE axis for everything at Wikidata5M scale, or --shard-axis."""
    from blp_amd import ranking
    if world <= 1:
        return "none"
    axis = ranking.choose_shard_axis(cfg["N"], cfg["D"], 2 * cfg["triples"], world)
    if cfg["N"] > 1_000_000:
        axis = "candidate"
    return axis if shard_axis == "auto" else shard_axis


def run_plan(world, shard_axis="auto"):
    """What `python bench.py --gpus world` is going to do, computed without a GPU (`--plan`; tests/test_bench_host.py): per
    sub-result its axis, every rank's shard of the table / of the triples, the collectives of ONE step with their byte counts
    (blp_amd.ranking.exchange_plan = what rank_triples issues) and the steps it times."""
    from blp_amd import ranking
    todo = [("fb15k237-transe", "auto", "fb15k237-transe")] + [(n, "auto", n) for n in SUB_RESULTS]
    if world > 1:
        todo += [(n, a, f"{n}@{a}") for n, a in SUB_RESULTS_EXTRA_SHARDED]
    # (the top level of the line is headline_workload(world); every other entry is a sub-result)
    plan = {}
    for name, forced, key in todo:
        cfg = WORKLOADS[name]
        axis = workload_axis(cfg, world, forced if forced != "auto" else shard_axis)
        N, T = cfg["N"], cfg["triples"]
        rows = [ranking.shard_bounds(N, world, r) if axis == "candidate" else (0, N) for r in range(world)]
        triples = [ranking.shard_bounds(T, world, r) if axis == "query" else (0, T) for r in range(world)]
        half = bool(cfg.get("table_dtype"))
        plan[key] = {"axis": axis, "table_rows_per_rank": rows, "triples_per_rank": triples,
                     "table_bytes_per_rank": [(hi - lo) * cfg["D"] * (2 if half else 4) for lo, hi in rows],
                     "exchanges_per_step": ranking.exchange_plan(N, cfg["D"], T, world, axis, half_table=half),
                     "timed_steps": heavy_steps(world) if cfg.get("heavy") else None}
    return plan


def heavy_steps(world):
    """(timed steps, warm-up steps) of the whole-evaluation workloads (> 1 s per step on one GPU): two timed steps on one GPU,
    one on several -- `python bench.py --gpus 8` times 12 sub-results with every rank in each, and stays under two minutes."""
    return (2, 1) if world == 1 else (1, 0)


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


class Job:
    """One workload prepared on this rank's device; step() is one whole evaluation."""

    def __init__(self, name, device, world=1, rank=0, shard_axis="auto", backend="nccl"):
        from blp_amd import models, ranking
        self.name, self.cfg, self.device, self.world, self.rank, self.backend = name, WORKLOADS[name], device, world, rank, backend
        cfg = self.cfg
        self.N, self.D, self.T = cfg["N"], cfg["D"], cfg["triples"]
        self.axis = workload_axis(cfg, world, shard_axis)  # (north_star: the 4.6 M-entity ranking along the candidate axis)
        self.lo, self.hi = ranking.shard_bounds(self.N, world, rank) if self.axis == "candidate" else (0, self.N)
        # a candidate shard of a Wikidata5M-scale table generates its own rows only; small tables are made whole (rank 0
        # keeps the whole one for the parity spot check) and sliced
        local_only = self.axis == "candidate" and self.N > 1_000_000
        table, rel_w, heads, tails, rels = make_data(cfg, device, sort=False, rows=(self.lo, self.hi) if local_only else None)
        self.model = models.LinkPrediction(cfg["D"], cfg["model"], "margin", cfg["R"], 0)
        self.model.rel_emb.weight.data = rel_w.cpu()
        self.model = self.model.to(device)
        self.triples = torch.stack((heads, tails, rels), dim=1).contiguous()   # (T, 3) entity / relation ids
        self.ent2idx = torch.arange(self.N, device=device)                    # ids are table rows in the synthetic sets
        # one-off cost outside the timed step, reported (the reference builds its networkx graph once as well, train.py:298-302)
        self.index = make_filter_index(cfg, heads, tails, rels)
        self.filter_index_build_ms, self.filter_index_first_build_ms = self.index.build_ms, self.index.first_build_ms
        self.index.segments(self.triples[:1], self.ent2idx, device)           # sorted arrays resident before the clock starts
        self.table_dtype = getattr(torch, cfg["table_dtype"]) if cfg.get("table_dtype") else torch.float32
        if self.table_dtype != torch.float32:  # the 16-bit copy IS the table of this workload; checks see it widened (exact)
            table = table.to(self.table_dtype)
        if local_only:
            self.full_table, self.table = None, table
        elif self.table_dtype != torch.float32:
            self.full_table = table.float() if self.N <= 100_000 and (rank == 0 or self.axis != "candidate") else None  # (the oracle's spot check only)
            self.table = table[self.lo:self.hi].contiguous() if self.axis == "candidate" else table
        else:
            self.full_table = table if rank == 0 or self.axis != "candidate" else None  # rank 0 keeps it for the parity check
            self.table = table[self.lo:self.hi].contiguous() if self.axis == "candidate" else table
        del table
        self.ranking = ranking
        self.heavy = bool(cfg.get("heavy"))
        # Initialisation, not warm-up: the first two evaluations of a process load the kernels' code objects and grow
        # the caching allocator to its steady state (a one-off of tens of ms shows up in the SECOND evaluation).
        for filtered in ((True,) if self.heavy else (True, True, False)):
            self.step(filtered)
        self.fence()

    def step(self, filtered=True, timing=None):
        from blp_amd import ops
        triples, counts, ids_ok = self.ranking.rank_triples(
            self.model, self.table, self.triples, self.ent2idx, self.index if filtered else None,
            num_entities=self.N, world=self.world, rank=self.rank, axis=self.axis if self.world > 1 else "candidate",
            block_size=self.cfg["block"], timing=timing)
        sums = ops.rank_metric_sums(counts)
        return triples, counts, sums

    def timed(self, steps, warmup, filtered=True):
        """K steps between barrier + synchronize on both sides; the MAX over ranks."""
        for _ in range(warmup):
            self.step(filtered)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = self.step(filtered)
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.backend == "gloo" else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        return elapsed, out

    def fence(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def kernel_ms(self, events, reps):
        """Duration of the ranking pass proper (pre-pass + exact refinement, or the exact kernel), measured with HIP
        events recorded by the library on the stream the kernels run on (blp_profile_next_rank_kernel arms the
        NEXT blp_rank_all of this thread: the first block of a step)."""
        from blp_amd import _lib
        pairs = []
        for _ in range(reps):
            a, b = events.pair()
            _lib.check(_lib.lib().blp_profile_next_rank_kernel(a, b), "blp_profile_next_rank_kernel")
            self.step()
            pairs.append((a, b))
        torch.cuda.synchronize()
        # a reference-batched evaluation (a table pass per `block` triples) may be ONE launch of a streaming kernel that
        # walks all its passes: the events then bracket all of them, and one pass is its share
        return sum(events.elapsed_ms(a, b) for a, b in pairs) / len(pairs) / self.passes_per_launch()

    def passes_per_launch(self):
        from blp_amd import _lib
        block = min(self.cfg["block"], self.T)
        if block >= self.T:
            return 1
        n_local = self.hi - self.lo
        model_id = {"transe": 0, "distmult": 1, "complex": 2, "simple": 3}[self.cfg["model"]]
        from blp_amd import ops
        return max(1, int(_lib.lib().blp_rank_all_batches_passes_per_launch(model_id, ops.TABLE_DTYPES[self.table_dtype], n_local, self.D,
                                                                               self.D, self.T, block, block)))

    def exchange_ms(self, reps):
        """Milliseconds per step inside the collectives of this rank (device events around each exchange on the stream
        the kernels run on; includes waiting for the slowest rank).  0 on one GPU."""
        if self.world == 1:
            return 0.0
        total = 0.0
        for _ in range(reps):
            timing = {}
            self.step(True, timing=timing)
            torch.cuda.synchronize()
            total += self.ranking.exchange_ms(timing)
        return total / reps

    def prepass_stats(self):
        """What the pre-pass leaves to the exact path on this workload's block (one GPU): the whole query block through a bare
        ops.rank_all on a workspace of our own, then include/blp_hip.h: blp_rank_all_prepass_stats on it.  Not timed."""
        from blp_amd import ops
        if self.world > 1 or self.cfg["block"] < self.T or self.table.dtype != torch.float32:
            return None
        rel_w = self.model.rel_emb.weight.detach()
        h, t, r = self.triples[:, 0], self.triples[:, 1], self.triples[:, 2]
        q_fixed, q_rel, true_row = build_queries(self.table, rel_w, h, t, r)
        if not ops.rank_all_supported(self.cfg["model"], self.D, self.T, self.T):
            return None
        ws = torch.empty(ops.rank_all_workspace_bytes(self.cfg["model"], self.N, self.D, self.T, self.T), dtype=torch.uint8, device=self.device)
        ops.rank_all(self.cfg["model"], self.table, q_fixed, q_rel, self.T, true_row=true_row, workspace=ws)
        return ops.prepass_stats(self.cfg["model"], self.N, self.D, self.T, self.T, ws)

    def loop_layout(self, reps=5):
        """The same evaluation handed over in the REFERENCE LOOP's layout (train.py:128-157: batch after batch of
        eval_batch_size = cfg['loop_batch'] triples, each [its head queries | its tail queries]): one blp_build_queries +
        ONE blp_rank_all_batches call for all batches (the library merges them into blocks), and the same call with a
        ranking pass per batch (block_triples = batch: the reference's pass structure, 3 launches per batch)."""
        from blp_amd import ops
        b, rel_w = self.cfg["loop_batch"], self.model.rel_emb.weight

        def run(block_triples):
            qb = ops.build_queries(self.triples, self.ent2idx, self.table, rel_w, b, index=self.index, gather=False, num_rows=self.N)
            counts = ops.rank_all_batches(self.cfg["model"], self.table, qb.fixed_row, rel_w, qb.rel_ids, qb.true_row, self.T, b,
                                          filter=qb.filter, block_triples=block_triples)
            return ops.rank_metric_sums(counts)

        out = {"eval_batch_size": b, "batches": -(-self.T // b)}
        for key, block_triples, n in (("ms_one_call_all_batches", 0, reps), ("ms_one_call_pass_per_batch", b, 2)):
            run(block_triples)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                sums = run(block_triples)
            torch.cuda.synchronize()
            out[key] = (time.perf_counter() - t0) / n * 1e3
            out[key.replace("ms_", "mrr_")] = sums[0].item() / (2 * self.T)
        return out

    def roofline(self, kernel_ms):
        """Roof of the dominant kernel: algorithmic work of ONE ranking launch on this rank / its time."""
        model, D, Q = self.cfg["model"], self.D, 2 * min(self.T, self.cfg["block"])
        if self.axis == "query":
            t_lo, t_hi = self.ranking.shard_bounds(self.T, self.world, self.rank)
            Q = 2 * (t_hi - t_lo)
        n_local = self.hi - self.lo
        alg_flops = 2.0 * n_local * D * Q          # TransE: subtract + |.|-accumulate per element; bilinear: the GEMM
        alg_bytes = n_local * D * self.table.element_size() + Q * (2 * D * 4 + 24)
        t_k = kernel_ms * 1e-3
        transe = model == "transe"
        peak_tf = SAD_PEAK_TOPS if transe else BF16X3_PEAK_TFLOPS
        if alg_bytes / (HBM_PEAK_GBPS * 1e9) >= alg_flops / (peak_tf * 1e12):
            ppl = self.passes_per_launch()
            ring = ppl > 1 or (not transe and (model == "complex" or n_local < 1_700_000)) or (transe and n_local < 1_700_000)
            dot = "approximate keys (a chain of fused multiply-adds) decided within a band, undecided rows re-scored in the reference's order; "
            roof = {"bound": "hbm", "achieved": alg_bytes / t_k / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "peak_measured": HBM_MEASURED_GBPS,  # (MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy)
                    "arith": ("f32, the reference's operations in its order (2-3 VALU operations per element and query)" if transe or not ring else
                              "f32 fused multiply-add chain per (row, query) decided within a proven band against the exact true key; "
                              "undecided rows re-scored in the reference's f32 order"),
                    "kernel": (((("rank_stream16_kernel" if transe else "rank_stream_dot16_kernel") + f" [a {self.cfg['table_dtype']} table, widened exactly] (" if self.table.element_size() == 2 else "") +
                               ("rank_stream_kernel (exact f32 keys; " if transe else "rank_stream_dot_kernel (" + dot)) +
                               "the table streamed once per pass through per-wave rings of 32-column pieces" +
                               (f"; all {ppl} passes of the step in one launch, kernel_ms = its share of one pass)" if ppl > 1 else ")")) if ring else
                              "rank_stream_wg_kernel (exact f32 keys; the table streamed once through workgroup tiles, two tiles in flight per workgroup)"}
        elif transe:
            wide = D not in (64, 128, 256)
            roof = {"bound": "valu", "achieved": alg_flops / t_k / 1e12, "peak": SAD_PUBLISHED_TOPS, "unit": "TFLOP/s",
                    "peak_measured": SAD_PEAK_TOPS,
                    "arith": "u16 fixed-point v_sad_u16 pre-pass (decides >= 99.5 % of the pairs within a proven band) + exact f32 "
                             "re-scoring of the rest in the reference's order; NOT the reference's f32 lane-ops",
                    "peak_source": "peak: MI355X_MICROARCH.md's VALU issue rate (one wave-instruction per 4 clocks and SIMD) -> 2 elements "
                                   "x 2 ops x 64 lanes x 1024 SIMDs x 2.4 GHz / 4 = 157.3 Tops/s.  peak_measured: the v_sad_u16 issue rate "
                                   "measured by tools/sad_ubench.hip -> profiles/r04/sad_ubench.log (4.23 - 4.48 cycles per instruction "
                                   "and SIMD at 4 waves/SIMD, taken as 4.3) -> 146.3 Tops/s",
                    "kernel": ("wide_rank_sad_kernel + wide_refine_* (any-width u16 v_sad_u16 pre-pass)" if wide else
                               "rank_sad_kernel<128> + sad_refine_* (u16 fixed-point v_sad_u16 pre-pass + band + exact f32 "
                               "refinement; range / quantise kernels included)"),
                    "note": "VALU roof (an L1 norm has no matrix-core form): 157.3 Tops/s is v_sad_u16 at the published 4 cycles per "
                            "instruction (2 elements x (subtract + |.|-accumulate) x 64 lanes); 146.3 at its measured 4.3 "
                            "(peak_measured); the exact f32 add/sub kernel tops out at half of that.  achieved = 2 ops x D x Q x N "
                            "/ time of the whole rank pass.  SURVEY 8(d)'s own accounting for the reference's f32 arithmetic "
                            "-- N x Q x D x c lane-ops, c = 3 on the head side (add r, subtract t, |.|-accumulate) and 2 on the "
                            "tail side, against the 78.6 T/s non-FMA f32 lane-op rate -- is in `survey_8d_lane_ops`: a frac "
                            "above 1 there says the kernel does NOT execute those f32 lane-ops (it decides >= 99.8 % of the "
                            "pairs in 16-bit fixed point, two elements per v_sad_u16, and re-scores the rest exactly).",
                    "survey_8d_lane_ops": {"lane_ops_per_launch": 2.5 * n_local * D * Q,
                                           "achieved_Tops": 2.5 * n_local * D * Q / t_k / 1e12, "peak_Tops": F32_PEAK_TFLOPS / 2,
                                           "frac": 2.5 * n_local * D * Q / t_k / 1e12 / (F32_PEAK_TFLOPS / 2)}}
        else:
            roof = {"bound": "mfma", "achieved": alg_flops / t_k / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                    "arith": "bf16 x 3 split products on v_mfma_f32_32x32x16_bf16 (f32 accumulate) decided within a band + exact "
                             "f32 re-scoring of the undecided pairs in the reference's order",
                    "kernel": f"rank_gemm_bf16_kernel<{model.upper()},128> + refine_* (bf16 x 3 split MFMA GEMM + band + exact "
                              "f32 refinement)",
                    "note": "achieved = 2 flops x D x Q x N (the f32 GEMM the reference's scores amount to) / time of GEMM "
                            "pre-pass + exact refinement.  Every f32 product is three bf16 MFMA products (hi*hi + hi*lo + "
                            "lo*hi), so the roof is the dense bf16 MFMA peak / 3 = 833 TF (5.3x the f32 MFMA peak).",
                    "mfma_busy": None}
        if self.table.element_size() == 2:
            roof["arith"] += f"; the candidate table is a {self.cfg['table_dtype']} copy, every element widened to f32 (exactly) before it is used"
        roof["frac"] = roof["achieved"] / roof["peak"]
        if "peak_measured" in roof:
            roof["frac_measured"] = roof["achieved"] / roof["peak_measured"]
        roof["kernel_ms"] = kernel_ms
        roof["passes_per_launch"] = self.passes_per_launch()  # > 1: kernel_ms, traffic and the algorithmic figures are one pass's share
        # HBM bytes / matrix-pipe busy fraction of the dominant kernel come from the committed rocprofv3 PMC passes (a
        # profiler cannot run inside this process): stamped with the round / commit they were taken at, and dropped
        # (null) on more than one GPU or when the profiled kernel no longer looks like the live one.
        roof["traffic"], roof["pmc_source"] = None, None
        pmc, why = load_pmc(self.name, live_pass_ms=kernel_ms if self.world == 1 else None)
        if pmc is not None:
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            if "mfma_busy" in roof:
                roof["mfma_busy"] = pmc.get("mfma_busy_frac_at_2.4GHz")
        roof["pmc_source"] = why
        roof["algorithmic_bytes_per_launch"] = alg_bytes
        roof["algorithmic_flops_per_launch"] = alg_flops
        return roof

    def measure(self, steps, warmup, events):
        """The JSON fields of this workload: whole evaluation (raw + filtered) and raw-only, the roofline."""
        if self.heavy:  # a step takes seconds: two timed steps (one on several GPUs), one of everything else
            cap_steps, cap_warmup = heavy_steps(self.world)
            steps, warmup = min(steps, cap_steps), min(warmup, cap_warmup)
        raw_steps = 1 if self.heavy else max(1, min(steps, 5))
        elapsed, (triples, counts, sums) = self.timed(steps, warmup, filtered=True)
        raw_elapsed, _ = self.timed(raw_steps, 0 if self.heavy else 1, filtered=False)
        kernel_ms = self.kernel_ms(events, 1 if self.heavy else max(1, min(steps, 10)))
        exchange = self.exchange_ms(1 if self.heavy else max(1, min(steps, 5)))
        scored = 2.0 * self.T * self.N
        per_rank, exchange_per_rank = [kernel_ms], [exchange]
        if self.world > 1:
            box = [None] * self.world
            dist.all_gather_object(box, (kernel_ms, exchange))
            per_rank, exchange_per_rank = [b[0] for b in box], [b[1] for b in box]
        sums = sums.cpu()
        out = {
            "value": scored * steps / elapsed,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_raw_only": raw_elapsed / raw_steps * 1e3, "timed_steps": steps,
            "mrr": sums[0].item() / (2 * self.T), "mrr_filtered": sums[1].item() / (2 * self.T),
            "hits@1,3,10": [x.item() / (2 * self.T) for x in sums[2:5]],
            "hits@1,3,10_filtered": [x.item() / (2 * self.T) for x in sums[5:8]],
            "roofline": self.roofline(kernel_ms),
            "kernel_ms_per_rank": per_rank,
            "exchange_ms": max(exchange_per_rank), "exchange_ms_per_rank": exchange_per_rank,
            "shard_axis": self.axis, "ranks": self.world,
            "filter_index_build_ms": self.filter_index_build_ms,  # once per evaluation set-up, NOT inside ms_per_step
            "filter_index_first_build_ms_in_this_process": self.filter_index_first_build_ms,  # (+ loading the sort's code objects)
            "filter_index_edges": self.index.num_edges,
        }
        return out, triples, counts


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(job, budget_s=10.0):
    """The reference's CPU path restated (oracle/ref_port.py: same torch CPU kernels, incl. the dense filter mask
    overwrite of train.py:159-171), timed on this box's host cores on a bounded sample of reference batches
    (eval_batch_size = 64 triples for FB15k-237, 2 for Wikidata5M: scripts/blp-transe-{fb15k237,wikidata5m}.sh:18),
    with all host threads and with one."""
    from oracle import ref_port
    cfg = job.cfg
    B = 64 if cfg["N"] < 1_000_000 else 2
    tab, rw = job.full_table.cpu(), job.model.rel_emb.weight.detach().cpu()
    trip = job.triples.cpu()
    T = trip.shape[0]
    ent2idx = torch.arange(cfg["N"])

    def batch(i):
        lo = (i * B) % max(T - B + 1, 1)
        t = trip[lo:lo + B]
        hf, tf = job.index.masks(t, cfg["N"], ent2idx)  # the vectorised filter, not the reference's networkx walk
        return ref_port.eval_batch(cfg["model"], tab, t[:, 0], t[:, 1], rw[t[:, 2]], filter_mask=torch.cat((hf, tf)))

    def run(budget, max_batches):
        batch(0)  # warm-up
        done, t0 = 0, time.perf_counter()
        while True:
            batch(done + 1)
            done += 1
            el = time.perf_counter() - t0
            if el >= budget or done >= max_batches:
                return done, el

    # best of a few thread counts: 128 threads on a 128-way box were SLOWER than one (oversubscribed reductions over a
    # (128, 14 541) matrix); each leg is bounded, ~20 s of CPU work in all
    all_threads = torch.get_num_threads()
    legs = {}
    for threads in sorted({1, 8, 32, all_threads}):
        if threads > all_threads:
            continue
        torch.set_num_threads(threads)
        done, el = run(budget_s / 2, 64 if threads > 1 else 6)
        legs[threads] = (2.0 * B * cfg["N"] * done / el, done, el)
    torch.set_num_threads(all_threads)
    best = max(legs, key=lambda k: legs[k][0])
    value, done, el = legs[best]

    # A second CPU figure, for scale: the order-exact C oracle (oracle/blp_oracle.c: scalar loops in the reference's operation
    # order, OpenMP over the queries of a side, CSR filter) on the same batches -- NOT what the reference runs (its CPU path is
    # the torch expressions above), but what a plain C restatement of it reaches on these host cores.
    def c_oracle_leg(budget):
        import numpy as np
        from oracle import oracle as orc
        tab_np, rw_np = tab.numpy(), rw.numpy()

        def cbatch(i):
            lo = (i * B) % max(T - B + 1, 1)
            t = trip[lo:lo + B]
            rowptr, col = job.index.csr(t, ent2idx)
            b = t.shape[0]
            rel = rw_np[t[:, 2].numpy()]
            orc.rank_counts(cfg["model"], orc.SIDE_HEAD, tab_np, tab_np[t[:, 1].numpy()], rel, true_row=t[:, 0].numpy(),
                            filt_rowptr=rowptr[:b + 1].numpy(), filt_col=col[:rowptr[b]].numpy())
            orc.rank_counts(cfg["model"], orc.SIDE_TAIL, tab_np, tab_np[t[:, 0].numpy()], rel, true_row=t[:, 1].numpy(),
                            filt_rowptr=(rowptr[b:] - rowptr[b]).numpy(), filt_col=col[rowptr[b]:].numpy())

        cbatch(0)
        n, t0 = 0, time.perf_counter()
        while True:
            cbatch(n + 1)
            n += 1
            e = time.perf_counter() - t0
            if e >= budget or n >= 256:
                break
        return {"value": 2.0 * B * cfg["N"] * n / e, "unit": "scored triples/s", "threads": min(B, os.cpu_count() or 1),
                "kind": "port: oracle/blp_oracle.c (scalar C in the reference's operation order, OpenMP over the <= "
                        f"{B} queries of a side; CSR filter)", "sample": f"{n} reference batches of {B} triples, {e:.1f} s"}

    try:
        c_leg = c_oracle_leg(3.0)
    except Exception as exc:  # (a baseline beside the baseline: never the reason a bench line is lost)
        c_leg = {"error": repr(exc)}
    return {"value": value, "unit": "scored triples/s", "cores": best, "c_oracle": c_leg,
            "kind": "port, vectorised filter (the reference's own networkx walk adds ~190 ms per batch: SURVEY.md 8a)",
            "value_by_threads": {str(k): v[0] for k, v in legs.items()}, "value_1_thread": legs[1][0],
            "cpu_model": cpu_model_name(), "logical_cpus": os.cpu_count(),
            "sample": f"{done} reference batches of {B} triples ({2 * B} queries) x {cfg['N']} candidates, raw + filtered, "
                      f"{el:.1f} s of torch-CPU work on {best} threads (the best of {sorted(legs)} threads, each leg bounded)"}


def torch_gpu_baseline(job, budget_s=3.0):
    """The reference's own expressions (torch broadcasting score_fn + get_metrics, restated in
    oracle/ref_port.py) on THIS GPU through stock PyTorch-ROCm, reference batch size, raw ranking only -- what the
    unmodified reference's scoring does on a MI355X.  Bounded sample; a reported baseline, not the target."""
    from oracle import ref_port
    cfg = job.cfg
    if cfg["N"] > 1_000_000:
        return None
    B, T = 64, job.T
    rel_w = job.model.rel_emb.weight.detach()

    def batch(i):
        lo = (i * B) % max(T - B + 1, 1)
        t = job.triples[lo:lo + B]
        out = ref_port.eval_batch(cfg["model"], job.full_table, t[:, 0], t[:, 1], rel_w[t[:, 2]])
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
            "PyTorch-ROCm (same GPU), raw ranking only", "sample": f"{done} reference batches of {B} triples, {el:.2f} s"}


def parity_spot_check(job, triples, counts, n=32):
    """First n head- and first n tail-queries of the measured evaluation (relation-sorted order) against the CPU
    oracle, raw and filtered counts."""
    import numpy as np
    from oracle import oracle as orc
    cfg = job.cfg
    if cfg["N"] > 100_000:
        return "skipped (table too large for the CPU oracle inside bench; see tests)"
    tab = job.full_table.cpu().numpy()
    t = triples[:n].cpu()
    rel = job.model.rel_emb.weight.detach().cpu()[t[:, 2]].numpy()
    rowptr, col = job.index.csr(t, torch.arange(cfg["N"]))
    b = t.shape[0]
    T = triples.shape[0]
    want_h = orc.rank_counts(cfg["model"], orc.SIDE_HEAD, tab, tab[t[:, 1].numpy()], rel, true_row=t[:, 0].numpy(),
                             filt_rowptr=rowptr[:b + 1].numpy(), filt_col=col[:rowptr[b]].numpy())
    want_t = orc.rank_counts(cfg["model"], orc.SIDE_TAIL, tab, tab[t[:, 0].numpy()], rel, true_row=t[:, 1].numpy(),
                             filt_rowptr=(rowptr[b:] - rowptr[b]).numpy(), filt_col=col[rowptr[b]:].numpy())
    ok = np.array_equal(want_h, counts[:b].cpu().numpy()) and np.array_equal(want_t, counts[T:T + b].cpu().numpy())
    return f"{2 * b} queries vs CPU oracle (raw + filtered): " + ("identical counts" if ok else "MISMATCH")


def load_pmc(workload, live_pass_ms=None):
    """The dominant kernel's figures from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py): HBM bytes per launch (FETCH_SIZE) and the matrix-pipe busy fraction
    (SQ_VALU_MFMA_BUSY_CYCLES).  Returns (entry or None, provenance): the entry is dropped when there is no live
    single-GPU pass time to hold it against or when the ranking pass of the profiled run (`bench_pass_ms`) differs
    from the live one by more than 10 % -- the counters then describe another kernel than the one just timed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    # (the whole-evaluation workloads run the very launch of the 64-pass ones, 3 447 passes long: without counters of their
    #  own, one pass's share of the 64-pass launch is the same figure)
    if workload.endswith("-full") and workload not in doc:
        workload = workload[:-5]
    entry, stamp = doc.get(workload), doc.get("_profile", {})
    where = f"profiles/{stamp.get('round', '?')}/{workload} (rocprofv3 --pmc, commit {stamp.get('commit', '?')})"
    if not entry:
        return None, f"no PMC pass recorded for {workload}"
    if live_pass_ms is None:
        return None, f"{where}: not applicable to this run (more than one GPU)"
    recorded = entry.get("bench_pass_ms")
    if recorded is None:
        return None, f"{where}: no pass time recorded with the counters"
    if abs(recorded - live_pass_ms) > 0.10 * live_pass_ms:
        return None, f"{where}: stale -- profiled pass {recorded:.4g} ms, live pass {live_pass_ms:.4g} ms"
    return entry, f"{where}; profiled pass {recorded:.4g} ms vs live {live_pass_ms:.4g} ms"


def hbm_probe(device, events, reps=30):
    """The exact ranking kernel at its HBM-bound operating point (BASELINE config 4 at 1 GPU): 4.6 M x 128 f32
    table = 2.355 GB, reference batching 2 triples = 4 queries per table pass, the bare C-ABI call."""
    from blp_amd import _lib, ops
    cfg = WORKLOADS["wikidata5m-transe"]
    table, rel_w, heads, tails, rels = make_data(cfg, device, seed=5)
    q_fixed, q_rel, true_row = build_queries(table, rel_w, heads[:2], tails[:2], rels[:2])
    for _ in range(150):  # (~55 ms: after the host-side table generation the clocks take tens of ms to come back up)
        ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
    pairs = []
    for _ in range(reps):
        a, b = events.pair()
        _lib.check(_lib.lib().blp_profile_next_rank_kernel(a, b), "blp_profile_next_rank_kernel")
        ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
        pairs.append((a, b))
    torch.cuda.synchronize()
    ms = sum(events.elapsed_ms(a, b) for a, b in pairs) / len(pairs)
    call_ms = float("inf")
    for _ in range(3):  # (the better of three loops: a one-off host stall inside one of them read as 2 ms per call)
        t0 = time.perf_counter()
        for _ in range(10):
            ops.rank_metrics(ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row))
        torch.cuda.synchronize()
        call_ms = min(call_ms, (time.perf_counter() - t0) / 10 * 1e3)
    alg_bytes = cfg["N"] * cfg["D"] * 4 + 4 * (2 * cfg["D"] * 4 + 24)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    pmc, why = load_pmc("wikidata5m-transe", live_pass_ms=ms)
    return {"workload": "wikidata5m-transe, 4 queries per table pass (bare blp_rank_all call)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "kernel_ms": ms,
            "whole_call_ms": call_ms, "scored_triples_per_s": 4.0 * cfg["N"] / (call_ms * 1e-3),
            "traffic": (pmc or {}).get("hbm_bytes_per_launch"), "pmc_source": why}


INBATCH_SHAPES = {
    # BASELINE config 1's training batch (scripts/blp-transe-fb15k237.sh: batch_size 64, num_negatives 64, dim 128)
    "inbatch-fb15k237": dict(model="transe", loss="margin", B=64, K=64, D=128, dtype="float32", reg=0.0),
    # BASELINE config 5: ComplEx, half-precision embeddings under autocast (f32 relation rows), per-GPU batch 128 -- and
    # the reference's whole Wikidata5M batch of 1 024 on one GPU (scripts/blp-complex-wikidata5m.sh)
    "inbatch-wikidata5m-complex-fp16": dict(model="complex", loss="margin", B=128, K=64, D=128, dtype="float16", reg=1e-3),
    "inbatch-wikidata5m-complex-fp16-b1024": dict(model="complex", loss="margin", B=1024, K=64, D=128, dtype="float16", reg=1e-3),
}


def inbatch_bench(device, iters=200):
    """The training-side kernel (compute_loss on in-batch negatives, models.py:51-70): one step
    = fused forward + backward, two launches (three when the forward has more than 96 scoring workgroups).  `us_per_step_kernels`: the raw C-ABI calls issued back to back (device
    events; what the kernels and their launch gaps cost); `us_per_step_autograd`: ops.inbatch_loss(...).backward() from
    Python, wall clock; `torch_us_per_step`: the reference's expressions (oracle/ref_port.py) through stock PyTorch-ROCm on
    the same tensors.  Launch / latency-bound (~100 KB of data): microseconds, not a roofline fraction."""
    from blp_amd import _lib, ops
    from oracle import ref_port
    out = {}
    # The engine's hand-over to its per-device worker thread and back is host behaviour, not ours, and it drifts: in a fresh
    # process a backward() of a node WITHOUT kernels costs 50 - 65 us for the first seconds and 24 - 27 us later (the idle
    # governor learns the worker's wake-up pattern; tools/autograd_floor_probe.py, profiles/r05/autograd_floor_probe.log).
    # Two seconds of such backward() calls go first, so that the shapes below are all measured in the settled state -- and
    # every figure has its floor measured right beside it (`us_node_cost` = the difference = what this package adds).
    if ops.torch_glue() is not None:
        e0 = torch.zeros(64, 2, 128, device=device, requires_grad=True)
        r0 = torch.zeros(64, 1, 128, device=device, requires_grad=True)
        n0 = torch.zeros(64, 64, 2, dtype=torch.int64, device=device)
        t_end = time.perf_counter() + 2.0
        while time.perf_counter() < t_end:
            e0.grad = r0.grad = None
            ops.torch_glue().autograd_floor(e0, r0, n0).backward()
        torch.cuda.synchronize()
    for name, c in INBATCH_SHAPES.items():
        g = torch.Generator(device=device).manual_seed(7)
        B, K, D = c["B"], c["K"], c["D"]
        dtype = getattr(torch, c["dtype"])
        ent = (torch.randn(B, 2, D, device=device, generator=g) * 0.4).to(dtype)
        rel = torch.randn(B, 1, D, device=device, generator=g) * 0.3
        neg_idx = torch.randint(0, 2 * B, (B, K, 2), device=device, generator=g)

        # leaves made once; a step = forward + backward into fresh .grad tensors (what compute_loss(...).backward() costs the
        # caller -- round 3 also timed two clone kernels per step here)
        e_leaf, r_leaf = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)

        def fused():
            e_leaf.grad = r_leaf.grad = None
            ops.inbatch_loss(c["model"], c["loss"], e_leaf, r_leaf, neg_idx, c["reg"]).backward()

        def forward_only():
            with torch.no_grad():
                ops.inbatch_loss(c["model"], c["loss"], e_leaf, r_leaf, neg_idx, c["reg"])

        def floor():  # a node of the same shape that launches nothing: autograd's own cost per step on this host
            e_leaf.grad = r_leaf.grad = None
            ops.torch_glue().autograd_floor(e_leaf, r_leaf, neg_idx).backward()

        def fused_in_graph():  # one stock node upstream, as an encoder would be (what the node costs inside a training graph)
            e_leaf.grad = r_leaf.grad = None
            ops.inbatch_loss(c["model"], c["loss"], e_leaf * 1.0, r_leaf, neg_idx, c["reg"]).backward()

        def floor_in_graph():
            e_leaf.grad = r_leaf.grad = None
            ops.torch_glue().autograd_floor(e_leaf * 1.0, r_leaf, neg_idx).backward()

        def stock():
            e_leaf.grad = r_leaf.grad = None
            ref_port.compute_loss(c["model"], c["loss"], e_leaf, r_leaf, neg_idx, c["reg"]).backward()

        def wall(fn, n, settle_s=0.25, rounds=3):
            # Steady state: the step is issued back to back for `settle_s` seconds before the clock starts, then the best of three
            # loops.  Round 4 timed 200 steps after 20: backward() hands its nodes to the engine's per-device worker thread, and
            # for the first ~100 ms of a burst that hand-over costs ~35 us more per step than afterwards (the worker's core comes
            # out of an idle state; tools/autograd_floor_probe.py: whichever variant is measured FIRST reads 60 us, every later
            # one 27) -- a training loop calls backward() continuously, so the settled figure is the one a step pays.
            t_end = time.perf_counter() + settle_s
            while time.perf_counter() < t_end:
                for _ in range(20):
                    fn()
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(rounds):
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n * 1e6)
            return best

        def wall_spread(fn, n, settles=5):
            # the same figure as a DISTRIBUTION: `settles` independent settle-then-time rounds, each the mean of n steps; the
            # median is what the line reports, best and p90 beside it (a fresh box once read 64 - 87 us where a settled one read
            # 39 - 42: VERDICT r05 -- one best-of-three hides which of the two a caller gets)
            rounds = sorted(wall(fn, n, rounds=1) for _ in range(settles))
            return {"median": rounds[len(rounds) // 2], "best": rounds[0], "p90": rounds[min(len(rounds) - 1, int(0.9 * len(rounds)))],
                    "rounds": rounds}

        def cold(fn, n=50):
            # ... and the first steps of a burst: after 50 ms without a backward() (the worker thread asleep), the mean of the next n
            torch.cuda.synchronize()
            time.sleep(0.05)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6

        def wall_single_thread(fn, n):
            # the same with the engine's device threads off (torch.autograd.set_multithreading_enabled(False): backward()
            # then runs its nodes on the calling thread -- no hand-over to the per-device worker and back, which is most of
            # what an empty node costs; a single-GPU training loop can run that way, nn.DataParallel cannot)
            with torch.autograd.set_multithreading_enabled(False):
                return wall(fn, n)

        # raw C-ABI, back to back on the current stream
        L = _lib.lib()
        loss = torch.empty((), dtype=torch.float32, device=device)
        pos = torch.empty(_lib.inbatch_save_floats(_lib.MODEL_IDS[c["model"]], B, K, D), dtype=torch.float32, device=device)
        ticket = torch.zeros(_lib.INBATCH_TICKET_INTS, dtype=torch.int32, device=device)  # zero on entry, left zero
        neg = torch.empty((B, K), dtype=torch.float32, device=device)
        g_ent, g_rel, one = torch.empty_like(ent), torch.empty(B, D, device=device), torch.ones((), device=device)
        rel2 = rel.reshape(B, D).contiguous()
        args = (_lib.MODEL_IDS[c["model"]], _lib.LOSS_IDS[c["loss"]], _lib.DTYPE_NAMES.index(c["dtype"]), 0)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def raw():
            _lib.check(L.blp_inbatch_loss_fwd(*args, ent.data_ptr(), rel2.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                                loss.data_ptr(), pos.data_ptr(), neg.data_ptr(), ticket.data_ptr(), device.index, stream), "fwd")
            _lib.check(L.blp_inbatch_loss_bwd(*args, ent.data_ptr(), rel2.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                                one.data_ptr(), pos.data_ptr(), neg.data_ptr(), g_ent.data_ptr(), g_rel.data_ptr(),
                                                device.index, stream), "bwd")

        for _ in range(20):
            raw()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            raw()
        b.record()
        torch.cuda.synchronize()
        spread = wall_spread(fused, iters)
        out[name] = {"rel_model": c["model"], "loss": c["loss"], "B": B, "K": K, "D": D, "storage": c["dtype"],
                     "launches_per_step": 1 + int(L.blp_inbatch_loss_fwd_launches(args[0], B, K, D, c["reg"])),
                     "us_per_step_kernels": a.elapsed_time(b) / iters * 1e3,
                     "autograd_plumbing": "C++ torch::autograd::Function (blp_amd/_torch_glue.so)" if ops.torch_glue() is not None
                                          else "Python autograd.Function + ctypes",
                     # median of 5 settle-then-time rounds (the spread beside it)
                     "us_per_step_autograd": spread["median"], "us_per_step_autograd_best": spread["best"],
                     "us_per_step_autograd_p90": spread["p90"], "us_per_step_autograd_rounds": spread["rounds"],
                     "us_forward_no_grad": wall(forward_only, iters),
                     "us_per_step_autograd_first_steps_of_a_burst": cold(fused),
                     "us_autograd_floor_no_kernels": wall(floor, iters) if ops.torch_glue() is not None else None,
                     "us_per_step_autograd_in_graph": wall(fused_in_graph, iters),
                     "us_autograd_floor_in_graph": wall(floor_in_graph, iters) if ops.torch_glue() is not None else None,
                     "us_per_step_autograd_engine_single_threaded": wall_single_thread(fused, iters),
                     "us_autograd_floor_engine_single_threaded": wall_single_thread(floor, iters) if ops.torch_glue() is not None else None,
                     "torch_us_per_step": wall(stock, max(20, iters // 4)),
                     "pairs_per_step": B * (K + 1)}
        o = out[name]  # what OUR node adds to a backward(): the step minus the same node without kernels, measured back to back
        if o["us_autograd_floor_no_kernels"] is not None:
            o["us_node_cost"] = o["us_per_step_autograd"] - o["us_autograd_floor_no_kernels"]
            o["us_node_cost_in_graph"] = o["us_per_step_autograd_in_graph"] - o["us_autograd_floor_in_graph"]
        if name == "inbatch-fb15k237":  # the reference's training wrapper (train.py:329-330,344): nn.DataParallel, here two replicas on this device
            out[name]["dataparallel_two_replicas"] = dataparallel_step_us(device, c, wall)
    return out


def dataparallel_step_us(device, c, wall):
    """One training step of TransductiveLinkPrediction under nn.DataParallel with two replicas on ONE device (scatter, replicate,
    one Python thread per replica calling the C-ABI concurrently, gather, mean, backward through both replicas): the fused loss
    against the same module with the reference's expressions (oracle/ref_port.py) through stock PyTorch-ROCm."""
    from blp_amd import models
    from oracle import ref_port

    class StockLoss(models.TransductiveLinkPrediction):
        def compute_loss(self, ent_embs, rels, neg_idx):
            return ref_port.compute_loss(self.rel_model, c["loss"], ent_embs, self.rel_emb(rels), neg_idx, self.regularizer)

    g = torch.Generator(device=device).manual_seed(9)
    B, K, D, E, R = c["B"], c["K"], c["D"], 14541, 237
    pairs = torch.randint(0, E, (2 * B, 2), device=device, generator=g)
    rels = torch.randint(0, R, (2 * B, 1), device=device, generator=g)
    negs = torch.cat([torch.randint(0, 2 * B, (B, K, 2), device=device, generator=g) for _ in range(2)])  # device-local indices (data.py:289-298)
    out = {"replicas": 2, "triples_per_replica": B}
    import warnings
    for key, cls in (("fused_us_per_step", models.TransductiveLinkPrediction), ("stock_us_per_step", StockLoss)):
        net = cls(D, c["model"], c["loss"], E, R, c["reg"]).to(device)
        dp = torch.nn.DataParallel(net, device_ids=[device.index, device.index])

        def step():
            net.zero_grad(set_to_none=True)
            dp(pairs, rels, negs).mean().backward()

        def alone():
            net.zero_grad(set_to_none=True)
            net(pairs[:B], rels[:B], negs[:B]).backward()

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # ("gather along dimension 0, but all input tensors were scalars")
            out[key] = wall(step, 50)
        out[key.replace("us_per_step", "one_replica_alone_us")] = wall(alone, 100)
    return out


def relaunch_with_ranks(n_gpus, timeout_s=None):
    """`python bench.py --gpus N` started without a launcher: start the N ranks ourselves (one process per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1 and a free port), hand rank 0's single JSON line through on stdout and
    return the launcher's exit status -- non-zero if any rank died (the launcher names the failed ranks on stderr, which is
    passed through).  The run is bounded: after BLP_BENCH_TIMEOUT_S seconds (default 900; a hung collective would otherwise
    hold the box until the driver's own limit) the launcher's process group is killed, status 124."""
    import signal
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")       # (the launcher would set 1 and say so on stderr; the CPU side is tiny either way)
    # dmabuf IPC.  Not measured by this package (no RCCL run of it has had two GPUs): the build environment's own note -- "the
    # host driver only supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing across processes fails with
    # hipIpcGetMemHandle: invalid argument"; it is exported on the GPU boxes already, this only keeps it when the caller's
    # environment was scrubbed.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    timeout_s = float(os.environ.get("BLP_BENCH_TIMEOUT_S", "900")) if timeout_s is None else timeout_s
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)  # (its own process group: the launcher AND its ranks can be stopped)
    try:
        return proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {n_gpus}-rank run did not finish within {timeout_s:.0f} s (BLP_BENCH_TIMEOUT_S): stopping it", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            proc.wait(timeout=20)
        except (subprocess.TimeoutExpired, ProcessLookupError):
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
        return 124


def sub_result(name, axis, device, world, rank, backend, steps, events):
    """One sub-result: the workload prepared, measured (the better of two timed runs on one GPU) and torn down."""
    sub = Job(name, device, world, rank, axis, backend)
    k_sub = max(2, min(steps, 5))
    # best of two timed runs of K steps: a one-off stall (a code object loaded on first use, the allocator growing)
    # inside a 5-step run once made a 1.2 ms evaluation read as 16 ms
    f, t, c = sub.measure(k_sub, 3 if world == 1 else 1, events)
    runs = 1
    if world == 1 and not sub.heavy:
        f2, t2, c2 = sub.measure(k_sub, 1, events)
        runs = 2
        if f2["ms_per_step"] < f["ms_per_step"]:
            f, t, c = f2, t2, c2
    f["steps"], f["timed_runs"], f["unit"] = f.pop("timed_steps"), runs, "scored triples/s"
    if rank == 0:
        f["parity_check"] = parity_spot_check(sub, t, c)
    if sub.cfg["block"] < sub.T:  # reference batching: a step is several table passes
        passes = -(-sub.T // sub.cfg["block"])
        f["table_passes_per_step"], f["ms_per_table_pass"] = passes, f["ms_per_step"] / passes
    if "loop_batch" in sub.cfg and world == 1:
        f["reference_loop_layout"] = sub.loop_layout()
    if world == 1 and (sub.cfg.get("clusters") or name in ("fb15k237-distmult", "fb15k237-transe")):
        stats = sub.prepass_stats()
        if stats:
            f["prepass"] = stats
            if stats["decided_frac"] is not None:
                f["decided_frac"] = stats["decided_frac"]
    del sub, t, c
    from blp_amd import ops
    ops.release_workspaces()
    torch.cuda.empty_cache()
    return f


def call_overhead(device, iters=3000):
    """Host time per call of the Python layer around the library, on a problem so small (8 queries x 640 rows) that the GPU
    keeps up: `ops_rank_all_us` = one ops.rank_all call issued (argument checks, workspace, marshalling AND the library's
    three launches); `library_call_us` = the same blp_rank_all call with its arguments prepared once (what the launch chain
    itself costs the host); their difference is the Python wrapper.  And one 128-query call (the reference's eval batch
    against the FB15k-237 table): wall time per call, completed."""
    from blp_amd import _lib, ops
    g = torch.Generator(device=device).manual_seed(11)
    table = torch.nn.functional.normalize(torch.randn(640, 128, device=device, generator=g), dim=-1)
    qf, qr = torch.randn(8, 128, device=device, generator=g), torch.randn(8, 128, device=device, generator=g)
    true = torch.zeros(8, dtype=torch.int64, device=device)
    out = torch.empty((8, 4), dtype=torch.int32, device=device)

    def host_us(fn, n, burst=32):
        """(host us per call ISSUED -- bursts of `burst` calls into an idle queue, so that the host never waits for the
        device --, wall us per call completed back to back)"""
        for _ in range(200):
            fn()
        issued = 0.0
        for _ in range(max(1, n // burst)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(burst):
                fn()
            issued += time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return issued / (max(1, n // burst) * burst) * 1e6, (time.perf_counter() - t0) / n * 1e6

    wrapped, wrapped_done = host_us(lambda: ops.rank_all("transe", table, qf, qr, 4, true_row=true, out=out), iters)
    L = _lib.lib()
    ws_bytes = L.blp_rank_all_workspace_bytes(0, 640, 128, 4, 4)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    stream = torch._C._cuda_getCurrentRawStream(device.index)
    args = (0, table.data_ptr(), 640, 128, 128, qf.data_ptr(), qr.data_ptr(), true.data_ptr(), None, 4, 4, None,
            out.data_ptr(), ws.data_ptr(), ws_bytes, device.index, stream)
    raw, _ = host_us(lambda: L.blp_rank_all(*args), iters)
    cfg = WORKLOADS["fb15k237-transe"]
    big, rel_w, heads, tails, rels = make_data(cfg, device)
    q_fixed, q_rel, true_row = build_queries(big, rel_w, heads[:64], tails[:64], rels[:64])
    _, call128 = host_us(lambda: ops.rank_all("transe", big, q_fixed, q_rel, 64, true_row=true_row), 500)
    return {"ops_rank_all_us": wrapped, "library_call_us": raw, "python_wrapper_us": wrapped - raw,
            "ops_rank_all_completed_us": wrapped_done, "launches_per_call": 3, "rank_all_128_queries_fb15k237_us": call128}


def table_build_bench(device):
    """The step BEFORE the path: the entity-table build's fused pieces against the stock PyTorch-ROCm
    modules, per emb_batch_size chunk -- the bag-of-words encoder's whole build (models.py:143-155 + F.normalize + the row
    assignment: blp_bow_rows) at the BERT word-embedding table's size, the DKRL encoder's whole build (blp_dkrl_rows), and the BERT encoders' last step (enc_linear +
    F.normalize + row assignment, models.py:110-111: blp_project_rows).  Floating point (tolerance in the tests); us per chunk."""
    from blp_amd import models, ops
    g = torch.Generator(device=device).manual_seed(21)
    out = {}

    def us(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    V, E = 28996, 768  # bert-base-cased word embeddings
    model = models.BOW("transe", "margin", 5, 0, embeddings=torch.randn(V, E) * 0.1).to(device)
    for name, n, L in (("bert-bow, 512 entities x 32 tokens (scripts/bert-bow-fb15k237.sh)", 512, 32),
                       ("bert-bow, 12 288 entities x 64 tokens (scripts/bert-bow-wikidata5m.sh)", 12288, 64)):
        tok = torch.randint(1, V, (n, L), device=device, generator=g)
        lengths = torch.randint(L // 2, L + 1, (n, 1), device=device, generator=g)
        mask = (torch.arange(L, device=device).unsqueeze(0) < lengths).float()
        rows = torch.empty(n, E, device=device)
        with torch.no_grad():
            stock = us(lambda: rows.copy_(model.encode(tok, mask)))
            fused = us(lambda: model.encode_into(rows, tok, mask, defer_check=True))
        model.check_tokens()
        out[name] = {"stock_us": stock, "fused_us": fused, "gathered_GBps": n * L * E * 4 / fused / 1e3, "kernel": "blp_bow_rows"}
    # the DKRL encoder's whole build (models.py:158-204 + F.normalize + the row assignment: blp_dkrl_rows; conv1 on the
    # matrix cores with f32 operands: `conv_TFLOPs` = its arithmetic against the 157.3 TF f32 MFMA peak)
    dkrl = models.DKRL(128, "transe", "margin", 5, 0, embeddings=model.embeddings.weight.detach().cpu()).to(device)
    for name, n, L in (("bert-dkrl, 512 entities x 32 tokens (scripts/bert-dkrl-fb15k237.sh)", 512, 32),
                       ("bert-dkrl, 12 288 entities x 64 tokens (scripts/bert-dkrl-wikidata5m.sh)", 12288, 64)):
        tok = torch.randint(1, V, (n, L), device=device, generator=g)
        lengths = torch.randint(L // 2, L + 1, (n, 1), device=device, generator=g)
        lengths[0] = L
        mask = (torch.arange(L, device=device).unsqueeze(0) < lengths).float()
        rows = torch.empty(n, 128, device=device)
        with torch.no_grad():
            stock = us(lambda: rows.copy_(dkrl.encode(tok, mask)), 10)
            fused = us(lambda: dkrl.encode_into(rows, tok, mask, defer_check=True), 10)
        dkrl.check_tokens()
        flops = 2.0 * n * L * 2 * E * 128
        out[name] = {"stock_us": stock, "fused_us": fused, "conv_TFLOPs": flops / fused / 1e6, "frac_of_f32_mfma_peak": flops / fused / 1e6 / F32_PEAK_TFLOPS,
                     "kernel": "blp_dkrl_rows"}
    del dkrl
    x = torch.randn(14541, 768, device=device, generator=g)
    w = torch.randn(128, 768, device=device, generator=g) * 0.03
    rows = torch.empty(14541, 128, device=device)
    with torch.no_grad():
        stock = us(lambda: rows.copy_(torch.nn.functional.normalize(torch.nn.functional.linear(x, w), dim=-1)))
        fused = us(lambda: ops.project_rows(x, w, rows, True))
    out["bert [CLS] rows 14 541 x 768 -> 128, normalised"] = {"stock_us": stock, "fused_us": fused, "kernel": "blp_project_rows"}
    return out


LINE_LIMIT = 6000               # bytes of the final stdout line (the driver's reader lost round 4's 26 KB line)
DETAILS_FILE = "bench_details.json"


def _sig(x, digits=6):
    """Floats to `digits` significant figures (the compact line); everything else as it is."""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _short(text, n):
    return text if text is None or len(text) <= n else text[:n - 1] + "\u2026"


def compact_roofline(roof):
    """The roofline object of the contract (bound / achieved / peak / unit / frac / traffic) + the kernel's name and time;
    the prose (`note`, `peak_source`, the long `arith`) stays in the details file."""
    if not roof:
        return roof
    out = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "passes_per_launch",
                                     "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch")}
    if "peak_measured" in roof:  # the published peak is `peak`; a measured derate sits beside it, never instead of it
        out["peak_measured"], out["frac_measured"] = roof["peak_measured"], roof.get("frac_measured")
    out["kernel"] = _short(roof.get("kernel"), 72)
    if roof.get("mfma_busy") is not None:
        out["mfma_busy"] = roof["mfma_busy"]
    if "survey_8d_lane_ops" in roof:
        out["survey_8d_lane_ops_frac"] = roof["survey_8d_lane_ops"]["frac"]
    out["pmc_source"] = _short(roof.get("pmc_source"), 96)
    return out


def compact_sub(sub, world):
    roof = sub.get("roofline") or {}
    out = {"value": sub.get("value"), "ms_per_step": sub.get("ms_per_step"), "frac": roof.get("frac"), "bound": roof.get("bound"),
           "kernel_ms": roof.get("kernel_ms")}
    if "decided_frac" in sub:
        out["decided_frac"] = sub["decided_frac"]
    if "vs_random_step" in sub:
        out["vs_random_step"] = sub["vs_random_step"]
    if world > 1:  # (13 sub-results on several ranks: the per-rank kernel time stays in the details file)
        out.pop("kernel_ms")
        out["exchange_ms"], out["axis"] = sub.get("exchange_ms"), sub.get("shard_axis")
    parity = sub.get("parity_check")
    if parity and not parity.startswith("skipped"):
        out["parity"] = "ok" if parity.endswith("identical counts") else "MISMATCH"
    return out


def compact_result(result, limit=LINE_LIMIT):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline`, `parity_check` and a few numbers per sub-result,
    under `limit` bytes for every N.  Everything else (per-rank lists, prose notes, the baselines' legs, table-build and
    call-overhead figures) is in DETAILS_FILE, which this line names.  If the line would still be too long, optional
    sections go, least important first; the contract's keys never do."""
    world = result.get("n_gpus", 1)
    line = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data") if k in result}
    line["arith"] = _short(result.get("arith"), 120)
    cfg = dict(result.get("config", {}))
    cfg.pop("step", None)
    line["config"] = cfg
    for k in ("ms_per_step_raw_only", "mrr", "mrr_filtered", "hits@1,3,10", "hits@1,3,10_filtered", "exchange_ms", "parity_check", "decided_frac",
              "filter_index_build_ms", "vs_1gpu", "kernel_ms_per_rank", "exchange_ms_per_rank"):
        if k in result:
            line[k] = result[k]
    line["roofline"] = compact_roofline(result.get("roofline"))
    cpu = result.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": _short(cpu["kind"], 40),
                                "sample": _short(cpu["sample"], 110), "cpu_model": cpu.get("cpu_model"),
                                "logical_cpus": cpu.get("logical_cpus"), "c_oracle_value": (cpu.get("c_oracle") or {}).get("value")}
    tg = result.get("torch_gpu_baseline")
    if tg:
        line["torch_gpu_baseline_value"] = tg["value"]
    if "sub_results" in result:
        line["sub_results"] = {name: compact_sub(sub, world) for name, sub in result["sub_results"].items()}
    if "inbatch_loss" in result:
        line["inbatch_loss"] = {name: {k: v.get(k) for k in ("launches_per_step", "us_per_step_kernels", "us_per_step_autograd", "us_per_step_autograd_p90",
                                                             "us_node_cost", "torch_us_per_step")}
                                for name, v in result["inbatch_loss"].items()}
    if "hbm_probe" in result:
        line["hbm_probe"] = {k: result["hbm_probe"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms", "traffic")}
    if "call_overhead" in result:
        line["call_overhead"] = {k: result["call_overhead"].get(k) for k in ("library_call_us", "rank_all_128_queries_fb15k237_us")}
    line["details"] = result.get("details")
    line = _sig(line)
    line["value"], line["ms_per_step"] = result["value"], result["ms_per_step"]  # (full precision: value == units / time, to the bit)
    for drop in ("call_overhead", "torch_gpu_baseline_value", "inbatch_loss", "hbm_probe", "hits@1,3,10", "hits@1,3,10_filtered"):
        if len(json.dumps(line)) <= limit:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > limit and "sub_results" in line:  # (many sharded sub-results: two numbers each)
        line["sub_results"] = {k: {"value": v["value"], "frac": v["frac"]} for k, v in line["sub_results"].items()}
    if len(json.dumps(line)) > limit:
        line.pop("sub_results", None)
    return line


def write_details(result, path):
    """The full object (what round 4 printed as one 26 KB line), indented, next to bench.py -- or where --details says."""
    try:
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
            f.write("\n")
        return path
    except OSError as exc:  # (a read-only tree: the compact line must still be printed)
        print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
        return None


def dry_run_collectives(device, world, rank, backend, result_fd, nbytes=1024, timeout_s=120):
    """`--dry-nccl`: the process group of an N > 1 run and the three collectives ONE evaluation issues (ranking.exchange_plan: the
    all-gather of table rows, the all-reduce of the queries' vectors, the all-gather of the int32 counts), 1 KB each, values
    checked.  Every rank reports on stderr; rank 0 prints one JSON line {"dry_nccl": "ok", "ranks": N, ...} on stdout.  A rank
    that fails names itself and the step ("rank 3: all_reduce: <error>") and the process exits with status 3 -- before the
    driver spends a scaling run on a box whose RCCL cannot come up."""
    import datetime
    from blp_amd import ranking
    step = "init_process_group"
    try:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = dict(timeout=datetime.timedelta(seconds=timeout_s))
            if backend == "gloo":
                dist.init_process_group("gloo", **kw)
            else:
                dist.init_process_group("nccl", device_id=device, **kw)
        group_size = dist.get_world_size() if world > 1 else 1
        n = nbytes // 4
        step = "all_gather (table rows)"
        rows = torch.full((n // 8, 8), float(rank), device=device)
        full = ranking.all_gather_rows(rows, world * (n // 8), world)
        assert full[:, 0].reshape(world, -1).mean(dim=1).tolist() == [float(r) for r in range(world)], "wrong values"
        step = "all_reduce (query vectors)"
        vec = torch.full((n,), float(rank + 1), device=device)
        if world > 1:
            ranking._all_reduce(vec)
        assert vec[0].item() == (world * (world + 1) / 2 if world > 1 else 1.0), "wrong sum"
        step = "all_gather (int32 counts)"
        part = torch.full((n,), rank, dtype=torch.int32, device=device)
        gathered = torch.empty(world * n, dtype=torch.int32, device=device)
        if world > 1:
            ranking._all_gather_into(gathered, part)
        else:
            gathered.copy_(part)
        assert gathered.reshape(world, n)[:, 0].tolist() == list(range(world)), "wrong values"
        if device.type == "cuda":
            torch.cuda.synchronize()
        print(f"bench.py --dry-nccl: rank {rank} of {group_size} on {device}: ok", file=sys.stderr, flush=True)
        if world > 1:
            dist.barrier()
        if rank == 0:
            os.write(result_fd, (json.dumps({"dry_nccl": "ok", "ranks": group_size, "backend": backend if world > 1 else None,
                                             "collectives": ["all_gather", "all_reduce", "all_gather"], "bytes_each": nbytes}) + "\n").encode())
        if world > 1:
            dist.destroy_process_group()
        return 0
    except BaseException as exc:  # noqa: BLE001 -- named, then a status the launcher reports
        print(f"bench.py --dry-nccl: rank {rank}: {step}: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
        return 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="the top-level workload (default: headline_workload(--gpus): fb15k237-transe on one GPU, wikidata5m-transe -- "
                         "candidate axis -- on several)")
    ap.add_argument("--dry-nccl", action="store_true", help="pre-flight of an N > 1 run: bring the process group up, run the three "
                    "collectives of an evaluation on 1 KB each, print one JSON line and exit (a failing rank is named, status 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true")
    ap.add_argument("--no-sub-results", action="store_true")
    ap.add_argument("--details", default=os.path.join(ROOT, DETAILS_FILE),
                    help="where rank 0 writes the FULL result object (the stdout line is its compact form, < 6 KB)")
    ap.add_argument("--shard-axis", default="auto", choices=["auto", "candidate", "query"],
                    help="N > 1: which axis to shard (default: blp_amd.ranking.choose_shard_axis; the Wikidata5M-scale "
                         "table always along the candidate axis)")
    ap.add_argument("--plan", action="store_true", help="print what a run on --gpus N ranks would do (axes, shards, the bytes of "
                    "every collective, steps of the long workloads) as JSON and exit; needs no GPU")
    args = ap.parse_args()
    default_line = args.workload is None
    if default_line:
        args.workload = headline_workload(args.gpus)

    if args.plan:
        print(json.dumps(run_plan(args.gpus, args.shard_axis), indent=1))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_with_ranks(args.gpus))  # no launcher around us: be our own
    # stdout carries ONE line, the JSON result, and nothing else: from here on file descriptor 1 is stderr -- RCCL prints a version
    # banner through C stdio when its first communicator comes up (nn.DataParallel's replicate on one GPU is enough; every rank
    # of an N > 1 run), which a pipe delivers at process exit, i.e. AFTER a line printed from Python -- and the result goes to
    # the saved descriptor at the very end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.dry_nccl and not torch.cuda.is_available() and os.environ.get("BLP_BENCH_BACKEND") == "gloo":
        # (the pre-flight's own plumbing can be exercised without a GPU: gloo, host tensors -- tests/test_bench_host.py)
        raise SystemExit(dry_run_collectives(torch.device("cpu"), world, rank, "gloo", result_fd))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the measured path)")
    # BLP_BENCH_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than ranks
    # (ranks share devices, collectives go through host memory).  Timings are then meaningless.
    backend = os.environ.get("BLP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {args.gpus}: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if args.dry_nccl:
        raise SystemExit(dry_run_collectives(device, world, rank, backend, result_fd))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        ranks_seen = dist.get_world_size()

    events = HipEvents()
    job = Job(args.workload, device, world, rank, args.shard_axis, backend)
    fields, triples, counts = job.measure(args.steps, args.warmup, events)
    timed_steps = fields.pop("timed_steps")  # (== --steps, except for the whole-evaluation workloads: two steps of > 1 s)
    cfg = job.cfg
    result = None
    if rank == 0:
        result = {
            "metric": "scored triples/sec, all-entity eval (raw + filtered ranks, MRR + Hits@k)",
            "value": fields.pop("value"),
            "unit": "scored triples/s",
            "n_gpus": world,
            "steps": timed_steps,
            "warmup": args.warmup,
            "ms_per_step": fields.pop("ms_per_step"),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "arith": fields["roofline"]["arith"],
            "data": "synthetic (seeded randn table, Xavier rel_emb, uniform random test triples, Zipf(0.8) filtering graph)",
            "config": {"workload": args.workload, "rel_model": cfg["model"], "entities": cfg["N"], "dim": cfg["D"],
                       "queries_per_step": 2 * cfg["triples"], "triples_per_ranking_call": min(cfg["triples"], cfg["block"]),
                       "filter_graph_edges": job.index.num_edges,
                       "step": "whole evaluation: id lookups + query gathers + filter segments + raw and "
                               "filtered ranking + metric sums",
                       "parallelism": f"{job.axis}-axis shards x{world}" if world > 1 else "single GPU",
                       "shard_axis": job.axis, "ranks_in_process_group": ranks_seen, "backend": backend if world > 1 else None},
        }
        result.update(fields)
        if world > 1:  # against the committed one-GPU run of the SAME workload (the driver computes its own efficiency from its N = 1 line)
            ref = n1_reference(args.workload)
            result["vs_1gpu"] = result["value"] / ref["value"] if ref else None
            result["one_gpu_reference"] = ref
        result["parity_check"] = parity_spot_check(job, triples, counts)
        if world == 1:
            result["prepass"] = job.prepass_stats()
            if result["prepass"] and result["prepass"]["decided_frac"] is not None:
                result["decided_frac"] = result["prepass"]["decided_frac"]
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(job)
            result["torch_gpu_baseline"] = torch_gpu_baseline(job)
    del triples, counts, job
    torch.cuda.empty_cache()
    if not args.no_sub_results and default_line:
        # The same names for every N (SCALE's N = 1 line against BENCH, field by field).  N > 1: every rank takes part in
        # every sub-result -- each step has its exchanges (replicated query vectors, one all-gather of counts).
        subs = {}
        todo = [(name, "auto", name) for name in SUB_RESULTS]
        if world > 1:  # (the one-GPU headline becomes a sub-result: on the axis a sharded evaluation of its shape takes, and on the candidate axis)
            todo = [("fb15k237-transe", "auto", "fb15k237-transe")] + todo
            todo += [(name, axis, f"{name}@{axis}") for name, axis in SUB_RESULTS_EXTRA_SHARDED]
        for name, axis, key in todo:
            f = sub_result(name, axis, device, world, rank, backend, args.steps, events)
            if world == 1:
                for k in ("kernel_ms_per_rank", "exchange_ms", "exchange_ms_per_rank", "shard_axis", "ranks"):
                    f.pop(k)
            subs[key] = f
        if rank == 0:  # the clustered workloads against the same shapes on i.i.d. random data
            for key, f in subs.items():
                twin = WORKLOADS.get(key, {}).get("random_twin")
                base = result["ms_per_step"] if twin == args.workload else subs.get(twin, {}).get("ms_per_step")
                if twin and base:
                    f["vs_random_step"] = f["ms_per_step"] / base
        # the training-side scoring is replicas only: every rank runs the same step, rank 0 reports its own
        inbatch = inbatch_bench(device)
        if rank == 0:
            result["sub_results"], result["inbatch_loss"] = subs, inbatch
            result["call_overhead"] = call_overhead(device)
            result["table_build"] = table_build_bench(device)
    if not args.no_hbm_probe and (default_line or not args.workload.startswith("wikidata5m")):
        if rank == 0:  # one GPU's HBM-bound operating point (the other ranks wait at the barrier below)
            result["hbm_probe"] = hbm_probe(device, events)
    if rank == 0:
        written = write_details(result, args.details)
        result["details"] = os.path.relpath(written, ROOT) if written else None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(compact_result(result)) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
