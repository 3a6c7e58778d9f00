"""GPU parity at the sizes that are TIMED: the exact tensors of bench.py's FB15k-237 workloads (105 740 queries x
14 541 candidates, the 310 116-edge filtering graph of SURVEY.md 8d config 2).  The pre-pass paths' behaviour depends
on the block size (queries per workgroup, grid shape, pair-list compaction, candidate slabs), so the configuration
that is measured is the one compared here:
  (i)   pre-pass path == the exact f32 kernels on ALL queries (raw and filtered counts);
  (ii)  bilinear models: bf16 x 3 pre-pass == the f32-chain MFMA pre-pass (a provable fma chain) on all queries;
  (iii) == the CPU oracle on 4 096 queries spread over the block (both sides, every 25th query or so);
  (iv)  blp_rank_metric_sums (what bench.py reports MRR / Hits from) == the oracle's metrics on those counts.
Integer counts: bit-exact.  MRR: f64 sums of identical f32 reciprocal ranks, compared to 1e-12 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODELS = ("transe", "distmult", "complex", "simple")


@pytest.fixture(scope="module")
def bench():
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module("bench")


def _block(bench, model):
    """The bench workload's tensors and ONE evaluation of it through the product path (ranking.rank_triples)."""
    job = bench.Job(f"fb15k237-{model}", torch.device("cuda", 0))
    triples, counts, sums = job.step(filtered=True)
    torch.cuda.synchronize()
    return job, triples, counts, sums


def _direct(job, triples, **knobs):
    """The same block through ops.rank_all with test knobs set (kernel selection), filtered."""
    from blp_amd import _lib, ops
    T = triples.shape[0]
    h, t, r = triples[:, 0], triples[:, 1], triples[:, 2]
    table = job.table
    rel = job.model.rel_emb.weight.detach()[r]
    seg = job.index.segments(triples, job.ent2idx, table.device)
    try:
        for k, v in knobs.items():
            _lib.set_knob(k, v)
        out = ops.rank_all(job.cfg["model"], table, torch.cat((table[t], table[h])), torch.cat((rel, rel)), T,
                           true_row=torch.cat((h, t)), filter=seg)
        torch.cuda.synchronize()
    finally:
        _lib.reset_knobs()
    return out


@pytest.mark.parametrize("model", MODELS)
def test_timed_block_equals_exact_kernels_and_oracle(bench, oracle, model):
    from blp_amd import ops
    job, triples, counts, sums = _block(bench, model)
    T, N = job.T, job.N
    assert counts.shape == (2 * T, 4)
    # (i) every query: the pre-pass path against the exact f32 kernels
    exact = _direct(job, triples, rank_kernel=1)
    assert torch.equal(counts, exact), f"{int((counts != exact).any(dim=1).sum())} queries differ from the exact kernels"
    # (ii) bilinear: the bf16 x 3 pre-pass against the f32 fma-chain pre-pass (its band is provable)
    if model != "transe":
        chain = _direct(job, triples, gemm_kernel=1)
        assert torch.equal(counts, chain)
    # filtered counts never exceed the raw ones, and a filtered candidate was counted raw
    c = counts.cpu().numpy()
    assert (c[:, 2] <= c[:, 0]).all() and (c[:, 3] <= c[:, 1]).all() and (c[:, 1] > c[:, 0]).all()
    assert (c[:, :2] != c[:, 2:]).any(), "the filter of this workload removes nothing?"
    # (iii) the CPU oracle on 2 048 triples spread over the block = 4 096 queries, both sides, with their filters
    pick = torch.arange(0, T, max(T // 2048, 1))[:2048]
    sub = triples.cpu()[pick]
    tab = job.full_table.cpu().numpy()
    rel = job.model.rel_emb.weight.detach().cpu()[sub[:, 2]].numpy()
    rowptr, col = job.index.csr(sub, torch.arange(N))
    b = sub.shape[0]
    want_h = oracle.rank_counts(model, oracle.SIDE_HEAD, tab, tab[sub[:, 1].numpy()], rel, true_row=sub[:, 0].numpy(),
                                filt_rowptr=rowptr[:b + 1].numpy(), filt_col=col[:rowptr[b]].numpy())
    want_t = oracle.rank_counts(model, oracle.SIDE_TAIL, tab, tab[sub[:, 0].numpy()], rel, true_row=sub[:, 1].numpy(),
                                filt_rowptr=(rowptr[b:] - rowptr[b]).numpy(), filt_col=col[rowptr[b]:].numpy())
    assert np.array_equal(c[pick.numpy()], want_h)
    assert np.array_equal(c[T + pick.numpy()], want_t)
    # (iv) the device-side metric sums against the oracle's get_metrics on all 105 740 counts
    got = sums.cpu().numpy()
    for v, (a, bcol) in enumerate(((0, 1), (2, 3))):
        rr, hits = oracle.metrics_from_counts(c[:, a], c[:, bcol])
        assert got[v] == pytest.approx(rr.astype(np.float64).sum(), rel=1e-12)
        assert got[2 + 3 * v: 5 + 3 * v].tolist() == hits.sum(axis=0).astype(np.float64).tolist()
    mrr = got[0] / (2 * T)
    assert abs(mrr - oracle.metrics_from_counts(c[:, 0], c[:, 1])[0].astype(np.float64).mean()) < 1e-9
    # and the rr / hits kernel agrees with the sums kernel
    rr_dev, hits_dev = ops.rank_metrics(counts)
    assert got[:2].tolist() == pytest.approx(rr_dev.double().sum(dim=0).tolist(), rel=1e-12)


@pytest.mark.parametrize("model", ["transe", "distmult"])
def test_clustered_block_equals_exact_kernels_and_oracle(bench, oracle, model):
    """Away from i.i.d. random tables (bench workloads fb15k237-*-clustered): 500 clusters of EXACT duplicate rows (entities
    with one description) and a trained model's triples (true tail among the 145 best-scoring entities) -- every query has
    ~29 candidates that tie with its true entity, the bilinear pre-pass's workgroups run out of list quota and flag
    half-segments by the hundred thousand, which flags_to_entries_kernel turns into entries of the pair pass.  All 105 740
    queries against the exact f32 kernels, 4 096 against the CPU oracle; what the pre-pass left undecided is reported."""
    from blp_amd import ops
    job = bench.Job(f"fb15k237-{model}-clustered", torch.device("cuda", 0))
    triples, counts, sums = job.step(filtered=True)
    torch.cuda.synchronize()
    T, N = job.T, job.N
    exact = _direct(job, triples, rank_kernel=1)
    assert torch.equal(counts, exact), f"{int((counts != exact).any(dim=1).sum())} queries differ from the exact kernels"
    c = counts.cpu().numpy()
    assert np.median(c[:, 1] - c[:, 0]) >= 20  # ties with the true entity: ge - gt = the size of its cluster (about 29)
    pick = torch.arange(0, T, max(T // 2048, 1))[:2048]
    sub = triples.cpu()[pick]
    tab = job.full_table.cpu().numpy()
    rel = job.model.rel_emb.weight.detach().cpu()[sub[:, 2]].numpy()
    rowptr, col = job.index.csr(sub, torch.arange(N))
    b = sub.shape[0]
    want_h = oracle.rank_counts(model, oracle.SIDE_HEAD, tab, tab[sub[:, 1].numpy()], rel, true_row=sub[:, 0].numpy(),
                                filt_rowptr=rowptr[:b + 1].numpy(), filt_col=col[:rowptr[b]].numpy())
    want_t = oracle.rank_counts(model, oracle.SIDE_TAIL, tab, tab[sub[:, 0].numpy()], rel, true_row=sub[:, 1].numpy(),
                                filt_rowptr=(rowptr[b:] - rowptr[b]).numpy(), filt_col=col[rowptr[b]:].numpy())
    assert np.array_equal(c[pick.numpy()], want_h)
    assert np.array_equal(c[T + pick.numpy()], want_t)
    st = job.prepass_stats()
    print(f"{model} clustered: decided {st['decided_frac']:.5f}, listed {st['listed']:,}, flagged rows {st['flagged_rows']:,} ({st['path']})")
    assert st["pairs"] == 2 * T * N and 0.98 < st["decided_frac"] < 0.9995  # (random data: 0.9988 / 0.9997)
    assert st["listed"] >= 2 * T * 20  # at least the ties
    if model == "distmult":  # the flags became entries: nothing is left for the one-wave-per-query sweep
        assert st["path"].startswith("bf16") and st["flagged_rows"] == 0


@pytest.mark.parametrize("model", ["transe", "distmult", "complex"])
def test_lists_run_full_and_the_exact_kernel_takes_over(bench, oracle, model):
    """20 clusters of exact duplicates: 5 % of the table ties with every query's true entity.  The pre-pass's lists run full, a
    device-side counter says so, the refinement kernels stand down and the exact kernel re-ranks the block (rank_common.h: Gate)
    -- a bounded worst case (pre-pass + exact kernel) where round 4 spent 130 / 70 ms re-scoring flagged tiles.  Counts equal the
    exact kernels' on all 105 740 queries and the CPU oracle's on 512; the step stays under 8 x the random-data step."""
    import time
    name = f"test-full-lists-{model}"
    bench.WORKLOADS[name] = dict(bench.WORKLOADS[f"fb15k237-{model}"], clusters=20, noise=0.0, top=145)  # (= bench's fb15k237-*-ties5pct)
    try:
        job = bench.Job(name, torch.device("cuda", 0))
        triples, counts, _ = job.step(filtered=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            job.step(filtered=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        exact = _direct(job, triples, rank_kernel=1)
        assert torch.equal(counts, exact), f"{int((counts != exact).any(dim=1).sum())} queries differ from the exact kernels"
        T, N = job.T, job.N
        pick = torch.arange(0, T, T // 256)[:256]
        sub = triples.cpu()[pick]
        tab = job.full_table.cpu().numpy()
        rel = job.model.rel_emb.weight.detach().cpu()[sub[:, 2]].numpy()
        rowptr, col = job.index.csr(sub, torch.arange(N))
        b = sub.shape[0]
        want_h = oracle.rank_counts(model, oracle.SIDE_HEAD, tab, tab[sub[:, 1].numpy()], rel, true_row=sub[:, 0].numpy(),
                                    filt_rowptr=rowptr[:b + 1].numpy(), filt_col=col[:rowptr[b]].numpy())
        want_t = oracle.rank_counts(model, oracle.SIDE_TAIL, tab, tab[sub[:, 0].numpy()], rel, true_row=sub[:, 1].numpy(),
                                    filt_rowptr=(rowptr[b:] - rowptr[b]).numpy(), filt_col=col[rowptr[b]:].numpy())
        c = counts.cpu().numpy()
        assert np.array_equal(c[pick.numpy()], want_h) and np.array_equal(c[T + pick.numpy()], want_t)
        assert np.median(c[:, 1] - c[:, 0]) >= 500  # ge - gt = the true entity's cluster (~ 727 rows)
        print(f"{model}, 20 clusters: {ms:.2f} ms per evaluation")
        assert ms < (30.0 if model == "transe" else 45.0)  # (measured 16 / 19 / 31 ms; without the fallback 132 / 68 / 71 ms)
    finally:
        del bench.WORKLOADS[name]


@pytest.mark.parametrize("model", ["distmult", "transe"])
def test_heavy_ties_overflow_every_list_and_stay_exact(bench, model):
    """100 clusters of exact duplicates: 145 rows tie with every query's true entity (1 % of the table) -- more than the
    workgroups' lists AND the spill region of the bilinear path hold, so flags survive the conversion and the one-wave-per-query
    sweeps run in earnest (TransE: the pair lists overflow into flagged tiles).  Slow (the cliff DESIGN 4.3 documents), exact:
    every one of the 105 740 queries against the exact f32 kernels."""
    name = f"test-heavy-ties-{model}"
    bench.WORKLOADS[name] = dict(bench.WORKLOADS[f"fb15k237-{model}"], clusters=100, noise=0.0, top=145)
    try:
        job = bench.Job(name, torch.device("cuda", 0))
        triples, counts, _ = job.step(filtered=True)
        torch.cuda.synchronize()
        exact = _direct(job, triples, rank_kernel=1)
        assert torch.equal(counts, exact), f"{int((counts != exact).any(dim=1).sum())} queries differ from the exact kernels"
        c = counts.cpu().numpy()
        assert np.median(c[:, 1] - c[:, 0]) >= 100  # ge - gt = the true entity's cluster
        st = job.prepass_stats()
        print(f"{model}, 100 clusters: decided {st['decided_frac']:.4f}, listed {st['listed']:,}, flagged rows {st['flagged_rows']:,}")
        assert st["decided_frac"] < 0.995 and st["listed"] + st["flagged_rows"] >= 2 * job.T * 100
        if model == "distmult":  # (TransE's per-workgroup lists are deeper: at 1 % ties they overflow only here and there)
            assert st["flagged_rows"] > 0
    finally:
        del bench.WORKLOADS[name]


def test_wikidata5m_block_prepass_equals_exact_kernel():
    """The Wikidata5M-scale block workload (13 788 queries x 4.6 M candidates, 30 candidate slabs of the fixed-point
    pre-pass, the split's own triples as the filtering graph): every count equals the exact f32 kernel's."""
    import bench
    job = bench.Job("wikidata5m-transe-block", torch.device("cuda", 0))
    triples, counts, _ = job.step(filtered=True)
    exact = _direct(job, triples, rank_kernel=1)
    assert torch.equal(counts, exact)


@pytest.fixture(scope="module")
def wikidata_table():
    g = torch.Generator().manual_seed(46)
    table = torch.randn(4_600_000, 128, generator=g)
    table[1_234_567] = table[7]          # an exact tie with a row some query may hold as its true entity
    table[4_599_999, 5] = float("inf")   # the last row of the last tile
    return table


@pytest.mark.parametrize("model", MODELS)
def test_wikidata5m_scale_passes_equal_the_oracle(oracle, wikidata_table, model):
    """The WHOLE 4.6 M x 128 table against the CPU oracle (until round 3 the oracle only ever saw 100 - 200 k-row slabs of
    it): the reference's evaluation batch (2 triples = 4 queries per table pass, scripts/blp-*-wikidata5m.sh:18), three
    triples -- a pass of two and a pass of one, both in one launch of the streaming kernel -- and the same queries as bare
    blp_rank_all calls (the workgroup-tile / ring kernels a single call takes); raw counts and a CSR filter."""
    from blp_amd import ops
    from test_gpu_parity import oracle_counts, random_csr
    N, D, T, batch = wikidata_table.shape[0], 128, 3, 2
    table = torch.nn.functional.normalize(wikidata_table, dim=-1) if model == "transe" else wikidata_table * 0.1
    g = torch.Generator().manual_seed(len(model))
    rel_w = (torch.rand(5, D, generator=g) - 0.5) * 0.25
    heads = torch.tensor([7, 3_000_000, 4_599_990])  # (7: ties with row 1 234 567)
    tails = torch.randint(0, N, (T,), generator=g)
    rels = torch.randint(0, 5, (T,), generator=g)
    # oracle: [all head-replacing queries | all tail-replacing queries]
    q_fixed = torch.cat((table[tails], table[heads]))
    q_rel = torch.cat((rel_w[rels], rel_w[rels]))
    true_row = torch.cat((heads, tails))
    rowptr, col = random_csr(2 * T, N, true_row.numpy(), seed=9)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, T, true_row=true_row, csr=(rowptr, col))
    dev_table, dev_rel = table.cuda(), rel_w.cuda()
    # (a) one bare call with all 3 + 3 queries
    got = ops.rank_all(model, dev_table, q_fixed.cuda(), q_rel.cuda(), T, true_row=true_row.cuda(),
                       filt_rowptr=torch.from_numpy(rowptr).cuda(), filt_col=torch.from_numpy(col).cuda()).cpu().numpy()
    assert np.array_equal(got, want)
    # (b) the reference loop's layout, a pass per batch of two triples: [h0 h1 | t0 t1] [h2 | t2]
    order = torch.tensor([0, 1, T + 0, T + 1, 2, T + 2])
    fixed_row = torch.cat((tails, heads))[order].cuda()
    raw = ops.rank_all_batches(model, dev_table, fixed_row, dev_rel, torch.cat((rels, rels))[order].cuda(), true_row[order].cuda(),
                               T, batch, block_triples=batch).cpu().numpy()
    assert np.array_equal(raw[:, :2], want[order.numpy(), :2])
    assert np.array_equal(raw[:, 2:], raw[:, :2])


def _dump_block(bench, model, data):
    """2 048 queries x all 14 541 candidates of the timed block through the bilinear pre-pass in dump mode: the
    kernel's own S~ and eps matrices, plus the inputs.  "scaled": every table element and every relation element
    scaled by its own power of two in 2^-8 .. 2^8 (exponent spreads inside a dot product, which the bench data lacks:
    a few products dominate the sum), plus two all-zero rows."""
    from blp_amd import _lib, ops
    job = bench.Job(f"fb15k237-{model}", torch.device("cuda", 0))
    table = job.table
    g = torch.Generator(device="cuda").manual_seed(17)
    if data == "scaled":
        table = table * torch.exp2(torch.randint(-8, 9, (job.N, job.D), device="cuda", generator=g).float())
        table[5] = 0.0
        table[4097] = 0.0
    pick = torch.arange(0, job.T, job.T // 1024, device="cuda")[:1024]
    t = job.triples[pick]
    rel = job.model.rel_emb.weight.detach()[t[:, 2]]
    fixed = torch.cat((table[t[:, 1]], table[t[:, 0]]))
    rel2 = torch.cat((rel, rel))
    if data == "scaled":
        rel2 = rel2 * torch.exp2(torch.randint(-8, 9, rel2.shape, device="cuda", generator=g).float())
    b, Q, N = t.shape[0], 2 * t.shape[0], job.N
    S = torch.full((Q, N), float("nan"), device="cuda")
    E = torch.full((Q, N), float("nan"), device="cuda")
    with _lib.use_hooks_library() as H:  # the dump hook exists in the hooks build only (same kernels, same band arithmetic)
        _lib.check(H.blp_debug_gemm_dump(S.data_ptr(), E.data_ptr()), "blp_debug_gemm_dump", H)
        ops.rank_all(model, table, fixed, rel2, b, true_row=torch.cat((t[:, 0], t[:, 1])))  # dump call: counts meaningless
        torch.cuda.synchronize()
    assert not torch.isnan(S).any() and not torch.isnan(E).any()
    return table, fixed, rel2, b, S, E


@pytest.mark.parametrize("data", ["bench", "scaled"])
@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
def test_bf16_band_margin(bench, model, data):
    """The whole band of the bilinear pre-pass on the kernel's own numbers: |S~ - S_ref| (S_ref = the
    reference's f32 score, bit-exact from blp_score_fwd) against the band half-width eps the kernel used for that
    pair.  Correctness needs ratio <= 1 for every pair.  On the bench data the largest ratio is ~0.08; on the
    adversarial set ~0.37, where the error is dominated by the terms the bf16 split drops (194 u of the band's 620 u,
    a rigorous bound that is nearly attained when a few same-sign products dominate the sum) -- the test fails above
    1/2.  The one empirical ingredient, the matrix pipe's internal rounding, is isolated in the next test."""
    from blp_amd import ops
    table, fixed, rel2, b, S, E = _dump_block(bench, model, data)
    ent = table.unsqueeze(0)
    ref = torch.cat((ops.score(model, ent, fixed[:b].unsqueeze(1), rel2[:b].unsqueeze(1)),
                     ops.score(model, fixed[b:].unsqueeze(1), ent, rel2[b:].unsqueeze(1))))
    err = (S.double() - ref.double()).abs()
    finite = torch.isfinite(E)
    assert finite.float().mean() > 0.99  # an infinite band (exact path) only for rows outside the relative bounds
    ratio = torch.where(finite & (E > 0), err / E.double(), torch.zeros_like(err))
    assert (err[E == 0] == 0).all()  # all-zero rows: no band needed, no error made
    worst = ratio.max().item()
    print(f"bf16 x 3 band, {model} / {data}: max |S~ - S_ref| / eps = {worst:.4f} over {S.numel():,} pairs")
    assert worst < 0.5, worst


@pytest.mark.parametrize("data", ["bench", "scaled"])
def test_mfma_accumulation_assumption(bench, data):
    """Assumption (A) of DESIGN.md 4.3, checked on the kernel's own output: the matrix pipe adds the products of a
    v_mfma_f32_32x32x16_bf16 to its accumulator no worse than one at a time with a truncating f32 addition (2 u each).
    With the kernel's order -- 32 cross-term MFMAs first (partial sums <= 2^-7 T), then the 16 main ones -- that bounds
    |S~ - S3| by 262 u T, where S3 = sum w_hi e_hi + w_hi e_lo + w_lo e_hi exactly (products of bf16 pairs are exact;
    computed here in f64 from the same round-to-nearest-even splits) and T = sum |w_k| |e_k|.  DistMult: W = f * r, one
    f32 product, the same on both sides.  The ratio must stay below 1; it is printed and fails above 1/2."""
    table, fixed, rel2, b, S, E = _dump_block(bench, "distmult", data)

    def split(x):
        hi = x.bfloat16().float()
        lo = (x - hi).bfloat16().float()
        return hi.double(), lo.double()

    w_hi, w_lo = split(fixed * rel2)
    e_hi, e_lo = split(table)
    s3 = w_hi @ e_hi.T + w_hi @ e_lo.T + w_lo @ e_hi.T
    T = (fixed * rel2).abs().double() @ table.abs().double().T
    u = 2.0 ** -24
    err = (S.double() - s3).abs()
    bound = 262.0 * u * T
    ok = bound > 0
    ratio = (err[ok] / bound[ok]).max().item()
    assert (err[~ok] == 0).all()
    print(f"MFMA accumulation, distmult / {data}: max |S~ - S3| / (262 u T) = {ratio:.4f} over {S.numel():,} pairs")
    assert ratio < 0.5, ratio
