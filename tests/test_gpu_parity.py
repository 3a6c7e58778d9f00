"""GPU parity: the HIP path (through the C-ABI) against the golden vectors and the CPU oracle.
Bit-exact for scores and rank counts; tolerance (stated per test) for the floating-point loss."""
import numpy as np
import pytest
import torch

from conftest import REL_MODELS, golden, golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from blp_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True, params=["shipped", "prepass"])
def routing(request):
    """Every test here runs twice.  "shipped": no knob at all -- libblp_hip.so, the dispatch production callers get
    (these tests use small tables, so most blocks then take the small-block / exact kernels).  "prepass": the hooks
    library with the knobs that send every block of >= 64 (TransE) / 32 (bilinear) queries to the pre-pass kernels, as
    at evaluation sizes.  Tests marked `default_routing` run at the shipped dispatch only (tests/conftest.py)."""
    from blp_amd import _lib
    _lib.reset_knobs()
    if request.param == "prepass":
        _lib.set_knob("sad_min_queries", 64)
        _lib.set_knob("small_kernel", 2)  # the MFMA pre-pass, not the small-block kernel, from 32 queries on
    yield request.param
    _lib.reset_knobs()


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def random_problem(model, N, D, q_head, q_tail, seed, nrel=11):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(N, D, generator=g)
    table = torch.nn.functional.normalize(table, dim=-1) if model == "transe" else table * 0.1
    rel_w = (torch.rand(nrel, D, generator=g) - 0.5) * 0.25
    Q = q_head + q_tail
    fixed_row = torch.randint(0, N, (Q,), generator=g)
    true_row = torch.randint(0, N, (Q,), generator=g)
    rels = torch.randint(0, nrel, (Q,), generator=g)
    return table, table[fixed_row].clone(), rel_w[rels].clone(), true_row


def oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=None, q_true=None, csr=None):
    t, f, r = table.numpy(), q_fixed.numpy(), q_rel.numpy()
    parts = []
    for side, sl in ((oracle.SIDE_HEAD, slice(0, q_head)), (oracle.SIDE_TAIL, slice(q_head, None))):
        if f[sl].shape[0] == 0:
            continue
        kw = {}
        if true_row is not None:
            kw["true_row"] = true_row.numpy()[sl]
        else:
            kw["q_true"] = q_true.numpy()[sl]
        if csr is not None:
            rowptr, col = csr
            lo, hi = (0, q_head) if side == oracle.SIDE_HEAD else (q_head, len(rowptr) - 1)
            kw["filt_rowptr"] = rowptr[lo:hi + 1] - rowptr[lo]
            kw["filt_col"] = col[rowptr[lo]:rowptr[hi]]
        parts.append(oracle.rank_counts(model, side, t, f[sl], r[sl], **kw))
    return np.concatenate(parts) if parts else np.zeros((0, 4), np.int32)


def random_csr(Q, N, true_row, seed, max_per_row=9):
    rng = np.random.default_rng(seed)
    rowptr = [0]
    cols = []
    for q in range(Q):
        k = int(rng.integers(0, max_per_row + 1))
        if q % 7 == 0:
            k = 0  # empty rows
        c = rng.choice(N, size=min(k, N), replace=False)
        c = c[c != int(true_row[q])]  # the true entity is never filtered (utils.py:71,78)
        cols.append(np.sort(c))
        rowptr.append(rowptr[-1] + len(c))
    return np.asarray(rowptr, np.int64), (np.concatenate(cols) if cols else np.zeros(0)).astype(np.int64)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("scores_"))
def test_golden_rank_counts_and_scores(ops, name):
    g = golden(name)
    model = name.split("_")[1]
    table = dev(g["table"])
    rel_w = dev(g["rel_w"])
    heads, tails, rels = (dev(g[k])[:, 0] for k in ("heads", "tails", "rels"))
    B = heads.shape[0]
    q_fixed = torch.cat((table[tails], table[heads]))
    q_rel = torch.cat((rel_w[rels], rel_w[rels]))
    true_row = torch.cat((heads, tails))
    counts = ops.rank_all(model, table, q_fixed, q_rel, B, true_row=true_row).cpu().numpy()
    assert np.array_equal(counts[:, 0], g["gt"])
    assert np.array_equal(counts[:, 1], g["ge"])
    assert np.array_equal(counts[:, 2:], counts[:, :2])
    rr, hits = ops.rank_metrics(dev(counts))
    assert np.array_equal(rr[:, 0].cpu().numpy().view(np.uint32), g["rr"][:, 0].view(np.uint32))
    assert np.array_equal(hits[:, 0].cpu().numpy(), g["hits"])
    # score_fn with the reference's own broadcast shapes (train.py:146-147): bit-identical scores
    ent = table.unsqueeze(0)
    hp = ops.score(model, ent, table[tails].unsqueeze(1), rel_w[rels].unsqueeze(1)).cpu().numpy()
    tp = ops.score(model, table[heads].unsqueeze(1), ent, rel_w[rels].unsqueeze(1)).cpu().numpy()
    assert np.array_equal(hp.view(np.uint32), g["head_pred"].view(np.uint32))
    assert np.array_equal(tp.view(np.uint32), g["tail_pred"].view(np.uint32))


@pytest.mark.parametrize("model", REL_MODELS)
@pytest.mark.parametrize("D", [64, 128, 256])
def test_random_vs_oracle_ragged(ops, oracle, model, D):
    """N not a multiple of the 64-row tile, more than one query chunk, both sides, CSR filter."""
    N, q_head, q_tail = 1000 + D // 64, 300, 41
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=D + len(model))
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.default_routing
@pytest.mark.parametrize("q_head,q_tail", [(70, 58), (300, 41), (1, 180)])
def test_transe_small_blocks_take_the_exact_kernels(ops, oracle, q_head, q_tail):
    """The library's own routing (no knob): 128 .. 341 TransE queries against a 2 000-row table are below the
    pre-pass's break-even and go to the small-block exact f32 kernel (rank_small.hip).  Counts against the oracle,
    with a filter."""
    from blp_amd import _lib
    L = _lib.lib()
    N, D = 2000, 128
    assert L.blp_rank_all_workspace_bytes(0, N, D, q_head, q_tail) < (q_head + q_tail) * 2 * D * 4 + N * D  # no table image
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=q_head)
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=q_tail)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)


SMALL_SHAPES = [(5, 7), (64, 64), (0, 33), (37, 0), (130, 129), (2, 2), (4, 3), (1, 0)]


@pytest.mark.default_routing
@pytest.mark.parametrize("model", REL_MODELS)
@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("q_head,q_tail", SMALL_SHAPES)
def test_small_block_kernel_vs_oracle(ops, oracle, knobs, model, D, q_head, q_tail):
    """rank_small.hip (coefficients computed in the kernel, LDS broadcasts, four TransE chains interleaved) on every
    block it can take: both sides, one side only, query counts that are no multiple of its group of four, a table that
    is no multiple of a workgroup's 256 rows (the last quad has an empty tile), CSR filter.  == oracle."""
    N = 1000 + D // 64 if q_head != 37 else 300
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=D + q_head + len(model))
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D + q_tail)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    # TransE: 1 = the tile in LDS + coefficients in scalar registers, 3 = the tile in registers
    for variant in ((1, 3) if model == "transe" else (1,)):
        knobs("small_kernel", variant)
        got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                           filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
        assert np.array_equal(got, want), variant


@pytest.mark.default_routing
@pytest.mark.parametrize("model", ["transe", "complex"])
@pytest.mark.parametrize("chunk", [1, 5, 12, 32, 50, 96])
def test_small_block_kernel_query_chunks(ops, oracle, knobs, model, chunk):
    """Any number of queries per workgroup (the host picks a multiple of 32 by the size of the block): partial rounds,
    waves with no query, chunks that straddle the head / tail boundary."""
    knobs("exact_query_chunk", chunk)
    N, D, q_head, q_tail = 777, 128, 45, 38
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=chunk)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
    for variant in ((1, 3) if model == "transe" else (1,)):
        knobs("small_kernel", variant)
        got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda()).cpu().numpy()
        assert np.array_equal(got, want), variant


@pytest.mark.default_routing
@pytest.mark.parametrize("model,q_head,q_tail", [("transe", 40, 50), ("distmult", 3, 9), ("simple", 70, 0)])
def test_small_block_kernel_table_of_more_tiles_than_slots(ops, oracle, knobs, model, q_head, q_tail):
    """Tables of more than 256 tiles.  The register-tile kernel: a workgroup walks several tiles (coefficients staged once
    for up to two rounds, again per tile otherwise) and its slot's partial counts add them up.  TransE's scalar-register
    kernel (tables of up to 1 024 tiles): one workgroup per (tile, chunk), the tiles of a slot ADD to counts the true-key
    launch has zeroed; beyond 1 024 tiles TransE is back on the register-tile kernel."""
    for N in (256 * 64 * 2 + 777, 1024 * 64 + 5):
        D = 64
        table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=q_head)
        rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=q_tail)
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
        for variant in ((1, 3) if model == "transe" else (1,)):
            knobs("small_kernel", variant)
            got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                               filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
            assert np.array_equal(got, want), (N, variant)
            got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                               filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()  # (a second call: the slots' counts start at zero again)
            assert np.array_equal(got, want), (N, variant, "second call")


@pytest.mark.default_routing
@pytest.mark.parametrize("model", REL_MODELS)
def test_small_block_kernel_ties_and_nonfinite(ops, oracle, knobs, model):
    """Duplicated rows (ties with the true entity: > vs >=), a NaN row, an infinite value, true entity given as a
    vector (candidate shards): the small-block kernel keeps the reference's counts."""
    N, D, q_head, q_tail = 600, 128, 21, 22
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=len(model))
    table[100:140] = table[true_row[:40]]           # exact duplicates of true entities
    table[7, 3] = float("nan")
    table[9, 5] = float("inf")
    table[11] = 0.0
    q_true = table[true_row].clone()
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, q_true=q_true)
    # TransE: the LDS-tile kernel with the coefficients in scalar registers (1) and the register-tile one (3)
    for variant in ((1, 3) if model == "transe" else (1,)):
        knobs("small_kernel", variant)
        got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, q_true=q_true.cuda()).cpu().numpy()
        assert np.array_equal(got, want), variant
    assert (want[:40, 1] > want[:40, 0]).all()      # the ties are really there


@pytest.mark.default_routing
@pytest.mark.parametrize("model", REL_MODELS)
def test_small_blocks_default_routing_equals_the_other_kernels(ops, model, knobs):
    """A reference-sized batch (64 triples = 128 queries) against an FB15k-237-sized table, the library's own routing
    (the small-block kernel) == the exact tile kernel == the pre-pass path."""
    N, D, q_head, q_tail = 14541, 128, 64, 64
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=5)
    args = (model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
    auto = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    knobs("rank_kernel", 1)
    tiles = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    knobs("rank_kernel", 0)
    knobs("small_kernel", 2)
    knobs("sad_min_queries", 64)
    prepass = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    assert np.array_equal(auto, tiles) and np.array_equal(auto, prepass)


@pytest.mark.parametrize("model,D", [("transe", 64), ("transe", 128), ("transe", 256), ("distmult", 64), ("distmult", 128),
                                     ("complex", 64), ("complex", 128), ("simple", 64), ("simple", 128), ("distmult", 256),
                                     ("complex", 256), ("simple", 256)])
@pytest.mark.parametrize("q_head,q_tail", [(2, 2), (4, 4), (0, 3), (1, 0), (3, 1)])
def test_stream_kernels_vs_oracle(ops, oracle, knobs, model, D, q_head, q_tail):
    """A handful of queries (the reference's Wikidata5M eval batch: 2 triples = 4 queries per table pass,
    scripts/blp-*-wikidata5m.sh:18) go to rank_stream.hip: TransE consumes the table 32 columns at a time from a load
    ring that runs across tile boundaries, the bilinear models through a workgroup's double-buffered LDS tile.  Tables of
    1 row .. many tiles per wave with a ragged last tile, a tie with the true entity, non-finite rows, a filter; counts
    against the oracle and against rank_tiles<STATIC> (knob)."""
    for N in (1, 63, 130, 70001):
        table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=N + D + q_head)
        if N > 100:
            table[17] = table[int(true_row[0])]  # an exact tie with query 0's true entity
            table[N - 1, 3] = float("inf")
            table[N // 2, 0] = float("nan")
        rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=N)
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
        args = (model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
        kw = dict(true_row=true_row.cuda(), filt_rowptr=dev(rowptr), filt_col=dev(col))
        knobs("stream_kernel", 0)
        got = ops.rank_all(*args, **kw).cpu().numpy()
        knobs("stream_kernel", 2)
        tiles = ops.rank_all(*args, **kw).cpu().numpy()
        assert np.array_equal(got, want), (N, got, want)
        assert np.array_equal(tiles, want)
        # both streaming kernels of the model (the library picks by table length); the bilinear models also with
        # order-exact keys only (5: no approximate keys)
        # (bilinear models at D = 256: 3 / 5 have no workgroup-tile kernel to force -- rank_tiles<STATIC> takes the block)
        for variant in (3, 4) if model == "transe" else (3, 4, 5):
            knobs("stream_kernel", variant)
            assert np.array_equal(ops.rank_all(*args, **kw).cpu().numpy(), want), (N, variant)


@pytest.mark.default_routing  # (the streaming kernels do not depend on the pre-pass knobs of the `routing` fixture: once is enough)
@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("kind", ["ties", "near", "scales", "tiny", "zero", "huge", "nonfinite", "constant"])
def test_stream_dot_band_adversarial(ops, oracle, knobs, model, D, kind):
    """The bilinear models' approximate keys (rank_stream.hip, DOT: a chain of fused multiply-adds decided against the true
    key within C u ||B_q|| ||e||, undecided rows re-scored in the reference's order) on inputs built to sit inside or
    break the band: duplicated rows, rows nudged by one ulp, per-row and per-column scales over 2^-40 .. 2^40, rows whose
    squares underflow, all-zero rows and queries, magnitudes near overflow, NaN / Inf, a constant table.  Both kernels
    (workgroup tile, ring), 2 + 2 and 4 + 3 queries, counts against the oracle."""
    N = 70001 + D
    for q_head, q_tail in ((2, 2), (4, 3)):
        table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=D + q_head + len(kind))
        g = torch.Generator().manual_seed(7 * D + q_tail)
        Q = q_head + q_tail
        if kind == "ties":
            for q in range(Q):
                table[torch.randint(0, N, (40,), generator=g)] = table[int(true_row[q])].clone()
        elif kind == "near":
            for q in range(Q):
                rows = torch.randint(0, N, (64,), generator=g)
                near = table[int(true_row[q])].repeat(64, 1)
                col = torch.randint(0, D, (64,), generator=g)
                bits = near[torch.arange(64), col].view(torch.int32) + torch.randint(-2, 3, (64,), generator=g, dtype=torch.int32)
                near[torch.arange(64), col] = bits.view(torch.float32)
                table[rows] = near
        elif kind == "scales":
            table *= torch.exp2(torch.randint(-40, 41, (N, 1), generator=g).float())
            table *= torch.exp2(torch.randint(-6, 7, (1, D), generator=g).float())
            q_rel *= torch.exp2(torch.randint(-20, 21, (Q, 1), generator=g).float())
            q_fixed = table[torch.randint(0, N, (Q,), generator=g)].clone()
        elif kind == "tiny":
            table[::3] *= 1e-17   # squares underflow to denormals / zero
            table[1::7] *= 1e-24
            q_fixed[0] *= 1e-20
        elif kind == "zero":
            table[::5] = 0.0
            q_fixed[0] = 0.0
            q_rel[-1] = 0.0
            true_row[1] = 5  # a true entity with an all-zero row
        elif kind == "huge":
            table[::4] *= 1e18
            table[2::9] *= 3e19
            q_rel[0] *= 1e19
            q_fixed[-1] *= 1e19
        elif kind == "nonfinite":
            table[torch.randint(0, N, (50,), generator=g), torch.randint(0, D, (50,), generator=g)] = float("inf")
            table[torch.randint(0, N, (50,), generator=g), torch.randint(0, D, (50,), generator=g)] = float("nan")
            q_rel[0, 3] = float("inf")
            q_fixed[-1, 1] = float("nan")
        elif kind == "constant":
            table[:] = 0.25
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
        args = (model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
        for variant in (0, 3, 4):
            knobs("stream_kernel", variant)
            got = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
            assert np.array_equal(got, want), (kind, variant, q_head, np.nonzero((got != want).any(1))[0][:8])


@pytest.mark.parametrize("D", [64, 128, 256])
def test_transe_exact_kernels_many_queries(ops, oracle, D, knobs):
    """Q >= 256 TransE normally takes the fixed-point pre-pass; the rank_kernel knob selects the exact f32
    kernel (hand-pipelined VALU) for the same block.  Same counts."""
    knobs("rank_kernel", 1)
    N, q_head, q_tail = 1000 + D // 64, 300, 41
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=D + 3)
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_sad_band_adversarial(ops, oracle, D):
    """The fixed-point pre-pass (TransE, Q >= 256) on inputs built to land inside its band: duplicated
    rows (exact ties with the true entity), dyadic values, rows nudged by 2^-20 (near-ties far below
    the 16-bit resolution), an all-zero row and an outlier row that stretches the quantisation range."""
    g = torch.Generator().manual_seed(D)
    N, q_head, q_tail = 900, 150, 170
    base = torch.randint(-8, 9, (N // 4, D), generator=g).float() / 8.0
    table = base.repeat(4, 1)[torch.randperm(4 * (N // 4), generator=g)]
    table[::7] *= (1.0 + 2.0 ** -20)
    table[3::11] *= (1.0 - 2.0 ** -20)
    table[5] = 0.0
    table[6] *= 40.0
    N = table.shape[0]
    Q = q_head + q_tail
    q_fixed = table[torch.randint(0, N, (Q,), generator=g)].clone()
    q_rel = torch.randint(-16, 17, (Q, D), generator=g).float() / 16.0
    true_row = torch.randint(0, N, (Q,), generator=g)
    rowptr, col = random_csr(Q, N, true_row.numpy(), seed=D + 1)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)
    assert (want[:, 1] - want[:, 0]).max() > 1  # real ties: the > / >= split must survive


@pytest.mark.parametrize("case", ["nan", "inf", "constant", "huge", "all_ties"])
def test_sad_degenerate_inputs_stay_exact(ops, oracle, case):
    """Non-finite values or a zero-width range switch the pre-pass off (every tile is refined exactly);
    values near FLT_MAX and an all-ties table go through it.  Counts equal the oracle's in all cases."""
    D, N, q_head, q_tail = 128, 700, 130, 140
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=31)
    if case == "nan":
        table[17, 3] = float("nan"); q_rel[7, 9] = float("nan")
    elif case == "inf":
        table[40, 5] = float("inf"); q_fixed[50] = table[40]
    elif case == "constant":
        table[:] = 0.25; q_fixed[:] = 0.25; q_rel[:] = 0.0
    elif case == "huge":
        table[11] = 1e37; table[12] = -1e37
    elif case == "all_ties":
        table[:] = table[0]
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row)
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)


def test_sad_candidate_slabs_and_shard_vectors(ops, oracle, knobs):
    """BLP_SAD_PASS_GROUPS=1 forces 512-candidate slabs (ragged last slab); q_true instead of true_row
    (the form a candidate shard sees) gives the same counts."""
    D, N, q_head, q_tail = 128, 2000 + 37, 170, 190
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=29)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row)
    args = ("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
    one_pass = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    by_vector = ops.rank_all(*args, q_true=table[true_row].cuda()).cpu().numpy()
    knobs("sad_pass_groups", 1)
    slabs = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    assert np.array_equal(one_pass, want)
    assert np.array_equal(by_vector, want)
    assert np.array_equal(slabs, want)


@pytest.mark.parametrize("D", [300, 768, 100, 4, 1024])
def test_transe_any_width_prepass(ops, oracle, D, knobs):
    """TransE at the widths of the reference's BOW / DKRL encoders (300 GloVe, 768 BERT) and other widths
    the register-resident kernels are not compiled for: the any-width fixed-point pre-pass (rank_sad_wide.hip),
    whatever the number of queries (the BOW scripts evaluate 16 or 32 triples at a time).  Counts identical to the
    oracle with a CSR filter, with the true entity given as a row or as a vector, with exact ties, and over
    candidate slabs."""
    assert not ops.dim_supported("transe", D) and ops.rank_all_supported("transe", D, 150, 170)
    assert ops.rank_all_supported("transe", D, 1, 0) and not ops.rank_all_supported("distmult", D, 150, 170)
    for q_head, q_tail in ((3, 4), (0, 33), (16, 16), (1, 0)):  # the reference's small eval batches
        N = 702 + 9 * q_head
        table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=D + q_tail)
        table[::9] = table[1::9]  # exact ties
        q_fixed = table[torch.arange(q_head + q_tail) * 3 % N].clone()
        rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D + 2)
        want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
        got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                           filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
        assert np.array_equal(got, want), (q_head, q_tail)
    N, q_head, q_tail = 1000 + 37, 150, 170
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=D)
    table[::9] = table[1::9][: table[::9].shape[0]]  # exact ties
    q_fixed = table[torch.arange(q_head + q_tail) * 3 % N].clone()
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D + 1)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    args = ("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
    got = ops.rank_all(*args, true_row=true_row.cuda(), filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)
    by_vector = ops.rank_all(*args, q_true=table[true_row].cuda(), filt_rowptr=dev(rowptr), filt_col=dev(col))
    assert np.array_equal(by_vector.cpu().numpy(), want)
    knobs("sad_pass_groups", 2)  # 512-candidate slabs
    slabs = ops.rank_all(*args, true_row=true_row.cuda(), filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(slabs, want)
    assert (want[:, 1] - want[:, 0]).max() > 1


@pytest.mark.default_routing
@pytest.mark.parametrize("D,N,q_head,q_tail", [(300, 9100, 40, 37), (768, 16500, 70, 61), (100, 8300, 5, 3)])
def test_transe_any_width_refinement_across_group_slices(ops, oracle, D, N, q_head, q_tail):
    """The any-width refinement (rank_sad_wide.hip: wide_refine_chunks_kernel) sorts a query chunk's undecided pairs by query
    in LDS, 32 candidate groups' regions at a time: tables of more than 32 groups (8 192 rows) take several slices per chunk.
    Near-duplicate rows (a few ulps off their neighbours: the fixed-point band cannot separate them) fill the regions --
    past their quota in places, which sends those tiles to the flag sweep -- and exact duplicates pin the tie rule; counts
    identical to the oracle, raw and with a CSR filter."""
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=N + D)
    g = torch.Generator().manual_seed(D)
    # clusters of near-identical candidates around some of the true entities
    for t in true_row[::3].tolist():
        rows = torch.randint(0, N, (180,), generator=g)
        table[rows] = table[t] * (1.0 + 1e-7 * torch.randn(180, 1, generator=g))
    table[5::1000] = table[4::1000][: table[5::1000].shape[0]]  # exact ties
    # ... and 200 CONSECUTIVE near-copies of one true entity: more undecided pairs than one (chunk, group) region lists
    t0 = int(true_row[1])
    base = (t0 + 1000) % (N - 300)
    table[base:base + 200] = table[t0] * (1.0 + 1e-7 * torch.randn(200, 1, generator=g))
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D + 5)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)
    assert (want[:, 1] - want[:, 0]).max() >= 1


@pytest.mark.parametrize("case", ["nan", "constant", "huge"])
def test_transe_any_width_degenerate_inputs(ops, oracle, case):
    D, N, q_head, q_tail = 300, 500, 130, 140
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=77)
    if case == "nan":
        table[17, 3] = float("nan"); q_rel[7, 9] = float("inf")
    elif case == "constant":
        table[:] = 0.25; q_fixed[:] = 0.25; q_rel[:] = 0.0
    else:
        table[11] = 1e37; table[12] = -1e37
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row)
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
@pytest.mark.parametrize("D", [64, 128])
def test_gemm_band_adversarial(ops, oracle, model, D):
    """The GEMM + error-band path (Q >= 64) on inputs built to land inside the band: duplicated rows
    (exact ties with the true entity), dyadic values (every order exact), rows scaled by 1 +- 2^-20
    (near-ties), a huge-norm row and an all-zero row.  Counts must still be bit-exact."""
    g = torch.Generator().manual_seed(D)
    N, q_head, q_tail = 700, 50, 46
    base = torch.randint(-8, 9, (N // 4, D), generator=g).float() / 8.0
    table = base.repeat(4, 1)[torch.randperm(4 * (N // 4), generator=g)]
    table[::7] *= (1.0 + 2.0 ** -20)
    table[3::11] *= (1.0 - 2.0 ** -20)
    table[5] = 0.0
    table[6] *= 1e4
    N = table.shape[0]
    Q = q_head + q_tail
    q_fixed = table[torch.randint(0, N, (Q,), generator=g)].clone()
    q_rel = torch.randint(-16, 17, (Q, D), generator=g).float() / 16.0
    true_row = torch.randint(0, N, (Q,), generator=g)
    rowptr, col = random_csr(Q, N, true_row.numpy(), seed=D + 1)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)
    # ties make the realistic rank a half-integer for some queries: the >/>= split must survive
    assert (want[:, 1] - want[:, 0]).max() > 1


@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
def test_gemm_candidate_slabs(ops, oracle, model, knobs):
    """When the flag bitmap / pair regions would exceed their cap the candidate axis is processed in
    slabs.  BLP_GEMM_PASS_WORDS=1 forces 512-candidate slabs: counts must equal the one-pass result
    and the oracle's (ragged last slab, true entities in every slab)."""
    D, N, q_head, q_tail = 128, 2000 + 37, 70, 90
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=23)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
    args = (model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
    one_pass = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    knobs("gemm_pass_words", 1)
    slabs = ops.rank_all(*args, true_row=true_row.cuda()).cpu().numpy()
    assert np.array_equal(one_pass, want)
    assert np.array_equal(slabs, want)


def test_gemm_pair_quota_overflow_falls_back_to_flags(ops, oracle):
    """Every candidate ties with the true entity (constant table): far more undecided pairs than a
    workgroup can list, so almost everything takes the flagged half-segment path."""
    D, N, Q = 128, 1000, 128
    table = torch.full((N, D), 0.25)
    q_fixed = torch.full((Q, D), 0.5)
    q_rel = torch.full((Q, D), -0.125)
    true_row = torch.arange(Q) * 7 % N
    for model in ("distmult", "complex", "simple"):
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, Q // 2, true_row=true_row)
        got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), Q // 2, true_row=true_row.cuda())
        assert np.array_equal(got.cpu().numpy(), want)
        assert (want[:, 1] == N).all() and (want[:, 0] == 0).all()


@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
def test_gemm_path_nonfinite_inputs_take_the_exact_path(ops, oracle, model):
    D, N, q_head, q_tail = 128, 300, 40, 40
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=9)
    table[17, 3] = float("inf")
    table[40, 5] = float("nan")
    q_rel[7, 9] = float("nan")
    q_fixed[50] = table[17]
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("sort", [False, True])
def test_transe_relation_id_hint(ops, oracle, D, sort):
    """rel_ids only changes HOW head-replacing TransE queries are scored (e + r shared per relation),
    never the counts -- in any order of the block, with a CSR filter, more than one query chunk."""
    g = torch.Generator().manual_seed(17 + D)
    N, q_head, q_tail, nrel = 1000, 300, 70, 7
    table = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1)
    rel_w = (torch.rand(nrel, D, generator=g) - 0.5) * 0.25
    rels = torch.randint(0, nrel, (q_head,), generator=g)
    if sort:
        rels = torch.sort(rels).values
    rels_t = torch.randint(0, nrel, (q_tail,), generator=g)
    fixed_row = torch.randint(0, N, (q_head + q_tail,), generator=g)
    true_row = torch.randint(0, N, (q_head + q_tail,), generator=g)
    q_fixed = table[fixed_row].clone()
    q_rel = torch.cat((rel_w[rels], rel_w[rels_t]))
    rel_ids = torch.cat((rels, rels_t))
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    for ids in (None, rel_ids.cuda()):
        got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                           filt_rowptr=dev(rowptr), filt_col=dev(col), rel_ids=ids)
        assert np.array_equal(got.cpu().numpy(), want), ("with ids" if ids is not None else "without ids")


@pytest.mark.parametrize("model,D", [("transe", 128), ("transe", 300), ("distmult", 128), ("complex", 64)])
def test_prepass_paths_are_repeatable(ops, model, D):
    """Integer counts cannot depend on the order in which workgroups finish: 12 runs of a block big enough
    for several waves of workgroups (pair lists, flag bitmap, LDS counters, global atomics) are identical."""
    N, q_head, q_tail = 5000 + 11, 900, 1000
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=D + 5)
    table[::13] = table[1::13][: table[::13].shape[0]]  # ties keep the refinement paths busy
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D)
    args = (model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head)
    kw = dict(true_row=true_row.cuda(), filt_rowptr=dev(rowptr), filt_col=dev(col))
    first = ops.rank_all(*args, **kw)
    for _ in range(11):
        assert torch.equal(ops.rank_all(*args, **kw), first)
    assert int(first[:, 1].min()) >= 1


@pytest.mark.parametrize("model,D", [("transe", 128), ("transe", 64), ("transe", 300), ("distmult", 128), ("complex", 64),
                                     ("simple", 128)])
@pytest.mark.parametrize("gemm_kernel", [None, "f32"])
def test_workspace_is_never_overrun(model, D, gemm_kernel, knobs):
    """blp_rank_all through the raw C-ABI with a workspace of exactly blp_rank_all_workspace_bytes bytes
    followed by a canary: every kernel path, for query-block shapes whose tile / chunk counts do not
    divide evenly (1 tile on one side, 17 on the other, ...), must leave the canary intact.  (A
    randomised soak found the bilinear pair list sized for the wrong chunking: a write past the end
    that only faulted when the allocation happened to end a mapped segment.)"""
    from blp_amd import _lib
    if gemm_kernel:
        if model == "transe":
            pytest.skip("bilinear knob")
        knobs("gemm_kernel", 1)
    L = _lib.lib()
    g = torch.Generator().manual_seed(D)
    guard = 1 << 16
    for N, q_head, q_tail in ((1692, 241, 208), (1668, 456, 4), (700, 4, 540), (2100, 33, 513), (97, 1, 300), (4099, 300, 300)):
        table = torch.randn(N, D, generator=g).cuda()
        dup = table[1::3]
        table[0:3 * dup.shape[0]:3] = dup  # ties: plenty of undecided pairs
        Q = q_head + q_tail
        q_fixed = table[torch.randint(0, N, (Q,), generator=g).cuda()].contiguous()
        q_rel = (torch.randn(Q, D, generator=g) * 0.1).cuda()
        true_row = torch.randint(0, N, (Q,), generator=g).cuda()
        counts = torch.empty((Q, 4), dtype=torch.int32, device="cuda")
        need = L.blp_rank_all_workspace_bytes(_lib.MODEL_IDS[model], N, D, q_head, q_tail)
        ws = torch.full((need + guard,), 0xA5, dtype=torch.uint8, device="cuda")
        rc = L.blp_rank_all(_lib.MODEL_IDS[model], table.data_ptr(), N, D, D, q_fixed.data_ptr(), q_rel.data_ptr(),
                            true_row.data_ptr(), None, q_head, q_tail, None, counts.data_ptr(), ws.data_ptr(), need,
                            0, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "blp_rank_all")
        torch.cuda.synchronize()
        assert bool((ws[need:] == 0xA5).all()), (N, q_head, q_tail)
        assert int(counts[:, 1].min()) >= 1


@pytest.mark.parametrize("model", REL_MODELS)
def test_edge_shapes(ops, oracle, model):
    D = 128
    for N, q_head, q_tail in ((1, 1, 1), (63, 0, 5), (64, 5, 0), (65, 2, 2), (129, 1, 0)):
        table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=N)
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
        got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda())
        assert np.array_equal(got.cpu().numpy(), want), (N, q_head, q_tail)
    # empty query block
    table = torch.randn(10, D).cuda()
    out = ops.rank_all(model, table, torch.empty(0, D).cuda(), torch.empty(0, D).cuda(), 0,
                       true_row=torch.empty(0, dtype=torch.int64).cuda())
    assert out.shape == (0, 4)
    # strided table rows (ld > D)
    big = torch.randn(70, D + 4)
    tab = big[:, :D]
    if model == "transe":
        tab = torch.nn.functional.normalize(tab, dim=-1)
        big[:, :D] = tab
    q_fixed, q_rel = tab[:6].contiguous(), torch.randn(6, D) * 0.1
    true_row = torch.arange(6)
    want = oracle_counts(oracle, model, tab.contiguous(), q_fixed, q_rel, 3, true_row=true_row)
    got = ops.rank_all(model, big.cuda()[:, :D], q_fixed.cuda(), q_rel.cuda(), 3, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)
    # ... and with enough queries for the pre-pass paths (their own table readers honour ld too)
    idx = torch.arange(300) % 70
    q_fixed, q_rel, true_row = tab[idx].contiguous(), torch.randn(300, D) * 0.1, (idx * 7 + 3) % 70
    want = oracle_counts(oracle, model, tab.contiguous(), q_fixed, q_rel, 140, true_row=true_row)
    got = ops.rank_all(model, big.cuda()[:, :D], q_fixed.cuda(), q_rel.cuda(), 140, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("model", REL_MODELS)
def test_candidate_shards_add_up(ops, oracle, model):
    """Candidate-axis sharding: per-shard counts (true entity given as a vector) sum to the unsharded
    counts, for any split -- the property the multi-GPU ranking relies on."""
    N, D, q_head, q_tail = 777, 128, 19, 23
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=5)
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=3)
    t = table.cuda()
    whole = ops.rank_all(model, t, q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                         filt_rowptr=dev(rowptr), filt_col=dev(col))
    q_true = table[true_row].cuda()
    total = torch.zeros_like(whole)
    for lo, hi in ((0, 100), (100, 101), (101, 640), (640, N)):
        # CSR columns restricted to the shard and rebased
        keep = (col >= lo) & (col < hi)
        rp = np.zeros(len(rowptr), np.int64)
        for q in range(len(rowptr) - 1):
            rp[q + 1] = rp[q] + keep[rowptr[q]:rowptr[q + 1]].sum()
        total += ops.rank_all(model, t[lo:hi], q_fixed.cuda(), q_rel.cuda(), q_head, q_true=q_true,
                              filt_rowptr=dev(rp), filt_col=dev(col[keep] - lo))
    assert torch.equal(total, whole)
    assert np.array_equal(whole.cpu().numpy(),
                          oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row,
                                        csr=(rowptr, col)))


@pytest.mark.parametrize("model", REL_MODELS)
def test_fb15k237_shape_vs_oracle(ops, oracle, model):
    """BASELINE config 2/3 table shape (14 541 x 128), two reference batches worth of queries."""
    N, D, q_head, q_tail = 14541, 128, 64, 64
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=237, nrel=237)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row)
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda())
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.default_routing
def test_wikidata5m_scale_properties(ops, oracle):
    """BASELINE config 4 scale (4.6 M x 128, reference batch of 2 triples = 4 queries): too big for
    the oracle to rank in full, so check (a) a 200k-row slab against the oracle, (b) additivity of
    the slab decomposition against the one-pass result, (c) idempotence."""
    N, D = 4_600_000, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    table = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=-1)
    rows = torch.tensor([17, N - 1, 2_300_000, 4_599_000], device="cuda")
    q_fixed = table[rows].clone()
    q_rel = (torch.rand(4, D, device="cuda", generator=g) - 0.5) * 0.25
    true_row = torch.tensor([N - 5, 123_456, 0, 3_999_999], device="cuda")
    whole = ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
    again = ops.rank_all("transe", table, q_fixed, q_rel, 2, true_row=true_row)
    assert torch.equal(whole, again)
    q_true = table[true_row].clone()
    total = torch.zeros_like(whole)
    bounds = [0, 200_000, 1_000_003, 2_777_777, N]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        part = ops.rank_all("transe", table[lo:hi], q_fixed, q_rel, 2, q_true=q_true)
        if lo == 0:
            slab = table[lo:hi].cpu()
            want = oracle_counts(oracle, "transe", slab, q_fixed.cpu(), q_rel.cpu(), 2, q_true=q_true.cpu())
            assert np.array_equal(part.cpu().numpy(), want)
        total += part
    assert torch.equal(total, whole)
    assert int(whole[:, 1].min()) >= 1  # the true entity itself is always >=


@pytest.mark.default_routing
def test_wikidata5m_scale_prepass_paths_equal_exact_kernels(ops, oracle, knobs):
    """4.6 M x 128 with enough queries for the pre-pass paths (TransE: fixed-point SAD, Q = 512;
    DistMult: bf16 x 3 GEMM, Q = 128).  Their counts must equal (a) the exact f32 kernels' on the whole
    table and (b) the oracle's on a 100k-row slab (true entities as vectors, the sharded form)."""
    N, D = 4_600_000, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    table = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=-1)
    for model, q_half in (("transe", 256), ("distmult", 64)):
        Q = 2 * q_half
        fixed_row = torch.randint(0, N, (Q,), device="cuda", generator=g)
        true_row = torch.randint(0, N, (Q,), device="cuda", generator=g)
        q_fixed = table[fixed_row].clone()
        q_rel = (torch.rand(Q, D, device="cuda", generator=g) - 0.5) * (0.25 if model == "transe" else 2.0)
        fast = ops.rank_all(model, table, q_fixed, q_rel, q_half, true_row=true_row)
        if model == "transe":
            knobs("rank_kernel", 1)
            exact = ops.rank_all(model, table, q_fixed, q_rel, q_half, true_row=true_row)
            knobs("rank_kernel", 0)
        else:  # blocks of < 32 queries take the exact VALU kernel
            parts_h = [ops.rank_all(model, table, q_fixed[a:a + 16], q_rel[a:a + 16], 16, true_row=true_row[a:a + 16])
                       for a in range(0, q_half, 16)]
            parts_t = [ops.rank_all(model, table, q_fixed[a:a + 16], q_rel[a:a + 16], 0, true_row=true_row[a:a + 16])
                       for a in range(q_half, Q, 16)]
            exact = torch.cat(parts_h + parts_t)
        assert torch.equal(fast, exact), model
        assert int(fast[:, 1].min()) >= 1
        q_true = table[true_row].clone()
        slab = table[:100_000]
        part = ops.rank_all(model, slab, q_fixed, q_rel, q_half, q_true=q_true)
        want = oracle_counts(oracle, model, slab.cpu(), q_fixed.cpu(), q_rel.cpu(), q_half, q_true=q_true.cpu())
        assert np.array_equal(part.cpu().numpy(), want), model


@pytest.mark.parametrize("per_group", ["16", "32", "64", "128", "256"])
@pytest.mark.parametrize("D", [64, 128, 256])
def test_sad_queries_per_workgroup(ops, oracle, D, per_group, knobs):
    """A workgroup of the TransE pre-pass takes 16 .. 256 queries (a cost model picks; BLP_SAD_QUERIES_PER_GROUP
    forces).  Ragged last chunk, exact ties (pair lists), a block of identical rows that overflows the pair quota
    (flags -> sweep), a CSR filter: same counts for every chunk length."""
    knobs("sad_queries_per_group", int(per_group))
    N, q_head, q_tail = 900 + 7, 256 * 3 + 5, 256 + 131
    table, q_fixed, q_rel, true_row = random_problem("transe", N, D, q_head, q_tail, seed=91)
    table[::5] = table[1::5]
    table[300:560] = table[300]          # 260 identical candidates ...
    true_row[40:300] = torch.arange(300, 560)  # ... each the true entity of one of 260 consecutive queries
    q_fixed[40:300] = q_fixed[40]
    q_rel[40:300] = q_rel[40]
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=6)
    want = oracle_counts(oracle, "transe", table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col))
    assert np.array_equal(got.cpu().numpy(), want)
    assert (want[40:300, 1] - want[40:300, 0]).min() >= 259


@pytest.mark.parametrize("tiles", ["16", "32", "64"])
@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
def test_gemm_query_chunk_lengths(ops, oracle, model, tiles, knobs):
    """A workgroup of the bf16 pre-pass takes 16, 32 or 64 query tiles (the longest that still fills the
    chip; BLP_GEMM_TILES_PER_CHUNK forces one).  Ragged sides (1 .. 8 tiles past a multiple of the chunk), exact
    ties (pair lists), a constant block (quota overflow -> flags) and a CSR filter: same counts."""
    knobs("gemm_tiles_per_chunk", int(tiles))
    D, N, q_head, q_tail = 128, 700 + 5, 32 * 67 + 3, 32 * 33 - 7
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=77)
    table[::5] = table[1::5]
    table[300:420] = 0.25   # many candidates tie exactly for the queries below
    q_fixed[100:160] = 0.5
    q_rel[100:160] = -0.125
    true_row[100:160] = torch.arange(300, 360)
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=5)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col))
    assert np.array_equal(got.cpu().numpy(), want)
    assert (want[100:160, 1] - want[100:160, 0]).min() >= 119


@pytest.mark.parametrize("model", ["distmult", "complex", "simple"])
def test_gemm_f32_chain_variant(ops, oracle, model, knobs):
    """BLP_GEMM_KERNEL=f32 selects the f32 MFMA chain pre-pass (band constant 320) instead of bf16 x 3."""
    knobs("gemm_kernel", 1)
    D, N, q_head, q_tail = 128, 1000 + 9, 150, 141
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=41)
    table[::5] = table[1::5]  # exact ties
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=3)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col))
    assert np.array_equal(got.cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------
@pytest.mark.default_routing
@pytest.mark.parametrize("name", golden_names("loss_"))
def test_golden_inbatch_loss(ops, name):
    """Loss within 1e-6 relative, gradients within rtol 1e-5 / atol 1e-7 of the reference (f32
    reductions in a different order; the reference's own CPU/GPU results differ by as much)."""
    g = golden(name)
    _, model, loss_fn, _ = name.split("_")
    ent = dev(g["ent_embs"]).requires_grad_(True)
    rel_w = dev(g["rel_w"]).requires_grad_(True)
    rels = dev(g["rels"])
    neg_idx = dev(g["neg_idx"])
    loss = ops.inbatch_loss(model, loss_fn, ent, rel_w[rels], neg_idx, float(g["regularizer"]))
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(ent.grad.cpu().numpy(), g["grad_ent"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rel_w.grad.cpu().numpy(), g["grad_rel_w"], rtol=1e-5, atol=1e-7)


@pytest.mark.default_routing
@pytest.mark.parametrize("model", REL_MODELS)
@pytest.mark.parametrize("loss_fn", ["margin", "nll"])
def test_inbatch_loss_training_shape_vs_port(ops, model, loss_fn):
    """B = 64, K = 64, D = 128 (scripts/blp-*-fb15k237.sh) against the torch port on CPU; backward is
    deterministic (bitwise equal across runs)."""
    from oracle import ref_port
    torch.manual_seed(3)
    B, K, D = 64, 64, 128
    ent = torch.randn(B, 2, D) * (1.0 if model == "transe" else 0.4)
    if model == "transe":
        ent = torch.nn.functional.normalize(ent, dim=-1)
    rel = torch.randn(B, 1, D) * 0.3
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    reg = 1e-3 if model == "complex" else 0.0
    e_ref, r_ref = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    ref = ref_port.compute_loss(model, loss_fn, e_ref, r_ref, neg_idx, reg)
    ref.backward()
    results = []
    for _ in range(2):
        e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
        loss = ops.inbatch_loss(model, loss_fn, e, r, neg_idx.cuda(), reg)
        (loss * 2.0).backward()  # upstream gradient is honoured
        results.append((loss.item(), e.grad.clone(), r.grad.clone()))
    assert results[0][0] == pytest.approx(ref.item(), rel=2e-6, abs=1e-7)
    np.testing.assert_allclose(results[0][1].cpu().numpy() / 2.0, e_ref.grad.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(results[0][2].cpu().numpy() / 2.0, r_ref.grad.numpy(), rtol=2e-5, atol=2e-7)
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][2], results[1][2])


@pytest.mark.default_routing
@pytest.mark.parametrize("model,B,K,skew", [("transe", 1024, 64, None), ("complex", 1024, 64, None), ("distmult", 300, 7, None),
                                            ("simple", 200, 33, None), ("transe", 3, 2, None), ("distmult", 64, 80, "two_rows"),
                                            ("complex", 96, 64, "one_row"), ("transe", 520, 9, "few_rows"),
                                            ("distmult", 2100, 3, None), ("transe", 64, 1000, None), ("simple", 700, 1, "few_rows")])
def test_inbatch_loss_backward_row_ownership_and_rounds(ops, model, B, K, skew):
    """The backward's work split (inbatch_loss.hip): S waves share an entity row and walk its negatives from the index of
    neg_idx the forward leaves (chunks of 1 024 entries sorted stably by the row they name) -- batches of 1 024 (128 chunks:
    two blocks of 64 chunk offsets per row; 2 048 rows: two passes of the index workgroups' 1 024-row histograms), 2 100 (five
    passes), sizes that leave short last workgroups, K = 1 and K = 1 000, and negatives that all name one, two or four rows
    (every lane of a slice names the same row: the in-slice ranks; lists of tens of thousands of entries per row).
    Against the torch port, the training-shape test's tolerances; and bit-reproducible."""
    from oracle import ref_port
    torch.manual_seed(B + K)
    D = 128
    ent = torch.randn(B, 2, D) * (1.0 if model == "transe" else 0.4)
    rel = torch.randn(B, 1, D) * 0.3
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    if skew == "two_rows":
        neg_idx = torch.randint(0, 2, (B, K, 2))
    elif skew == "one_row":
        neg_idx[..., 0] = 5
    elif skew == "few_rows":
        neg_idx = torch.randint(100, 104, (B, K, 2))
    e_ref, r_ref = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    ref = ref_port.compute_loss(model, "margin", e_ref, r_ref, neg_idx, 1e-3)
    ref.backward()
    grads = []
    for _ in range(2):
        e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
        loss = ops.inbatch_loss(model, "margin", e, r, neg_idx.cuda(), 1e-3)
        loss.backward()
        grads.append((e.grad.clone(), r.grad.clone()))
    assert loss.item() == pytest.approx(ref.item(), rel=2e-6, abs=1e-7)
    # sums of up to 2 B K signed terms per row: a few ulps of the largest partial sum in absolute terms, and the more terms a
    # row adds the more roundings both sides make (K = 1 000: 2 000 terms of 1 / (B K) per TransE row, the reference's
    # index_put accumulation and this kernel's entry order differ by up to 1e-7 absolute on gradients of 0.025)
    scale = float(e_ref.grad.abs().max())
    atol = 2e-6 * max(scale, 1e-3) * max(1.0, K / 250.0)
    np.testing.assert_allclose(grads[0][0].cpu().numpy(), e_ref.grad.numpy(), rtol=2e-5, atol=atol)
    np.testing.assert_allclose(grads[0][1].cpu().numpy(), r_ref.grad.numpy(), rtol=2e-5, atol=atol)
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.default_routing
@pytest.mark.parametrize("model,B,K,skew", [("transe", 64, 64, None), ("distmult", 700, 5, None), ("transe", 1024, 64, None),
                                            ("distmult", 40, 60, "one_row"), ("transe", 300, 11, "dups")])
def test_inbatch_forward_leaves_a_stable_index_of_neg_idx(model, B, K, skew):
    """What the backward walks (include/blp_hip.h: save_pos; inbatch_loss.hip: SaveLayout): for every chunk of 1 024 consecutive
    entries of neg_idx.view(-1) the entries grouped by the row they name, IN ENTRY ORDER within a row, each with the row the
    other slot of its pair names, and the chunk's exclusive row offsets -- read back here from the raw C-ABI call and held against a stable sort on the host, entry for
    entry; and the ticket counter is left zero."""
    import ctypes
    from blp_amd import _lib
    torch.manual_seed(B * K)
    D = 128
    ent = (torch.randn(B, 2, D) * 0.4).cuda()
    rel = (torch.randn(B, D) * 0.3).cuda()
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    if skew == "one_row":
        neg_idx[..., 1] = 7
    elif skew == "dups":  # runs of equal values inside 64-entry slices, other values in between
        flat = neg_idx.view(-1)
        flat[::3] = flat[0]
        flat[1::7] = 2 * B - 1
    mid = _lib.MODEL_IDS[model]
    L = _lib.lib()
    n = _lib.inbatch_save_floats(mid, B, K, D)
    save = torch.full((n,), float("nan"), device="cuda")
    neg = torch.empty(B, K, device="cuda")
    loss = torch.empty((), device="cuda")
    ticket = torch.zeros(_lib.INBATCH_TICKET_INTS, dtype=torch.int32, device="cuda")
    dev_idx = neg_idx.cuda()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):  # (twice: the ticket comes back zeroed)
        _lib.check(L.blp_inbatch_loss_fwd(mid, 0, 0, 0, ent.data_ptr(), rel.data_ptr(), dev_idx.data_ptr(), B, K, D, 1e-3, loss.data_ptr(),
                                          save.data_ptr(), neg.data_ptr(), ticket.data_ptr(), 0, stream), "fwd")
    torch.cuda.synchronize()
    assert ticket.tolist() == [0] * _lib.INBATCH_TICKET_INTS and bool(torch.isfinite(loss))
    per_block = 64 if model == "transe" else 8          # pairs per scoring workgroup at D = 128 (4 / 32 lanes per pair)
    pair_blocks = -(-B * (K + 1) // per_block)
    partials_at = (B + 1) // 2 * 2
    regsh_at = partials_at + 4 * (max(pair_blocks, B) + (B + 3) // 4)   # (sized for the packed and for the row-per-workgroup slot mapping)
    off_at = (regsh_at + B + 1) // 2 * 2 + 6 * 64
    chunk, entries = 1024, 2 * B * K
    C = -(-entries // chunk)
    sorted_at = (off_at + C * (2 * B + 1) + 1) // 2 * 2          # int2 {entry, the row the other slot of its pair names}
    assert sorted_at + 2 * entries == n
    ints = save.view(torch.int32).cpu()
    off = ints[off_at:off_at + C * (2 * B + 1)].reshape(C, 2 * B + 1)
    got = ints[sorted_at:sorted_at + 2 * entries].reshape(entries, 2)
    flat = neg_idx.view(-1)
    for c in range(C):
        lo, hi = c * chunk, min((c + 1) * chunk, entries)
        vals = flat[lo:hi]
        order = torch.sort(vals, stable=True).indices + lo        # by row, entry order within a row
        assert torch.equal(got[lo:hi, 0].long(), order), (c, "entries")
        assert torch.equal(got[lo:hi, 1].long(), flat[order ^ 1]), (c, "partner rows")
        counts = torch.bincount(vals, minlength=2 * B)
        want_off = torch.cat((torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)))
        assert torch.equal(off[c].long(), want_off), (c, "offsets")
    assert launches_match(L, mid, B, K, D)


def launches_match(L, mid, B, K, D):
    """One forward launch iff at most 96 scoring + regulariser workgroups: with the slots packed, or -- when a row's K + 1 pairs
    fit one workgroup -- with a workgroup per row of the batch."""
    lanes = 4 if mid == 0 else 32
    reg = (B + 3) // 4
    packed = -(-B * (K + 1) // (256 // lanes)) + reg
    per_row = B + reg if (K + 1 <= 256 and (K + 1) * lanes <= 1024) else packed
    return L.blp_inbatch_loss_fwd_launches(mid, B, K, D, 1e-3) == (1 if min(packed, per_row) <= 96 else 2)


@pytest.mark.parametrize("model,D", [("distmult", 100), ("complex", 200), ("simple", 300), ("transe", 1000),
                                     ("distmult", 1100)])
@pytest.mark.default_routing
def test_inbatch_loss_any_width(ops, model, D):
    """The reference takes any `dim` (glove-bow 300, bert-bow 768, a user's 100 / 200): the fused loss does too --
    reduction widths off the 32-grid or past torch.sum's first cascade (>= 512) and rows wider than one 768-element
    sweep of the backward.  Same tolerances as the training-shape test."""
    from oracle import ref_port
    torch.manual_seed(D)
    B, K = 16, 8
    ent = torch.randn(B, 2, D) * (1.0 if model == "transe" else 0.4)
    rel = torch.randn(B, 1, D) * 0.3
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    e_ref, r_ref = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    ref = ref_port.compute_loss(model, "margin", e_ref, r_ref, neg_idx, 1e-2)
    ref.backward()
    e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
    loss = ops.inbatch_loss(model, "margin", e, r, neg_idx.cuda(), 1e-2)
    loss.backward()
    assert loss.item() == pytest.approx(ref.item(), rel=2e-6, abs=1e-7)
    np.testing.assert_allclose(e.grad.cpu().numpy(), e_ref.grad.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(r.grad.cpu().numpy(), r_ref.grad.numpy(), rtol=2e-5, atol=2e-7)
    # and the scores themselves are the reference's bit for bit at these widths
    pos = ops.score(model, e.detach()[:, 0], e.detach()[:, 1], r.detach()[:, 0]).cpu()
    want = ref_port.SCORE_FNS[model](ent[:, 0], ent[:, 1], rel[:, 0])
    assert torch.equal(pos, want)


@pytest.mark.parametrize("model", REL_MODELS)
@pytest.mark.parametrize("loss_fn", ["margin", "nll"])
@pytest.mark.parametrize("dtype,rel_f32", [(torch.float16, False), (torch.float16, True), (torch.bfloat16, False),
                                            (torch.bfloat16, True)])
@pytest.mark.default_routing
def test_inbatch_loss_half_storage(ops, model, loss_fn, dtype, rel_f32):
    """BASELINE config 5 shape per GPU (B = 128, K = 64, D = 128) with half-precision embeddings (what
    the encoder emits under autocast; relation rows optionally still f32).  The reference has no half
    path: the oracle is the f32 torch port on the SAME values widened to f32.  Operands are widened
    exactly and accumulated in f32, so the loss matches like the f32 path (rel 2e-6); gradients are f32
    values rounded once to the storage type: rtol 2^-10 (f16) / 2^-7 (bf16)."""
    from oracle import ref_port
    torch.manual_seed(5)
    B, K, D = 128, 64, 128
    ent = torch.randn(B, 2, D) * (1.0 if model == "transe" else 0.4)
    if model == "transe":
        ent = torch.nn.functional.normalize(ent, dim=-1)
    ent = ent.to(dtype)
    rel = torch.randn(B, 1, D) * 0.3
    rel = rel if rel_f32 else rel.to(dtype)
    neg_idx = torch.randint(0, 2 * B, (B, K, 2))
    reg = 1e-3 if model == "complex" else 0.0
    e_ref, r_ref = ent.float().clone().requires_grad_(True), rel.float().clone().requires_grad_(True)
    ref = ref_port.compute_loss(model, loss_fn, e_ref, r_ref, neg_idx, reg)
    ref.backward()
    e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
    loss = ops.inbatch_loss(model, loss_fn, e, r, neg_idx.cuda(), reg)
    loss.backward()
    assert loss.dtype == torch.float32 and e.grad.dtype == dtype and r.grad.dtype == rel.dtype
    assert loss.item() == pytest.approx(ref.item(), rel=2e-6, abs=1e-7)
    def tol(t):
        return {torch.float32: 2e-5, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[t]
    np.testing.assert_allclose(e.grad.float().cpu().numpy(), e_ref.grad.numpy(), rtol=tol(dtype), atol=1e-7)
    np.testing.assert_allclose(r.grad.float().cpu().numpy(), r_ref.grad.numpy(), rtol=tol(rel.dtype), atol=1e-7)


@pytest.mark.default_routing
def test_compute_loss_under_autocast(ops):
    """LinkPrediction.compute_loss with f16 embeddings and f32 relation rows (the autocast mix) takes the
    fused path and matches its own f32 evaluation of the widened inputs."""
    from blp_amd import models
    torch.manual_seed(9)
    B, K, D = 64, 64, 128
    model = models.LinkPrediction(D, "complex", "margin", 11, 1e-3).cuda()
    ent = (torch.randn(B, 2, D, device="cuda") * 0.4).half()
    rels = torch.randint(0, 11, (B, 1), device="cuda")
    neg_idx = torch.randint(0, 2 * B, (B, K, 2), device="cuda")
    with torch.autocast("cuda", dtype=torch.float16):
        half_loss = model.compute_loss(ent, rels, neg_idx)
    full_loss = model.compute_loss(ent.float(), rels, neg_idx)
    assert half_loss.dtype == torch.float32
    assert half_loss.item() == pytest.approx(full_loss.item(), rel=1e-6)


@pytest.mark.default_routing
@pytest.mark.parametrize("model", REL_MODELS)
def test_score_fn_training_broadcast_and_grad(ops, model):
    """score_fn on (B, K, D) x (B, 1, D) (models.py:67): golden-exact forward, analytic backward vs
    autograd of the torch port (rtol 1e-5)."""
    from oracle import ref_port
    g = golden("score_pairs")
    h, t, r = (torch.from_numpy(g[f"train_{k}_{model}"]) for k in ("h", "t", "r"))
    hd, td, rd = (x.cuda().requires_grad_(True) for x in (h, t, r))
    out = ops.score(model, hd, td, rd)
    assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), g[f"train_{model}"].view(np.uint32))
    w = torch.randn(out.shape)
    (out * w.cuda()).sum().backward()
    hc, tc, rc = (x.clone().requires_grad_(True) for x in (h, t, r))
    (ref_port.SCORE_FNS[model](hc, tc, rc) * w).sum().backward()
    # the (B, 1, D) operand's gradient is a sum over K of signed terms: reduction order differs from
    # torch-CPU autograd, so allow a few ulps of the summands (|w| ~ 1) in absolute terms
    for got, want in ((hd.grad, hc.grad), (td.grad, tc.grad), (rd.grad, rc.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.default_routing
def test_transe_wide_rows_score_fn(ops):
    """TransE on BOW widths (300 GloVe / 768 BERT embeddings): runtime-D path, golden-exact."""
    g = golden("score_pairs")
    for d in (300, 768):
        out = ops.score("transe", dev(g[f"h_{d}"]), dev(g[f"t_{d}"]), dev(g[f"r_{d}"]))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), g[f"transe_{d}"].view(np.uint32))


@pytest.mark.default_routing
def test_errors_are_loud(ops):
    table = torch.randn(8, 96).cuda()  # D = 96 is not a compiled ranking width (TransE alone is taken at any D % 4 == 0)
    q = torch.randn(2, 96).cuda()
    with pytest.raises(RuntimeError, match="UNSUPPORTED_DIM"):
        ops.rank_all("distmult", table, q, q, 1, true_row=torch.zeros(2, dtype=torch.int64).cuda())
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.rank_all("transe", torch.randn(8, 128), torch.randn(2, 128), torch.randn(2, 128), 1,
                     true_row=torch.zeros(2, dtype=torch.int64))
    with pytest.raises(KeyError):
        ops.score("rotate", table, table, table)


def test_two_threads_two_streams_concurrently(ops, oracle):
    """The nn.DataParallel calling pattern (train.py:329-330, 344: one Python thread per replica) on one device:
    two threads, each on its own stream, hammer blp_rank_all (pre-pass paths: workspace + pair lists + flags per
    call) and the fused loss forward / backward at the same time.  ctypes releases the GIL during the calls, so they
    really overlap.  Every iteration's counts must be the oracle's and the loss / gradients the single-threaded
    ones; an argument error raised in one thread must not leak into the other's blp_last_error."""
    import threading
    from blp_amd import _lib
    from oracle import ref_port
    problems = {}
    for i, model in enumerate(("transe", "distmult")):
        N, D, q_head, q_tail = 2000 + 37 * i, 128, 170, 150
        table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=50 + i)
        rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=60 + i)
        want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
        g = torch.Generator().manual_seed(70 + i)
        B, K = 32, 16
        ent = torch.randn(B, 2, D, generator=g) * 0.4
        rel = torch.randn(B, 1, D, generator=g) * 0.3
        neg_idx = torch.randint(0, 2 * B, (B, K, 2), generator=g)
        e_ref, r_ref = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
        ref = ref_port.compute_loss(model, "margin", e_ref, r_ref, neg_idx, 1e-3)
        ref.backward()
        problems[i] = dict(model=model, q_head=q_head, want=want, ref=ref.item(), grad=e_ref.grad.numpy(),
                           dev=[x.cuda() for x in (table, q_fixed, q_rel, true_row, torch.from_numpy(rowptr),
                                                   torch.from_numpy(col), ent, rel, neg_idx)])
    torch.cuda.synchronize()
    failures, start = [], threading.Barrier(2)

    def worker(i):
        try:
            p = problems[i]
            table, q_fixed, q_rel, true_row, rowptr, col, ent, rel, neg_idx = p["dev"]
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream):
                for it in range(25):
                    got = ops.rank_all(p["model"], table, q_fixed, q_rel, p["q_head"], true_row=true_row,
                                       filt_rowptr=rowptr, filt_col=col)
                    e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
                    loss = ops.inbatch_loss(p["model"], "margin", e, r, neg_idx, 1e-3)
                    loss.backward()
                    if it % 5 == i:  # an error in THIS thread only
                        rc = _lib.lib().blp_rank_all(9 + i, None, 0, 128, 128, None, None, None, None, 1, 1, None,
                                                     None, None, 0, 0, None)
                        assert rc == -1 and f"unknown model {9 + i}".encode() in _lib.lib().blp_last_error()
                    stream.synchronize()
                    if not np.array_equal(got.cpu().numpy(), p["want"]):
                        failures.append(f"thread {i} iteration {it}: counts differ")
                    if abs(loss.item() - p["ref"]) > 2e-6 * max(1.0, abs(p["ref"])):
                        failures.append(f"thread {i} iteration {it}: loss {loss.item()} vs {p['ref']}")
                    if not np.allclose(e.grad.cpu().numpy(), p["grad"], rtol=2e-5, atol=2e-7):
                        failures.append(f"thread {i} iteration {it}: gradients differ")
        except Exception as exc:  # noqa: BLE001 -- reported by the main thread
            failures.append(f"thread {i}: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not failures, failures[:5]
