"""Torch-facing operators over the C-ABI (libblp_hip.so).  PyTorch is plumbing here: it owns the
device memory and the stream; every arithmetic step of the hot path runs in the HIP kernels.

    rank_all(...)       all-entities ranking counts        (train.py:146-171, utils.py:103-105)
    rank_metrics(...)   counts -> reciprocal ranks, hits   (utils.py:104-109)
    score(...)          score_fn(heads, tails, rels)       (models.py:222-248), differentiable
    inbatch_loss(...)   compute_loss on in-batch negatives (models.py:51-70), differentiable

Tensors must live on a HIP device; there is no CPU implementation behind these functions (CPU
tensors are handled one level up, in blp_amd.models, with plain torch expressions).
"""
import collections
import ctypes

import torch

from . import _lib

_K_VALUES = (ctypes.c_int32 * 3)


def _stream(device):
    # (the raw hipStream_t of torch's current stream on `device`; the public torch.cuda.current_stream() builds a Stream
    # object around the same handle at ~2 us per call)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device.index))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _addr(t):
    """The device address as a plain int (what a c_void_p argument takes without building a ctypes object), None = NULL."""
    return None if t is None else t.data_ptr()


def _filter_spec(filter, Q, dev):
    """SegmentFilter -> byref(blp_filter) (the tensors stay referenced by the caller's ``filter`` for the duration of the call)."""
    seg = [None if x is None else (x if x.dtype is torch.int64 and x.device == dev and x.is_contiguous()
                                   else x.to(device=dev, dtype=torch.int64).contiguous())
           for x in (filter.seg_lo, filter.seg_hi, filter.values, filter.exclude, filter.ent2idx)]
    if seg[0].shape[0] != Q or seg[1].shape[0] != Q or (seg[3] is not None and seg[3].shape[0] != Q):
        raise ValueError("SegmentFilter: seg_lo, seg_hi and exclude need one entry per query")
    spec = _lib.BlpFilter(seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr() if seg[2].numel() else seg[0].data_ptr(),
                          None if seg[3] is None else seg[3].data_ptr(), None if seg[4] is None else seg[4].data_ptr(),
                          0 if seg[4] is None else seg[4].shape[0], int(filter.row_base))
    spec._keep = seg  # converted copies must outlive the (asynchronous) call's argument marshalling
    return ctypes.byref(spec)


# Ranking workspaces, one per (device, stream), grown on demand and kept: every kernel of a call runs on the caller's stream,
# so the next call on that stream is ordered behind it and may reuse the buffer (a torch.empty + a size query per call were
# ~4 us of a 20 us small call).  The sizes come from the library (blp_rank_all*_workspace_bytes), memoised per shape.
_workspaces = {}
_ws_bytes_memo = {}
# The cache is bounded two ways (round 4 kept the largest workspace per stream for the life of the process, so an evaluation's
# scratch -- hundreds of MB of operand images and pair lists -- stayed pinned through the training steps that followed it and a
# configuration that fitted before could run out of memory): (a) ranking.eval_link_prediction calls release_workspaces() when
# it is done, so nothing of an evaluation outlives it; (b) a workspace above WORKSPACE_CACHE_MAX_BYTES (the Wikidata5M-scale
# blocks: GBs) is never kept at all -- a per-call torch.empty that goes back to the caching allocator when the call returns
# (stream-ordered: the allocator hands the block to later work on the same stream only).  Everything below the cap is kept
# WITHIN an evaluation: allocating the FB15k-237 block's 250 MB per call cost the evaluation step 40 us and, when the allocator
# had split the block in between, a hipMalloc of ~9 ms now and then (profiles/r05/clustered_sweep.log, first version).
WORKSPACE_CACHE_MAX_BYTES = 1 << 30


def _workspace(dev, stream, nbytes):
    if nbytes > WORKSPACE_CACHE_MAX_BYTES:
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    key = (dev.index, stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _workspaces[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    return buf


def release_workspaces():
    """Drop the cached ranking workspaces (they hold the largest requirement seen per stream)."""
    _workspaces.clear()


def _rank_ws_bytes(L, model, N, D, q_head, q_tail):
    if L is not _lib._product:  # the hooks build (tests, tools): its routing -- hence the size -- moves with the knobs
        return L.blp_rank_all_workspace_bytes(model, N, D, q_head, q_tail)
    key = (model, N, D, q_head, q_tail)
    n = _ws_bytes_memo.get(key)
    if n is None:
        if len(_ws_bytes_memo) > 4096:
            _ws_bytes_memo.clear()
        n = _ws_bytes_memo[key] = L.blp_rank_all_workspace_bytes(model, N, D, q_head, q_tail)
    return n


def _i64_vector(t):
    """(Q,) int64, contiguous -- as given when it already is."""
    if t.dtype is torch.int64 and t.dim() == 1 and t.is_contiguous():
        return t
    return t.reshape(-1).to(torch.int64).contiguous()


def _f32_matrix(t, D, name):
    """(Q, D) float32, contiguous -- as given when it already is."""
    if t.dtype is torch.float32 and t.dim() == 2 and t.shape[1] == D and t.is_contiguous():
        return t
    return _f32_rows(t, name).reshape(-1, D).contiguous()


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("blp_amd.ops works on HIP device tensors only (got a CPU tensor); "
                               "there is no CPU fallback in the product path")


def _f32_rows(t, name):
    """float32, unit stride along the last dim (rows may be strided)."""
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        t = t.contiguous()
    return t


TABLE_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}  # BLP_DTYPE_* of include/blp_hip.h


def _table_rows(t, name):
    """A candidate table: float32, or the 16-bit copy the table build can emit next to it (float16 / bfloat16: rows start on
    16-byte boundaries); unit stride along the last dim."""
    if t.dtype not in TABLE_DTYPES:
        raise TypeError(f"{name} must be float32, float16 or bfloat16, got {t.dtype}")
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        t = t.contiguous()
    if t.dtype != torch.float32 and t.dim() == 2 and (t.stride(0) % 8 or t.data_ptr() % 16):
        # (a width that is not a multiple of 8 -- the bag-of-words 300 --: rows padded to the next multiple, a copy)
        wide = torch.zeros((t.shape[0], (t.shape[1] + 7) // 8 * 8), dtype=t.dtype, device=t.device)
        wide[:, :t.shape[1]] = t
        t = wide[:, :t.shape[1]]
    return t


# The filtered setting as segments of a sorted index of the filtering graph (include/blp_hip.h: blp_filter;
# built by blp_amd.utils.FilterIndex.segments): seg_lo, seg_hi (Q,) int64 slices of `values` (entity ids),
# exclude (Q,) the triple's own entity, ent2idx the id -> table row map (None: values are rows), row_base the
# first global row of a candidate shard.  Everything stays on the device; nothing is materialised per batch.
SegmentFilter = collections.namedtuple("SegmentFilter", "seg_lo seg_hi values exclude ent2idx row_base")


_selftested = set()


def ensure_selftest(device, rel_model=None):
    """The matrix-pipe self-test of ``device`` (include/blp_hip.h: blp_selftest), once per process and device, from HERE -- a
    set-up step that waits for the stream -- so that no ranking call ever synchronises inside the C-ABI (which it would, under a
    process-wide mutex, if the first bilinear block of a device found the verdict open).  TransE never needs it."""
    if rel_model == "transe":
        return
    index = device.index if isinstance(device, torch.device) else int(device)
    if index not in _selftested:
        if torch.cuda.is_current_stream_capturing():  # (the test has to be waited for: the library then takes its provable pre-pass)
            return
        _lib.selftest(index, torch._C._cuda_getCurrentRawStream(index))
        _selftested.add(index)


def dim_supported(rel_model, dim):
    """True if the fused ranking kernels are compiled for this embedding width."""
    return bool(_lib.lib().blp_dim_supported(_lib.MODEL_IDS[rel_model], int(dim)))


def rank_all_supported(rel_model, dim, q_head, q_tail):
    """True if rank_all takes a block of this shape (fused kernels at 64 / 128 / 256, or the any-width
    TransE pre-pass at any D % 4 == 0 up to 1024); otherwise use score + rank_from_scores."""
    return bool(_lib.lib().blp_rank_all_supported(_lib.MODEL_IDS[rel_model], int(dim), int(q_head), int(q_tail)))


def rank_all(rel_model, table, q_fixed, q_rel, q_head, true_row=None, q_true=None,
             filt_rowptr=None, filt_col=None, out=None, rel_ids=None, filter=None, workspace=None):
    """Rank-count every query against every row of ``table`` in one pass over the table.

    table (N, D) f32; q_fixed, q_rel (Q, D) f32; queries [0, q_head) replace the head (q_fixed is the
    tail embedding), queries [q_head, Q) replace the tail (q_fixed is the head embedding).
    true_row (Q,) int64 rows of the true entities in ``table``  -- or -- q_true (Q, D) their vectors
    (sharded ranking).  filt_rowptr (Q+1,), filt_col (nnz,) int64: CSR of filtered rows per query -- or --
    filter: a SegmentFilter (slices of a FilterIndex on the device; no per-batch list).
    rel_ids: accepted for compatibility and ignored (it was an optimisation hint of the C-ABI before 6.0.0).
    workspace: optional uint8 device tensor of >= rank_all_workspace_bytes(...) bytes to use as the call's scratch (a caller
    that wants to look at it afterwards: prepass_stats); default: this module's own.
    Returns counts (Q, 4) int32: gt, ge, gt_filtered, ge_filtered.
    """
    _require_device(table, q_fixed, q_rel, true_row, q_true, filt_rowptr, filt_col, rel_ids)
    table = _f32_rows(table, "table")
    if table.dim() != 2:
        raise ValueError(f"table must be (N, D), got {tuple(table.shape)}")
    N, D = table.shape
    q_fixed = _f32_matrix(q_fixed, D, "q_fixed")
    q_rel = _f32_matrix(q_rel, D, "q_rel")
    Q = q_fixed.shape[0]
    if q_rel.shape[0] != Q:
        raise ValueError("q_fixed and q_rel must have the same number of rows")
    if not 0 <= q_head <= Q:
        raise ValueError(f"q_head = {q_head} outside [0, {Q}]")
    if (true_row is None) == (q_true is None):
        raise ValueError("give exactly one of true_row / q_true")
    if true_row is not None:
        true_row = _i64_vector(true_row)
        if true_row.shape[0] != Q:
            raise ValueError("true_row must have one entry per query")
    else:
        q_true = _f32_matrix(q_true, D, "q_true")
        if q_true.shape[0] != Q:
            raise ValueError("q_true must have one row per query")
    if rel_ids is not None:
        rel_ids = _i64_vector(rel_ids)
        if rel_ids.shape[0] != Q:
            raise ValueError("rel_ids must have one entry per query")
    if (filt_rowptr is None) != (filt_col is None):
        raise ValueError("filt_rowptr and filt_col go together")
    if filter is not None and filt_rowptr is not None:
        raise ValueError("give the filter as a CSR or as a SegmentFilter, not both")
    if filt_rowptr is not None:
        filt_rowptr = filt_rowptr.to(torch.int64).contiguous()
        filt_col = filt_col.to(torch.int64).contiguous()
        if filt_rowptr.shape[0] != Q + 1:
            raise ValueError("filt_rowptr must have Q + 1 entries")
        if filt_col.numel() == 0:  # empty CSR: keep a non-NULL pointer for the C-ABI
            filt_col = torch.zeros(1, dtype=torch.int64, device=filt_col.device)
    dev = table.device
    counts = out if out is not None else torch.empty((Q, 4), dtype=torch.int32, device=dev)
    if counts.shape != (Q, 4) or counts.dtype != torch.int32 or not counts.is_contiguous():
        raise ValueError("out must be a contiguous (Q, 4) int32 tensor")
    if Q == 0:
        return counts
    ensure_selftest(dev, rel_model)
    L = _lib.lib()
    model = _lib.MODEL_IDS[rel_model]
    ws_bytes = _rank_ws_bytes(L, model, N, D, q_head, Q - q_head)
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    if workspace is None:
        workspace = _workspace(dev, stream, ws_bytes)
    elif not workspace.is_cuda or workspace.dtype != torch.uint8 or workspace.numel() < ws_bytes or workspace.data_ptr() % 256:
        raise ValueError(f"workspace must be a 256-byte aligned uint8 device tensor of >= {ws_bytes} bytes")
    if filter is not None:
        spec = _filter_spec(filter, Q, dev)
    elif filt_rowptr is not None:  # a CSR is the segment form with seg_lo = rowptr, seg_hi = rowptr + 1, values = col
        csr = _lib.BlpFilter(filt_rowptr.data_ptr(), filt_rowptr.data_ptr() + 8, filt_col.data_ptr(), None, None, 0, 0)
        spec = ctypes.byref(csr)
    else:
        spec = None
    status = L.blp_rank_all(model, table.data_ptr(), N, D, table.stride(0) if N > 0 else D, q_fixed.data_ptr(), q_rel.data_ptr(),
                            _addr(true_row), _addr(q_true), q_head, Q - q_head, spec, counts.data_ptr(), workspace.data_ptr(),
                            ws_bytes, dev.index, stream)
    if status:
        _lib.check(status, "blp_rank_all")
    return counts


def table16_is_read_directly(rel_model, dtype, N, D, n_triples, batch, block_triples=None, ld=None):
    """include/blp_hip.h: blp_rank_all_batches_native16 -- True if rank_all_batches over a float16 / bfloat16 table of these sizes
    streams the 16-bit rows themselves (half the bytes per pass); False if the library would widen the table to f32 first.
    ``ld``: the row stride (elements) the call will be made with; default = what rank_all_batches passes for a contiguous
    ``table.to(dtype)`` copy (D, or D padded to a multiple of 8 by _table_rows when D % 8 != 0)."""
    if dtype not in TABLE_DTYPES or dtype == torch.float32:
        return False
    if ld is None:
        ld = (D + 7) // 8 * 8
    return bool(_lib.lib().blp_rank_all_batches_native16(_lib.MODEL_IDS[rel_model], TABLE_DTYPES[dtype], int(N), int(D), ld, int(n_triples),
                                                         int(batch), int(batch if block_triples is None else block_triples)))


def rank_all_workspace_bytes(rel_model, N, D, q_head, q_tail):
    return int(_lib.lib().blp_rank_all_workspace_bytes(_lib.MODEL_IDS[rel_model], int(N), int(D), int(q_head), int(q_tail)))


PREPASS_PATHS = {0: "exact kernels (no pre-pass)", 1: "v_sad_u16 pre-pass", 2: "bf16 x 3 MFMA pre-pass", 3: "f32-chain MFMA pre-pass",
                 4: "any-width v_sad_u16 pre-pass"}


def prepass_stats(rel_model, N, D, q_head, q_tail, workspace):
    """include/blp_hip.h: blp_rank_all_prepass_stats -- what the pre-pass of the last rank_all(..., workspace=workspace) of this
    shape left to the exact path.  Waits for the stream.  {"pairs", "listed", "flagged_rows", "path", "decided_frac"};
    decided_frac is None on the paths that are not counted."""
    out = (ctypes.c_int64 * 4)()
    dev = workspace.device
    L = _lib.lib()
    _lib.check(L.blp_rank_all_prepass_stats(_lib.MODEL_IDS[rel_model], int(N), int(D), int(q_head), int(q_tail), workspace.data_ptr(),
                                            workspace.numel(), out, dev.index, torch._C._cuda_getCurrentRawStream(dev.index)),
               "blp_rank_all_prepass_stats")
    pairs, listed, flagged, path = (int(x) for x in out)
    undecided = None if listed < 0 else listed + flagged
    return {"pairs": pairs, "listed": listed, "flagged_rows": flagged, "path": PREPASS_PATHS[path],
            "decided_frac": None if undecided is None or pairs == 0 else 1.0 - undecided / pairs}


def rank_all_idx(rel_model, table, fixed_row, rel_emb, rel_ids, q_head, true_row, filter=None, out=None):
    """rank_all with the queries left as INDICES (blp_rank_all_shard with source == table): query q's fixed-entity vector is row
    ``fixed_row[q]`` of ``table``, its relation vector row ``rel_ids[q]`` of ``rel_emb`` (R, D) -- what the reference
    gathers into ``ent_emb[tails]`` / ``rel_emb(rels)`` (train.py:141-145), un-gathered: no (Q, D) arrays are built or
    streamed.  All indices must be in range (build_queries clamps and flags bad ids).  Same counts as rank_all."""
    return rank_all_shard(rel_model, table, table, fixed_row, rel_emb, rel_ids, q_head, true_row, filter=filter, out=out)


def rank_all_shard(rel_model, table, source, fixed_row, rel_emb, rel_ids, q_head, true_row, filter=None, out=None):
    """One shard of the candidate axis (blp_rank_all_shard): ``table`` (N, D) holds this rank's rows of the entity table --
    the candidates --, ``source`` (S, D) the vectors the queries themselves are made of (the whole table, or the vectors
    of the entities in the triples: gather_triple_vectors), replicated on every rank; fixed_row / true_row (Q,) index
    ``source``, rel_ids (Q,) index ``rel_emb`` (R, D).  A SegmentFilter's row_base = this shard's first global row.
    Counts (Q, 4) int32 of this shard; the shards' counts add up to the unsharded ones.  ``source is table``: the
    unsharded index form (rank_all_idx)."""
    _require_device(table, source, fixed_row, rel_emb, rel_ids, true_row)
    table = _f32_rows(table, "table")
    source = table if source is table else _f32_rows(source, "source")
    rel_emb = _f32_rows(rel_emb, "rel_emb").contiguous()
    N, D = table.shape
    if source.dim() != 2 or source.shape[1] != D:
        raise ValueError(f"source must be (S, {D}), got {tuple(source.shape)}")
    fixed_row, rel_ids, true_row = _i64_vector(fixed_row), _i64_vector(rel_ids), _i64_vector(true_row)
    Q = fixed_row.shape[0]
    if rel_ids.shape[0] != Q or true_row.shape[0] != Q or rel_emb.shape[1] != D:
        raise ValueError("fixed_row, rel_ids and true_row need one entry per query; rel_emb must be (R, D)")
    if not 0 <= q_head <= Q:
        raise ValueError(f"q_head = {q_head} outside [0, {Q}]")
    dev = table.device
    counts = out if out is not None else torch.empty((Q, 4), dtype=torch.int32, device=dev)
    if counts.shape != (Q, 4) or counts.dtype != torch.int32 or not counts.is_contiguous():
        raise ValueError("out must be a contiguous (Q, 4) int32 tensor")
    if Q == 0:
        return counts
    ensure_selftest(dev, rel_model)
    L = _lib.lib()
    model = _lib.MODEL_IDS[rel_model]
    ws_bytes = _rank_ws_bytes(L, model, N, D, q_head, Q - q_head)
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    workspace = _workspace(dev, stream, ws_bytes)
    spec = None if filter is None else _filter_spec(filter, Q, dev)
    status = L.blp_rank_all_shard(model, table.data_ptr(), N, D, table.stride(0) if N > 1 else D, source.data_ptr(), source.shape[0],
                                  source.stride(0) if source.shape[0] > 1 else D, fixed_row.data_ptr(), rel_emb.data_ptr(),
                                  rel_emb.shape[0], rel_ids.data_ptr(), true_row.data_ptr(), q_head, Q - q_head, spec, counts.data_ptr(),
                                  workspace.data_ptr(), ws_bytes, dev.index, stream)
    if status:
        _lib.check(status, "blp_rank_all_shard")
    return counts


def rank_all_batches(rel_model, table, fixed_row, rel_emb, rel_ids, true_row, num_triples, batch, filter=None, out=None, source=None,
                     block_triples=0):
    """Every batch of the reference's evaluation loop in ONE call (blp_rank_all_batches): fixed_row / rel_ids / true_row
    (2 num_triples,), a SegmentFilter's seg_lo / seg_hi / exclude and the returned counts (2 num_triples, 4) are laid out
    batch after batch of ``batch`` triples, each batch as [its head queries | its tail queries] -- build_queries'
    layout with block_size = batch, i.e. what a loop over the reference's DataLoader sees.  Same counts as one
    rank_all_idx call per batch; the library ranks ``block_triples`` triples at a time (0: 65 536; <= batch: one pass per
    batch, the reference's own pass structure, issued back to back by the library)."""
    source = table if source is None else source
    _require_device(table, source, fixed_row, rel_emb, rel_ids, true_row)
    # (a 16-bit table -- float16 / bfloat16 --: the candidates only; `source`, the queries' own
    #  vectors, is float32: gather_triple_vectors widens them)
    table, rel_emb = _table_rows(table, "table"), _f32_rows(rel_emb, "rel_emb").contiguous()
    if source is table and table.dtype != torch.float32:
        raise TypeError("a 16-bit table needs a float32 `source` for the queries' vectors (ops.gather_triple_vectors + build_queries(by_position=True))")
    source = table if source is table else _f32_rows(source, "source")
    N, D = table.shape
    tdt = TABLE_DTYPES[table.dtype]
    ld = table.stride(0) if tdt or N > 1 else D
    n, Q = int(num_triples), 2 * int(num_triples)
    fixed_row, rel_ids, true_row = _i64_vector(fixed_row), _i64_vector(rel_ids), _i64_vector(true_row)
    if fixed_row.shape[0] != Q or rel_ids.shape[0] != Q or true_row.shape[0] != Q or rel_emb.shape[1] != D or source.shape[1] != D:
        raise ValueError("fixed_row, rel_ids, true_row need 2 * num_triples entries; rel_emb / source must be (., D)")
    dev = table.device
    counts = out if out is not None else torch.empty((Q, 4), dtype=torch.int32, device=dev)
    if counts.shape != (Q, 4) or counts.dtype != torch.int32 or not counts.is_contiguous():
        raise ValueError("out must be a contiguous (2 * num_triples, 4) int32 tensor")
    if Q == 0:
        return counts
    ensure_selftest(dev, rel_model)
    L = _lib.lib()
    model = _lib.MODEL_IDS[rel_model]
    ws_bytes = L.blp_rank_all_batches_workspace_bytes(model, tdt, N, D, ld, n, int(batch), int(block_triples))
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    workspace = _workspace(dev, stream, ws_bytes)
    spec = None if filter is None else _filter_spec(filter, Q, dev)
    status = L.blp_rank_all_batches(model, table.data_ptr(), tdt, N, D, ld, source.data_ptr(), source.shape[0],
                                      source.stride(0) if source.shape[0] > 1 else D, fixed_row.data_ptr(), rel_emb.data_ptr(), rel_emb.shape[0],
                                      rel_ids.data_ptr(), true_row.data_ptr(), n, int(batch), int(block_triples), spec, counts.data_ptr(),
                                      workspace.data_ptr(), ws_bytes, dev.index, stream)
    if status:
        _lib.check(status, "blp_rank_all_batches")
    return counts


def gather_triple_vectors(triples, ent2idx, table, row_base=0):
    """(2n, D) f32: row t = the vector of triple t's head, row n + t = of its tail (train.py:141-142 for the whole set),
    for the entities whose global row ent2idx[id] lies in [row_base, row_base + len(table)); zeros for the others, so
    that ONE all-reduce over the ranks of a candidate-axis shard replicates every vector (blp_gather_triple_vectors)."""
    _require_device(triples, ent2idx, table)
    table = _table_rows(table, "table")  # (float16 / bfloat16 rows come out widened, exactly: the result is float32 always)
    triples = triples.to(torch.int64).contiguous()
    n, (N, D) = triples.shape[0], table.shape
    tdt = TABLE_DTYPES[table.dtype]
    out = torch.empty((2 * n, D), dtype=torch.float32, device=table.device)
    if ent2idx is not None:
        ent2idx = ent2idx.to(torch.int64).contiguous()
    ld = table.stride(0) if tdt or N > 1 else D
    status = _lib.lib().blp_gather_triple_vectors(_ptr(triples), n, _ptr(ent2idx), 0 if ent2idx is None else ent2idx.shape[0],
                                                    _ptr(table), tdt, N, D, ld, int(row_base), _ptr(out),
                                                    table.device.index, _stream(table.device))
    _lib.check(status, "blp_gather_triple_vectors")
    return out


def rank_from_scores(scores, true_idx=None, true_score=None, filt_rowptr=None, filt_col=None):
    """Counts (Q, 4) int32 from a dense (Q, N) score matrix on the device (utils.py:103-105 +
    train.py:159-167): the true entity as column index ``true_idx`` (Q,) or as score ``true_score`` (Q,)."""
    _require_device(scores, true_idx, true_score, filt_rowptr, filt_col)
    if scores.dtype != torch.float32 or scores.dim() != 2:
        raise TypeError("scores must be a 2-D float32 tensor")
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    if (true_idx is None) == (true_score is None):
        raise ValueError("give exactly one of true_idx / true_score")
    Q, N = scores.shape
    dev = scores.device
    if true_idx is not None:
        true_idx = true_idx.reshape(-1).to(torch.int64).contiguous()
    else:
        true_score = true_score.reshape(-1).to(torch.float32).contiguous()
    if filt_rowptr is not None:
        filt_rowptr = filt_rowptr.to(torch.int64).contiguous()
        filt_col = filt_col.to(torch.int64).contiguous()
        if filt_col.numel() == 0:
            filt_col = torch.zeros(1, dtype=torch.int64, device=dev)
    counts = torch.empty((Q, 4), dtype=torch.int32, device=dev)
    status = _lib.lib().blp_rank_from_scores(_ptr(scores), Q, N, scores.stride(0) if Q else N, _ptr(true_idx),
                                             _ptr(true_score), _ptr(filt_rowptr), _ptr(filt_col), _ptr(counts),
                                             dev.index, _stream(dev))
    _lib.check(status, "blp_rank_from_scores")
    return counts


def rank_metrics(counts, k_values=(1, 3, 10)):
    """counts (Q, 4) int32 -> rr (Q, 2) f32 [raw, filtered], hits (Q, 2, 3) bool."""
    _require_device(counts)
    Q = counts.shape[0]
    dev = counts.device
    rr = torch.empty((Q, 2), dtype=torch.float32, device=dev)
    hits = torch.empty((Q, 2, 3), dtype=torch.uint8, device=dev)
    if Q:
        status = _lib.lib().blp_rank_metrics(_ptr(counts.contiguous()), Q, _K_VALUES(*k_values), _ptr(rr),
                                             _ptr(hits), dev.index, _stream(dev))
        _lib.check(status, "blp_rank_metrics")
    return rr, hits.bool()


def rank_metric_sums(counts, k_values=(1, 3, 10)):
    """counts (Q, 4) int32 -> (8,) f64: [sum rr raw, sum rr filtered, hits@k raw x 3, hits@k filtered x 3]
    (train.py:152-157 accumulated on the device, fixed summation order)."""
    _require_device(counts)
    sums = torch.empty(_lib.METRIC_SUMS_DOUBLES, dtype=torch.float64, device=counts.device)  # 8 results + scratch
    status = _lib.lib().blp_rank_metric_sums(_ptr(counts.contiguous()), counts.shape[0], _K_VALUES(*k_values),
                                             _ptr(sums), counts.device.index, _stream(counts.device))
    _lib.check(status, "blp_rank_metric_sums")
    return sums[:8]


# ------------------------------------------------------------------------------- evaluation prelude
QueryBlock = collections.namedtuple("QueryBlock", "q_fixed q_rel fixed_row true_row rel_ids ids_min filter")


def build_queries(triples, ent2idx, table, rel_emb, block_size, index=None, row_base=0, gather=True, by_position=False,
                  num_rows=None):
    """train.py:132-145 (+ utils.py:46-83 with ``index``) for a whole set of triples in one kernel (blp_build_queries).

    triples (n, 3) int64 rows (head id, tail id, relation id) on the table's device; ent2idx: id -> table row (-1: not a
    candidate) or None (ids are rows); table (N, D) f32; rel_emb (R, D) f32 (model.rel_emb.weight); queries come out
    block after block of ``block_size`` triples, each block as [head-replacing | tail-replacing] queries.
    index: a utils.FilterIndex -> ``filter`` is the SegmentFilter of all 2n queries (slice it per block).
    gather=False: no vectors are built (q_fixed = q_rel = None); rank_all_idx takes fixed_row / rel_ids instead.
    by_position: ``table`` is the (2n, D) array of gather_triple_vectors and fixed_row / true_row come out as positions in
    it (head of triple t = t, tail = n + t); ``num_rows`` = rows of the whole entity table (the id check's bound).
    Returns QueryBlock(q_fixed (2n, D), q_rel (2n, D), fixed_row (2n,), true_row (2n,), rel_ids (2n,), ids_min (0-dim
    int32: -1 if any id had no row -- the reference's assertion train.py:137-138, left on the device), filter or
    None)."""
    _require_device(triples, ent2idx, table, rel_emb)
    dev = table.device
    table = _f32_rows(table, "table")
    rel_emb = _f32_rows(rel_emb, "rel_emb").contiguous()
    triples = triples.to(torch.int64).contiguous()
    n, D = triples.shape[0], table.shape[1]
    q_fixed = torch.empty((2 * n, D), dtype=torch.float32, device=dev) if gather else None
    q_rel = torch.empty((2 * n, D), dtype=torch.float32, device=dev) if gather else None
    fixed_row = torch.empty(2 * n, dtype=torch.int64, device=dev)
    true_row = torch.empty(2 * n, dtype=torch.int64, device=dev)
    rel_ids = torch.empty(2 * n, dtype=torch.int64, device=dev)
    ids_min = torch.empty((), dtype=torch.int32, device=dev)
    a = _lib.BlpQueries()
    a.triples, a.n, a.block = triples.data_ptr(), n, int(block_size)
    if ent2idx is not None:
        ent2idx = ent2idx.to(torch.int64).contiguous()
        a.ent2idx, a.ent2idx_len = ent2idx.data_ptr(), ent2idx.shape[0]
    a.source, a.src_rows, a.ld, a.D = table.data_ptr(), table.shape[0], table.stride(0) if table.shape[0] > 1 else D, D
    if by_position:
        if table.shape[0] != 2 * n or num_rows is None:
            raise ValueError("by_position: the source is the (2n, D) array of gather_triple_vectors; give num_rows")
        a.by_position, a.src_rows = 1, int(num_rows)
    a.rel_emb, a.R = rel_emb.data_ptr(), rel_emb.shape[0]
    if gather:
        a.q_fixed, a.q_rel = q_fixed.data_ptr(), q_rel.data_ptr()
    a.fixed_row, a.true_row, a.rel_ids, a.ids_min = fixed_row.data_ptr(), true_row.data_ptr(), rel_ids.data_ptr(), ids_min.data_ptr()
    filt = None
    if index is not None:
        heads_key, tails_key, values, n_head_vals = index.device_arrays(dev)
        seg_lo = torch.empty(2 * n, dtype=torch.int64, device=dev)
        seg_hi = torch.empty_like(seg_lo)
        exclude = torch.empty_like(seg_lo)
        a.heads_key, a.n_heads, a.tails_key, a.n_tails, a.index_R = (heads_key.data_ptr(), heads_key.shape[0],
                                                                     tails_key.data_ptr(), tails_key.shape[0], index.R)
        a.seg_lo, a.seg_hi, a.exclude = seg_lo.data_ptr(), seg_hi.data_ptr(), exclude.data_ptr()
        assert n_head_vals == heads_key.shape[0]
        filt = SegmentFilter(seg_lo, seg_hi, values, exclude, ent2idx, row_base)
    status = _lib.lib().blp_build_queries(ctypes.byref(a), dev.index, _stream(dev))
    _lib.check(status, "blp_build_queries")
    return QueryBlock(q_fixed, q_rel, fixed_row, true_row, rel_ids, ids_min, filt)


# ------------------------------------------------------------------------------- table build epilogue
def project_rows_supported(hidden_size, dim):
    """True if blp_project_rows takes an (n, hidden_size) -> (n, dim) projection."""
    return bool(_lib.lib().blp_project_rows_supported(int(hidden_size), int(dim)))


def project_rows(x, weight, out, normalize):
    """out[i] = x[i] @ weight.T, L2-normalised if ``normalize`` -- enc_linear (models.py:110-111) + F.normalize
    (models.py:40-41) + the row assignment into the entity table (train.py:109-113) in one kernel.  ``x`` (n, E) f32
    may be a strided view (the [CLS] rows of an encoder output), ``out`` (n, D) f32 rows of the table (shard).
    Inference only (no autograd)."""
    _require_device(x, weight, out)
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or out.dtype != torch.float32:
        raise TypeError("project_rows works on float32 tensors")
    if x.dim() != 2 or weight.dim() != 2 or out.dim() != 2 or x.shape[1] != weight.shape[1] or \
            out.shape != (x.shape[0], weight.shape[0]):
        raise ValueError(f"shapes do not match: x {tuple(x.shape)}, weight {tuple(weight.shape)}, out {tuple(out.shape)}")
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    if out.stride(1) != 1:
        raise ValueError("out must have contiguous rows")
    weight = weight.contiguous()
    status = _lib.lib().blp_project_rows(_ptr(x), x.shape[0], x.stride(0) if x.shape[0] > 1 else x.shape[1], _ptr(weight),
                                         x.shape[1], weight.shape[0], int(bool(normalize)), _ptr(out),
                                         out.stride(0) if out.shape[0] > 1 else out.shape[1], out.device.index,
                                         _stream(out.device))
    _lib.check(status, "blp_project_rows")
    return out


def bow_rows_supported(embedding_dim):
    """True if blp_bow_rows takes word vectors of this width."""
    return bool(_lib.lib().blp_bow_rows_supported(int(embedding_dim)))


def bow_rows(text_tok, text_mask, weight, out, normalize, bad_flag=None):
    """out[i] = sum_l mask[i, l] * weight[tok[i, l]] / sum_l mask[i, l], L2-normalised if ``normalize`` -- the bag-of-words
    encoder (models.py:143-155) + F.normalize (models.py:40-41) + the row assignment into the entity table
    (train.py:109-113) in one kernel that reads every gathered word vector once (blp_bow_rows).  text_tok (n, L) int64,
    text_mask (n, L) float or None (all ones), weight (V, E) f32 (embeddings.weight), out (n, E) f32 rows of the table.
    Inference only (no autograd).  Token ids outside [0, V) raise IndexError, as nn.Embedding does -- checked after the
    launch with one host read of a flag, or, with ``bad_flag`` (a 0-dim int32 device tensor the caller zeroed once), left to
    the caller: the kernel sets it to -1 and nothing synchronises here (models.BOW.check_tokens reads it after a table build)."""
    _require_device(text_tok, text_mask, weight, out)
    if weight.dtype != torch.float32 or out.dtype != torch.float32:
        raise TypeError("bow_rows works on float32 tables")
    if text_tok.dim() != 2 or weight.dim() != 2 or out.dim() != 2 or out.shape != (text_tok.shape[0], weight.shape[1]):
        raise ValueError(f"shapes do not match: tok {tuple(text_tok.shape)}, weight {tuple(weight.shape)}, out {tuple(out.shape)}")
    if out.stride(1) != 1:
        raise ValueError("out must have contiguous rows")
    tok = text_tok if text_tok.dtype is torch.int64 and text_tok.is_contiguous() else text_tok.to(torch.int64).contiguous()
    mask = None
    if text_mask is not None:
        if text_mask.shape != text_tok.shape:
            raise ValueError("text_mask must have text_tok's shape")
        mask = text_mask if text_mask.dtype is torch.float32 and text_mask.is_contiguous() else text_mask.to(torch.float32).contiguous()
    weight = weight.contiguous()
    n, L = tok.shape
    dev = out.device
    bad = bad_flag if bad_flag is not None else torch.zeros((), dtype=torch.int32, device=dev)
    status = _lib.lib().blp_bow_rows(tok.data_ptr(), _addr(mask), n, L, weight.data_ptr(), weight.shape[0], weight.shape[1],
                                     int(bool(normalize)), out.data_ptr(), out.stride(0) if n > 1 else weight.shape[1],
                                     bad.data_ptr(), dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
    _lib.check(status, "blp_bow_rows")
    if bad_flag is None and bad.item() < 0:
        raise IndexError("bow_rows: a token id is outside the embedding table")
    return out


def dkrl_rows_supported(embedding_dim, dim, num_tokens):
    """True if blp_dkrl_rows takes word vectors of this width, this output width and descriptions of this many tokens."""
    return bool(_lib.lib().blp_dkrl_rows_supported(int(embedding_dim), int(dim), int(num_tokens)))


def dkrl_rows(text_tok, text_mask, weight, conv1, conv2, out, normalize, bad_flag=None):
    """The DKRL encoder (models.py:158-204: lookup, Conv1d(E, dim, 2), mask, max-pool 4, tanh, Conv1d(dim, dim, 2), masked
    mean, tanh) + F.normalize (models.py:40-41) + the row assignment into the entity table (train.py:109-113) in one kernel
    (blp_dkrl_rows).  text_tok (n, L) int64, text_mask (n, L) float or None, weight (V, E) f32 (embeddings.weight), conv1 /
    conv2 the two nn.Conv1d modules (f32), out (n, dim) f32 rows of the table.  Inference only.  bad_flag as for bow_rows."""
    _require_device(text_tok, text_mask, weight, out)
    w1, b1, w2, b2 = conv1.weight, conv1.bias, conv2.weight, conv2.bias
    if any(t.dtype != torch.float32 for t in (weight, w1, b1, w2, b2, out)):
        raise TypeError("dkrl_rows works on float32 tables and weights")
    n, L = text_tok.shape
    dim, E = w1.shape[0], weight.shape[1]
    if w1.shape != (dim, E, 2) or w2.shape != (dim, dim, 2) or out.shape != (n, dim) or out.stride(1) != 1:
        raise ValueError(f"shapes do not match: conv1 {tuple(w1.shape)}, conv2 {tuple(w2.shape)}, weight {tuple(weight.shape)}, out {tuple(out.shape)}")
    tok = text_tok if text_tok.dtype is torch.int64 and text_tok.is_contiguous() else text_tok.to(torch.int64).contiguous()
    mask = None
    if text_mask is not None:
        if text_mask.shape != text_tok.shape:
            raise ValueError("text_mask must have text_tok's shape")
        mask = text_mask if text_mask.dtype is torch.float32 and text_mask.is_contiguous() else text_mask.to(torch.float32).contiguous()
    weight, w1, b1, w2, b2 = weight.contiguous(), w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
    dev = out.device
    bad = bad_flag if bad_flag is not None else torch.zeros((), dtype=torch.int32, device=dev)
    status = _lib.lib().blp_dkrl_rows(tok.data_ptr(), _addr(mask), n, L, weight.data_ptr(), weight.shape[0], E, w1.data_ptr(),
                                      b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), dim, int(bool(normalize)), out.data_ptr(),
                                      out.stride(0) if n > 1 else dim, bad.data_ptr(), dev.index,
                                      torch._C._cuda_getCurrentRawStream(dev.index))
    _lib.check(status, "blp_dkrl_rows")
    if bad_flag is None and bad.item() < 0:
        raise IndexError("dkrl_rows: a token id is outside the embedding table")
    return out


# ------------------------------------------------------------------------------------- score_fn
def _collapse(sizes, strides):
    """Collapse a group of dims into (size, stride) or None if not expressible with one stride."""
    size, stride = 1, 0
    for n, st in zip(sizes, strides):
        if n == 1:
            continue
        if size == 1:
            size, stride = n, st
        elif stride == st * n:   # outer stride == inner stride * inner size
            size, stride = size * n, st
        else:
            return None
    return size, stride


def _two_level(shape, operands):
    """Find a split of the broadcast leading shape into (M0, M1) such that every operand's rows are
    base + i0*s0 + i1*s1.  Returns (M0, M1, [(tensor, s0, s1), ...]); copies an operand only if
    its layout cannot be expressed that way."""
    nd = len(shape)
    expanded = [op.expand(*shape, op.shape[-1]) for op in operands]
    for split in range(nd + 1):
        plan = []
        for op in expanded:
            outer = _collapse(shape[:split], op.stride()[:split])
            inner = _collapse(shape[split:], op.stride()[split:nd])
            if outer is None or inner is None:
                plan = None
                break
            plan.append((op, outer[1], inner[1]))
        if plan is not None:
            m0 = 1
            for n in shape[:split]:
                m0 *= n
            m1 = 1
            for n in shape[split:]:
                m1 *= n
            return m0, m1, plan
    total = 1
    for n in shape:
        total *= n
    D = operands[0].shape[-1]
    return total, 1, [(op.contiguous(), D, 0) for op in expanded]


class _ScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rel_model, heads, tails, rels):
        _require_device(heads, tails, rels)
        D = heads.shape[-1]
        if tails.shape[-1] != D or rels.shape[-1] != D:
            raise ValueError("heads, tails, rels must share the last dimension")
        ops = [_f32_rows(x, n) for x, n in ((heads, "heads"), (tails, "tails"), (rels, "rels"))]
        shape = torch.broadcast_shapes(*(x.shape[:-1] for x in ops))
        m0, m1, plan = _two_level(tuple(shape), ops)
        dev = heads.device
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        (h, hs0, hs1), (t, ts0, ts1), (r, rs0, rs1) = plan
        status = _lib.lib().blp_score_fwd(_lib.MODEL_IDS[rel_model], D, m0, m1, _ptr(h), hs0, hs1,
                                          _ptr(t), ts0, ts1, _ptr(r), rs0, rs1, _ptr(out), dev.index,
                                          _stream(dev))
        _lib.check(status, "blp_score_fwd")
        ctx.rel_model = rel_model
        ctx.shapes = (heads.shape, tails.shape, rels.shape)
        ctx.geometry = (m0, m1, D, tuple(shape))
        ctx.strides = ((hs0, hs1), (ts0, ts1), (rs0, rs1))
        ctx.save_for_backward(h, t, r)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, t, r = ctx.saved_tensors
        m0, m1, D, shape = ctx.geometry
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        need = ctx.needs_input_grad[1:4]
        grads = [torch.empty((*shape, D), dtype=torch.float32, device=dev) if n else None for n in need]
        (hs0, hs1), (ts0, ts1), (rs0, rs1) = ctx.strides
        status = _lib.lib().blp_score_bwd(_lib.MODEL_IDS[ctx.rel_model], D, m0, m1, _ptr(h), hs0, hs1,
                                          _ptr(t), ts0, ts1, _ptr(r), rs0, rs1, _ptr(grad_out),
                                          _ptr(grads[0]), _ptr(grads[1]), _ptr(grads[2]), dev.index,
                                          _stream(dev))
        _lib.check(status, "blp_score_bwd")
        outs = [g.sum_to_size(s) if g is not None else None for g, s in zip(grads, ctx.shapes)]
        return (None, *outs)


def score(rel_model, heads, tails, rels):
    """score_fn(heads, tails, rels) with the reference's broadcasting; bit-identical forward."""
    return _ScoreFn.apply(rel_model, heads, tails, rels)


# --------------------------------------------------------------------------------- in-batch loss
_tickets = {}


def inbatch_ticket(dev, stream):
    """The forward's ticket counter (include/blp_hip.h: BLP_INBATCH_TICKET_INTS int32, zero on entry, left zero by the kernel):
    one per (device, stream), zeroed once -- calls on one stream are ordered and share it."""
    key = (dev.index, stream)
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(_lib.INBATCH_TICKET_INTS, dtype=torch.int32, device=dev)
    return t


class _InBatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rel_model, loss_fn, regularizer, ent_embs, rel_vecs, neg_idx):
        _require_device(ent_embs, rel_vecs, neg_idx)
        if ent_embs.dim() != 3 or ent_embs.shape[1] != 2:
            raise ValueError(f"ent_embs must be (B, 2, D), got {tuple(ent_embs.shape)}")
        B, _, D = ent_embs.shape
        if neg_idx.dim() != 3 or neg_idx.shape[0] != B or neg_idx.shape[2] != 2:
            raise ValueError(f"neg_idx must be (B, K, 2), got {tuple(neg_idx.shape)}")
        K = neg_idx.shape[1]
        names = {getattr(torch, n): i for i, n in enumerate(_lib.DTYPE_NAMES)}
        if ent_embs.dtype not in names or rel_vecs.dtype not in (ent_embs.dtype, torch.float32):
            raise TypeError("ent_embs must be float32 / float16 / bfloat16 and rel_vecs of the same dtype or float32 "
                            f"(got {ent_embs.dtype}, {rel_vecs.dtype})")
        dtypes = (names[ent_embs.dtype], names[rel_vecs.dtype])
        ent = ent_embs.contiguous()
        rel = rel_vecs.reshape(B, D).contiguous()
        idx = neg_idx.to(torch.int64).contiguous()
        dev = ent.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        # blp_inbatch_loss_save_floats: positives' scores, the workgroups' partial loss sums, the index of neg_idx for the backward
        pos = torch.empty(_lib.inbatch_save_floats(_lib.MODEL_IDS[rel_model], B, K, D), dtype=torch.float32, device=dev)
        neg = torch.empty((B, K), dtype=torch.float32, device=dev)
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        status = _lib.lib().blp_inbatch_loss_fwd(_lib.MODEL_IDS[rel_model], _lib.LOSS_IDS[loss_fn], *dtypes,
                                                   _ptr(ent), _ptr(rel), _ptr(idx), B, K, D, float(regularizer),
                                                   _ptr(loss), _ptr(pos), _ptr(neg), _ptr(inbatch_ticket(dev, stream)), dev.index, stream)
        _lib.check(status, "blp_inbatch_loss_fwd")
        ctx.meta = (rel_model, loss_fn, float(regularizer), B, K, D, rel_vecs.shape, dtypes)
        ctx.save_for_backward(ent, rel, idx, pos, neg)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        ent, rel, idx, pos, neg = ctx.saved_tensors
        rel_model, loss_fn, regularizer, B, K, D, rel_shape, dtypes = ctx.meta
        dev = ent.device
        grad_loss = grad_loss.to(torch.float32).contiguous()
        grad_ent = torch.empty((B, 2, D), dtype=ent.dtype, device=dev)
        grad_rel = torch.empty((B, D), dtype=rel.dtype, device=dev)
        status = _lib.lib().blp_inbatch_loss_bwd(_lib.MODEL_IDS[rel_model], _lib.LOSS_IDS[loss_fn], *dtypes,
                                                   _ptr(ent), _ptr(rel), _ptr(idx), B, K, D, regularizer,
                                                   _ptr(grad_loss), _ptr(pos), _ptr(neg), _ptr(grad_ent),
                                                   _ptr(grad_rel), dev.index, _stream(dev))
        _lib.check(status, "blp_inbatch_loss_bwd")
        return None, None, None, grad_ent, grad_rel.reshape(rel_shape), None


_glue_module = False  # False: not looked for yet; None: not available (the Python autograd.Function serves)


def torch_glue():
    """blp_amd/_torch_glue.so (csrc/torch_glue.cpp: the in-batch loss's autograd plumbing as a torch::autograd::Function, no
    kernels) bound to the product library's entry points -- or None when it is not built / not loadable with this torch,
    in which case _InBatchLoss above does the same job from Python (same C-ABI calls, ~100 us more host time per step)."""
    global _glue_module
    if _glue_module is False:
        try:
            from . import _torch_glue as g
            L = _lib.lib()
            if getattr(g, "abi_version", 0) // 10000 != L.blp_version() // 10000:  # a stale build against an older header
                raise ImportError(f"blp_amd/_torch_glue.so was built for include/blp_hip.h {getattr(g, 'abi_version', '< 6.0.0')}, the library is "
                                  f"{L.blp_version()}; rebuild with python -m blp_amd.build")
            g.bind(*(ctypes.cast(fn, ctypes.c_void_p).value for fn in (L.blp_inbatch_loss_fwd, L.blp_inbatch_loss_bwd, L.blp_last_error,
                                                                       L.blp_inbatch_loss_save_floats)))
            _glue_module = g
        except Exception as exc:  # not built, another torch, a stale build, a failed bind: the Python autograd.Function serves
            import warnings
            if not isinstance(exc, ModuleNotFoundError):
                warnings.warn(f"blp_amd: C++ autograd glue not used ({exc}); falling back to the Python autograd.Function")
            _glue_module = None
    return _glue_module


def inbatch_loss(rel_model, loss_fn, ent_embs, rel_vecs, neg_idx, regularizer=0.0):
    """compute_loss on in-batch negatives (fused forward + deterministic backward)."""
    glue = torch_glue()
    if glue is None:
        return _InBatchLoss.apply(rel_model, loss_fn, regularizer, ent_embs, rel_vecs, neg_idx)
    _require_device(ent_embs, rel_vecs, neg_idx)
    return glue.inbatch_loss(ent_embs, rel_vecs, neg_idx, _lib.MODEL_IDS[rel_model], _lib.LOSS_IDS[loss_fn], float(regularizer),
                             torch._C._cuda_getCurrentRawStream(ent_embs.device.index))
