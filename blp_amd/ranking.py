"""All-entities ranking evaluation: the reference's eval_link_prediction (/root/reference/train.py:57-243)
restated around the fused HIP ranking, plus candidate-axis sharding across GPUs (new capability).

What changes relative to the reference, and what does not:
  * same inputs, same metric names, same numbers: raw and filtered MRR / Hits@{1,3,10}, the by-new-
    position split, the relation-category split, ``(mrr, ent_emb)`` return value;
  * the (2B, N) score matrix, the dense (2B, N) filter mask and the per-batch ``.item()`` syncs are
    gone: a block of triples becomes a (2B, 4) int32 count tensor on the device (blp_amd.ops.rank_all
    with a CSR filter from utils.FilterIndex), metrics are reduced once at the end;
  * with a process group, every rank encodes only rows [lo, hi) of the entity table; the ranking is
    then sharded along one of two axes (choose_shard_axis):
      - "candidate" (big tables, e.g. Wikidata5M): every rank keeps its rows, query / true-entity
        vectors are replicated by one exchange, and the per-shard counts of the whole evaluation
        are combined by ONE all-gather + sum;
      - "query" (small tables, many test triples, e.g. FB15k-237): the table shards are all-gathered
        once, every rank ranks its slice of the test triples against the full table, and the
        per-triple counts are all-gathered (no sum).  Per-query costs shrink with the world size too.
    RCCL over xGMI on GPUs; gloo in the CPU tests.

CPU tensors (the reference's CPU-runnable smoke configuration) take the reference's own dense route
through score_fn + get_metrics.
"""
import copy
import math

import torch
import torch.distributed as dist

from . import models, multidevice, ops, utils

HIT_POSITIONS = (1, 3, 10)


# ----------------------------------------------------------------------------------- one block
def _rank_block_dense(score_fn, table, q_fixed, q_rel, q_head, true_scores_from, filt_rowptr, filt_col):
    """Reference route (train.py:146-171) for CPU tensors: dense scores, then counts.
    ``true_scores_from`` is either ('row', true_row) or ('vec', q_true)."""
    ent = table.unsqueeze(0)
    fixed, rel = q_fixed.unsqueeze(1), q_rel.unsqueeze(1)
    pred = torch.cat((score_fn(ent, fixed[:q_head], rel[:q_head]), score_fn(fixed[q_head:], ent, rel[q_head:])))
    kind, value = true_scores_from
    if kind == "row":
        true = pred.gather(1, value.reshape(-1, 1))
    else:
        tv = value.unsqueeze(1)
        true = torch.cat((score_fn(tv[:q_head], fixed[:q_head], rel[:q_head]),
                          score_fn(fixed[q_head:], tv[q_head:], rel[q_head:])))
    gt_mask, ge_mask = pred > true, pred >= true
    counts = torch.empty((pred.shape[0], 4), dtype=torch.int32)
    counts[:, 0] = gt_mask.sum(1)
    counts[:, 1] = ge_mask.sum(1)
    if filt_rowptr is not None:
        keep = torch.ones_like(gt_mask)
        rows = torch.repeat_interleave(torch.arange(pred.shape[0]), filt_rowptr[1:] - filt_rowptr[:-1])
        keep[rows, filt_col] = False
        counts[:, 2] = (gt_mask & keep).sum(1)
        counts[:, 3] = (ge_mask & keep).sum(1)
    else:
        counts[:, 2:] = counts[:, :2]
    return counts


def fused_ranking_takes(model, table, num_queries):
    """True if a block of ``num_queries`` (half per side) against ``table`` goes through blp_rank_all (which takes
    the filter as device-side segments); otherwise the block takes a dense route that wants a CSR."""
    half = num_queries // 2
    return table.is_cuda and ops.rank_all_supported(model.rel_model, table.shape[1], half, num_queries - half)


def rank_block(model, table, q_fixed, q_rel, q_head, true_row=None, q_true=None, filt_rowptr=None, filt_col=None,
               rel_ids=None, filter=None, out=None):
    """Counts (Q, 4) int32 {gt, ge, gt_filtered, ge_filtered} for a block of queries against ``table``.
    Queries [0, q_head) replace the head, the rest replace the tail (train.py:149 order).  The filter is a CSR
    (filt_rowptr, filt_col) or, on a HIP device, an ops.SegmentFilter (``filter``)."""
    if table.is_cuda and not ops.rank_all_supported(model.rel_model, table.shape[1], q_head, q_fixed.shape[0] - q_head):
        if filter is not None:
            raise ValueError("the dense any-width route takes the filter as a CSR (utils.FilterIndex.csr)")
        counts = _rank_block_generic_width(model, table, q_fixed, q_rel, q_head, true_row, q_true, filt_rowptr, filt_col)
        return counts if out is None else out.copy_(counts)
    if table.is_cuda:
        dev = table.device
        return ops.rank_all(model.rel_model, table, q_fixed, q_rel, q_head,
                            true_row=None if true_row is None else true_row.to(dev),
                            q_true=q_true,
                            filt_rowptr=None if filt_rowptr is None else filt_rowptr.to(dev),
                            filt_col=None if filt_col is None else filt_col.to(dev),
                            rel_ids=None if rel_ids is None else rel_ids.to(dev), filter=filter, out=out)
    if filter is not None:
        raise ValueError("SegmentFilter is the HIP path's filter form; CPU tensors take a CSR")
    source = ("row", true_row) if true_row is not None else ("vec", q_true)
    counts = _rank_block_dense(model.score_fn, table, q_fixed, q_rel, q_head, source, filt_rowptr, filt_col)
    return counts if out is None else out.copy_(counts)


def _rank_block_generic_width(model, table, q_fixed, q_rel, q_head, true_row, q_true, filt_rowptr, filt_col,
                              max_matrix_bytes=1 << 30):
    """HIP route for embedding widths the fused ranking kernels are not compiled for (300 / 768-wide
    bag-of-words encoders): order-exact dense scores from blp_score_fwd in query slabs of bounded size,
    then blp_rank_from_scores.  Same counts; the (Q, N) slab does go through HBM here."""
    dev = table.device
    n = table.shape[0]
    q_total = q_fixed.shape[0]
    slab = max(1, min(q_total, max_matrix_bytes // max(4 * n, 1)))
    ent = table.unsqueeze(0)
    out = torch.empty((q_total, 4), dtype=torch.int32, device=dev)
    if filt_rowptr is not None:
        filt_rowptr, filt_col = filt_rowptr.to(dev), filt_col.to(dev)
    for lo in range(0, q_total, slab):
        hi = min(lo + slab, q_total)
        parts = []
        for a, b, head in ((lo, min(hi, q_head), True), (max(lo, q_head), hi, False)):
            if a < b:
                fixed, rel = q_fixed[a:b].unsqueeze(1), q_rel[a:b].unsqueeze(1)
                parts.append(model.score_fn(ent, fixed, rel) if head else model.score_fn(fixed, ent, rel))
        scores = torch.cat(parts) if len(parts) > 1 else parts[0]
        kw = {}
        if true_row is not None:
            kw["true_idx"] = true_row[lo:hi].to(dev)
        else:
            tv = q_true[lo:hi]
            h_end = max(min(hi, q_head) - lo, 0)
            kw["true_score"] = torch.cat((model.score_fn(tv[:h_end], q_fixed[lo:lo + h_end], q_rel[lo:lo + h_end]),
                                          model.score_fn(q_fixed[lo + h_end:hi], tv[h_end:], q_rel[lo + h_end:hi])))
        if filt_rowptr is not None:
            kw["filt_rowptr"] = filt_rowptr[lo:hi + 1] - filt_rowptr[lo]
            kw["filt_col"] = filt_col[filt_rowptr[lo]:filt_rowptr[hi]]
        out[lo:hi] = ops.rank_from_scores(scores, **kw)
    return out


def metrics_from_counts(counts, k_values=HIT_POSITIONS):
    """utils.py:104-109 from counts: rr (Q, 2) f32 [raw, filtered], hits (Q, 2, 3) bool."""
    if counts.is_cuda:
        return ops.rank_metrics(counts, k_values)
    c = counts.to(torch.int64)
    avg = torch.stack((c[:, 0] + 1 + c[:, 1], c[:, 2] + 1 + c[:, 3]), dim=1).float() * 0.5
    k = torch.tensor(k_values, dtype=torch.float32).view(1, 1, -1)
    return avg.reciprocal(), avg.unsqueeze(-1) <= k


# ----------------------------------------------------------------------------------- sharding
def shard_bounds(num_rows, world_size, rank):
    per = (num_rows + world_size - 1) // world_size
    return min(rank * per, num_rows), min((rank + 1) * per, num_rows)


QUERY_AXIS_MAX_TABLE_BYTES = 256 << 20   # replicate the table only if it is this small ...
QUERY_AXIS_MIN_QUERIES = 256             # ... and every rank still gets a many-query block


def choose_shard_axis(num_rows, dim, num_queries, world_size):
    """"query" when replicating the table is cheap and each rank keeps >= 256 queries (the regime
    where per-query costs, not the table pass, dominate a candidate shard); else "candidate"."""
    if world_size <= 1:
        return "candidate"
    small_table = num_rows * dim * 4 <= QUERY_AXIS_MAX_TABLE_BYTES
    return "query" if small_table and num_queries // world_size >= QUERY_AXIS_MIN_QUERIES else "candidate"


def replicates_whole_table(num_entities, num_triples, half_table=False):
    """Candidate-axis shards need the vectors the queries are made of on every rank: the whole table by ONE all-gather when it is
    smaller than the 2T vectors of the triples, else those 2T vectors by ONE all-reduce of an owner-filled array."""
    return num_entities <= 2 * num_triples and not half_table


def exchange_plan(num_entities, dim, num_triples, world, axis, half_table=False):
    """The collectives ONE evaluation of num_triples triples (2 * num_triples queries) issues on `world` ranks, as
    rank_triples below issues them (its fused path; same conditions): a list of {"op", "what", "bytes_per_rank",
    "bytes_total"} -- bytes_per_rank = what one rank contributes, bytes_total = what every rank holds afterwards.  SURVEY
    8e: the candidate axis ends in ONE all-gather of the int32 counts, G x Q x 16 bytes in all.  No GPU needed (tests, bench's
    plan of an N-GPU run)."""
    if world <= 1:
        return []
    Q = 2 * num_triples
    plan = []
    if axis == "candidate":
        if replicates_whole_table(num_entities, num_triples, half_table):
            per = (num_entities + world - 1) // world  # all_gather_rows pads every shard to the longest
            plan.append({"op": "all_gather", "what": "table rows (the queries' vectors: the table is smaller than they are)",
                         "bytes_per_rank": per * dim * 4, "bytes_total": world * per * dim * 4})
        else:
            plan.append({"op": "all_reduce", "what": "(2T, D) f32 vectors of the triples' entities, each filled by the shard that owns the row",
                         "bytes_per_rank": Q * dim * 4, "bytes_total": Q * dim * 4})
        plan.append({"op": "all_gather", "what": "(2T, 4) int32 rank counts of every shard, then a sum",
                     "bytes_per_rank": Q * 16, "bytes_total": world * Q * 16})
    elif axis == "query":
        per = (num_triples + world - 1) // world
        plan.append({"op": "all_gather", "what": "(T / G, 8) int32 counts of each rank's triples (head and tail query side by side)",
                     "bytes_per_rank": per * 32, "bytes_total": world * per * 32})
    else:
        raise ValueError(f"unknown shard axis {axis!r}")
    return plan


def _via_host(tensor, group):
    """gloo carries host memory only: device tensors go through a host copy there (functional runs of the N > 1
    path on fewer GPUs than ranks, tests); RCCL takes them as they are."""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def _threads(group):
    """True if ``group`` is a member of a one-process / one-thread-per-device group (multidevice.DeviceGroup) rather than
    a torch.distributed process group: the same two collectives, issued for all devices from one thread."""
    return isinstance(group, multidevice.Member)


def _all_reduce(tensor, group=None):
    if _threads(group):
        group.all_reduce(tensor)
    elif _via_host(tensor, group):
        host = tensor.cpu()
        dist.all_reduce(host, group=group)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, group=group)


def _all_gather_into(full, part, group=None):
    if _threads(group):
        group.all_gather_into(full, part)
    elif _via_host(part, group):
        host = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(host, part.cpu(), group=group)
        full.copy_(host)
    else:
        dist.all_gather_into_tensor(full, part, group=group)


def all_gather_rows(local, num_rows, world_size, group=None):
    """Row shards (shard_bounds layout) -> the full (num_rows, ...) tensor on every rank: ONE all-gather."""
    if world_size == 1:
        return local
    per = (num_rows + world_size - 1) // world_size
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    full = torch.empty((world_size * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_into(full, padded, group)
    return full[:num_rows]


class ShardedRanker:
    """Candidate-axis sharded ranking.  Rank r owns table rows [lo, hi); the per-shard
    count tensors add up exactly to the unsharded counts, so one all-gather of int32 counts per
    evaluation is the only data-path collective."""

    def __init__(self, model, local_table, num_rows, group=None):
        self.model = model
        self.table = local_table
        self.num_rows = int(num_rows)
        self.group = group
        if _threads(group):
            self.world, self.rank = group.world, group.rank
        else:
            self.world = dist.get_world_size(group) if self._distributed() else 1
            self.rank = dist.get_rank(group) if self._distributed() else 0
        self.lo, self.hi = shard_bounds(self.num_rows, self.world, self.rank)
        if local_table.shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo}, {self.hi}) of the table, "
                             f"got {local_table.shape[0]} rows")
        self._blocks = []

    def _distributed(self):
        return dist.is_available() and dist.is_initialized()

    def gather_rows(self, rows):
        """Vectors of the given global table rows on every rank: owners fill, one all-reduce (adding
        exact zeros) replicates.  Called once per evaluation for the entities of the test triples."""
        rows = rows.to(device=self.table.device, dtype=torch.long)
        out = torch.zeros((rows.shape[0], self.table.shape[1]), dtype=self.table.dtype, device=self.table.device)
        if self.hi > self.lo:  # no host decision on the data: rows of other shards read row 0 and are zeroed
            mine = (rows >= self.lo) & (rows < self.hi)
            local = torch.where(mine, rows - self.lo, torch.zeros_like(rows))
            out = torch.where(mine.unsqueeze(1), self.table[local], out)
        if self.world > 1:
            _all_reduce(out, self.group)
        return out

    def rank_block(self, q_fixed, q_rel, q_true, q_head, filt_rowptr=None, filt_col=None, rel_ids=None, filter=None):
        """Local counts of one query block against this rank's shard (queued for the final exchange).
        filt_col are GLOBAL table rows; only the ones this shard owns are kept (a SegmentFilter carries this
        shard's first row as row_base and the kernel skips the rows of other shards)."""
        if filter is not None:
            filter = filter._replace(row_base=self.lo)
        if filt_rowptr is not None and self.table.is_cuda:  # global rows minus this shard's first row; others are skipped
            filt_col = filt_col.to(self.table.device) - self.lo
        elif filt_rowptr is not None:
            owned = (filt_col >= self.lo) & (filt_col < self.hi)
            per_row = torch.zeros(filt_rowptr.shape[0] - 1, dtype=torch.long)
            rows = torch.repeat_interleave(torch.arange(per_row.shape[0]), filt_rowptr[1:] - filt_rowptr[:-1])
            per_row.index_add_(0, rows[owned], torch.ones(int(owned.sum()), dtype=torch.long))
            filt_rowptr = torch.cat((torch.zeros(1, dtype=torch.long), torch.cumsum(per_row, 0)))
            filt_col = filt_col[owned] - self.lo
        counts = rank_block(self.model, self.table, q_fixed, q_rel, q_head, q_true=q_true,
                            filt_rowptr=filt_rowptr, filt_col=filt_col, rel_ids=rel_ids, filter=filter)
        self._blocks.append(counts)
        return counts

    def finish(self):
        """Global counts of every queued block: ONE all-gather of the (Q, 4) int32 tensor, then a sum."""
        local = torch.cat(self._blocks) if self._blocks else torch.zeros((0, 4), dtype=torch.int32, device=self.table.device)
        self._blocks = []
        if self.world == 1:
            return local
        gathered = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        _all_gather_into(gathered.view(-1), local.contiguous().view(-1), self.group)
        return gathered.sum(dim=0, dtype=torch.int32)


# ----------------------------------------------------------------------------------- a set of triples
class _Stopwatch:
    """Device-side timing of the collectives of one evaluation (bench.py's ``exchange_ms``): events around each exchange
    on the current stream, read after the caller's synchronisation.  ``timing`` = None: nothing is recorded."""

    def __init__(self, timing, device):
        self.timing = timing if (timing is not None and device.type == "cuda") else None

    def __enter__(self):
        if self.timing is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.stop = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.timing is not None:
            self.stop.record()
            self.timing.setdefault("exchange_events", []).append((self.start, self.stop))
        return False


def exchange_ms(timing):
    """Milliseconds the recorded exchanges of ``timing`` took (call after a device synchronisation)."""
    return sum(a.elapsed_time(b) for a, b in timing.get("exchange_events", []))


def rank_triples(model, table, triples, ent2idx, index=None, *, num_entities=None, group=None, world=1, rank=0,
                 axis="candidate", block_size=65536, timing=None):
    """The reference's evaluation loop body (train.py:128-171) for a whole set of triples at once: rank counts of
    every triple's head and tail query against all ``num_entities`` candidates, raw and filtered.

    triples   (T, 3) int64 on the table's device, rows (head, tail, rel) entity / relation ids, any order
    ent2idx   utils.make_ent2idx map (entity id -> table row) on the same device
    index     utils.FilterIndex of the filtering graph, or None (filtered counts = raw counts)
    table     world == 1 or axis == "query": the full (num_entities, D) table; axis == "candidate": this rank's
              rows shard_bounds(num_entities, world, rank) of it
    timing    optional dict: the collectives are bracketed by device events (exchange_ms(timing) after a synchronise)
    Returns (triples in evaluation order -- as given --, counts (2T, 4) int32 in that order with every head query first,
    ids_ok: 0-dim bool tensor or None -- the reference's assertion train.py:137-138, left on the device).

    Everything runs on the table's device without a host round trip: the id lookups (train.py:134-135), the layout of
    the queries -- left as (vector row, relation) index pairs -- and the filter segments are one kernel
    (ops.build_queries); the blocks are then ONE library call (blp_rank_all_batches: per block <= 8 launches, none of them
    torch's).
    Collectives: candidate axis -- the vectors the queries are made of are replicated ONCE (the whole table by one
    all-gather when it is smaller than the 2T vectors of the triples, else those vectors by one all-reduce of an
    owner-filled array) and the (2T, 4) counts of every shard are combined by ONE all-gather + sum; query axis -- ONE
    all-gather of the per-triple counts.  No size is decided on the device (no torch.unique, no .item())."""
    device = table.device
    num_entities = table.shape[0] if num_entities is None else num_entities
    num_triples = triples.shape[0]
    by_query = world > 1 and axis == "query"
    by_candidate = world > 1 and not by_query
    t_lo, t_hi = shard_bounds(num_triples, world, rank) if by_query else (0, num_triples)
    n = t_hi - t_lo
    mine = slice(t_lo, t_hi)
    # (whether the fused path takes a block depends on the model and the width only; WHICH kernels rank it is decided per
    # block inside blp_rank_all_batches -- the short last block of a call may well take another route than the full ones)
    fused = n > 0 and model.rel_emb.weight.dtype == torch.float32 and fused_ranking_takes(model, table, 2 * min(block_size, n))
    row_lo = shard_bounds(num_entities, world, rank)[0] if by_candidate else 0
    head_pos = tail_pos = None

    def block_positions():
        idx = torch.arange(n, device=device)
        first = torch.div(idx, block_size, rounding_mode="floor") * block_size       # first triple of the block
        hp = first + idx                                                                # = 2 * first + (idx - first)
        return idx, hp, hp + torch.clamp(n - first, max=block_size)

    if fused:
        # The queries stay INDICES: (row of `source`, row of rel_emb) -- no (2n, D) arrays at all.  `source` holds the
        # vectors the queries are made of: the table itself, or on a candidate shard a replicated copy of what the
        # triples need from the other shards.
        rel_w = model.rel_emb.weight
        source, by_position = table, False
        half_table = table.dtype != torch.float32  # the 16-bit copy of the table (ops.rank_all_batches): candidates only
        if by_candidate and replicates_whole_table(num_entities, num_triples, half_table):  # small table: all of it, one all-gather
            with _Stopwatch(timing, device):
                source = all_gather_rows(table, num_entities, world, group)
        elif by_candidate or half_table:                          # big table: the 2T vectors of the triples, one all-reduce
            # (a 16-bit table on any axis: the queries' own vectors are its rows WIDENED to float32 -- exactly --, which this
            #  gather does on the way; nothing to exchange unless the candidate axis is sharded)
            source = ops.gather_triple_vectors(triples[mine], ent2idx, table, row_base=row_lo)
            if by_candidate:
                with _Stopwatch(timing, device):
                    _all_reduce(source, group)
            by_position = True
        # one kernel (blp_build_queries) instead of ~35 small torch kernels: lookups, layout, binary searches
        qb = ops.build_queries(triples[mine], ent2idx, source, rel_w, block_size, index=index, gather=False, row_base=row_lo,
                               by_position=by_position, num_rows=num_entities)
        ids_ok = qb.ids_min >= 0
        # every block in ONE library call (blp_rank_all_batches with a ranking pass per block: the blocks are issued back to
        # back by the library -- at the reference's Wikidata5M batching, 2 triples per table pass, a Python-level loop over
        # the blocks costs as much host time per pass as the pass takes on a 1/8 shard)
        counts = ops.rank_all_batches(model.rel_model, table, qb.fixed_row, rel_w, qb.rel_ids, qb.true_row, n, block_size,
                                      filter=qb.filter, source=source, block_triples=block_size)
        if by_candidate:  # ONE all-gather for the whole set; the shards' counts add up exactly
            with _Stopwatch(timing, device):
                gathered = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=device)
                _all_gather_into(gathered.view(-1), counts.view(-1), group)
                counts = gathered.sum(dim=0, dtype=torch.int32)
    else:
        # CPU tensors, half-precision relation tables, widths the fused kernels do not take: the torch prelude and
        # gathered query vectors (dense routes; a CSR filter per block).
        if table.dtype != torch.float32:
            table = table.float()  # (a 16-bit table on a dense route: widened once, exactly)
        ranker = ShardedRanker(model, table, num_entities, group) if by_candidate else None
        heads = ent2idx[triples[:, 0]]
        tails = ent2idx[triples[:, 1]]
        ids_ok = torch.minimum(heads.min(), tails.min()) >= 0 if num_triples else None
        heads, tails = heads.clamp_min(0), tails.clamp_min(0)  # a bad id reads row 0 instead of out of bounds; ids_ok tells
        if by_candidate:  # replicate the vectors of the entities in the triples: one exchange
            with _Stopwatch(timing, device):
                source = ranker.gather_rows(torch.cat((heads, tails)))
            head_ref = torch.arange(num_triples, device=device)
            tail_ref = head_ref + num_triples
        else:
            source, head_ref, tail_ref = table, heads[mine], tails[mine]
        idx, head_pos, tail_pos = block_positions()
        fixed_src = torch.empty(2 * n, dtype=torch.long, device=device)
        true_src = torch.empty_like(fixed_src)
        rel_ids = torch.empty_like(fixed_src)
        order_src = torch.empty_like(fixed_src)  # position in the [all heads | all tails] order of utils.FilterIndex
        fixed_src[head_pos], fixed_src[tail_pos] = tail_ref, head_ref
        true_src[head_pos], true_src[tail_pos] = head_ref, tail_ref
        rel_ids[head_pos] = rel_ids[tail_pos] = triples[mine, 2]
        order_src[head_pos], order_src[tail_pos] = idx, idx + n
        q_fixed = source[fixed_src]
        q_rel = model.rel_emb(rel_ids)
        q_true = source[true_src] if by_candidate else None
        seg = None
        if index is not None and n > 0 and fused_ranking_takes(model, table, 2 * min(block_size, n)):  # slices of the sorted index: nothing is listed per batch
            seg = index.segments(triples[mine], ent2idx, device)
            seg = seg._replace(seg_lo=seg.seg_lo[order_src], seg_hi=seg.seg_hi[order_src], exclude=seg.exclude[order_src])
        counts = torch.empty((2 * n, 4), dtype=torch.int32, device=device)
        for start in range(0, n, block_size):
            b = min(start + block_size, n) - start
            sl = slice(2 * start, 2 * (start + b))
            filt = {}
            if seg is not None:
                filt = dict(filter=seg._replace(seg_lo=seg.seg_lo[sl], seg_hi=seg.seg_hi[sl], exclude=seg.exclude[sl]))
            elif index is not None:  # dense any-width routes take a CSR
                rowptr, col = index.csr(triples[t_lo + start: t_lo + start + b], ent2idx, device)
                filt = dict(filt_rowptr=rowptr, filt_col=col)
            if by_candidate:
                ranker.rank_block(q_fixed[sl], q_rel[sl], q_true[sl], b, rel_ids=rel_ids[sl], **filt)
            else:
                rank_block(model, table, q_fixed[sl], q_rel[sl], b, true_row=true_src[sl], rel_ids=rel_ids[sl], out=counts[sl],
                           **filt)
        if by_candidate:
            with _Stopwatch(timing, device):
                counts = ranker.finish()  # ONE all-gather for the whole set; blocks in the order they were queued
    if n > block_size:  # every head query first, like one big batch
        if head_pos is None:
            _, head_pos, tail_pos = block_positions()
        counts = counts[torch.cat((head_pos, tail_pos))]
    if by_query:  # per-triple counts of every rank's slice: ONE all-gather for the whole set
        with _Stopwatch(timing, device):
            both = all_gather_rows(torch.cat((counts[:n], counts[n:]), dim=1), num_triples, world, group)
        counts = torch.cat((both[:, :4], both[:, 4:])).contiguous()
    return triples, counts, ids_ok


# ----------------------------------------------------------------------------------- evaluation
def _module(model):
    wrappers = (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)
    return model.module if isinstance(model, wrappers) else model


def _loader_triples(loader, max_num_batches):
    """Every triple the reference's loop ``for i, triples in enumerate(loader)`` would see before
    ``i == max_num_batches``.  A sequential default-collate DataLoader over a dataset that holds its
    triples as one tensor (data.GraphDataset and subclasses: the reference's eval loaders,
    train.py:124-128) is read as a slice instead of 800+ Python-level batches."""
    from torch.utils.data import SequentialSampler
    from torch.utils.data.dataloader import default_collate
    dataset = getattr(loader, "dataset", None)
    stored = getattr(dataset, "triples", None)
    batch_size = getattr(loader, "batch_size", None)
    if (isinstance(stored, torch.Tensor) and stored.dim() == 2 and batch_size
            and isinstance(getattr(loader, "sampler", None), SequentialSampler)
            and getattr(loader, "collate_fn", None) is default_collate and len(dataset) == stored.shape[0]):
        n = stored.shape[0]
        if getattr(loader, "drop_last", False):
            n = n // batch_size * batch_size
        if max_num_batches is not None:
            n = min(n, max_num_batches * batch_size)
        return stored[:n]
    batches = []
    for i, triples in enumerate(loader):
        if max_num_batches is not None and i == max_num_batches:
            break
        batches.append(triples)
    return torch.cat(batches) if batches else torch.zeros((0, 3), dtype=torch.long)


@torch.no_grad()
def build_entity_table(model, text_dataset, entities, emb_batch_size, device, log=None, rows=None):
    """train.py:96-121: encode ``entities`` (or the slice ``rows`` = (lo, hi) of them) in chunks of
    emb_batch_size into an (n, dim) f32 table, in the order of ``entities``."""
    lo, hi = rows if rows is not None else (0, entities.shape[0])
    n = hi - lo
    table = torch.zeros((n, model.dim), dtype=torch.float, device=device)
    num_iters = math.ceil(n / emb_batch_size) if n else 0
    report = max(1, math.ceil(0.2 * num_iters))
    for it, idx in enumerate(range(0, n, emb_batch_size)):
        batch_ents = entities[lo + idx: lo + min(idx + emb_batch_size, n)]
        if isinstance(model, models.InductiveLinkPrediction):
            text_tok, text_mask, _ = text_dataset.get_entity_description(batch_ents)
            # the model writes its rows itself: the BERT encoders fuse enc_linear + normalise + this assignment
            # (models.BertEmbeddingsLP.encode_into -> blp_project_rows); same values as model(tok, mask), train.py:109
            rows, tok, mask = table[idx: idx + batch_ents.shape[0]], text_tok.to(device), text_mask.to(device)
            if hasattr(model, "check_tokens"):  # (the fused bag-of-words / DKRL builds: their token-id check once per table, below)
                model.encode_into(rows, tok, mask, defer_check=True)
            else:
                model.encode_into(rows, tok, mask)
        else:
            table[idx: idx + batch_ents.shape[0]] = model(batch_ents.to(device))
        if log is not None and (it + 1) % report == 0:
            log.info(f"[{idx + batch_ents.shape[0]:,}/{n:,}]")
    if hasattr(model, "check_tokens"):  # the fused bag-of-words build leaves its token-id check on the device until here
        model.check_tokens()
    return table


def table16_everywhere(rel_model, dtype, num_entities, dim, num_triples, block_size, world, axis):
    """Whether EVERY rank's shard of an evaluation would stream a 16-bit copy of the table itself (ops.table16_is_read_directly),
    evaluated for all ranks' (rows, triples) from the shard layout alone -- so that every rank takes the same decision
    without an exchange.  A per-rank decision can differ between ranks (the last candidate shard is shorter and the
    library's routing has a row threshold): some ranks would then rank the rounded copy and others the float32 table,
    the summed counts would mix two inputs, and -- the copy deciding which collective replicates the queries' vectors
    (replicates_whole_table) -- the ranks could issue mismatched collectives and hang."""
    by_query = world > 1 and axis == "query"
    for r in range(max(world, 1)):
        lo, hi = (0, num_entities) if (world <= 1 or by_query) else shard_bounds(num_entities, world, r)
        t_lo, t_hi = shard_bounds(num_triples, world, r) if by_query else (0, num_triples)
        if not ops.table16_is_read_directly(rel_model, dtype, hi - lo, dim, t_hi - t_lo, block_size):
            return False
    return True


def _shard_of_evaluation(model, text_dataset, entities, emb_batch_size, device, _log, triples, ent2idx, index, *, group, world, rank,
                         shard_axis, block_size, rank_table_dtype):
    """What ONE rank (a process of a torch.distributed group, or a device thread of a multidevice.DeviceGroup) does of an
    evaluation: encode its rows of the entity table (train.py:96-121), rank the triples against them (rank_triples; the
    collectives of SURVEY.md 8e inside).  Returns (table -- this rank's rows, or all rows on the query axis --, axis,
    triples, counts (2T, 4) int32 of the WHOLE evaluation, ids_ok)."""
    num_entities, num_triples = entities.shape[0], triples.shape[0]
    lo, hi = shard_bounds(num_entities, world, rank)
    table = build_entity_table(model, text_dataset, entities, emb_batch_size, device, _log if rank == 0 else None, rows=(lo, hi))
    triples = triples.to(device)  # in loader order (train.py:128-131)
    axis = shard_axis if shard_axis != "auto" else choose_shard_axis(num_entities, table.shape[1], 2 * num_triples, world)
    if world > 1 and axis == "query":  # full table everywhere, each rank takes a slice of the triples
        table = all_gather_rows(table, num_entities, world, group)
    rank_table = table
    if rank_table_dtype not in (None, torch.float32) and table.is_cuda:
        if table16_everywhere(model.rel_model, rank_table_dtype, num_entities, table.shape[1], num_triples, block_size, world, axis):
            rank_table = table.to(rank_table_dtype)
        else:
            # (the library would widen a 16-bit copy back to float32: more memory and time than the float32 table and other
            #  metrics for nothing; and the reference's pass structure -- a table pass per eval batch -- only pays with it)
            block_size = max(block_size, 65536)
            if rank == 0:
                _log.info(f"rank_table_dtype={rank_table_dtype}: not used for this evaluation (dim {table.shape[1]}, {num_entities:,} rows "
                          f"on {world} rank(s): the library would widen a 16-bit copy back to float32); ranking the float32 table "
                          f"in blocks of {block_size:,} triples")
    triples, counts, ids_ok = rank_triples(model, rank_table, triples, ent2idx.to(device), index, num_entities=num_entities,
                                           group=group, world=world, rank=rank, axis=axis, block_size=block_size)
    return table, axis, triples, counts, ids_ok


@torch.no_grad()
def eval_link_prediction(model, triples_loader, text_dataset, entities, epoch, emb_batch_size, _run, _log,
                         prefix="", max_num_batches=None, filtering_graph=None, new_entities=None,
                         return_embeddings=False, device=None, group=None, block_size=65536, shard_axis="auto",
                         eval_mode=False, rank_table_dtype=None, devices=None):
    """Drop-in for train.eval_link_prediction (same positional arguments, metric names and return
    value).  ``device`` defaults to the model's device.  Several GPUs, two ways (both: SURVEY.md 8e -- every shard encodes
    and keeps its own rows, ONE all-gather of the int32 counts; ``shard_axis`` = "candidate" | "query" | "auto"):

      * ``devices`` = [device, ...]: THIS process, one Python thread per device -- the reference's own process model
        (nn.DataParallel, train.py:329-330,344).  Every thread gets a replica of the model, builds its row shard, ranks on its
        own stream; the counts are combined by RCCL calls issued for all devices from one thread (multidevice.DeviceGroup).
        A device may repeat (two shards on one GPU).  The metrics, and the returned table, live on devices[0].
      * ``group`` (or an initialised default process group): one PROCESS per device under torch.distributed.run.

    ``eval_mode``: the reference never leaves train mode (train.py:57-121 has no model.eval()), so its entity table is
    built with BERT's dropout active and its metrics differ run to run.  The default (False) keeps whatever mode the
    model is in -- the reference's behaviour, so that this function drops in without changing a result's law;
    ``eval_mode=True`` puts the encoder in eval mode for the table build and restores it afterwards (deterministic
    tables; train.py's config key ``eval_dropout=False``).

    ``rank_table_dtype`` (torch.float16 / torch.bfloat16; default None = the reference's float32): rank against a 16-bit COPY of
    the table ("emit fp16 copy").  A deliberate change of the INPUT, not of the arithmetic: the candidates are
    the rounded rows, scored in f32 in the reference's order (counts = the reference's on the rounded table, bit for bit).
    Honoured ONLY where it pays: at the reference's Wikidata5M batching (``block_size`` <= 4 triples per table pass, dim 128 /
    256, a table long enough for the streaming kernels ON EVERY RANK) the passes read the 16-bit rows themselves, half the
    bytes; for any other shape the float32 table is ranked in the default 65 536-triple blocks and a log line says so
    (train.py: ``rank_table_dtype=float16``).  The returned embeddings stay float32."""
    model = _module(model)
    if device is None:
        device = next(model.parameters()).device
    compute_filtered = filtering_graph is not None
    dataset = triples_loader.dataset
    if devices is not None and group is not None:
        raise ValueError("give `devices` (threads of this process) or `group` (a process group), not both")
    if devices is not None:
        threads = multidevice.DeviceGroup(devices)
        device = threads.devices[0]
    else:
        threads = None

    if isinstance(model, models.InductiveLinkPrediction):
        if compute_filtered:
            index = filtering_graph if isinstance(filtering_graph, utils.FilterIndex) else utils.FilterIndex(filtering_graph, device=device)
            max_ent_id = max(index.max_node, int(entities.max()))
        else:
            index, max_ent_id = None, int(entities.max())
        ent2idx = utils.make_ent2idx(entities, max_ent_id)
    else:
        entities = torch.arange(model.ent_emb.num_embeddings)
        ent2idx = entities
        index = None
        if compute_filtered:
            index = filtering_graph if isinstance(filtering_graph, utils.FilterIndex) else utils.FilterIndex(filtering_graph, device=device)
    num_entities = entities.shape[0]

    was_training = model.training
    if eval_mode:
        model.eval()
    triples = _loader_triples(triples_loader, max_num_batches)
    num_triples = triples.shape[0]
    common = dict(shard_axis=shard_axis, block_size=block_size, rank_table_dtype=rank_table_dtype)
    if threads is not None:
        # one replica per device thread, made here on the calling thread (a module is not copied while another thread runs it);
        # the model itself serves the first thread on its own device
        home = next(model.parameters()).device
        replicas, used_home = [], False
        for d in threads.devices:
            if d == home and not used_home:
                replicas.append(model)
                used_home = True
            else:
                replicas.append(copy.deepcopy(model).to(d))
        for d in set(threads.devices):
            if d.type == "cuda":
                ops.ensure_selftest(d, model.rel_model)  # (a set-up step that waits: here, not in whichever thread ranks first)
                if index is not None:
                    index.device_arrays(d)  # moved once, here, not by whichever thread asks first
        world = threads.world
        _log.info(f"Evaluating on {world} device thread(s): {', '.join(str(d) for d in threads.devices)} ({threads.exchange} exchange)")
        shards = threads.run(lambda m: _shard_of_evaluation(replicas[m.rank], text_dataset, entities, emb_batch_size, m.device, _log, triples,
                                                            ent2idx, index, group=m, world=world, rank=m.rank, **common))
        del replicas
        table, axis, triples, counts, ids_ok = shards[0]
        sharded = world > 1
    else:
        sharded = group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        world = dist.get_world_size(group) if sharded else 1
        rank = dist.get_rank(group) if sharded else 0
        shards = None
        table, axis, triples, counts, ids_ok = _shard_of_evaluation(model, text_dataset, entities, emb_batch_size, device, _log, triples,
                                                                    ent2idx, index, group=group, world=world, rank=rank, **common)
    _log.info("Computing metrics on set of triples")
    if table.is_cuda:
        ops.release_workspaces()  # the evaluation's scratch does not stay pinned through the training steps that follow
    rr, hits = metrics_from_counts(counts)
    num_predictions = 2 * num_triples
    _log.info(f"The total number of predictions is {num_predictions:,}")
    denom = max(num_predictions, 1)
    sums = rr.double().sum(dim=0).tolist() if num_predictions else [0.0, 0.0]
    hit_sums = hits.double().sum(dim=0).tolist() if num_predictions else [[0.0] * 3, [0.0] * 3]
    assert ids_ok is None or bool(ids_ok), "a test triple names an entity that is not among the candidates"
    mrr, mrr_filt = sums[0] / denom, (sums[1] / denom if compute_filtered else 0.0)

    log_str = f"{prefix} mrr: {mrr:.4f}  "
    _run.log_scalar(f"{prefix}_mrr", mrr, epoch)
    for j, k in enumerate(HIT_POSITIONS):
        value = hit_sums[0][j] / denom
        log_str += f"hits@{k}: {value:.4f}  "
        _run.log_scalar(f"{prefix}_hits@{k}", value, epoch)
    if compute_filtered:
        log_str += f"mrr_filt: {mrr_filt:.4f}  "
        _run.log_scalar(f"{prefix}_mrr_filt", mrr_filt, epoch)
        for j, k in enumerate(HIT_POSITIONS):
            value = hit_sums[1][j] / denom
            log_str += f"hits@{k}_filt: {value:.4f}  "
            _run.log_scalar(f"{prefix}_hits@{k}_filt", value, epoch)
    _log.info(log_str)

    if compute_filtered and new_entities is not None:
        by_pos, pos_counts = utils.split_by_new_position(triples, rr[:, 1], new_entities)
        pos_counts[pos_counts < 1.0] = 1.0
        by_pos = by_pos / pos_counts
        log_str = ""
        for i, name in enumerate((f"{prefix}_mrr_filt_both_new", f"{prefix}_mrr_filt_head_new",
                                  f"{prefix}_mrr_filt_tail_new")):
            value = by_pos[i].item()
            log_str += f"{name}: {value:.4f}  "
            _run.log_scalar(name, value, epoch)
        _log.info(log_str)

    if compute_filtered and getattr(dataset, "has_rel_categories", False):
        from .data import CATEGORY_IDS
        by_cat, cat_count = utils.split_by_category(triples, rr[:, 1], dataset.rel_categories)
        cat_count[cat_count < 1.0] = 1.0
        by_cat = by_cat / cat_count
        for i, case in enumerate(["pred_head", "pred_tail"]):
            log_str = f"{case} "
            for cat, cat_id in CATEGORY_IDS.items():
                log_str += f"{cat}_mrr: {by_cat[i, cat_id]:.4f}  "
            _log.info(log_str)

    if was_training and eval_mode:
        model.train()
    if return_embeddings:
        if sharded and world > 1 and axis != "query":  # the full table only on request
            if shards is not None:  # device threads: the shards come to devices[0] by peer copies
                table = torch.cat([s[0].to(device) for s in shards])
            else:                   # processes: all-gather of the shards
                table = all_gather_rows(table, num_entities, world, group)
        return mrr, table.unsqueeze(0)
    return mrr, None
