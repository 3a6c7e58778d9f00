"""Host-side helpers with the reference's names and semantics (/root/reference/utils.py), plus the
vectorised filter index the HIP ranking consumes.

    get_model            utils.py:6-28     string -> model class
    make_ent2idx         utils.py:31-43
    get_triple_filters   utils.py:46-83    dense (B, N) masks -- kept for drop-in callers
    FilterIndex          (new)             one-time sorted index of the filtering graph -> per-batch CSR
    get_metrics          utils.py:86-111
    split_by_new_position / split_by_category   utils.py:114-168
    get_logger           utils.py:171-182
"""
import logging

import torch

from . import models


def get_model(model, dim, rel_model, loss_fn, num_entities, num_relations, encoder_name, regularizer):
    if model == "blp":
        return models.BertEmbeddingsLP(dim, rel_model, loss_fn, num_relations, encoder_name, regularizer)
    elif model == "bert-bow":
        return models.BOW(rel_model, loss_fn, num_relations, regularizer, encoder_name=encoder_name)
    elif model == "bert-dkrl":
        return models.DKRL(dim, rel_model, loss_fn, num_relations, regularizer, encoder_name=encoder_name)
    elif model == "glove-bow":
        return models.BOW(rel_model, loss_fn, num_relations, regularizer, embeddings="data/glove/glove.6B.300d.pt")
    elif model == "glove-dkrl":
        return models.DKRL(dim, rel_model, loss_fn, num_relations, regularizer,
                           embeddings="data/glove/glove.6B.300d.pt")
    elif model == "transductive":
        return models.TransductiveLinkPrediction(dim, rel_model, loss_fn, num_entities, num_relations, regularizer)
    else:
        raise ValueError(f"Unkown model {model}")


def make_ent2idx(entities, max_ent_id):
    """Tensor indexed by entity id holding the entity's position in ``entities`` (-1 if absent).

    >>> make_ent2idx(torch.tensor([4, 5, 0]), 5)
    tensor([ 2, -1, -1, -1,  0,  1])
    """
    idx = torch.arange(entities.shape[0])
    ent2idx = torch.empty(int(max_ent_id) + 1, dtype=torch.long).fill_(-1)
    ent2idx.scatter_(0, entities, idx)
    return ent2idx


class FilterIndex:
    """Sorted index of the filtering graph: (head, rel) -> known tails and (tail, rel) -> known heads.

    The reference walks a networkx MultiDiGraph in Python for every evaluation batch (utils.py:46-83,
    ~190 ms per 64 triples at FB15k-237 size).  This builds two sorted key arrays once -- a sort + unique of the
    packed (key, value) pairs ON THE DEVICE the evaluation runs on when ``device`` is given (the host sort cost
    75-84 ms per FB15k-237 graph next to a 3 ms evaluation, and seconds for the 20 M edges of a Wikidata5M training
    graph), else on the host -- and answers a batch with searchsorted + one gather, producing either the reference's
    dense masks or the CSR lists / segments the HIP ranking kernel takes.  Parallel edges collapse (a mask entry
    is set once however many times the edge occurs).
    """

    def __init__(self, edges, num_relations=None, device=None):
        """edges: (E, 3) int64 rows (head, tail, rel) -- the reference's triple column order
        (data.py:128) -- or a networkx graph built with add_weighted_edges_from(triples).
        device: where to build (and keep) the index; None = where ``edges`` lives (a CPU tensor: the host)."""
        if not isinstance(edges, torch.Tensor):
            rows = [(h, t, w) for h, t, w in edges.edges(data="weight")]
            edges = torch.tensor(rows, dtype=torch.long).reshape(-1, 3)
        edges = edges.to(device=device if device is not None else edges.device, dtype=torch.long)
        self.home = edges.device
        self.num_edges = edges.shape[0]
        self.max_node = int(edges[:, :2].max()) if self.num_edges else -1
        seen = int(edges[:, 2].max()) + 1 if self.num_edges else 1
        # keys pack (entity, relation) as entity * R + relation; a queried relation >= R (one the graph never
        # saw, e.g. Wikidata5M's per-split graphs) matches nothing instead of aliasing onto another key
        self.R = max(int(num_relations), seen) if num_relations is not None else seen
        h, t, r = edges[:, 0], edges[:, 1], edges[:, 2]
        self.tails_key, self.tails_val = self._build(h * self.R + r, t)   # (h, r) -> tails
        self.heads_key, self.heads_val = self._build(t * self.R + r, h)   # (t, r) -> heads

    @staticmethod
    def _build(key, val):
        if key.numel() == 0:
            return key, val
        span = int(val.max()) + 1
        packed = torch.unique(key * span + val)  # sorted, duplicates (parallel edges) removed
        return torch.div(packed, span, rounding_mode="floor"), packed % span

    @staticmethod
    def _lookup(sorted_key, sorted_val, key, exclude, ent2idx):
        """For each query key: the values under it, minus `exclude`, mapped through ent2idx, minus -1.
        Returns (counts per query, flat mapped values).  Device-agnostic: all tensors on one device."""
        dev = key.device
        lo = torch.searchsorted(sorted_key, key, right=False)
        hi = torch.searchsorted(sorted_key, key, right=True)
        n = hi - lo
        total = int(n.sum())
        if total == 0:
            return torch.zeros_like(n), torch.empty(0, dtype=torch.long, device=dev)
        owner = torch.repeat_interleave(torch.arange(key.shape[0], device=dev), n, output_size=total)
        first = torch.cumsum(n, 0) - n
        pos = (lo - first)[owner] + torch.arange(total, device=dev)
        vals = sorted_val[pos]
        in_range = vals < ent2idx.shape[0]
        mapped = torch.where(in_range, ent2idx[torch.where(in_range, vals, torch.zeros_like(vals))],
                             torch.full_like(vals, -1))
        keep = (vals != exclude[owner]) & (mapped >= 0)
        owner, mapped = owner[keep], mapped[keep]
        counts = torch.bincount(owner, minlength=key.shape[0])
        return counts, mapped

    def _keys(self, triples):
        """Packed lookup keys of a batch: (tail, rel) for the head side, (head, rel) for the tail side; -1 (which no
        edge has) for relations outside the index."""
        h, t, r = triples[:, 0], triples[:, 1], triples[:, 2]
        known = (r >= 0) & (r < self.R)
        return torch.where(known, t * self.R + r, -1), torch.where(known, h * self.R + r, -1)

    def segments(self, triples, ent2idx, device, row_base=0):
        """The filter of a batch as a blp_amd.ops.SegmentFilter over 2B queries in the ranking order (B head-
        replacing, then B tail-replacing): four binary searches on the device, no list is built and the host
        never waits.  ``ent2idx`` should already live on ``device`` (it is moved if not)."""
        from .ops import SegmentFilter
        device = torch.device(device)
        heads_key, tails_key, values, n_head_vals = self.device_arrays(device)
        triples = triples.to(device=device, dtype=torch.long)
        key_head, key_tail = self._keys(triples)
        lo = torch.cat((torch.searchsorted(heads_key, key_head, right=False),
                        torch.searchsorted(tails_key, key_tail, right=False) + n_head_vals))
        hi = torch.cat((torch.searchsorted(heads_key, key_head, right=True),
                        torch.searchsorted(tails_key, key_tail, right=True) + n_head_vals))
        exclude = torch.cat((triples[:, 0], triples[:, 1]))
        return SegmentFilter(lo, hi, values, exclude, ent2idx.to(device), row_base)

    def device_arrays(self, device):
        """(heads_key, tails_key, [heads' values | tails' values], number of head values) on ``device``, cached: what
        the segment form of the filter indexes (ops.SegmentFilter, blp_build_queries)."""
        cache = self.__dict__.setdefault("_segment_cache", {})
        device = torch.device(device)
        if device not in cache:
            heads_key, heads_val, tails_key, tails_val = self._on(device)
            cache[device] = (heads_key.contiguous(), tails_key.contiguous(), torch.cat((heads_val, tails_val)).contiguous(),
                             heads_val.shape[0])
        return cache[device]

    def _on(self, device):
        """The four sorted arrays on ``device`` (moved once, cached)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self.home:
            return self.heads_key, self.heads_val, self.tails_key, self.tails_val
        cache = self.__dict__.setdefault("_device_cache", {})
        if device not in cache:
            cache[device] = tuple(x.to(device) for x in (self.heads_key, self.heads_val, self.tails_key, self.tails_val))
        return cache[device]

    def csr(self, triples, ent2idx, device="cpu"):
        """CSR over 2B queries in the ranking order (B head-replacing queries, then B tail-replacing):
        rowptr (2B + 1,), cols (nnz,) = table rows removed by the filtered setting for each query.
        With a HIP ``device`` the lookup (searchsorted + gathers) runs there and the CSR stays there."""
        heads_key, heads_val, tails_key, tails_val = self._on(device)
        triples = triples.to(device=device, dtype=torch.long)
        ent2idx = ent2idx.to(device)
        h, t = triples[:, 0], triples[:, 1]
        key_head, key_tail = self._keys(triples)
        n_head, col_head = self._lookup(heads_key, heads_val, key_head, h, ent2idx)
        n_tail, col_tail = self._lookup(tails_key, tails_val, key_tail, t, ent2idx)
        rowptr = torch.zeros(2 * triples.shape[0] + 1, dtype=torch.long, device=triples.device)
        rowptr[1:] = torch.cumsum(torch.cat((n_head, n_tail)), 0)
        return rowptr, torch.cat((col_head, col_tail))

    def masks(self, triples, num_ents, ent2idx):
        rowptr, cols = self.csr(triples, ent2idx)
        b = triples.shape[0]
        mask = torch.zeros((2 * b, num_ents), dtype=torch.bool)
        rows = torch.repeat_interleave(torch.arange(2 * b), rowptr[1:] - rowptr[:-1])
        mask[rows, cols] = True
        return mask[:b], mask[b:]


def get_triple_filters(triples, graph, num_ents, ent2idx):
    """Dense (B, num_ents) bool masks of known-true replacement heads / tails (reference semantics:
    same relation, different entity than the triple's own, candidates only).  ``graph`` may be the
    reference's networkx MultiDiGraph or a FilterIndex (built once per evaluation)."""
    index = graph if isinstance(graph, FilterIndex) else _cached_index(graph)
    return index.masks(triples, num_ents, ent2idx)


def _cached_index(graph):
    index = graph.graph.get("_blp_filter_index") if hasattr(graph, "graph") else None
    if index is None or index.num_edges != graph.number_of_edges():
        index = FilterIndex(graph)
        if hasattr(graph, "graph"):
            graph.graph["_blp_filter_index"] = index
    return index


def get_metrics(pred_scores, true_idx, k_values):
    """Reciprocal of the realistic rank (mean of optimistic and pessimistic) and hits@k.

    pred_scores (B, N), true_idx (B, 1), k_values (1, k) -> reciprocals (B, 1) f32, hits (B, k) bool.
    Dense-matrix form kept for drop-in callers; the evaluation loop of this package never builds the
    (B, N) matrix on a HIP device (blp_amd.ranking).
    """
    true_scores = pred_scores.gather(dim=1, index=true_idx)
    best_rank = (pred_scores > true_scores).sum(dim=1, keepdim=True) + 1
    worst_rank = (pred_scores >= true_scores).sum(dim=1, keepdim=True)
    average_rank = (best_rank + worst_rank).float() * 0.5
    return average_rank.reciprocal(), average_rank <= k_values


def split_by_new_position(triples, mrr_values, new_entities):
    """MRR broken down by where the new (unseen-in-training) entity sits: both / head / tail (utils.py:114-148).
    mrr_values has 2B entries (head-replacing first); a triple's value is the mean of its two.  Runs on the
    device of ``mrr_values`` (one masked sum instead of a Python loop with two .item() calls per triple)."""
    num_triples = triples.shape[0]
    dev = mrr_values.device
    values = mrr_values.detach().to(torch.float64).reshape(-1)
    per_triple = (values[:num_triples] + values[num_triples:2 * num_triples]) / 2.0
    new = torch.tensor(sorted(new_entities), dtype=torch.long, device=dev)
    triples = triples.to(dev)
    head_new = torch.isin(triples[:, 0], new)
    tail_new = torch.isin(triples[:, 1], new)
    groups = torch.stack((head_new & tail_new, head_new & ~tail_new, ~head_new & tail_new)).to(torch.float64)
    return (groups @ per_triple).float(), groups.sum(dim=1).float()


def split_by_category(triples, mrr_values, rel_categories):
    num_triples = triples.shape[0]
    device = mrr_values.device
    cats = rel_categories.to(device)[triples[:, 2].to(device)]
    mrr_by_category = torch.zeros([2, 4], device=device)
    mrr_by_category[0].index_add_(0, cats, mrr_values[:num_triples].float())
    mrr_by_category[1].index_add_(0, cats, mrr_values[num_triples:2 * num_triples].float())
    mrr_cat_count = torch.zeros([1, 4], dtype=torch.float, device=device)
    mrr_cat_count[0].index_add_(0, cats, torch.ones(num_triples, device=device))
    return mrr_by_category, mrr_cat_count


def get_logger():
    """Root logger with an HH:MM:SS timestamp (utils.py:171-182)."""
    logger = logging.getLogger("")
    logger.handlers = []
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(name)s - %(message)s", datefmt="%H:%M:%S"))
    logger.addHandler(handler)
    logger.setLevel("INFO")
    return logger
