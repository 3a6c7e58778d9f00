"""Worker of tests/test_gpu_shard.py::test_candidate_axis_two_gloo_ranks_take_the_fused_path, one process per rank
(torch.distributed.run, gloo backend: the ranks share one GPU and the collectives go through host memory -- the code path
is the RCCL one otherwise).  Prints one JSON line per rank.
    python -m torch.distributed.run --nproc-per-node 2 ... tests/shard_worker.py MODEL N T BLOCK"""
import json
import os
import sys

import torch
import torch.distributed as dist
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# aten ops that launch no kernel: views, metadata, allocation
FREE_OPS = {"slice", "view", "reshape", "empty", "empty_strided", "alias", "detach", "as_strided", "select", "unsqueeze", "squeeze",
            "expand", "_unsafe_view", "t", "transpose", "permute", "lift_fresh", "unbind", "sym_size", "sym_stride", "sym_numel",
            "sym_storage_offset", "is_pinned", "_local_scalar_dense_placeholder"}


class BlockLoopOps(TorchDispatchMode):
    """Logs the aten ops dispatched while the blocks are being ranked (ops.rank_all_batches: one library call for all the
    blocks of an evaluation) that are not views / allocations -- every one of them would be a torch kernel (a copy
    included)."""

    def __init__(self):
        super().__init__()
        self.inside, self.seen = False, []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)  # e.g. "aten.slice.Tensor"
        parts = name.split(".")
        base = parts[1] if len(parts) > 1 else name
        if self.inside and base not in FREE_OPS:
            self.seen.append(name)
        return func(*args, **(kwargs or {}))


def main():
    model_name, N, T, block = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from blp_amd import models, ops, ranking, utils
    from test_gpu_shard import _problem
    D, R = 128, 5
    table, rel_w, ent2idx, triples, edges = _problem(model_name, N, D, T, R, seed=11)
    index = utils.FilterIndex(edges, num_relations=R)
    model = models.LinkPrediction(D, model_name, "margin", R, 0)
    model.rel_emb.weight.data = rel_w.clone()
    model = model.cuda()
    dev_table, dev_triples, dev_e2i = table.cuda(), triples.cuda(), ent2idx.cuda()
    _, single, _ = ranking.rank_triples(model, dev_table, dev_triples, dev_e2i, index, block_size=block)
    lo, hi = ranking.shard_bounds(N, world, rank)
    shard = dev_table[lo:hi].contiguous()
    kw = dict(num_entities=N, world=world, rank=rank, axis="candidate", block_size=block)
    _, counts, ok = ranking.rank_triples(model, shard, dev_triples, dev_e2i, index, **kw)  # (also warms the allocator up)
    torch.cuda.synchronize()
    equal = bool(torch.equal(counts, single))

    # (0) the collectives of one evaluation: op and bytes one rank hands in, in the order issued (== ranking.exchange_plan)
    issued = []
    real_reduce, real_gather = ranking._all_reduce, ranking._all_gather_into

    def logged_reduce(tensor, group=None):
        issued.append(["all_reduce", tensor.numel() * tensor.element_size()])
        return real_reduce(tensor, group)

    def logged_gather(full, part, group=None):
        issued.append(["all_gather", part.numel() * part.element_size()])
        return real_gather(full, part, group)

    ranking._all_reduce, ranking._all_gather_into = logged_reduce, logged_gather
    try:
        ranking.rank_triples(model, shard, dev_triples, dev_e2i, index, **kw)
    finally:
        ranking._all_reduce, ranking._all_gather_into = real_reduce, real_gather
    torch.cuda.synchronize()

    # (1) which aten ops run while the blocks are issued
    log = BlockLoopOps()
    real = ops.rank_all_batches
    calls = []

    def traced(*a, **k):
        log.inside = True
        try:
            calls.append(1)
            return real(*a, **k)
        finally:
            log.inside = False

    ops.rank_all_batches = traced
    try:
        with log:
            ranking.rank_triples(model, shard, dev_triples, dev_e2i, index, **kw)
    finally:
        ops.rank_all_batches = real
    torch.cuda.synchronize()

    # (2) the device kernels of ONE block, from the profiler's trace of this process
    per_block, names = None, []
    try:
        from torch.profiler import ProfilerActivity, profile
        b = min(T, block)
        qb = ops.build_queries(dev_triples[:b], dev_e2i, dev_table, model.rel_emb.weight, block, index=index, gather=False,
                               row_base=lo)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            ops.rank_all_shard(model_name, shard, dev_table, qb.fixed_row, model.rel_emb.weight, qb.rel_ids, b, qb.true_row,
                               filter=qb.filter)
            torch.cuda.synchronize()
        kernels = [e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")
                   and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
        if kernels:
            per_block, names = len(kernels), [e.name[:60] for e in kernels]
    except Exception as exc:  # no device tracing in this build of torch: (1) still holds
        names = [f"profiler unavailable: {exc!r}"[:200]]
    source = "table" if N <= 2 * T else "vectors"
    print(json.dumps({"rank": rank, "counts_equal_single_process": equal, "ids_ok": bool(ok), "source": source,
                      "library_calls_for_all_blocks": len(calls), "collectives": issued, "torch_compute_ops_in_block_loop": sorted(set(log.seen)),
                      "kernels_per_block": per_block, "kernel_names": names}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
