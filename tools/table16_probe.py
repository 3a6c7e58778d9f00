"""Reference-batched passes over a Wikidata5M-scale table in float32 / float16 / bfloat16 storage: us per table pass.
    python tools/table16_probe.py [rows] [triples] [batch]
(blp_rank_all_batches, block_triples = batch: a pass per batch, all passes in one launch of a ring kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import _lib, ops  # noqa: E402
if os.environ.get("BLP_PROBE_LIB"):  # a variant build of the library (blp_amd.build.build(variant=...))
    _lib.LIB_PATH = os.path.abspath(os.environ["BLP_PROBE_LIB"])


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    D, R = 128, 822
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for model in ("transe", "complex", "distmult"):
        table = torch.randn(N, D, device=dev, generator=g)
        table = torch.nn.functional.normalize(table, dim=-1) if model == "transe" else table * 0.1
        rel_w = (torch.rand(R, D, device=dev, generator=g) - 0.5) * 0.25
        triples = torch.stack((torch.randint(0, N, (T,), device=dev, generator=g), torch.randint(0, N, (T,), device=dev, generator=g),
                               torch.randint(0, R, (T,), device=dev, generator=g)), dim=1)
        base = None
        for dtype in (torch.float32, torch.float16, torch.bfloat16):
            tab = table if dtype is torch.float32 else table.to(dtype)
            source = ops.gather_triple_vectors(triples, None, tab)
            qb = ops.build_queries(triples, None, source, rel_w, batch, gather=False, by_position=True, num_rows=N)

            def step():
                return ops.rank_all_batches(model, tab, qb.fixed_row, rel_w, qb.rel_ids, qb.true_row, T, batch, source=source,
                                            block_triples=batch)
            counts = step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                a.record()
                step()
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            passes = -(-T // batch)
            us = best * 1e3 / passes
            gb = N * D * tab.element_size() / 1e9
            if dtype is torch.float32:
                base = us
            print(f"{model:9s} {str(dtype):15s} {us:8.1f} us per pass  {gb / (us * 1e-6) / 1e3:6.2f} TB/s of table  "
                  f"{base / us:5.2f}x float32   mean gt {counts[:, 0].double().mean().item():.1f}", flush=True)
            del tab


if __name__ == "__main__":
    main()
