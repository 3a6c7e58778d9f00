"""End-to-end timing of blp_amd.ranking.eval_link_prediction (raw + filtered) at FB15k-237 size on one
GPU: 14 541 entities (transductive embedding table, so no text encoder in the way), 52 870 test
triples, 310 116 filter edges.  Shows what the host side (filter CSR, batching, metric splits) costs
next to the ranking kernels, i.e. SURVEY 8f rows 1 and 3.
    python tools/bench_eval_loop.py [rel_model]"""
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import models, ranking, utils  # noqa: E402


class _Run:
    def log_scalar(self, *a):
        pass


class _Triples(torch.utils.data.Dataset):
    """The shape of data.GraphDataset as the eval loop sees it: one (T, 3) tensor, indexed per triple."""

    def __init__(self, triples):
        self.triples = triples

    def __getitem__(self, i):
        return self.triples[i]

    def __len__(self):
        return self.triples.shape[0]


def main():
    rel_model = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "transe"
    N, R, T, E, D = 14541, 237, 52870, 310116, 128
    g = torch.Generator().manual_seed(0)
    model = models.TransductiveLinkPrediction(D, rel_model, "margin", N, R, 0).cuda().eval()
    test = torch.stack((torch.randint(0, N, (T,), generator=g), torch.randint(0, N, (T,), generator=g),
                        torch.randint(0, R, (T,), generator=g)), dim=1)
    edges = torch.cat((test, torch.stack((torch.randint(0, N, (E - T,), generator=g),
                                          torch.randint(0, N, (E - T,), generator=g),
                                          torch.randint(0, R, (E - T,), generator=g)), dim=1)))
    loader = torch.utils.data.DataLoader(_Triples(test), batch_size=64)  # train.py:126-128
    t0 = time.perf_counter()
    index = utils.FilterIndex(edges, num_relations=R)
    t_index = time.perf_counter() - t0
    log = logging.getLogger("bench_eval")
    for name, filt in (("raw", None), ("raw + filtered", index)):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mrr, _ = ranking.eval_link_prediction(model, loader, None, None, 0, 512, _Run(), log, prefix="test",
                                                  filtering_graph=filt)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{rel_model:9s} {name:15s}: {dt * 1e3:8.1f} ms for {T} triples x {N} entities "
              f"({2 * T * N / dt / 1e9:6.1f} G scored triples/s end to end), mrr {mrr:.5f}")
    print(f"FilterIndex build ({E} edges): {t_index * 1e3:.1f} ms (once per evaluation graph)")
    if "--profile" in sys.argv:  # where the host side of one evaluation goes
        import cProfile
        import pstats
        prof = cProfile.Profile()
        torch.argsort(test[:, 2], stable=True)  # (the first CPU sort inside a profiler pays a one-off 80 ms)
        prof.enable()
        ranking.eval_link_prediction(model, loader, None, None, 0, 512, _Run(), log, prefix="test", filtering_graph=index)
        torch.cuda.synchronize()
        prof.disable()
        pstats.Stats(prof).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
