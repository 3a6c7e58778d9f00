"""bench.py's host logic that needs no GPU: the self-launch of `python bench.py --gpus N` and the rank-local table rows."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_local_table_chunks_add_up_to_the_whole_table():
    """A candidate-axis rank generates ONLY its rows of a Wikidata5M-scale table (seeded chunks): the shards of any world
    size, concatenated, are the table one rank generates -- so `bench.py --gpus N` ranks the same data for every N."""
    import bench
    from blp_amd import ranking
    cfg = dict(bench.WORKLOADS["wikidata5m-transe"], N=3 * bench.TABLE_CHUNK_ROWS + 1234, D=8)
    dev = torch.device("cpu")
    whole = bench.make_table_rows(cfg, dev, 0, cfg["N"], seed=1)
    assert whole.shape == (cfg["N"], 8) and torch.allclose(whole.norm(dim=1), torch.ones(cfg["N"]), atol=1e-5)
    for world in (2, 3, 8):
        parts = [bench.make_table_rows(cfg, dev, *ranking.shard_bounds(cfg["N"], world, r), seed=1) for r in range(world)]
        assert torch.equal(torch.cat(parts), whole), world
    assert not torch.equal(bench.make_table_rows(cfg, dev, 0, 100, seed=2), whole[:100])
    bilinear = dict(cfg, model="complex")
    assert bench.make_table_rows(bilinear, dev, 5, 9, seed=1).abs().max() < 1.0  # 0.1 * randn, not normalised


def test_sub_result_names_are_the_same_for_every_world_size():
    import bench
    assert all(name in bench.WORKLOADS for name in bench.SUB_RESULTS)
    assert {"wikidata5m-transe-full", "wikidata5m-complex-full", "wikidata5m-protocol", "wikidata5m-complex-block"} <= set(bench.SUB_RESULTS)
    full = bench.WORKLOADS["wikidata5m-transe-full"]
    assert full["triples"] == 6894 and full["block"] == 2 and full["N"] == 4_600_000  # 3 447 table passes per step


@pytest.mark.skipif(torch.cuda.is_available(), reason="on a GPU box the launched ranks would run the whole bench (test_gpu_eval covers it)")
def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it (how the driver's bench command is shaped) re-executes
    itself under torch.distributed.run; here, without a GPU, the ranks get as far as the device check and say so (the
    launcher stops the others when the first one exits), and the launcher's non-zero status comes back -- round 3's bench.py gave up before starting anything."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert run.returncode != 0
    assert run.stderr.count("bench.py needs a HIP device") >= 1 and "torch/distributed/elastic" in run.stderr, run.stderr[-3000:]
    assert "needs torch.distributed.run" not in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
    # a launcher's world size that contradicts --gpus is still refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "does not match --gpus" in bad.stderr
