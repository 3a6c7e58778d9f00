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


def _fabricated_result(world):
    """A full bench result with every section at its longest: all sub-results with their prose, per-rank lists of `world` ranks."""
    import bench
    roof = {"bound": "valu", "achieved": 128.5, "peak": bench.SAD_PEAK_TOPS, "unit": "TFLOP/s", "arith": "a" * 200, "peak_source": "p" * 300,
            "kernel": "k" * 150, "note": "n" * 800, "survey_8d_lane_ops": {"lane_ops_per_launch": 4.9e11, "achieved_Tops": 160.6, "peak_Tops": 78.65, "frac": 2.04},
            "frac": 0.8785, "kernel_ms": 3.0626399, "passes_per_launch": 1, "traffic": 317288448.0, "pmc_source": "s" * 140,
            "algorithmic_bytes_per_launch": 118260512, "algorithmic_flops_per_launch": 393616727040.0, "mfma_busy": 0.4971234}
    sub = {"value": 54206474300.589836, "ms_per_step": 1170.05949599843, "ms_per_step_raw_only": 1166.325209022034, "mrr": 4.377748104640011e-06,
           "mrr_filtered": 4.377748104640011e-06, "hits@1,3,10": [0.0, 0.0, 0.0], "hits@1,3,10_filtered": [0.0, 0.0, 0.0], "roofline": roof,
           "steps": 2, "timed_runs": 1, "unit": "scored triples/s", "parity_check": "64 queries vs CPU oracle (raw + filtered): identical counts",
           "table_passes_per_step": 3447, "ms_per_table_pass": 0.3394428476931912, "kernel_ms_per_rank": [0.123456789] * world,
           "exchange_ms": 1.23456789, "exchange_ms_per_rank": [1.23456789] * world, "shard_axis": "candidate", "ranks": world,
           "decided_frac": 0.87654321, "vs_random_step": 2.3456789}
    names = list(bench.SUB_RESULTS) + ([f"{n}@{a}" for n, a in bench.SUB_RESULTS_EXTRA_SHARDED] if world > 1 else [])
    result = {"metric": "scored triples/sec, all-entity eval (raw + filtered ranks, MRR + Hits@k)", "value": 489012345678.9012, "unit": "scored triples/s",
              "n_gpus": world, "steps": 10, "warmup": 2, "ms_per_step": 3.1443210987, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
              "dtype": "f32", "arith": roof["arith"], "data": "d" * 104,
              "config": {"workload": "fb15k237-transe", "rel_model": "transe", "entities": 14541, "dim": 128, "queries_per_step": 105740,
                         "triples_per_ranking_call": 52870, "filter_graph_edges": 310116, "step": "s" * 120, "parallelism": f"query-axis shards x{world}",
                         "shard_axis": "query", "ranks_in_process_group": world, "backend": "nccl"},
              "ms_per_step_raw_only": 3.0123456, "mrr": 0.000675208123, "mrr_filtered": 0.000675211123, "hits@1,3,10": [1.89143e-05] * 3,
              "hits@1,3,10_filtered": [1.89143e-05] * 3, "roofline": roof, "kernel_ms_per_rank": [2.9] * world, "exchange_ms": 0.5,
              "exchange_ms_per_rank": [0.5] * world, "shard_axis": "query", "ranks": world,
              "parity_check": "64 queries vs CPU oracle (raw + filtered): identical counts",
              "cpu_baseline": {"value": 3452465.3194224145, "unit": "scored triples/s", "cores": 1, "c_oracle": {"value": 20975265.59, "kind": "c" * 150, "sample": "x" * 50},
                               "kind": "port, vectorised filter (" + "k" * 90, "value_by_threads": {"1": 3.4e6, "8": 2.5e6, "32": 3.1e6, "128": 1.5e6},
                               "value_1_thread": 3.4e6, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "logical_cpus": 256, "sample": "s" * 200},
              "torch_gpu_baseline": {"value": 1.2e10, "unit": "scored triples/s", "kind": "k" * 90, "sample": "s" * 50},
              "sub_results": {n: dict(sub) for n in names},
              "inbatch_loss": {n: {"us_per_step_kernels": 22.123456, "us_per_step_autograd": 81.54321, "torch_us_per_step": 2959.36, "other": "o" * 300}
                               for n in bench.INBATCH_SHAPES},
              "call_overhead": {"ops_rank_all_us": 15.1, "library_call_us": 10.9519, "python_wrapper_us": 4.2, "rank_all_128_queries_fb15k237_us": 27.6805},
              "table_build": {"t" * 70: {"stock_us": 1.0, "fused_us": 0.5}},
              "hbm_probe": {"workload": "w" * 80, "bound": "hbm", "achieved": 6756.49, "peak": 8000.0, "unit": "GB/s", "frac": 0.844562, "kernel_ms": 0.348584,
                            "whole_call_ms": 0.36, "scored_triples_per_s": 5.1e10, "traffic": 2353090000.0, "pmc_source": "p" * 140},
              "details": "bench_details.json"}
    return result


@pytest.mark.parametrize("world", [1, 2, 8])
def test_stdout_line_stays_under_the_drivers_limit(world):
    """Round 4's single JSON line was 26 KB and the driver's reader lost it (BENCH_r04.parsed null).  The stdout line is now
    bench.compact_result(full): under 6 KB for every N, the contract's keys intact, `roofline` with bound / achieved / peak /
    unit / frac / traffic, `cpu_baseline` with value / unit / cores / kind / sample, each sub-result by value, ms_per_step and
    roofline fraction; the full object goes to the file the line names."""
    import json
    import bench
    full = _fabricated_result(world)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_result(full)
    text = json.dumps(line)
    assert len(text) < 6144, len(text)
    assert json.loads(text) == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check", "sub_results", "details"):
        assert key in line, key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["n_gpus"] == world
    assert line["config"]["workload"] == "fb15k237-transe" and "step" not in line["config"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"} and "note" not in line["roofline"]
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert set(line["sub_results"]) == set(full["sub_results"])
    for sub in line["sub_results"].values():
        assert sub["value"] > 0 and sub["frac"] > 0
        assert ("exchange_ms" in sub) == (world > 1)
    # a limit that cannot hold the optional sections: they go, the contract's keys stay
    small = bench.compact_result(full, limit=2500)
    assert len(json.dumps(small)) <= 2500 and "roofline" in small and "cpu_baseline" in small and small["value"] == full["value"]


def test_details_file_holds_the_full_result(tmp_path):
    import json
    import bench
    full = _fabricated_result(1)
    path = bench.write_details(full, str(tmp_path / "d.json"))
    assert json.load(open(path)) == full
    assert bench.write_details(full, str(tmp_path / "no" / "such" / "dir" / "d.json")) is None  # the line is still printed


def test_eight_rank_plan_without_a_gpu():
    """`python bench.py --gpus 8 --plan`: what the driver's 8-GPU run will do, computed here without a GPU -- the same
    sub-result names as one rank (+ the headline on the candidate axis), everything at Wikidata5M scale on the north_star's
    candidate axis with disjoint shards that cover the table, and per step exactly the exchanges of SURVEY 8e: the (2T, D) f32
    vectors of the queries replicated ONCE (one all-reduce) and ONE all-gather of the (2T, 4) int32 counts = G x Q x 16 bytes
    (1.76 MB at Q = 13 788, G = 8); the FB15k-237 evaluations on the query axis with one all-gather of per-triple counts."""
    import bench
    plan = bench.run_plan(8)
    assert set(plan) == {"fb15k237-transe", *bench.SUB_RESULTS, "fb15k237-transe@candidate"}
    for name, p in plan.items():
        cfg = bench.WORKLOADS[name.split("@")[0]]
        N, T, D = cfg["N"], cfg["triples"], cfg["D"]
        big = N > 1_000_000
        assert p["axis"] == ("candidate" if big or "@candidate" in name else "query"), name
        ex = p["exchanges_per_step"]
        if p["axis"] == "candidate":
            rows = p["table_rows_per_rank"]
            assert rows[0][0] == 0 and rows[-1][1] == N and all(a[1] == b[0] for a, b in zip(rows, rows[1:])), name
            assert max(hi - lo for lo, hi in rows) - min(hi - lo for lo, hi in rows) <= 8, name  # balanced shards
            assert all(t == [0, T] or t == (0, T) for t in p["triples_per_rank"]), name  # every rank ranks every query
            assert [e["op"] for e in ex] == (["all_reduce", "all_gather"] if big else ["all_gather", "all_gather"]), name
            assert ex[-1]["bytes_per_rank"] == 2 * T * 16 and ex[-1]["bytes_total"] == 8 * 2 * T * 16, name
            if big:
                assert ex[0]["bytes_total"] == 2 * T * D * 4, name  # the replicated vectors never grow with the table
                assert sum(p["table_bytes_per_rank"]) == N * D * (2 if cfg.get("table_dtype") else 4), name
        else:
            assert [e["op"] for e in ex] == ["all_gather"] and ex[0]["bytes_total"] == 8 * -(-T // 8) * 32, name
            trip = p["triples_per_rank"]
            assert trip[0][0] == 0 and trip[-1][1] == T and all(a[1] == b[0] for a, b in zip(trip, trip[1:])), name
    assert plan["wikidata5m-transe-block"]["exchanges_per_step"][-1]["bytes_total"] == 8 * 13788 * 16 == 1764864
    assert plan["wikidata5m-transe-full"]["timed_steps"] == [1, 0] or plan["wikidata5m-transe-full"]["timed_steps"] == (1, 0)
    assert bench.run_plan(1)["wikidata5m-transe-full"]["timed_steps"] in ([2, 1], (2, 1))
    assert all(p["exchanges_per_step"] == [] and p["axis"] == "none" for p in bench.run_plan(1).values())


def test_plan_flag_prints_json_without_a_gpu():
    import json
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plan"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    plan = json.loads(run.stdout)
    assert plan["wikidata5m-complex"]["axis"] == "candidate" and len(plan["wikidata5m-complex"]["table_rows_per_rank"]) == 4


def test_self_launched_run_is_bounded_in_time(tmp_path, monkeypatch):
    """bench.relaunch_with_ranks gives the launcher BLP_BENCH_TIMEOUT_S seconds: a rank that never finishes (a hung collective)
    is stopped with its whole process group, status 124, and the reason is on stderr."""
    import time
    import bench
    hang = tmp_path / "hang.py"
    hang.write_text("import time\ntime.sleep(600)\n")
    monkeypatch.setattr(bench, "__file__", str(hang))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    t0 = time.time()
    status = bench.relaunch_with_ranks(2, timeout_s=8)
    assert status == 124 and time.time() - t0 < 60


def test_multi_gpu_headline_is_the_candidate_axis_wikidata5m_ranking():
    """for N > 1 the TOP LEVEL of the line is the quantity the north_star's >= 6x target is defined on --
    Wikidata5M-scale TransE, the reference's batching, CANDIDATE axis, one all-gather of the int32 counts -- with `vs_1gpu`
    against the committed one-GPU figure; FB15k-237 (query axis and candidate axis) are sub-results.  One GPU: BASELINE.json's
    FB15k-237 configuration, as before."""
    import json
    import bench
    assert bench.headline_workload(1) == "fb15k237-transe"
    for world in (2, 4, 8):
        name = bench.headline_workload(world)
        assert name == "wikidata5m-transe"
        cfg = bench.WORKLOADS[name]
        assert cfg["N"] == 4_600_000 and cfg["D"] == 128 and cfg["block"] == 2 and cfg["model"] == "transe"
        assert bench.workload_axis(cfg, world) == "candidate"
        p = bench.run_plan(world)
        assert p[name]["axis"] == "candidate"
        assert [e["op"] for e in p[name]["exchanges_per_step"]] == ["all_reduce", "all_gather"]
        assert p[name]["exchanges_per_step"][-1]["bytes_total"] == world * 2 * cfg["triples"] * 16
        assert p["fb15k237-transe"]["axis"] == "query" and p["fb15k237-transe@candidate"]["axis"] == "candidate"
    ref = bench.n1_reference("wikidata5m-transe")
    assert ref and ref["value"] > 4.4e10 and "source" in ref  # (the 70 %-of-HBM target is 4.4e10 scored triples/s)
    committed = json.load(open(os.path.join(ROOT, bench.N1_REFERENCE_FILE)))
    assert {"wikidata5m-transe", "fb15k237-transe"} <= set(committed)
    # the compact line carries the new top-level fields
    result = _fabricated_result(8)
    result.update(vs_1gpu=6.4, kernel_ms_per_rank=[0.05] * 8, exchange_ms_per_rank=[0.1] * 8, filter_index_build_ms=1.5)
    line = bench.compact_result(result)
    assert line["vs_1gpu"] == 6.4 and len(line["kernel_ms_per_rank"]) == 8 and line["filter_index_build_ms"] == 1.5


def test_roofline_peaks_are_the_published_ones_with_the_measured_derate_beside():
    """`peak` = the guide's figure (157.3 Tops/s for the v_sad_u16 path = its VALU issue rate, 8 TB/s HBM, 2.5
    PF bf16 / 3), the builder-measured derate as `peak_measured` / `frac_measured` next to it."""
    import bench
    assert bench.SAD_PUBLISHED_TOPS == pytest.approx(157.3, abs=0.05)
    assert bench.SAD_PEAK_TOPS == pytest.approx(146.3, abs=0.05)
    assert bench.HBM_PEAK_GBPS == 8000.0 and bench.HBM_MEASURED_GBPS == 6290.0 and bench.BF16X3_PEAK_TFLOPS == pytest.approx(2500 / 3)
    roof = {"bound": "valu", "achieved": 129.2, "peak": bench.SAD_PUBLISHED_TOPS, "peak_measured": bench.SAD_PEAK_TOPS, "unit": "TFLOP/s",
            "frac": 129.2 / bench.SAD_PUBLISHED_TOPS, "frac_measured": 129.2 / bench.SAD_PEAK_TOPS, "kernel": "k", "arith": "a"}
    line = bench.compact_roofline(roof)
    assert line["peak"] == bench.SAD_PUBLISHED_TOPS and line["frac"] == pytest.approx(0.821, abs=2e-3)
    assert line["peak_measured"] == bench.SAD_PEAK_TOPS and line["frac_measured"] == pytest.approx(0.883, abs=2e-3)


def test_dry_nccl_preflight_under_gloo():
    """`bench.py --gpus 3 --dry-nccl` (here: gloo, host tensors): the group comes up, the three collectives of an evaluation
    run on 1 KB each with checked values, every rank reports on stderr, rank 0 prints one JSON line; a world size the launcher
    contradicts is refused as ever."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BLP_BENCH_BACKEND="gloo", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-nccl"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"dry_nccl": "ok", "ranks": 3, "backend": "gloo", "collectives": ["all_gather", "all_reduce", "all_gather"], "bytes_each": 1024}
    for r in range(3):
        assert f"rank {r} of 3 on cpu: ok" in run.stderr


def test_dry_nccl_names_the_failing_rank(monkeypatch, capsys):
    """A collective that fails is reported as "rank R: <step>: <error>" and the status is 3."""
    import bench
    from blp_amd import ranking

    def broken(tensor, group=None):
        raise RuntimeError("xGMI link down")

    monkeypatch.setattr(ranking, "all_gather_rows", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("xGMI link down")))
    status = bench.dry_run_collectives(torch.device("cpu"), 1, 0, "gloo", 1)
    assert status == 3
    assert "rank 0: all_gather (table rows): RuntimeError: xGMI link down" in capsys.readouterr().err
