"""GPU: the dispatch the library SHIPS with.  No knob is set anywhere in this file: every call goes to libblp_hip.so and
takes whatever kernel the thresholds in rank_common.h / sad_common.h / rank_stream.hip / rank_gemm.hip select -- block
shapes that sit ON each threshold (one query or one tile either side), a seeded slice of the randomised soak
(tools/fuzz_parity.py --shipped), and the golden vectors -- all against the CPU oracle (utils.py:103-105 counts,
train.py:159-167 filtered)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REL_MODELS
from test_gpu_parity import dev, oracle_counts, random_csr, random_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def product_library_only():
    from blp_amd import _lib
    _lib.reset_knobs()
    yield
    assert not _lib._knobs_set and _lib.lib() is _lib._product


# (model, D, N, q_head, q_tail): what takes the block on either side of a threshold
BOUNDARIES = [
    # TransE, table of <= 1 024 tiles: small-block scalar-register kernel below 6 M pairs, SAD pre-pass from there
    ("transe", 128, 14541, 200, 212), ("transe", 128, 14541, 200, 213),      # 5.991 M | 6.005 M pairs
    ("transe", 128, 14477, 200, 213), ("transe", 128, 14605, 200, 213),      # one tile fewer | more at 413 queries
    ("transe", 64, 65536, 45, 46), ("transe", 64, 65536, 46, 46),            # 1 024 tiles: 5.96 M | 6.03 M pairs
    # more than 1 024 tiles: register-tile small kernel below 4 M pairs; above: rank_tiles under 64 queries, SAD from 64 on
    ("transe", 64, 65600, 30, 30), ("transe", 64, 65600, 30, 31), ("transe", 64, 65600, 32, 32),
    # the 64-query floor of the SAD pre-pass on a table that qualifies by pairs
    ("transe", 128, 100000, 31, 32), ("transe", 128, 100000, 32, 32),
    # bilinear: small-block kernel below 0.4 M pairs; above: rank_tiles under 32 queries, MFMA pre-pass from 32 on
    ("distmult", 128, 14541, 13, 14), ("distmult", 128, 14541, 14, 14), ("complex", 128, 14541, 15, 16),
    ("simple", 128, 14541, 16, 16), ("complex", 64, 5000, 40, 39), ("distmult", 64, 5000, 40, 40),
    # <= 4 + 4 queries: streaming kernels on tables of more than 256 tiles, small-block kernel up to 256 tiles
    ("transe", 128, 16384, 4, 4), ("transe", 128, 16385, 4, 4), ("complex", 128, 16384, 4, 4), ("complex", 128, 16385, 4, 4),
    ("transe", 128, 20000, 5, 4), ("distmult", 128, 20000, 4, 5), ("transe", 128, 500000, 5, 4), ("simple", 64, 300000, 4, 4),
    # 256-tile slot limit of the bilinear small-block kernel (tiles share slots above it)
    ("distmult", 128, 16384, 10, 10), ("distmult", 128, 16385, 10, 10), ("simple", 128, 16448, 9, 12),
    # true keys: one lane per query up to 2 048 queries (in the prelude launch of the MFMA path), the cooperative kernel above
    ("transe", 128, 3000, 1024, 1024), ("transe", 128, 3000, 1024, 1025),
    ("distmult", 128, 3000, 1024, 1024), ("complex", 128, 3000, 1025, 1024),
    # 4 096 queries: the most the small-block kernels take whatever the table
    ("transe", 128, 70, 2048, 2048), ("transe", 128, 70, 2048, 2049), ("simple", 128, 70, 2048, 2048), ("simple", 128, 70, 2049, 2048),
    # D = 256: TransE SAD / exact kernels, bilinear on rank_tiles
    ("transe", 256, 9000, 300, 300), ("complex", 256, 3000, 40, 40),
]


@pytest.mark.parametrize("model,D,N,q_head,q_tail", BOUNDARIES)
def test_blocks_on_the_dispatch_thresholds_vs_oracle(oracle, model, D, N, q_head, q_tail):
    from blp_amd import ops
    table, q_fixed, q_rel, true_row = random_problem(model, N, D, q_head, q_tail, seed=N % 1000 + q_head)
    table[N // 3] = table[true_row[0]]  # a tie with a true entity: the >= count must see it on every path
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=D + q_tail)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, true_row=true_row.cuda(),
                       filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("N", [1_700_000 - 64, 1_700_000, 1_700_000 + 64])
def test_transe_stream_kernel_switch_at_1_7_million_rows(oracle, N):
    """4 queries against a long table: the per-wave ring kernel below 1.7 M rows, the workgroup-tile kernel from there on."""
    from blp_amd import ops
    D, q_head, q_tail = 64, 2, 2
    g = torch.Generator(device="cuda").manual_seed(N)
    table = torch.nn.functional.normalize(torch.randn(N, D, device="cuda", generator=g), dim=-1)
    fixed_row = torch.tensor([5, N - 1, N // 2, 77], device="cuda")
    true_row = torch.tensor([N - 2, 3, 1_000_000, N - 64], device="cuda")
    q_rel = (torch.rand(4, D, device="cuda", generator=g) - 0.5) * 0.2
    got = ops.rank_all("transe", table, table[fixed_row].contiguous(), q_rel, q_head, true_row=true_row).cpu().numpy()
    host = table.cpu()
    want = oracle_counts(oracle, "transe", host, host[fixed_row.cpu()], q_rel.cpu(), q_head, true_row=true_row.cpu())
    assert np.array_equal(got, want)


def test_seeded_slice_of_the_randomised_soak_at_the_shipped_dispatch(capsys):
    """320 cases of tools/fuzz_parity.py in its --shipped mode (no knobs; shapes drawn around the thresholds; data kinds:
    ties, NaN / Inf, outliers, tiny / mixed scales, constant tables; CSR filters, strided tables, vector / index query
    forms): every count equals the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    n, bad = fuzz_parity.run(budget=None, seed0=31000, max_cases=320, shipped=True)
    out = capsys.readouterr().out
    assert n == 320 and bad == 0, out[-3000:]
