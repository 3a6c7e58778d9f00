"""GPU: the run-time guard behind the bilinear pre-pass's one empirical assumption (how
v_mfma_f32_32x32x16_bf16 rounds its accumulation).  The product no longer depends on a TEST having measured the matrix pipe:
once per device the library runs a self-test through the pre-pass's own MFMA sequence (include/blp_hip.h: blp_selftest) and
serves bilinear blocks by the f32-chain pre-pass -- provable without (A) -- when it fails or cannot be run."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_gpu_parity import dev, oracle_counts, random_csr, random_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BILINEAR = ("distmult", "complex", "simple")


@pytest.fixture(autouse=True)
def knobs_back_to_automatic():
    from blp_amd import _lib
    yield
    _lib.reset_knobs()
    if _lib._hooks is not None:  # leave the hooks library's verdict as a real one for the tests that follow
        _lib._hooks.blp_debug_reset_selftest(0)


def test_this_device_passes_and_says_so():
    from blp_amd import _lib
    assert _lib.selftest(0) is True
    caps = _lib.device_caps(0)
    print(f"matrix-pipe self-test: {caps['mfma_bf16_accum']}, worst |S~ - S3| / (262 u T) = {caps['mfma_bf16_accum_worst']:.4f}")
    assert caps["mfma_bf16_accum"] == "passed" and 0.0 <= caps["mfma_bf16_accum_worst"] < 0.5
    assert _lib.selftest(0) is True  # a second call returns the verdict, nothing runs


def test_first_bilinear_block_of_a_process_tests_the_device_itself():
    """A fresh process: no verdict; one DistMult block through ops.rank_all (no set-up call) -> the verdict is there, and the
    counts are the oracle's.  TransE blocks never ask for it."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from blp_amd import _lib, ops
from oracle import oracle as orc
from test_gpu_parity import oracle_counts, random_problem
orc.build()
assert _lib.device_caps(0)["mfma_bf16_accum"] == "not tested yet"
for model in ("transe", "distmult"):
    table, q_fixed, q_rel, true_row = random_problem(model, 6000, 128, 70, 60, seed=5)
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), 70, true_row=true_row.cuda()).cpu().numpy()
    assert np.array_equal(got, oracle_counts(orc, model, table, q_fixed, q_rel, 70, true_row=true_row)), model
    print(model, _lib.device_caps(0)["mfma_bf16_accum"])
''' % (ROOT, os.path.join(ROOT, "tests"))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith(("transe", "distmult"))]
    assert lines == ["transe not tested yet", "distmult passed"], run.stdout


@pytest.mark.parametrize("model", BILINEAR)
def test_forced_violation_routes_bilinear_blocks_to_the_f32_chain(oracle, knobs, model):
    """Knob mfma_selftest = 1 (hooks build): the self-test reports a violation whatever it measured.  Bilinear blocks then take
    the f32-chain pre-pass: identical counts (ties, a filter), blp_device_caps says FAILED, and the dump hook -- which
    exists in the bf16 kernel only -- is refused.  Back at 0 and tested again: the bf16 pre-pass serves, the same counts."""
    from blp_amd import _lib, ops
    N, q_head, q_tail = 5000, 330, 310
    table, q_fixed, q_rel, true_row = random_problem(model, N, 128, q_head, q_tail, seed=17)
    table[N // 3] = table[true_row[0]]
    table[N // 2] = table[true_row[q_head]]
    rowptr, col = random_csr(q_head + q_tail, N, true_row.numpy(), seed=3)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q_head, true_row=true_row, csr=(rowptr, col))
    t, f, r, tr = table.cuda(), q_fixed.cuda(), q_rel.cuda(), true_row.cuda()

    def run():
        return ops.rank_all(model, t, f, r, q_head, true_row=tr, filt_rowptr=dev(rowptr), filt_col=dev(col)).cpu().numpy()

    def dump_is_taken():
        s = torch.zeros(q_head + q_tail, N, device="cuda")
        e = torch.zeros_like(s)
        H = _lib.hooks_lib()
        _lib.check(H.blp_debug_gemm_dump(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(e.data_ptr())), "blp_debug_gemm_dump", H)
        try:
            ops.rank_all(model, t, f, r, q_head, true_row=tr)
        except RuntimeError:
            return False
        torch.cuda.synchronize()
        return bool(e.abs().sum() > 0)

    knobs("mfma_selftest", 1)
    H = _lib.hooks_lib()
    assert H.blp_debug_reset_selftest(0) == 0
    assert _lib.device_caps(0)["mfma_bf16_accum"] == "not tested yet"
    assert np.array_equal(run(), want)
    caps = _lib.device_caps(0)
    assert caps["mfma_bf16_accum"].startswith("FAILED") and caps["mfma_bf16_accum_worst"] == float("inf")
    assert _lib.selftest(0) is False
    assert not dump_is_taken()
    assert np.array_equal(run(), want)
    # the verdict is per process and device, not per call: still the f32 chain with the knob gone ...
    knobs("mfma_selftest", 0)
    with _lib.use_hooks_library():
        assert _lib.device_caps(0)["mfma_bf16_accum"].startswith("FAILED")
        # ... until the device is tested again: passes, bf16 pre-pass, the same counts
        assert H.blp_debug_reset_selftest(0) == 0
        assert np.array_equal(run(), want)
        assert _lib.device_caps(0)["mfma_bf16_accum"] == "passed"
        assert dump_is_taken()


def test_no_verdict_while_the_stream_is_captured(oracle):
    """A bilinear block captured into a graph before the device was ever tested: the self-test cannot wait for a capturing
    stream, so that call is served by the f32-chain pre-pass (no host synchronisation inside the capture), the verdict stays
    open, and the replayed graph gives the oracle's counts."""
    from blp_amd import _lib, ops
    model, N, q = "complex", 4000, 96
    table, q_fixed, q_rel, true_row = random_problem(model, N, 128, q, q, seed=23)
    want = oracle_counts(oracle, model, table, q_fixed, q_rel, q, true_row=true_row)
    bufs = [x.cuda() for x in (table, q_fixed, q_rel, true_row)]
    out = torch.full((2 * q, 4), -1, dtype=torch.int32, device="cuda")
    with _lib.use_hooks_library():
        H = _lib.hooks_lib()
        ops.rank_all(model, *bufs[:3], q, true_row=bufs[3], out=out)  # (workspace grown, code objects loaded before the capture)
        torch.cuda.synchronize()
        assert H.blp_debug_reset_selftest(0) == 0
        out.fill_(-1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                ops.rank_all(model, *bufs[:3], q, true_row=bufs[3], out=out)
        assert _lib.device_caps(0)["mfma_bf16_accum"] == "not tested yet"
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        assert _lib.selftest(0) is True
