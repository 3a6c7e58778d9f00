"""The C-ABI library builds, loads and exports exactly what include/blp_hip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from blp_amd import build
    build.build_hooks()
    return build.build()


def _declared_symbols(hooks=False):
    """Functions include/blp_hip.h declares: outside (product) or inside (test build) its #ifdef BLP_TEST_HOOKS block."""
    text = open(os.path.join(ROOT, "include", "blp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef BLP_TEST_HOOKS(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef BLP_TEST_HOOKS.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blp_[a-z_0-9]+)\s*\(", inside if hooks else outside)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("blp_"))


def test_header_symbols_match_binding(built_lib):
    from blp_amd import _lib
    assert _declared_symbols() == sorted(_lib.SYMBOLS)
    assert _declared_symbols(hooks=True) == sorted(_lib.HOOK_SYMBOLS)


def test_product_library_exports_no_test_hooks(built_lib):
    """libblp_hip.so exports exactly the production ABI -- no knob, no dump hook, hence no mutable process-wide state;
    libblp_hip.hooks.so (tests / tools) adds the two hooks."""
    from blp_amd import _lib
    assert _exported(_lib.LIB_PATH) == sorted(_lib.SYMBOLS)
    assert _exported(_lib.HOOKS_LIB_PATH) == sorted(_lib.SYMBOLS + _lib.HOOK_SYMBOLS)


def test_library_loads_and_exports_every_symbol(built_lib):
    from blp_amd import _lib
    L = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.blp_version() == 60000
    assert L.blp_dim_supported(0, 128) == 1
    assert L.blp_dim_supported(0, 100) == 0
    assert L.blp_rank_all_workspace_bytes(0, 14541, 128, 64, 64) >= 128 * (256 * 4 + 4 + 8)  # coefficients, key, accumulator
    # blocks blp_rank_all takes: the compiled widths always; TransE at any D % 4 == 0 up to 1024
    assert L.blp_rank_all_supported(0, 128, 1, 1) == 1 and L.blp_rank_all_supported(2, 128, 1, 1) == 1
    assert L.blp_rank_all_supported(0, 300, 150, 170) == 1 and L.blp_rank_all_supported(0, 768, 256, 0) == 1
    assert L.blp_rank_all_supported(0, 300, 10, 10) == 1 and L.blp_rank_all_supported(1, 300, 150, 170) == 0
    assert L.blp_rank_all_supported(0, 302, 150, 170) == 0 and L.blp_rank_all_supported(0, 2048, 150, 170) == 0
    assert L.blp_rank_all_workspace_bytes(0, 14541, 768, 300, 300) > 14541 * 768 * 2  # holds the 2-byte table image


def test_bad_arguments_return_status_not_crash(built_lib):
    from blp_amd import _lib
    L = _lib.lib()
    # argument validation happens before any device call, so this runs without a GPU
    rc = L.blp_rank_all(7, None, 0, 128, 128, None, None, None, None, 1, 1, None, None, None, 0, 0, None)
    assert rc == -1 and b"unknown model" in L.blp_last_error()
    rc = L.blp_rank_all(1, None, 0, 100, 100, None, None, None, None, 1, 1, None, None, None, 0, 0, None)
    assert rc == -2  # (DistMult at D = 100; TransE is taken at any D % 4 == 0)
    rc = L.blp_rank_all(0, None, 0, 102, 104, None, None, None, None, 1, 1, None, None, None, 0, 0, None)
    assert rc == -2
    rc = L.blp_score_fwd(1, 100, 1, 1, None, 0, 0, None, 0, 0, None, 0, 0, None, 0, None)
    assert rc == -1 and b"NULL" in L.blp_last_error()  # any width is scored; the pointers are what is wrong here
    rc = L.blp_score_fwd(2, 101, 1, 1, None, 0, 0, None, 0, 0, None, 0, 0, None, 0, None)
    assert rc == -2 and b"halves" in L.blp_last_error()  # ComplEx / SimplE split the vector
    one = ctypes.c_void_p(16)  # non-NULL placeholders: validation fails before anything is dereferenced
    rc = L.blp_inbatch_loss_fwd(0, 0, 5, 0, one, one, one, 4, 4, 128, 0.0, one, one, one, one, 0, None)
    assert rc == -1 and b"dtype" in L.blp_last_error()
    rc = L.blp_inbatch_loss_fwd(0, 0, 0, 1, one, one, one, 4, 4, 128, 0.0, one, one, one, one, 0, None)
    assert rc == -1  # relation rows narrower than the embeddings
    rc = L.blp_inbatch_loss_fwd(0, 0, 0, 0, one, one, one, 4, 4, 128, 0.0, one, one, one, None, 0, None)
    assert rc == -1 and b"ticket" in L.blp_last_error()  # the forward's last-workgroup ticket is the caller's (zeroed, per stream)
    rc = L.blp_rank_metric_sums(None, 5, None, None, 0, None)
    assert rc == -1
    # int32 counts: N >= 2^31 is refused; q_fixed / q_rel feed 16-byte vector loads
    rc = L.blp_rank_all(0, one, 1 << 31, 128, 128, one, one, one, None, 1, 1, None, one, one, 1 << 40, 0, None)
    assert rc == -1 and b"2^31" in L.blp_last_error()
    rc = L.blp_rank_all(0, one, 10, 128, 128, ctypes.c_void_p(20), one, one, None, 1, 1, None, one, one, 1 << 40, 0, None)
    assert rc == -1 and b"aligned" in L.blp_last_error()


def test_no_scalar_load_result_is_touched_before_its_wait(built_lib):
    """The hand-pipelined kernels issue scalar loads and wait for them in SEPARATE asm statements (rank_common.h: sload16 /
    sdrain; rank_stream.hip: sload16_pinned / sdrain_pinned).  Between the two the destination registers hold nothing yet,
    but the compiler -- which takes the request's result for available -- is free to copy or spill them there (the hardware
    has no interlock on scalar-load destinations).  The disassembly of every ranking object must show no instruction that
    names a destination register of an s_load_dwordx16 before the next full lgkmcnt wait."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from blp_amd import build
    checked = 0
    for src in ("rank_all.hip", "rank_small.hip", "rank_stream.hip", "rank_stream16.hip", "rank_sad.hip", "rank_sad_wide.hip", "rank_gemm.hip"):
        obj = os.path.join(build.OBJ, src + ".o")
        bad = kernel_resources.early_uses_of_scalar_loads(obj)
        assert not bad, (src, bad[:3])
        checked += sum(1 for l in kernel_resources.disassembly(obj) if l.startswith("s_load_dwordx16"))
    assert checked > 1000  # (the objects were really disassembled)


def test_hand_scheduled_valu_keeps_its_sgpr_wait_states(built_lib):
    """gfx950: a VALU instruction reading an SGPR pair a VALU instruction wrote needs two wait states in between.  The
    compiler pads its own code with s_nop; the hand-scheduled K-steps of rank_gemm.hip (v_cmp -> carry pair -> v_addc_co,
    inline asm) space themselves -- checked here on the disassembly of every ranking object, and the checker on itself
    (with four wait states demanded it must find decide_pair's compare -> add-with-carry pairs, three apart)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from blp_amd import build
    for src in ("rank_all.hip", "rank_small.hip", "rank_stream.hip", "rank_stream16.hip", "rank_sad.hip", "rank_sad_wide.hip", "rank_gemm.hip"):
        bad = kernel_resources.valu_sgpr_hazards(os.path.join(build.OBJ, src + ".o"))
        assert not bad, (src, bad[:3])
    stricter = kernel_resources.valu_sgpr_hazards(os.path.join(build.OBJ, "rank_gemm.hip.o"), min_gap=4)
    assert any(w.startswith("v_cmp") and r.startswith("v_addc_co") for w, r, _ in stricter)


def test_passes_per_launch_is_host_logic(built_lib):
    """blp_rank_all_batches_passes_per_launch (bench.py's roofline divides a bracketed launch by it): all passes of a
    reference-batched evaluation against a long table are one launch of a streaming kernel; blocks ranked as blocks, short
    tables (the small-block kernels) and batches of more than 4 triples are not.  Pure host logic: runs without a GPU."""
    from blp_amd import _lib
    L = _lib.lib()
    def f(model, *sizes):  # (the float32 table: BLP_DTYPE_F32)
        return L.blp_rank_all_batches_passes_per_launch(model, 0, *sizes)

    for model in range(4):
        assert f(model, 4_600_000, 128, 128, 128, 2, 2) == 64      # Wikidata5M, eval_batch_size 2 (scripts/blp-*-wikidata5m.sh:18)
        assert f(model, 575_000, 128, 128, 9, 4, 4) == 3            # a 1/8 shard, 4 + 4 queries, the last pass one triple
        assert f(model, 4_600_000, 128, 128, 128, 2, 0) == 1        # the library's own blocking: one block
        assert f(model, 4_600_000, 128, 128, 128, 64, 64) == 1      # a pass per 64-triple batch: not the streaming kernels
        assert f(model, 2000, 128, 128, 128, 2, 2) == 1             # short table: the small-block kernels, a launch per pass
        assert f(model, 4_600_000, 128, 128, 2, 2, 2) == 1          # a single pass
    assert f(0, 4_600_000, 128, 128, 0, 2, 2) == 0 and f(9, 100, 128, 128, 4, 2, 2) == 0
    assert f(1, 4_600_000, 256, 256, 128, 2, 2) == 64 and f(1, 4_600_000, 100, 100, 128, 2, 2) == 1


def test_typed_batches_routes_and_workspace_are_host_logic(built_lib):
    """blp_rank_all_batches with a 16-bit table: the reference-batched passes over a long table read it as it is (every pass
    in one launch, the workspace of the passes only); every other shape of call holds a widened float32 copy at the end of its
    workspace (N x D x 4 bytes on top of what the float32 call needs); BLP_DTYPE_F32 is the float32 entry.  No GPU needed."""
    from blp_amd import _lib
    L = _lib.lib()
    ws, ppl = L.blp_rank_all_batches_workspace_bytes, L.blp_rank_all_batches_passes_per_launch

    def f32(model, N, D, n, batch, block):
        return ws(model, 0, N, D, D, n, batch, block)

    for model in range(4):
        for dt in (1, 2):
            assert ppl(model, dt, 4_600_000, 128, 128, 128, 2, 2) == 64 and ppl(model, dt, 4_600_000, 256, 256, 9, 4, 4) == 3
            assert ppl(model, dt, 4_600_000, 128, 128, 2, 2, 2) == 1       # a single pass: the ring all the same (one pass per launch)
            assert ws(model, dt, 4_600_000, 128, 128, 128, 2, 2) < (1 << 20)  # no widened copy
            assert ppl(model, dt, 4_600_000, 128, 128, 128, 2, 0) == 1     # one block: a widened copy
            assert ws(model, dt, 4_600_000, 128, 128, 128, 2, 0) >= f32(model, 4_600_000, 128, 128, 2, 0) + 4_600_000 * 128 * 4
            assert ws(model, dt, 2000, 128, 128, 128, 2, 2) >= 2000 * 128 * 4   # short table: small-block kernels on the copy
            assert ws(model, dt, 4_600_000, 64, 64, 128, 2, 2) >= 4_600_000 * 64 * 4  # D = 64: not the 16-bit ring
            assert ws(model, dt, 4_600_000, 128, 132, 128, 2, 2) >= 4_600_000 * 128 * 4  # (ld % 8 != 0 is refused by the call itself)
        assert ppl(model, 0, 4_600_000, 128, 128, 128, 2, 2) == 64
    assert ws(0, 5, 100, 128, 128, 4, 2, 2) == 0 and ppl(0, 5, 100, 128, 128, 4, 2, 2) == 0


def test_batches_workspace_covers_every_block_of_the_call(built_lib):
    """blp_rank_all_workspace_bytes is not monotone in the query count (the route changes with Q x N), so the short LAST
    block of a blp_rank_all_batches call can need more than a full one (round 3: TransE D = 64, N = 7 400, 65 536 + 54 592
    triples reported 85 MB where the tail block carved 124 MB -- silent out-of-bounds writes).  The call's workspace must
    hold every block's own requirement, plus the permutation arrays behind it when batches are merged.  Host logic only."""
    from blp_amd import _lib
    L = _lib.lib()
    one = L.blp_rank_all_workspace_bytes

    def many(model, N, D, n, batch, block):
        return L.blp_rank_all_batches_workspace_bytes(model, 0, N, D, D, n, batch, block)

    cases = 0
    for model in range(4):
        for D in (64, 128):
            for N in (7400, 14541, 40943, 575_000, 4_600_000):
                for batch, block in ((64, 0), (512, 0), (65536, 0), (64, 64), (2, 2), (64, 4096), (16, 48)):
                    sup = (block if block else 65536) // batch * batch
                    sup = max(sup, batch)
                    for tail in (1, 2, 3, 33, 257, 4097, 54592, 65376, sup - 1):
                        if not 0 < tail < sup:
                            continue
                        for full_blocks in (1, 2):
                            n = full_blocks * sup + tail
                            got = many(model, N, D, n, batch, block)
                            merged = sup > batch
                            extra = 0
                            if merged:  # permuted index / segment arrays (6 x 2m int64) + permuted counts (2m x 16 B), 256-aligned
                                extra = 6 * 2 * sup * 8 + 2 * sup * 16
                            for m in (sup, tail):
                                need = one(model, N, D, m, m)
                                need = (need + 255) // 256 * 256 if merged else need
                                assert got >= need + extra, (model, D, N, n, batch, block, m, got, need + extra)
                            cases += 1
    assert cases > 2000
    # the reviewer's cases
    assert many(0, 7400, 64, 65536 + 54592, 65536, 65536) >= one(0, 7400, 64, 54592, 54592)
    assert many(0, 14541, 64, 512 + 257, 512, 512) >= one(0, 14541, 64, 257, 257)
    # block sizes whose permutation grid / query count would overflow are refused, not truncated
    p16 = ctypes.c_void_p(16)
    rc = L.blp_rank_all_batches(0, p16, 0, 100, 128, 128, p16, 100, 128, p16, p16, 5, p16, p16, 1 << 31, 1 << 30, 1 << 30,
                                None, p16, p16, 1 << 40, 0, None)
    assert rc == -1 and b"2^29" in L.blp_last_error()


def test_project_rows_argument_checks(built_lib):
    from blp_amd import _lib
    L = _lib.lib()
    assert L.blp_project_rows_supported(768, 128) == 1 and L.blp_project_rows_supported(770, 128) == 0
    assert L.blp_project_rows_supported(768, 100) == 0
    assert L.blp_project_rows(None, 0, 768, None, 768, 128, 1, None, 128, 0, None) == _lib.BLP_OK      # nothing to do
    assert L.blp_project_rows(None, 4, 768, None, 768, 128, 1, None, 128, 0, None) == -1
    assert L.blp_project_rows(None, 4, 768, None, 768, 96, 1, None, 96, 0, None) == -2
    assert L.blp_project_rows(None, -1, 768, None, 768, 128, 1, None, 128, 0, None) == -1
    assert b"blp_project_rows" in L.blp_last_error()


def test_bow_rows_argument_checks(built_lib):
    from blp_amd import _lib
    L = _lib.lib()
    assert L.blp_bow_rows_supported(300) == 1 and L.blp_bow_rows_supported(768) == 1 and L.blp_bow_rows_supported(1024) == 1
    assert L.blp_bow_rows_supported(302) == 0 and L.blp_bow_rows_supported(1028) == 0 and L.blp_bow_rows_supported(0) == 0
    p16 = ctypes.c_void_p(16)
    assert L.blp_bow_rows(None, None, 0, 32, None, 100, 300, 1, None, 300, None, 0, None) == _lib.BLP_OK     # nothing to do
    assert L.blp_bow_rows(None, None, 4, 32, p16, 100, 300, 1, p16, 300, p16, 0, None) == -1               # no tokens
    assert L.blp_bow_rows(p16, None, 4, 32, p16, 100, 302, 1, p16, 304, p16, 0, None) == -2
    assert L.blp_bow_rows(p16, None, 4, 32, ctypes.c_void_p(20), 100, 300, 1, p16, 300, p16, 0, None) == -1  # alignment
    assert L.blp_bow_rows(p16, None, 4, 32, p16, 100, 300, 1, p16, 296, p16, 0, None) == -1                # ldo < E
    assert b"blp_bow_rows" in L.blp_last_error()


def test_knobs_are_named_and_reset(built_lib):
    from blp_amd import _lib
    product = _lib.lib()
    before = product.blp_rank_all_workspace_bytes(0, 14541, 128, 300, 300)
    _lib.set_knob("rank_kernel", 1)  # exact f32 kernels: no images, no pair lists -- and calls now go to the hooks build
    H = _lib.lib()
    assert H is _lib.hooks_lib() and H is not product
    assert H.blp_rank_all_workspace_bytes(0, 14541, 128, 300, 300) < before
    assert product.blp_rank_all_workspace_bytes(0, 14541, 128, 300, 300) == before  # the product library has no knobs
    _lib.reset_knobs()
    assert _lib.lib() is product
    assert H.blp_rank_all_workspace_bytes(0, 14541, 128, 300, 300) == before
    with pytest.raises(RuntimeError, match="unknown knob"):
        _lib.set_knob("no_such_knob", 1)
    with pytest.raises(RuntimeError, match="BLP_ERR_BAD_ARG"):
        _lib.check(-1, "demo")


def test_metric_sums_buffer_size_matches_the_header():
    import re
    from blp_amd import _lib
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "blp_hip.h")).read()
    assert int(re.search(r"#define BLP_METRIC_SUMS_DOUBLES (\d+)", header).group(1)) == _lib.METRIC_SUMS_DOUBLES


def test_torch_glue_builds_loads_and_agrees_with_the_header(built_lib):
    """blp_amd/_torch_glue.so (csrc/torch_glue.cpp: the in-batch loss's autograd plumbing as a C++ torch extension -- host
    code only) builds against the installed torch, imports, binds the product library's entry points (it asks the library's
    blp_inbatch_loss_save_floats for the size of `save_pos`) and was written against this header's major version; it refuses
    CPU tensors like the Python plumbing."""
    import torch
    from blp_amd import _lib, build, ops
    assert os.path.exists(build.build_glue())
    glue = ops.torch_glue()
    assert glue is not None and glue.__file__.endswith("_torch_glue.so")
    header = open(os.path.join(ROOT, "include", "blp_hip.h")).read()
    assert glue.abi_version // 10000 == int(re.search(r"#define BLP_HIP_VERSION (\d+)", header).group(1)) // 10000 == _lib.lib().blp_version() // 10000
    assert int(re.search(r"#define BLP_INBATCH_TICKET_INTS (\d+)", header).group(1)) == _lib.INBATCH_TICKET_INTS
    # save_pos: the B positives' scores + two f64 partial sums per forward workgroup + the index of neg_idx (offsets per chunk
    # and row + one int per entry) -- grows with B K, never with a float argument
    for model in range(4):
        for B, K, D in ((1, 1, 128), (64, 64, 128), (1024, 64, 128), (64, 64, 300), (4096, 64, 128)):
            n = _lib.inbatch_save_floats(model, B, K, D)
            chunk = 1024
            chunks = -(-2 * B * K // chunk)
            assert n >= B + 4 * B * K + chunks * (2 * B + 1), (model, B, K, D, n)
            assert n <= B + 4 + 4 * B * K + chunks * (2 * B + 1) + 4 * (-(-B * (K + 1) // 8) + (B + 3) // 4) + 8 + B + 2 + 6 * 64  # (+ the many-workgroup forward's regulariser shares and reduction scratch)
    assert _lib.inbatch_save_floats(7, 64, 64, 128) == 0 and _lib.inbatch_save_floats(0, 0, 64, 128) == 0
    with pytest.raises(RuntimeError, match="HIP device tensors only"):
        glue.inbatch_loss(torch.zeros(4, 2, 8), torch.zeros(4, 1, 8), torch.zeros(4, 3, 2, dtype=torch.long), 0, 0, 0.0, 0)


def test_no_kernel_of_the_in_batch_loss_uses_scratch(built_lib):
    """Every inbatch_* kernel keeps its working set in registers / LDS: no VGPR spills, and .private_segment_fixed_size == 0
    in the gfx950 code object's notes (tools/kernel_resources.py) for the forward and for the backward at the scripts' widths
    (rows of up to 128 elements).  The backward's WIDE-row shape (<.., 8>: D > 128 or D % 4 != 0) of the bilinear models
    declares 20 bytes -- the slot the compiler reserves for the VGPR its spilled SCALAR registers are parked in -- and never
    touches them: its disassembly holds no scratch_* / buffer_* instruction (checked here)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from blp_amd import build
    obj = os.path.join(build.OBJ, "inbatch_loss.hip.o")
    kernels = kernel_resources.kernels_of(obj)
    names = [k for k in kernels if "inbatch" in k]
    assert len(names) >= 2 * 4 * 5  # forward + grad (two row shapes) per (model, storage types)
    declared = []
    for name in names:
        f = kernels[name]
        # (scalar registers parked in VGPR lanes -- sgpr_spill_count -- touch no memory and are not scratch)
        assert f.get("vgpr_spill_count", 0) == 0, (name, f)
        if f["private_segment_fixed_size"] != 0:
            assert f["private_segment_fixed_size"] <= 32 and "inbatch_grad_kernel" in name and "Li8EEEv" in name, (name, f)
            declared.append(name)
    if declared:
        lines = kernel_resources.disassembly(obj)
        starts = {i: l for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)}
        order = sorted(starts)
        for name in declared:
            begin = next(i for i in order if f"<{name}>" in starts[i])
            end = next((i for i in order if i > begin), len(lines))
            body = lines[begin:end]
            assert len(body) > 500 and not [l for l in body if "scratch_" in l or "buffer_" in l], name


def test_no_ranking_kernel_uses_scratch_memory(built_lib):
    """Every kernel of the product library keeps its working set in registers (the accumulator half of the unified file
    included: a non-zero vgpr_spill_count next to a zero private segment is parked there, not in memory) and LDS --
    .private_segment_fixed_size == 0 in the gfx950 code objects' notes -- with ONE named exception: blp_score_fwd's
    order-exact torch.sum at widths off the 32-grid (score_direct.h: torch_inner_sum_any keeps 128 partial sums in
    scratch; the bilinear models' scripts never use such a width).  Round 2: rank_tiles_kernel<0,128,false> and every
    D = 256 instantiation spilled 51-732 registers to scratch."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from blp_amd import build
    kernels = kernel_resources.kernels_of(build.OBJ)
    assert len(kernels) > 150
    nice = dict(zip(kernels, kernel_resources.demangle(list(kernels))))
    # (+ the in-batch loss backward's wide-row shape: a reserved, never touched 20-byte slot -- the test above reads its disassembly)
    offenders = {nice[k]: f["private_segment_fixed_size"] for k, f in kernels.items()
                 if f["private_segment_fixed_size"] != 0 and "score_fwd_kernel" not in nice[k]
                 and not ("inbatch_grad_kernel" in k and "Li8EEEv" in k)}
    assert not offenders, offenders
    exempt = [nice[k] for k, f in kernels.items() if f["private_segment_fixed_size"] != 0 and "inbatch_grad_kernel" not in k]
    assert all("score_fwd_kernel<" in n and "score_fwd_kernel<0>" not in n for n in exempt), exempt
