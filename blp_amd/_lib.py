"""ctypes binding of libblp_hip.so (include/blp_hip.h).

The library is the product: there is no Python or CPU fallback behind it.  If it is missing the
import of anything that needs it raises ``HipLibraryError`` telling the user to build it
(``python -m blp_amd.build``).  ctypes releases the GIL during calls, so nn.DataParallel replica
threads overlap their launches.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libblp_hip.so")
HOOKS_LIB_PATH = os.path.join(_HERE, "libblp_hip.hooks.so")  # the -DBLP_TEST_HOOKS build (tests / tools only)

MODEL_IDS = {"transe": 0, "distmult": 1, "complex": 2, "simple": 3}
LOSS_IDS = {"margin": 0, "nll": 1}

BLP_OK = 0
STATUS_NAMES = {0: "BLP_OK", -1: "BLP_ERR_BAD_ARG", -2: "BLP_ERR_UNSUPPORTED_DIM", -3: "BLP_ERR_HIP",
                -4: "BLP_ERR_WORKSPACE"}

# every symbol include/blp_hip.h declares (tests check the .so exports exactly these)
SYMBOLS = ("blp_version", "blp_last_error", "blp_device_caps", "blp_selftest", "blp_dim_supported",
           "blp_rank_all_workspace_bytes", "blp_rank_all_supported", "blp_rank_all", "blp_rank_all_shard", "blp_gather_triple_vectors",
           "blp_rank_all_batches", "blp_rank_all_batches_workspace_bytes", "blp_rank_all_batches_passes_per_launch",
           "blp_rank_all_batches_native16", "blp_profile_next_rank_kernel", "blp_rank_all_prepass_stats", "blp_rank_from_scores",
           "blp_rank_metrics", "blp_rank_metric_sums", "blp_score_fwd", "blp_score_bwd", "blp_inbatch_loss_save_floats",
           "blp_inbatch_loss_fwd_launches", "blp_inbatch_loss_fwd", "blp_inbatch_loss_bwd", "blp_project_rows_supported", "blp_project_rows", "blp_bow_rows_supported", "blp_bow_rows", "blp_dkrl_rows_supported", "blp_dkrl_rows",
           "blp_build_queries")
HOOK_SYMBOLS = ("blp_debug_set_knob", "blp_debug_gemm_dump", "blp_debug_reset_selftest")  # libblp_hip.hooks.so only
KNOBS = ("rank_kernel", "gemm_kernel", "sad_queries_per_group", "sad_pass_groups", "sad_min_queries",
         "gemm_pass_words", "gemm_tiles_per_chunk", "exact_query_chunk",
         "small_kernel", "stream_kernel", "dkrl_split", "mfma_selftest", "inbatch_probe", "inbatch_shares")  # blp_amd/csrc/knobs.h


DTYPE_NAMES = ("float32", "float16", "bfloat16")  # BLP_DTYPE_* of include/blp_hip.h
INBATCH_TICKET_INTS = 4  # BLP_INBATCH_TICKET_INTS of include/blp_hip.h


def inbatch_save_floats(model_id, B, K, D):
    """blp_inbatch_loss_save_floats: floats the `save_pos` argument of blp_inbatch_loss_fwd needs (positives' scores, partial
    loss sums, the index of neg_idx the backward walks)."""
    return int(lib().blp_inbatch_loss_save_floats(int(model_id), int(B), int(K), int(D)))


METRIC_SUMS_DOUBLES = 520  # BLP_METRIC_SUMS_DOUBLES: room the `sums` argument of blp_rank_metric_sums needs


class HipLibraryError(RuntimeError):
    pass


class BlpCaps(ctypes.Structure):
    _fields_ = [("compute_units", ctypes.c_int), ("wavefront_size", ctypes.c_int),
                ("lds_bytes_per_cu", ctypes.c_int), ("clock_mhz", ctypes.c_int),
                ("hbm_bytes", ctypes.c_int64), ("arch", ctypes.c_char * 32), ("mfma_bf16_accum", ctypes.c_int),
                ("mfma_bf16_accum_worst", ctypes.c_float)]


class BlpFilter(ctypes.Structure):  # blp_filter of include/blp_hip.h
    _fields_ = [("seg_lo", ctypes.c_void_p), ("seg_hi", ctypes.c_void_p), ("values", ctypes.c_void_p),
                ("exclude", ctypes.c_void_p), ("ent2idx", ctypes.c_void_p), ("ent2idx_len", ctypes.c_int64),
                ("row_base", ctypes.c_int64)]


class BlpQueries(ctypes.Structure):  # blp_queries of include/blp_hip.h
    _fields_ = [("triples", ctypes.c_void_p), ("n", ctypes.c_int64), ("block", ctypes.c_int64),
                ("ent2idx", ctypes.c_void_p), ("ent2idx_len", ctypes.c_int64),
                ("source", ctypes.c_void_p), ("src_rows", ctypes.c_int64), ("ld", ctypes.c_int64), ("D", ctypes.c_int),
                ("rel_emb", ctypes.c_void_p), ("R", ctypes.c_int64),
                ("heads_key", ctypes.c_void_p), ("n_heads", ctypes.c_int64),
                ("tails_key", ctypes.c_void_p), ("n_tails", ctypes.c_int64), ("index_R", ctypes.c_int64),
                ("q_fixed", ctypes.c_void_p), ("q_rel", ctypes.c_void_p), ("true_row", ctypes.c_void_p),
                ("rel_ids", ctypes.c_void_p), ("ids_min", ctypes.c_void_p),
                ("seg_lo", ctypes.c_void_p), ("seg_hi", ctypes.c_void_p), ("exclude", ctypes.c_void_p),
                ("fixed_row", ctypes.c_void_p), ("by_position", ctypes.c_int64)]


_lib = None       # the library calls go to: the product, or the hooks build while a test holds a knob
_product = None
_hooks = None
_vp, _i, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float


def lib():
    """The library every call goes to: libblp_hip.so, loaded once.  Raises HipLibraryError (never falls back) if it is
    not built.  (While a test holds a knob -- set_knob -- calls go to the hooks build instead.)"""
    global _lib, _product
    if _lib is not None:
        return _lib
    if _product is None:
        _product = _load(LIB_PATH, hooks=False)
    _lib = _product
    return _lib


def hooks_lib():
    """libblp_hip.hooks.so (tests / tools): the same kernels with blp_debug_set_knob / blp_debug_gemm_dump."""
    global _hooks
    if _hooks is None:
        _hooks = _load(HOOKS_LIB_PATH, hooks=True)
    return _hooks


def _load(path, hooks):
    # torch first: its bundled libamdhip64 (SONAME libamdhip64.so.7) must already be mapped so that the
    # dynamic loader binds libblp_hip.so to the SAME HIP runtime instance torch's streams and
    # allocations belong to (two runtimes in one process cannot share hipStream_t handles).
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found: the HIP extension is not built. Run `python -m blp_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback for "
            "CUDA/HIP tensors.")
    try:
        L = ctypes.CDLL(path)
    except OSError as exc:
        raise HipLibraryError(f"cannot load {path}: {exc}") from exc
    L.blp_version.restype = _i
    L.blp_version.argtypes = []
    L.blp_last_error.restype = ctypes.c_char_p
    L.blp_last_error.argtypes = []
    L.blp_device_caps.restype = _i
    L.blp_device_caps.argtypes = [_i, ctypes.POINTER(BlpCaps)]
    L.blp_rank_all_prepass_stats.restype = _i
    L.blp_rank_all_prepass_stats.argtypes = [_i, _i64, _i, _i64, _i64, _vp, _sz, ctypes.POINTER(ctypes.c_int64), _i, _vp]
    L.blp_rank_all_batches_native16.restype = _i
    L.blp_rank_all_batches_native16.argtypes = [_i, _i, _i64, _i, _i64, _i64, _i64, _i64]
    L.blp_selftest.restype = _i
    L.blp_selftest.argtypes = [_i, _vp]
    L.blp_dim_supported.restype = _i
    L.blp_dim_supported.argtypes = [_i, _i]
    L.blp_rank_all_workspace_bytes.restype = _sz
    L.blp_rank_all_workspace_bytes.argtypes = [_i, _i64, _i, _i64, _i64]
    L.blp_rank_all_supported.restype = _i
    L.blp_rank_all_supported.argtypes = [_i, _i, _i64, _i64]
    L.blp_rank_all.restype = _i
    L.blp_rank_all.argtypes = [_i, _vp, _i64, _i, _i64, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.POINTER(BlpFilter),
                               _vp, _vp, _sz, _i, _vp]
    L.blp_rank_all_shard.restype = _i
    L.blp_rank_all_shard.argtypes = [_i, _vp, _i64, _i, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64,
                                     ctypes.POINTER(BlpFilter), _vp, _vp, _sz, _i, _vp]
    L.blp_gather_triple_vectors.restype = _i
    L.blp_gather_triple_vectors.argtypes = [_vp, _i64, _vp, _i64, _vp, _i, _i64, _i, _i64, _i64, _vp, _i, _vp]
    L.blp_rank_all_batches_workspace_bytes.restype = _sz
    L.blp_rank_all_batches_workspace_bytes.argtypes = [_i, _i, _i64, _i, _i64, _i64, _i64, _i64]
    L.blp_rank_all_batches_passes_per_launch.restype = _i64
    L.blp_rank_all_batches_passes_per_launch.argtypes = [_i, _i, _i64, _i, _i64, _i64, _i64, _i64]
    L.blp_rank_all_batches.restype = _i
    L.blp_rank_all_batches.argtypes = [_i, _vp, _i, _i64, _i, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64,
                                       ctypes.POINTER(BlpFilter), _vp, _vp, _sz, _i, _vp]
    L.blp_profile_next_rank_kernel.restype = _i
    L.blp_profile_next_rank_kernel.argtypes = [_vp, _vp]
    L.blp_rank_from_scores.restype = _i
    L.blp_rank_from_scores.argtypes = [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]
    L.blp_rank_metrics.restype = _i
    L.blp_rank_metrics.argtypes = [_vp, _i64, ctypes.POINTER(ctypes.c_int32), _vp, _vp, _i, _vp]
    L.blp_rank_metric_sums.restype = _i
    L.blp_rank_metric_sums.argtypes = [_vp, _i64, ctypes.POINTER(ctypes.c_int32), _vp, _i, _vp]
    L.blp_score_fwd.restype = _i
    L.blp_score_fwd.argtypes = [_i, _i, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                _vp, _i, _vp]
    L.blp_score_bwd.restype = _i
    L.blp_score_bwd.argtypes = [_i, _i, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                _vp, _vp, _vp, _vp, _i, _vp]
    L.blp_inbatch_loss_save_floats.restype = ctypes.c_size_t
    L.blp_inbatch_loss_save_floats.argtypes = [_i, _i, _i, _i]
    L.blp_inbatch_loss_fwd_launches.restype = _i
    L.blp_inbatch_loss_fwd_launches.argtypes = [_i, _i, _i, _i, _f]
    L.blp_inbatch_loss_fwd.restype = _i
    L.blp_inbatch_loss_fwd.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp]
    L.blp_inbatch_loss_bwd.restype = _i
    L.blp_inbatch_loss_bwd.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp,
                                         _i, _vp]
    L.blp_build_queries.restype = _i
    L.blp_build_queries.argtypes = [ctypes.POINTER(BlpQueries), _i, _vp]
    L.blp_project_rows_supported.restype = _i
    L.blp_project_rows_supported.argtypes = [_i, _i]
    L.blp_project_rows.restype = _i
    L.blp_project_rows.argtypes = [_vp, _i64, _i64, _vp, _i, _i, _i, _vp, _i64, _i, _vp]
    L.blp_bow_rows_supported.restype = _i
    L.blp_bow_rows_supported.argtypes = [_i]
    L.blp_bow_rows.restype = _i
    L.blp_bow_rows.argtypes = [_vp, _vp, _i64, _i, _vp, _i64, _i, _i, _vp, _i64, _vp, _i, _vp]
    L.blp_dkrl_rows_supported.restype = _i
    L.blp_dkrl_rows_supported.argtypes = [_i, _i, _i]
    L.blp_dkrl_rows.restype = _i
    L.blp_dkrl_rows.argtypes = [_vp, _vp, _i64, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _vp, _i, _vp]
    if hooks:
        L.blp_debug_gemm_dump.restype = _i
        L.blp_debug_gemm_dump.argtypes = [_vp, _vp]
        L.blp_debug_set_knob.restype = _i
        L.blp_debug_set_knob.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
        L.blp_debug_reset_selftest.restype = _i
        L.blp_debug_reset_selftest.argtypes = [_i]
    return L


_knobs_set = {}


def set_knob(name, value):
    """Test / A-B hook (include/blp_hip.h: blp_debug_set_knob, hooks build only); 0 restores the automatic choice.
    While any knob is non-zero every call of this process goes to libblp_hip.hooks.so; with all knobs back at 0 the
    product library serves again -- so a test without knobs always exercises the shipped dispatch."""
    global _lib
    H = hooks_lib()
    check(H.blp_debug_set_knob(name.encode(), int(value)), "blp_debug_set_knob", H)
    if int(value):
        _knobs_set[name] = int(value)
    else:
        _knobs_set.pop(name, None)
    _lib = H if _knobs_set else None


def reset_knobs():
    global _lib
    if _hooks is not None:
        for name in KNOBS:
            check(_hooks.blp_debug_set_knob(name.encode(), 0), "blp_debug_set_knob", _hooks)
    _knobs_set.clear()
    _lib = None


class use_hooks_library:
    """Context manager: route the calls inside it to the hooks build (for blp_debug_gemm_dump, which needs no knob)."""
    def __enter__(self):
        global _lib
        self._prev = _lib
        _lib = hooks_lib()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev if _knobs_set else None
        return False


def check(status, what, library=None):
    """Map a non-zero C status to RuntimeError(blp_last_error()) (SURVEY 8b error convention)."""
    if status != BLP_OK:
        msg = (library or lib()).blp_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed with {STATUS_NAMES.get(status, status)}: {msg}")


def device_caps(device=0):
    caps = BlpCaps()
    check(lib().blp_device_caps(device, ctypes.byref(caps)), "blp_device_caps")
    return {"compute_units": caps.compute_units, "wavefront_size": caps.wavefront_size,
            "lds_bytes_per_cu": caps.lds_bytes_per_cu, "clock_mhz": caps.clock_mhz,
            "hbm_bytes": caps.hbm_bytes, "arch": caps.arch.decode(),
            "mfma_bf16_accum": {0: "not tested yet", 1: "passed", 2: "FAILED: bilinear blocks use the f32-chain pre-pass"}[caps.mfma_bf16_accum],
            "mfma_bf16_accum_worst": caps.mfma_bf16_accum_worst}


def selftest(device=0, stream=None):
    """include/blp_hip.h: blp_selftest -- the matrix-pipe accumulation self-test behind the bilinear pre-pass, once per device
    (a set-up call: it waits for the stream).  Returns True if the device passed (bf16 x 3 MFMA pre-pass), False if bilinear
    blocks are served by the f32-chain pre-pass."""
    import torch
    raw = torch.cuda.current_stream(device).cuda_stream if stream is None else stream
    state = lib().blp_selftest(int(device), ctypes.c_void_p(raw))
    if state < 0:
        check(state, "blp_selftest")
    return state == 1
