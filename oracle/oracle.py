"""ctypes binding of oracle/blp_oracle.c (order-exact C restatement of the reference arithmetic).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED -- every function here is
checked bit-for-bit against the imported reference (models.*_score, utils.get_metrics,
train.eval_link_prediction) through the golden vectors in tests/golden/ (generator:
tests/golden/make_golden.py, run in the build container where /root/reference is mounted).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libblp_oracle.so")

MODEL_IDS = {"transe": 0, "distmult": 1, "complex": 2, "simple": 3}
SIDE_HEAD, SIDE_TAIL = 0, 1

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "blp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libblp_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.blp_oracle_torch_inner_sum.restype = ctypes.c_float
        L.blp_oracle_torch_inner_sum.argtypes = [_f32p, ctypes.c_int64]
        L.blp_oracle_score_one.restype = ctypes.c_float
        L.blp_oracle_score_one.argtypes = [ctypes.c_int, _f32p, _f32p, _f32p, ctypes.c_int]
        L.blp_oracle_score_pairs.restype = None
        L.blp_oracle_score_pairs.argtypes = [ctypes.c_int, _f32p, _f32p, _f32p, ctypes.c_int64,
                                             ctypes.c_int, _f32p]
        L.blp_oracle_score_all.restype = None
        L.blp_oracle_score_all.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_int64, _f32p, _f32p,
                                           ctypes.c_int64, _f32p]
        L.blp_oracle_rank_counts.restype = None
        L.blp_oracle_rank_counts.argtypes = [ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_int64, _f32p, _f32p, _i64p,
                                             _f32p, ctypes.c_int64, _i64p, _i64p, _i32p]
        L.blp_oracle_metrics_from_counts.restype = None
        L.blp_oracle_metrics_from_counts.argtypes = [_i32p, _i32p, ctypes.c_int64, ctypes.c_int64,
                                                     _i32p, ctypes.c_int, _f32p, _u8p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _model_id(model):
    return MODEL_IDS[model] if isinstance(model, str) else int(model)


def torch_inner_sum(x):
    """torch.sum(x, dim=-1) order on the torch CPU backend, for a 1-D float32 array."""
    x = _f32(x)
    return np.float32(lib().blp_oracle_torch_inner_sum(_ptr(x, _f32p), x.shape[0]))


def score_pairs(model, heads, tails, rels):
    """score_fn on aligned rows: heads, tails, rels are (M, D) -> (M,)."""
    h, t, r = _f32(heads), _f32(tails), _f32(rels)
    assert h.shape == t.shape == r.shape and h.ndim == 2
    out = np.empty(h.shape[0], np.float32)
    lib().blp_oracle_score_pairs(_model_id(model), _ptr(h, _f32p), _ptr(t, _f32p), _ptr(r, _f32p),
                                 h.shape[0], h.shape[1], _ptr(out, _f32p))
    return out


def score_all(model, side, table, q_fixed, q_rel):
    """(Q, N) scores of every table row as replacement head (side=0) / tail (side=1)."""
    table, q_fixed, q_rel = _f32(table), _f32(q_fixed), _f32(q_rel)
    N, D = table.shape
    Q = q_fixed.shape[0]
    out = np.empty((Q, N), np.float32)
    lib().blp_oracle_score_all(_model_id(model), side, _ptr(table, _f32p), N, D, D,
                               _ptr(q_fixed, _f32p), _ptr(q_rel, _f32p), Q, _ptr(out, _f32p))
    return out


def rank_counts(model, side, table, q_fixed, q_rel, true_row=None, q_true=None,
                filt_rowptr=None, filt_col=None):
    """(Q, 4) int32 {gt, ge, gt_filt, ge_filt}; see blp_oracle_rank_counts in blp_oracle.c."""
    table, q_fixed, q_rel = _f32(table), _f32(q_fixed), _f32(q_rel)
    N, D = table.shape
    Q = q_fixed.shape[0]
    assert (true_row is None) != (q_true is None)
    if true_row is not None:
        true_row = np.ascontiguousarray(true_row, dtype=np.int64).reshape(-1)
    if q_true is not None:
        q_true = _f32(q_true)
    if filt_rowptr is not None:
        filt_rowptr = np.ascontiguousarray(filt_rowptr, dtype=np.int64)
        filt_col = np.ascontiguousarray(filt_col, dtype=np.int64)
    counts = np.empty((Q, 4), np.int32)
    lib().blp_oracle_rank_counts(_model_id(model), side, _ptr(table, _f32p), N, D, D,
                                 _ptr(q_fixed, _f32p), _ptr(q_rel, _f32p), _ptr(true_row, _i64p),
                                 _ptr(q_true, _f32p), Q, _ptr(filt_rowptr, _i64p),
                                 _ptr(filt_col, _i64p), _ptr(counts, _i32p))
    return counts


def metrics_from_counts(gt, ge, k_values=(1, 3, 10)):
    """utils.get_metrics tail: (rr float32 (Q,), hits bool (Q, nk)) from the two counts."""
    gt = np.ascontiguousarray(gt, dtype=np.int32).reshape(-1)
    ge = np.ascontiguousarray(ge, dtype=np.int32).reshape(-1)
    k = np.ascontiguousarray(k_values, dtype=np.int32)
    Q = gt.shape[0]
    rr = np.empty(Q, np.float32)
    hits = np.empty((Q, k.shape[0]), np.uint8)
    lib().blp_oracle_metrics_from_counts(_ptr(gt, _i32p), _ptr(ge, _i32p), Q, 1, _ptr(k, _i32p),
                                         k.shape[0], _ptr(rr, _f32p), _ptr(hits, _u8p))
    return rr, hits.astype(bool)
