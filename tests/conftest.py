import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REL_MODELS = ("transe", "distmult", "complex", "simple")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_routing: only at the shipped dispatch (no routing knobs), the product library")


def pytest_collection_modifyitems(config, items):
    """GPU parity tests run twice (fixture `routing` of test_gpu_parity.py / test_gpu_eval.py): at the SHIPPED dispatch on
    the product library, and with the knobs that keep the pre-pass kernels in play on small tables (hooks library).  Tests
    marked default_routing make sense at the shipped dispatch only: their "prepass" instance is dropped here."""
    keep = []
    for item in items:
        spec = getattr(item, "callspec", None)
        if item.get_closest_marker("default_routing") and spec is not None and spec.params.get("routing") == "prepass":
            continue
        keep.append(item)
    items[:] = keep


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    return dict(np.load(path, allow_pickle=False))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture
def knobs():
    """Set test knobs (blp_debug_set_knob of the hooks build, libblp_hip.hooks.so -- which then serves every call until
    the knobs are reset) for one test; every knob is automatic again afterwards and the product library serves."""
    from blp_amd import _lib
    yield _lib.set_knob
    _lib.reset_knobs()
