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
    config.addinivalue_line("markers", "default_routing: keep the library's own choice of TransE kernel (no sad_min_queries knob)")


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
    """Set test knobs of libblp_hip.so (blp_debug_set_knob) for one test; every knob is automatic again afterwards."""
    from blp_amd import _lib
    yield _lib.set_knob
    _lib.reset_knobs()
