import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The modules compute the last stage for the batch's rows only while the batch lists at most a fifth of the table's rows (a
# profitability cut-off); the test graphs are small next to their batches, so the tests lift the cut-off to keep exercising
# that path (test_sparse_last_stage_cutoff checks the default).
os.environ.setdefault("NGACF_SPARSE_MAX_FRACTION", "1e9")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def params_of(g, prefix="p."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
