import os
import sys

import numpy as np
import pytest
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def csr_from(npz, prefix):
    shape = tuple(int(v) for v in npz[prefix + "_shape"])
    return sparse.csr_matrix((npz[prefix + "_data"], npz[prefix + "_indices"], npz[prefix + "_indptr"]),
                             shape=shape)


def rel_err(y, ref):
    """max|y - ref| / max|ref| - the tolerance metric of BASELINE.md (outputs span many decades)."""
    ref = np.asarray(ref)
    den = np.max(np.abs(ref)) if ref.size else 1.0
    if den == 0:
        den = 1.0
    return float(np.max(np.abs(np.asarray(y) - ref)) / den) if ref.size else 0.0


@pytest.fixture(scope="session")
def golden_sensor123():
    return load_golden("sensor123.npz")


@pytest.fixture(scope="session")
def golden_logo():
    return load_golden("logo_heat50.npz")


@pytest.fixture(scope="session")
def golden_lap4():
    return load_golden("laplacians4.npz")


@pytest.fixture(scope="session")
def golden_doctest():
    return load_golden("doctest_sensor30.npz")


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("ops_sensor123.npz")


@pytest.fixture(scope="session")
def golden_knn():
    return load_golden("knn.npz")
