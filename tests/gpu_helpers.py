"""Shared helpers of the GPU parity tests (tests/test_gpu_*.py)."""
import numpy as np
import pytest
from scipy import sparse

from pygsp_amd import engine

TOL = {np.dtype(np.float64): 1e-11, np.dtype(np.float32): 2e-5}
BAR = {np.dtype(np.float64): 1e-5, np.dtype(np.float32): 1e-3}  # BASELINE.json north_star

DEFAULT_OPTIONS = (("kernel", 0), ("vec", 0), ("rows_per_wave", 0), ("xcd_remap", 1), ("combine", 0),
                   ("max_batch", 0), ("narrow_g_log2", -1), ("tile_gather", 1), ("graph_launch", 2),
                   ("alternate_sweep", 1), ("synthesis", 0), ("fuse_input", 1), ("tile_workgroups", 0),
                   ("tile_nt", -1), ("ws_limit_mb", 65536), ("host_pipeline", 1), ("host_batch", 0),
                   ("host_threads", 0), ("host_edge", 0))


@pytest.fixture(scope="module")
def ctx():
    """The process-wide default context, its options back at their defaults afterwards."""
    c = engine.default_context(0)
    yield c
    for key, val in DEFAULT_OPTIONS:
        c.set_option(key, val)


def random_graph(n, avg_deg, seed, hub=False, isolated=0):
    """Random symmetric weighted graph with ragged rows; optional hub row and isolated vertices."""
    rng = np.random.default_rng(seed)
    m = int(n * avg_deg / 2)
    r = rng.integers(0, n, m)
    c = rng.integers(0, n, m)
    keep = r != c
    r, c = r[keep], c[keep]
    w = rng.uniform(0.1, 1.0, r.size)
    if hub:
        hc = rng.choice(n - 1, size=min(n - 1, 300), replace=False) + 1
        r = np.concatenate([r, np.zeros(hc.size, dtype=r.dtype)])
        c = np.concatenate([c, hc])
        w = np.concatenate([w, rng.uniform(0.1, 1.0, hc.size)])
    A = sparse.coo_matrix((w, (r, c)), shape=(n, n)).tocsr()
    W = sparse.csr_matrix(A + A.T)
    if isolated:
        iso = rng.choice(n, size=isolated, replace=False)
        mask = np.ones(n)
        mask[iso] = 0
        D = sparse.diags(mask)
        W = sparse.csr_matrix(D @ W @ D)
        W.eliminate_zeros()
    W.sum_duplicates()
    W.sort_indices()
    return W


def upper_lmax(W):
    return 2.0 * float(np.ravel(W.sum(axis=0)).max()) + 1e-9
