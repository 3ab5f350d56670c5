"""Row tiles of the LDS-staged recurrence step: structure invariants and a numpy emulation of the kernel's
fast path against the explicit sparse product.  CPU only."""
import numpy as np
import pytest
from scipy import sparse

from oracle import cheby_oracle as orc
from pygsp_amd import engine, graphs, tiling


def host_internal_csr(L, perm):
    """The engine's internal padded CSR (diag slot first, rest ascending, padded to 4 with col=N),
    rebuilt on the host for a small graph."""
    N = L.shape[0]
    inv = np.empty(N, dtype=np.int64)
    inv[perm] = np.arange(N)
    L = sparse.csr_matrix(L)
    rptr = [0]
    rcol, rval = [], []
    for i in range(N):
        old = perm[i]
        cols = inv[L.indices[L.indptr[old]:L.indptr[old + 1]]]
        vals = L.data[L.indptr[old]:L.indptr[old + 1]]
        d = vals[cols == i].sum() if (cols == i).any() else 0.0
        m = cols != i
        order = np.argsort(cols[m])
        c = [i] + cols[m][order].tolist()
        v = [d] + vals[m][order].tolist()
        while len(c) % 4:
            c.append(N)
            v.append(0.0)
        rcol += c
        rval += v
        rptr.append(len(rcol))
    return np.array(rptr, dtype=np.int32), np.array(rcol, dtype=np.int32), np.array(rval)


@pytest.mark.parametrize("block_rows", [8, 32, 64])
def test_tile_structure_invariants(block_rows):
    """S1 of a block is ascending, contains the block's own rows, and every entry's position points back at its
    column; pads carry PAD; entry 0 of a row is its diagonal slot."""
    W, coords = graphs.sensor_weights(700, k=6, seed=4)
    L = orc.laplacian(W)
    N = L.shape[0]
    perm = engine.locality_order(W, coords).astype(np.int64)
    rptr, rcol, rval = host_internal_csr(L, perm)
    t = tiling.build_tiles(rptr, rcol, N, block_rows)
    nb = t["nb"]
    assert nb == (N + block_rows - 1) // block_rows
    for b in (0, nb // 2, nb - 1):
        s1 = t["s1rows"][t["s1ptr"][b]:t["s1ptr"][b + 1]]
        own = np.arange(b * block_rows, min((b + 1) * block_rows, N))
        assert np.all(np.diff(s1) > 0)
        assert np.isin(own, s1).all()
        for i in own:
            e = np.arange(rptr[i], rptr[i + 1])
            real = rcol[e] < N
            assert np.array_equal(s1[t["lidx1"][e][real]], rcol[e][real])
            assert np.all(t["lidx1"][e][~real] == tiling.PAD)
            assert s1[t["lidx1"][e[0]]] == i  # entry 0 is the diagonal slot
    assert 0 < t["max_n1"] < tiling.PAD and t["mean_n1"] <= t["max_n1"]


def test_gather_tiles_reproduce_one_step():
    """A numpy model of k_step_tile's fast path - stage the S1 rows, gather by position - equals the sparse
    product."""
    W, coords = graphs.sensor_weights(900, k=7, seed=6)
    L = orc.laplacian(W)
    N = L.shape[0]
    perm = engine.locality_order(W, coords)
    rptr, rcol, rval = host_internal_csr(L, perm)
    t1 = tiling.build_tiles(rptr, rcol, N, 64)
    x = np.random.default_rng(0).standard_normal((N, 3))
    xi = x[perm]
    out = np.zeros_like(xi)
    lidx = t1["lidx1"].astype(np.int64)
    lidx[lidx == tiling.PAD] = 0  # what DeviceGraph.enable_gather_tiles uploads
    for b in range(t1["nb"]):
        s1 = t1["s1rows"][t1["s1ptr"][b]:t1["s1ptr"][b + 1]]
        assert (np.diff(s1) > 0).all() and s1.size <= 65535
        tile = xi[s1]
        for i in range(b * 64, min((b + 1) * 64, N)):
            s, e = rptr[i], rptr[i + 1]
            out[i] = (rval[s:e, None] * tile[lidx[s:e]]).sum(axis=0)
            assert s1[lidx[s]] == i  # entry 0 is the diagonal slot: the row itself
    ref = np.empty_like(out)
    ref[:] = L.dot(x)[perm]
    assert np.max(np.abs(out - ref)) < 1e-12
