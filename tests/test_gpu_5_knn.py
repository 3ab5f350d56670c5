"""Device nearest-neighbour graph construction against the reference's golden
vectors (tests/golden/knn.npz) and the KD-tree oracle (oracle/knn_oracle.py).  Needs a real
MI355X: `-m gpu`.

The bar here is integer/bit work for the neighbour lists and distances (same arithmetic as the
KD-tree: bit-exact), 1e-12 relative for the weights (the mean distance sigma is summed in a different order, and it sits
in the exponent).
"""
import numpy as np
import pytest

from conftest import csr_from
from oracle import knn_oracle as knn
from pygsp_amd import engine, graphs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return engine.default_context(0)


def check_against_oracle(ctx, X, k, sigma=None):
    W, sg, info = engine.knn_graph(X, k, sigma, ctx=ctx, neighbors=True)
    Wr, sr, NNr, Dr = knn.knn_weights(X, k, sigma)
    np.testing.assert_array_equal(info["NN"], NNr)
    np.testing.assert_array_equal(info["D"], Dr)           # bit for bit
    assert abs(sg - sr) <= 1e-13 * sr                       # summation order of the mean
    assert W.nnz == Wr.nnz
    assert W.has_sorted_indices
    np.testing.assert_array_equal(W.indptr, Wr.indptr)
    np.testing.assert_array_equal(W.indices, Wr.indices)
    assert np.max(np.abs(W.data - Wr.data) / Wr.data) < 1e-12
    assert abs(W - W.T).max() == 0
    return W


def test_golden(ctx, golden_knn):
    g = golden_knn
    G = graphs.NNGraph(g["X3"], k=5)
    np.testing.assert_allclose(G.coords, g["X3_coords"], rtol=0, atol=1e-14)
    Wref = csr_from(g, "W3")
    assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-15
    assert abs(G.sigma - float(g["sigma3"])) < 1e-15
    G1 = graphs.NNGraph(g["X1"], k=3, center=False, rescale=False, sigma=0.7)
    assert abs(G1.W - csr_from(g, "W1")).max() < 1e-15
    Gs = graphs.Sensor(123, seed=42)
    np.testing.assert_array_equal(Gs.coords, g["sensor_coords"])
    assert Gs.W.nnz == csr_from(g, "Wsensor").nnz and abs(Gs.W - csr_from(g, "Wsensor")).max() < 1e-15
    Gd = graphs.Sensor(144, k=4, distributed=True, seed=7)
    assert abs(Gd.W - csr_from(g, "Wdist")).max() < 1e-15
    with pytest.raises(ValueError):
        graphs.Sensor(10, k=10)
    with pytest.raises(ValueError):
        graphs.Sensor(10, distributed=True)
    with pytest.raises(NotImplementedError):
        graphs.NNGraph(g["X3"], dist_type="minkowski", order=2.5)
    with pytest.raises(ValueError):
        engine.knn_graph(np.zeros((100, 65)), 3, ctx=ctx)    # 65 dimensions: not covered, says so


@pytest.mark.parametrize("d", [1, 2, 3])
def test_oracle_uniform(ctx, d):
    rng = np.random.default_rng(10 + d)
    for N, k in ((3000, 6), (50000, 8), (20000, 20), (5000, 40)):
        check_against_oracle(ctx, rng.uniform(0, 1, (N, d)), k)


def test_oracle_clustered_and_degenerate(ctx):
    rng = np.random.default_rng(99)
    # strongly non-uniform: most points in a small blob (many points per grid cell)
    X = np.concatenate([rng.normal(0.5, 0.01, (4000, 2)), rng.uniform(0, 1, (1000, 2))])
    check_against_oracle(ctx, X, 7)
    # points on a line inside the plane (one extent is zero)
    X = np.stack([rng.uniform(0, 1, 2000), np.full(2000, 0.25)], axis=1)
    check_against_oracle(ctx, X, 4)
    # tiny graphs, k = N - 1
    check_against_oracle(ctx, rng.uniform(0, 1, (9, 2)), 8)
    check_against_oracle(ctx, rng.uniform(0, 1, (2, 3)), 1)
    # given sigma
    check_against_oracle(ctx, rng.uniform(0, 1, (1000, 3)), 5, sigma=0.05)
    # exact duplicates: every distance list still sorted, no self loops, symmetric
    X = rng.uniform(0, 1, (500, 2))
    X[100:110] = X[0]
    W, _, info = engine.knn_graph(X, 5, ctx=ctx, neighbors=True)
    assert (np.diff(info["D"], axis=1) >= 0).all() and W.diagonal().max() == 0
    assert (info["NN"] != np.arange(500)[:, None]).all() and abs(W - W.T).max() == 0


def test_sensor_headline_scale(ctx):
    """The bench graph: 1M points, k = 8 - identical to the host construction."""
    W, coords = graphs.sensor_weights(1000000, k=8, seed=42)
    Wd, sigma, info = engine.knn_graph(coords, 8, ctx=ctx)
    assert Wd.nnz == W.nnz
    np.testing.assert_array_equal(Wd.indices, W.indices)
    assert np.max(np.abs(Wd.data - W.data) / W.data) < 1e-12
    assert info["build_ms"] < 2000


def test_curve_keys_match_host_orders(ctx):
    """gspx_curve_keys (device) against the numpy Hilbert / Morton orders of engine.py: the same keys up
    to the rounding of (c - lo) / span, i.e. the same order except for a handful of points that sit
    on a cell boundary."""
    import ctypes
    from scipy import sparse
    from pygsp_amd import _capi
    rng = np.random.default_rng(5)
    N = 50000
    W = sparse.identity(N, format="csr")
    for d, curve in ((2, "hilbert"), (2, "morton"), (3, "morton")):
        X = rng.uniform(-3, 7, (N, d))
        keys = np.empty(N, dtype=np.uint64)
        _capi.check(_capi.load().gspx_curve_keys(ctx._h, N, d, _capi.ptr(X), 1 if curve == "hilbert" else 0,
                                                 _capi.ptr(keys)))
        dev_perm = np.argsort(keys, kind="stable")
        host_perm = engine.hilbert_order(X) if curve == "hilbert" else None
        if host_perm is None:
            # the numpy Morton branch of locality_order (small-N path): call it through a 4000-point slice
            # is not comparable; rebuild the host code here
            c = X[:, :3]
            lo = c.min(axis=0)
            span = c.max(axis=0) - lo
            bits = 21 if d == 3 else 31
            q = np.minimum(((c - lo) / span * (2 ** bits - 1)).astype(np.uint64), 2 ** bits - 1)
            code = np.zeros(N, dtype=np.uint64)
            for b in range(bits):
                for k in range(d):
                    code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * d + k)
            host_perm = np.argsort(code, kind="stable")
        assert np.mean(dev_perm == host_perm) > 0.999, (d, curve)
        assert sorted(dev_perm.tolist()) == list(range(N))
    # and the default order of a large 2-D graph is the Hilbert one
    Wk, coords = graphs.sensor_weights(20000, k=6, seed=1)
    p = engine.locality_order(Wk, coords)
    assert np.mean(p == engine.hilbert_order(coords)) > 0.999


# ---------------------------------------------------------------------------------------------
# stochastic block model / Erdos-Renyi sampler on the device (second half of SURVEY 8(f) row 4)
# ---------------------------------------------------------------------------------------------
def _block_counts(W, z, k):
    coo = W.tocoo()
    up = coo.row > coo.col
    m = np.zeros((k, k))
    np.add.at(m, (z[coo.row[up]], z[coo.col[up]]), 1)
    return m + m.T - np.diag(np.diag(m))  # unordered pairs per block pair


def test_sbm_sampler_distribution(ctx):
    """Not bit-comparable with the reference (different random stream): checked as a distribution -
    edge counts per block pair within 5 sigma of Binomial(#pairs, M[a, b]), symmetric, no self loops,
    no duplicates, deterministic per seed; cross-checked against the numpy sampler's counts."""
    rng = np.random.default_rng(0)
    N, k = 60000, 4
    z = np.sort(rng.integers(0, k, N))
    M = np.array([[3e-4, 2e-5, 0, 1e-5], [2e-5, 5e-4, 4e-5, 0], [0, 4e-5, 2e-4, 3e-5], [1e-5, 0, 3e-5, 6e-4]])
    W, ms = engine.sbm_graph(z, M, seed=123, ctx=ctx)
    assert W.shape == (N, N) and W.has_sorted_indices and W.has_canonical_format
    assert W.diagonal().max() == 0 and abs(W - W.T).max() == 0
    assert W.data.min() == 1 and W.data.max() == 1          # unit weights, no duplicate pair
    sizes = np.bincount(z, minlength=k).astype(np.float64)
    pairs = np.outer(sizes, sizes)
    pairs[np.diag_indices(k)] = sizes * (sizes - 1) / 2
    cnt = _block_counts(W, z, k)
    mean, sd = pairs * M, np.sqrt(pairs * M * (1 - M))
    assert (np.abs(cnt - mean) <= 5 * sd + 1e-9).all(), (cnt, mean)
    assert (cnt[M == 0] == 0).all()
    # same seed -> same graph; another seed -> another graph with the same statistics
    W2, _ = engine.sbm_graph(z, M, seed=123, ctx=ctx)
    assert abs(W - W2).max() == 0
    W3, _ = engine.sbm_graph(z, M, seed=124, ctx=ctx)
    assert abs(W - W3).nnz > 0 and (np.abs(_block_counts(W3, z, k) - mean) <= 5 * sd + 1e-9).all()
    # unsorted labels: the block structure follows z, not the vertex numbering
    zp = rng.permutation(z)
    Wp, _ = engine.sbm_graph(zp, M, seed=5, ctx=ctx)
    assert (np.abs(_block_counts(Wp, zp, k) - mean) <= 5 * sd + 1e-9).all()
    # degrees of an Erdos-Renyi graph: Binomial(N - 1, p), no vertex left out of the sampling
    G = graphs.ErdosRenyi(200000, p=2e-5, seed=3)
    d = np.ravel(G.W.sum(axis=0))
    assert abs(d.mean() - 199999 * 2e-5) < 5 * np.sqrt(199999 * 2e-5 / 200000) * 1.5
    assert abs(d.var() - 199999 * 2e-5) < 0.1 * 199999 * 2e-5
    first, last = d[:1000].mean(), d[-1000:].mean()         # the triangle's ends are sampled like its middle
    assert abs(first - 4.0) < 0.5 and abs(last - 4.0) < 0.5
    # dense probabilities and p = 1 (every pair): small graph
    Wd, _ = engine.sbm_graph(np.zeros(300, dtype=np.int64), np.array([[1.0]]), seed=1, ctx=ctx)
    assert Wd.nnz == 300 * 299
    Wh, _ = engine.sbm_graph(np.repeat([0, 1], 150), np.array([[0.7, 0.1], [0.1, 0.7]]), seed=1, ctx=ctx)
    c2 = _block_counts(Wh, np.repeat([0, 1], 150), 2)
    assert abs(c2[0, 0] - 0.7 * 150 * 149 / 2) < 5 * np.sqrt(0.21 * 150 * 149 / 2)
    assert abs(c2[0, 1] - 0.1 * 150 * 150) < 5 * np.sqrt(0.09 * 150 * 150)
    with pytest.raises(ValueError):
        engine.sbm_graph(z, np.array([[2.0]]), ctx=ctx)


def test_sbm_sampler_directed_self_loops_connected(ctx):
    """The other switches of stochasticblockmodel.py:69-72, 125-157, as distributions: directed (every ORDERED pair
    an entry of its own, asymmetric M allowed), self_loops (pairs r == c take part; an undirected loop is one
    entry), connected (fresh samples until one is connected, ValueError after n_try)."""
    rng = np.random.default_rng(1)
    N, k = 40000, 3
    z = np.sort(rng.integers(0, k, N))
    sizes = np.bincount(z, minlength=k).astype(np.float64)
    M = np.array([[4e-4, 3e-5, 0], [6e-5, 5e-4, 2e-5], [1e-5, 0, 3e-4]])  # not symmetric
    with pytest.raises(ValueError):
        engine.sbm_graph(z, M, seed=1, ctx=ctx)  # undirected graphs need a symmetric M

    def ordered_counts(W):
        coo = W.tocoo()
        off = coo.row != coo.col
        m = np.zeros((k, k))
        np.add.at(m, (z[coo.row[off]], z[coo.col[off]]), 1)
        return m, int(np.count_nonzero(~off)), np.bincount(z[coo.row[~off]], minlength=k)

    # directed, no self-loops: ordered pairs (r, c), r != c, each with probability M[z_r, z_c]
    W, _ = engine.sbm_graph(z, M, seed=7, ctx=ctx, directed=True)
    assert W.has_sorted_indices and W.data.min() == 1 and W.data.max() == 1 and W.diagonal().max() == 0
    pairs = np.outer(sizes, sizes) - np.diag(sizes)
    cnt, loops, _ = ordered_counts(W)
    mean, sd = pairs * M, np.sqrt(pairs * M * (1 - M))
    assert loops == 0 and (np.abs(cnt - mean) <= 5 * sd + 1e-9).all(), (cnt, mean)
    assert (cnt[M == 0] == 0).all() and abs(W - W.T).nnz > 0
    both = W.multiply(W.T).nnz  # reciprocated pairs: independent, so about sum p_ab p_ba over ordered pairs
    assert abs(both - np.sum(pairs * M * M.T)) <= 5 * np.sqrt(np.sum(pairs * M * M.T)) + 3
    W2, _ = engine.sbm_graph(z, M, seed=7, ctx=ctx, directed=True)
    assert abs(W - W2).max() == 0
    # directed with self-loops: the diagonal takes part with probability M[a, a]
    W, _ = engine.sbm_graph(z, M, seed=8, ctx=ctx, directed=True, self_loops=True)
    cnt, loops, per_block = ordered_counts(W)
    assert (np.abs(cnt - mean) <= 5 * sd + 1e-9).all()
    lm = sizes * np.diag(M)
    assert (np.abs(per_block - lm) <= 5 * np.sqrt(lm) + 1e-9).all() and loops == per_block.sum()
    # undirected with self-loops: symmetric, every loop stored once
    Ms = (M + M.T) / 2
    W, _ = engine.sbm_graph(z, Ms, seed=9, ctx=ctx, self_loops=True)
    assert abs(W - W.T).max() == 0 and W.data.max() == 1
    cnt, loops, per_block = ordered_counts(W)
    up = np.outer(sizes, sizes)
    up[np.diag_indices(k)] = sizes * (sizes - 1) / 2
    und = cnt + cnt.T  # both triangles were counted: every unordered pair twice
    und[np.diag_indices(k)] = np.diag(cnt)
    und = und / 2
    assert (np.abs(und - up * Ms) <= 5 * np.sqrt(up * Ms * (1 - Ms)) + 1e-9).all()
    lm = sizes * np.diag(Ms)
    assert (np.abs(per_block - lm) <= 5 * np.sqrt(lm) + 1e-9).all()
    # dense, tiny: p = 1 gives every ordered pair (and every loop)
    Wd, _ = engine.sbm_graph(np.zeros(50, dtype=np.int64), np.array([[1.0]]), seed=1, ctx=ctx, directed=True)
    assert Wd.nnz == 50 * 49
    Wd, _ = engine.sbm_graph(np.zeros(50, dtype=np.int64), np.array([[1.0]]), seed=1, ctx=ctx, directed=True,
                             self_loops=True)
    assert Wd.nnz == 50 * 50
    Wd, _ = engine.sbm_graph(np.zeros(50, dtype=np.int64), np.array([[1.0]]), seed=1, ctx=ctx, self_loops=True)
    assert Wd.nnz == 50 * 50 and abs(Wd - Wd.T).max() == 0
    # the generator classes: directed graphs take the host route of Graph.__init__ (symmetrised Laplacian)
    G = graphs.StochasticBlockModel(3000, k=3, p=0.02, q=0.002, directed=True, seed=2)
    assert G.is_directed() and G.Ne == G.W.nnz and G.W.dtype == np.int64 and G.W.diagonal().max() == 0
    assert list(G.info) == ["node_com", "comm_sizes", "world_rad"] and G.info["comm_sizes"].sum() == 3000
    G = graphs.ErdosRenyi(2000, p=0.01, self_loops=True, seed=3)
    assert not G.is_directed() and G.W.diagonal().sum() > 0
    assert G.Ne == (G.W.nnz - np.count_nonzero(G.W.diagonal())) // 2 + np.count_nonzero(G.W.diagonal())
    # connected=True: a dense enough model is connected at the first try, a sparse one never is
    G = graphs.ErdosRenyi(3000, p=0.01, connected=True, seed=4)
    assert G.is_connected() and G.connected
    G = graphs.StochasticBlockModel(2000, k=2, p=0.02, q=0.002, directed=True, connected=True, n_try=20, seed=5)
    assert G.is_connected() and G.is_directed()
    with pytest.raises(ValueError, match="could not be connected after 3 trials"):
        graphs.ErdosRenyi(3000, p=1e-4, connected=True, n_try=3, seed=6)
    assert graphs.Graph(np.array([[0, 3, 0, 0], [3, 0, 4, 0], [0, 4, 0, 2], [0, 0, 2, 0.0]])).is_connected()
    assert not graphs.Graph(np.array([[0, 3, 0, 0], [3, 0, 4, 0], [0, 0, 0, 2], [0, 0, 2, 0.0]])).is_connected()
    G = graphs.StochasticBlockModel(2000, k=3, seed=7)       # defaults p = 0.7, q = 0.1
    assert G.W.shape == (2000, 2000) and np.array_equal(G.z, np.sort(np.random.default_rng(7).integers(0, 3, 2000)))


# ---------------------------------------------------------------------------------------------
# radius graphs (NNtype='radius', nngraph.py:228-287)
# ---------------------------------------------------------------------------------------------
def test_radius_graphs(ctx, golden_knn):
    g = golden_knn
    G = graphs.NNGraph(g["Xr"], NNtype="radius", epsilon=0.35)
    Wref = csr_from(g, "Wr")
    assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-14 and abs(G.sigma - float(g["sigma_r"])) < 1e-14
    G2 = graphs.NNGraph(g["X2r"], NNtype="radius", epsilon=0.08, center=False, rescale=False, sigma=0.01)
    assert G2.W.nnz == csr_from(g, "W2r").nnz and abs(G2.W - csr_from(g, "W2r")).max() < 1e-15
    rng = np.random.default_rng(17)
    for d, N, eps in ((1, 2000, 0.002), (2, 3000, 0.03), (3, 2500, 0.09), (2, 500, 0.6)):
        X = rng.uniform(0, 1, (N, d))
        W, sigma, _ = engine.radius_graph(X, eps, ctx=ctx)
        Wr, sr = knn.radius_weights(X, eps)
        assert W.nnz == Wr.nnz, (d, N)
        np.testing.assert_array_equal(W.indices, Wr.indices)   # the same neighbour sets
        assert abs(sigma - sr) <= 1e-13 * sr
        assert np.max(np.abs(W.data - Wr.data) / Wr.data) < 1e-12
        assert abs(W - W.T).max() == 0 and W.diagonal().max() == 0
    with pytest.raises(ValueError):
        engine.radius_graph(rng.uniform(0, 1, (50, 2)) * 100, 0.01, ctx=ctx)   # "No neighbors found"
    with pytest.raises(ValueError):
        graphs.NNGraph(g["Xr"], NNtype="hexagon")


def test_other_metrics(ctx, golden_knn):
    """dist_type 'manhattan' / 'max_dist' (and 'minkowski' of order 1, 2, inf), nngraph.py:139-145."""
    g = golden_knn
    G = graphs.NNGraph(g["X3"], k=6, dist_type="manhattan")
    assert G.W.nnz == csr_from(g, "W3_manhattan").nnz and abs(G.W - csr_from(g, "W3_manhattan")).max() < 1e-14
    G = graphs.NNGraph(g["X3"], k=4, dist_type="max_dist", center=False, rescale=False)
    assert abs(G.W - csr_from(g, "W3_maxdist")).max() < 1e-14
    G = graphs.NNGraph(g["X3"], k=4, dist_type="minkowski", order=np.inf, center=False, rescale=False)
    assert abs(G.W - csr_from(g, "W3_maxdist")).max() < 1e-14
    G = graphs.NNGraph(g["X2r"], NNtype="radius", epsilon=0.07, dist_type="manhattan", center=False, rescale=False)
    assert G.W.nnz == csr_from(g, "W2r_manhattan").nnz and abs(G.W - csr_from(g, "W2r_manhattan")).max() < 1e-14
    with pytest.raises(NotImplementedError):
        graphs.NNGraph(g["X3"], dist_type="minkowski", order=3)
    rng = np.random.default_rng(8)
    for metric in ("manhattan", "max_dist"):
        for d, N, k in ((1, 1500, 3), (2, 20000, 7), (3, 8000, 12)):
            X = rng.uniform(0, 1, (N, d))
            W, sg, info = engine.knn_graph(X, k, ctx=ctx, neighbors=True, metric=metric)
            Wr, sr, NNr, Dr = knn.knn_weights(X, k, dist_type=metric)
            np.testing.assert_array_equal(info["NN"], NNr)
            np.testing.assert_array_equal(info["D"], Dr)
            assert abs(sg - sr) <= 1e-13 * sr and np.max(np.abs(W.data - Wr.data) / Wr.data) < 1e-12
        X = rng.uniform(0, 1, (2500, 2))
        W, sg, _ = engine.radius_graph(X, 0.03, ctx=ctx, metric=metric)
        Wr, sr = knn.radius_weights(X, 0.03, dist_type=metric)
        assert W.nnz == Wr.nnz
        np.testing.assert_array_equal(W.indices, Wr.indices)


def test_symmetrize_types(ctx, golden_knn):
    """symmetrize_type of NNGraph: utils.symmetrize 'average' / 'maximum' / 'fill' / 'tril' / 'triu'."""
    g = golden_knn
    for st in ("maximum", "fill", "tril", "triu"):
        G = graphs.NNGraph(g["X3"], k=5, symmetrize_type=st)
        Wref = csr_from(g, "W3_" + st)
        Wref.eliminate_zeros()
        assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-14, st
        assert abs(G.W - G.W.T).max() == 0
    with pytest.raises(ValueError):
        graphs.NNGraph(g["X3"], symmetrize_type="sum")
    rng = np.random.default_rng(4)
    X = rng.uniform(0, 1, (20000, 2))
    for st in ("maximum", "tril", "triu"):
        W, _, _ = engine.knn_graph(X, 6, ctx=ctx, symmetrize=st)
        Wr = knn.knn_weights(X, 6, symmetrize_type=st)[0]
        assert W.nnz == Wr.nnz, st
        np.testing.assert_array_equal(W.indices, Wr.indices)
        assert np.max(np.abs(W.data - Wr.data) / Wr.data) < 1e-12


# ---- more than three dimensions: tiled brute force, pair distances on the matrix cores (gspx_knn_bf.hip.h) ----------
def _search_stats(ctx, X, k, metric="euclidean"):
    import ctypes
    from pygsp_amd import _capi
    lib = _capi.load()
    h = ctypes.c_void_p()
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    _capi.check(lib.gspx_knn_build(ctx._h, Xc.shape[0], Xc.shape[1], _capi.ptr(Xc), k, 0.0, engine.METRICS[metric], 0,
                                   ctypes.byref(h)))
    out = np.zeros(4)
    _capi.check(lib.gspx_knn_search_stats(h, _capi.ptr(out)))
    lib.gspx_knn_destroy(h)
    return dict(sample=int(out[0]), capacity=int(out[1]), mean_candidates=float(out[2]), exact_scans=int(out[3]))


def test_highdim_golden(ctx):
    """The reference's NNGraph on 9- and 25-dimensional patch clouds, a 6-D cloud (manhattan; 'maximum'
    symmetrisation) and a 40-D one (tests/golden/knn_highdim.npz, generated from the real pygsp)."""
    from conftest import load_golden
    g = load_golden("knn_highdim.npz")
    for tag, k in (("p9", 8), ("p25", 10)):
        G = graphs.NNGraph(g["X_" + tag], k=k)
        np.testing.assert_allclose(G.coords, g["coords_" + tag], rtol=0, atol=1e-13)
        Wref = csr_from(g, "W_" + tag)
        assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-14
        assert abs(G.sigma - float(g["sigma_" + tag])) < 1e-13
    G = graphs.NNGraph(g["X6"], k=7, dist_type="manhattan")
    assert abs(G.W - csr_from(g, "W6_manhattan")).max() < 1e-14
    G = graphs.NNGraph(g["X6"], k=5, symmetrize_type="maximum", center=False, rescale=False)
    assert abs(G.W - csr_from(g, "W6_maximum")).max() < 1e-14
    G = graphs.NNGraph(g["X40"], k=12, sigma=2.5)
    assert abs(G.W - csr_from(g, "W40")).max() < 1e-14
    # radius graphs beyond three dimensions (nngraph.py:228-287)
    G = graphs.NNGraph(g["X6"], NNtype="radius", epsilon=0.125)
    Wref = csr_from(g, "W6_radius")
    assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-12 and abs(G.sigma - float(g["sigma6_radius"])) < 1e-13
    G = graphs.NNGraph(g["X_p9"], NNtype="radius", epsilon=0.3, dist_type="manhattan")
    Wref = csr_from(g, "Wp9_radius_manhattan")
    assert G.W.nnz == Wref.nnz and abs(G.W - Wref).max() < 1e-12


def test_highdim_radius_oracle(ctx):
    """Ball queries in 5 to 40 dimensions against the KD-tree oracle: identical neighbour sets (the membership test runs
    in the KD-tree's arithmetic on the candidates of the MFMA sweep), weights to 1e-12; a radius that catches nobody is
    the reference's ValueError."""
    rng = np.random.default_rng(17)
    for N, d, eps in ((8000, 5, 0.6), (5000, 12, 2.2), (3000, 40, 6.4), (200, 6, 1.5)):
        X = rng.standard_normal((N, d))
        W, sg, _ = engine.radius_graph(X, eps, ctx=ctx)
        Wr, sr = knn.radius_weights(X, eps)
        assert W.nnz == Wr.nnz and W.nnz > 0, (N, d, W.nnz, Wr.nnz)
        assert np.array_equal(W.indptr, Wr.indptr) and np.array_equal(W.indices, Wr.indices)
        assert abs(sg - sr) <= 1e-12 * sr and np.max(np.abs(W.data - Wr.data) / Wr.data) < 1e-11
    with pytest.raises(ValueError, match="No neighbors"):
        engine.radius_graph(rng.standard_normal((500, 8)), 1e-3, ctx=ctx)


@pytest.mark.parametrize("N,d,k", [(20000, 4, 6), (20000, 8, 10), (12000, 25, 10), (6000, 33, 17), (3000, 64, 40),
                                   (70, 5, 9), (1000, 16, 1)])
def test_highdim_oracle(ctx, N, d, k):
    """Neighbours and distances equal scipy's KD-tree bit for bit in 4 to 64 dimensions (random clouds: no ties),
    through the MFMA candidate sweep for the large clouds and the exact scan for the tiny one."""
    rng = np.random.default_rng(N + d)
    X = rng.standard_normal((N, d)) * rng.uniform(0.5, 2.0, d) + rng.uniform(-1, 1, d)
    # the candidate sweep on the fp64 matrix cores, on the fp32 ones (forced; the default picks it when its rounding
    # margin is small against the bounds - it is, on these clouds), and the default: the selection is exact either way
    for f32 in (0, 2, 1):
        ctx.set_option("knn_f32", f32)
        try:
            check_against_oracle(ctx, X, k)
            st = _search_stats(ctx, X, k)
        finally:
            ctx.set_option("knn_f32", 1)
        if N > 4 * k + 64:  # (a query in the thin tail of the cloud may outgrow its list and take the exact scan)
            assert st["exact_scans"] <= N // 1000 and k <= st["mean_candidates"] <= st["capacity"], (f32, st)


def test_highdim_clustered_ties_and_other_metrics(ctx):
    """Tight clusters (most candidates of a query pass its bound: lists overflow, those queries take the exact
    scan), exact ties (a lattice: equal distances are ordered by index, like the grid search does in 1-3-D and the
    KD-tree oracle confirms on these clouds), duplicates, and the metrics without a product form."""
    rng = np.random.default_rng(8)
    centres = rng.standard_normal((5, 7)) * 50
    Xc = centres[rng.integers(0, 5, 9000)] + 1e-3 * rng.standard_normal((9000, 7))
    check_against_oracle(ctx, Xc, 8)
    # (|x|^2 ~ 1e4 against neighbour distances of 1e-3: the default keeps the fp64 sweep here; forcing the fp32 one
    # admits whole clusters, the lists overflow, the exact scan takes over - slower, never wrong)
    ctx.set_option("knn_f32", 2)
    try:
        check_against_oracle(ctx, Xc, 8)
    finally:
        ctx.set_option("knn_f32", 1)
    for metric in ("manhattan", "max_dist"):
        X = rng.standard_normal((4000, 6))
        W, sg, info = engine.knn_graph(X, 7, ctx=ctx, neighbors=True, metric=metric)
        D, NN = knn.knn_query(X, 7, metric)
        np.testing.assert_array_equal(info["NN"], NN[:, 1:])
        np.testing.assert_array_equal(info["D"], D[:, 1:])
    # a 4-D lattice with jitter-free integer coordinates: many exactly equal distances
    g4 = np.stack(np.meshgrid(*[np.arange(7.0)] * 4, indexing="ij"), -1).reshape(-1, 4)
    W, sg, info = engine.knn_graph(g4, 8, ctx=ctx, neighbors=True)
    D, NN = knn.knn_query(g4, 8)
    np.testing.assert_array_equal(info["D"], D[:, 1:])       # the distances are unambiguous
    assert (np.sort(info["NN"][:, :1], axis=1) >= 0).all()
    # every reported neighbour really is at the reported distance, and ties are resolved by index
    d_chk = np.sqrt(((g4[:, None, :] - g4[info["NN"]]) ** 2).sum(-1))
    np.testing.assert_array_equal(d_chk, info["D"])
    same = info["D"][:, 1:] == info["D"][:, :-1]
    assert (info["NN"][:, 1:][same] > info["NN"][:, :-1][same]).all()


def test_highdim_scale(ctx):
    """100k points in 16 dimensions, k = 10: the size where a KD-tree query takes tens of seconds on the host."""
    import time
    rng = np.random.default_rng(1)
    X = rng.standard_normal((100000, 16))
    X = X[np.argsort(X[:, 0])]  # an ORDERED cloud (like patches in raster order): the first sweep's subset is strided
    t0 = time.perf_counter()
    W, sg, info = engine.knn_graph(X, 10, ctx=ctx, neighbors=True)
    dt = time.perf_counter() - t0
    st = _search_stats(ctx, X, 10)
    assert st["exact_scans"] <= 100
    # oracle on a slice of the queries (the full KD-tree query is what we are replacing)
    from scipy import spatial
    rows = rng.choice(100000, 300, replace=False)
    D, NN = spatial.KDTree(X).query(X[rows], k=11)
    np.testing.assert_array_equal(info["NN"][rows], NN[:, 1:])
    np.testing.assert_array_equal(info["D"][rows], D[:, 1:])
    assert abs(W - W.T).max() == 0 and dt < 30
