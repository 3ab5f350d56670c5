"""Graph set-up on the device (gspx_graph_setup, gspx_curve_order): the checks of Graph.__init__
(pygsp/graphs/graph.py:98-134), the directedness test (graph.py:357-405), the internal vertex order and the
Laplacian (graph.py:510-630) in one call on the uploaded matrix - against the host route (numpy / scipy, the
reference's steps one by one) and the oracle.  `-m gpu`."""
import logging
import time

import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err  # noqa: F401
from gpu_helpers import ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu


def _same_csr(A, B):
    A, B = sparse.csr_matrix(A), sparse.csr_matrix(B)
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(A.data, B.data))


def test_curve_order_is_the_stable_argsort_of_the_keys(ctx):
    """gspx_curve_order (radix sort on the device) == numpy.argsort(keys, kind='stable') of gspx_curve_keys: uniform
    clouds, and clustered / quantised ones where thousands of points share a key, so that stability is tested."""
    rng = np.random.default_rng(11)
    lib = _capi.load()
    for N, d, curve, kind in ((50000, 2, 1, "uniform"), (50000, 2, 0, "uniform"), (70001, 3, 0, "uniform"),
                              (40000, 2, 1, "lattice"), (40000, 2, 0, "lattice"), (4097, 3, 0, "lattice"),
                              (300000, 2, 1, "uniform")):
        X = rng.uniform(-2, 5, (N, d))
        if kind == "lattice":
            X = np.round(X * 3) / 3  # ~ 20 distinct values per axis: long runs of equal keys
        keys = np.empty(N, dtype=np.uint64)
        perm = np.empty(N, dtype=np.int32)
        _capi.check(lib.gspx_curve_keys(ctx._h, N, d, _capi.ptr(X), curve, _capi.ptr(keys)))
        _capi.check(lib.gspx_curve_order(ctx._h, N, d, _capi.ptr(X), curve, _capi.ptr(perm)))
        assert np.array_equal(perm, np.argsort(keys, kind="stable")), (N, d, curve, kind)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_setup_equals_host_route(ctx, dtype):
    """A clean undirected CSR graph with coordinates: the one-call device set-up gives the Laplacian of the oracle bit
    for bit, the edge count and directedness of the host route, the vertex order of engine.auto_order, and the
    same filter output."""
    W, coords = graphs.sensor_weights(50000, k=7, seed=5)
    G = graphs.Graph(W, coords=coords, compute_dtype=dtype)
    rep = G.setup_report
    assert rep["built"] and rep["reordered"] and rep["asymmetric"] == 0 and rep["self_loops"] == 0
    assert rep["locality_curve"] > rep["locality_own"] + 0.05
    assert not G.is_directed() and G.n_edges == W.nnz // 2 and G.N == 50000
    L = orc.laplacian(W).astype(dtype)
    # (float32: the device sums a row's weights in float32, the oracle in float64 and then rounds)
    assert _same_csr(G.L, L) if dtype == np.float64 else abs(G.L - L).max() < 5e-6
    assert np.array_equal(G._internal_order(), engine.auto_order(W, coords, ctx=ctx))
    # the host route on the same matrix (a COO input does not qualify for the fast path)
    H = graphs.Graph(sparse.coo_matrix(W), coords=coords, compute_dtype=dtype)
    assert not hasattr(H, "setup_report") and H.n_edges == G.n_edges
    assert np.array_equal(H._internal_order(), G._internal_order()) and _same_csr(H.L, G.L)
    G.estimate_lmax("bounds")
    H.estimate_lmax("bounds")
    x = np.random.default_rng(2).standard_normal((G.N, 8))
    yg, yh = filters.Heat(G, 10).filter(x, order=15), filters.Heat(H, 10).filter(x, order=15)
    assert np.array_equal(yg, yh)
    # compute_laplacian() afterwards rebuilds through the ordinary path with the order chosen on the device
    G.compute_laplacian("normalized")
    assert abs(G.L - orc.laplacian(W, "normalized")).max() < (1e-14 if dtype == np.float64 else 1e-6)
    assert np.array_equal(G._internal_order(), H._internal_order())


def test_what_the_device_inspection_reports(ctx, caplog):
    """NaN / inf raise the reference's ValueError (graph.py:112-117); self-loops and negative weights warn
    (graph.py:119-134) and self-loops count as edges; a directed graph, explicit zeros and a non-canonical CSR go
    to the host route and come out as the reference has them."""
    W = random_graph(3000, 6, 3)
    base = orc.laplacian(W)
    G = graphs.Graph(W)
    assert G.setup_report["built"] and _same_csr(G.L, base) and G.n_edges == W.nnz // 2
    bad = W.copy()
    bad.data[5] = np.nan
    with pytest.raises(ValueError, match="Not a Number"):
        graphs.Graph(bad)
    bad.data[5] = np.inf
    with pytest.raises(ValueError, match="infinite"):
        graphs.Graph(bad)
    # self-loops and negative weights: warnings, the loops are edges of their own
    loops = np.arange(0, 3000, 100)
    Wl = sparse.csr_matrix(W + sparse.coo_matrix((np.full(30, 2.0), (loops, loops)), shape=W.shape))
    with caplog.at_level(logging.WARNING, logger="pygsp_amd.graphs"):
        caplog.clear()
        Gl = graphs.Graph(Wl)
    assert Gl.setup_report["built"] and Gl.setup_report["self_loops"] == 30
    assert any("self-loops" in r.message for r in caplog.records)
    assert Gl.n_edges == (Wl.nnz - 30) // 2 + 30 and abs(Gl.L - orc.laplacian(Wl)).max() < 1e-13
    Wn = W.copy()
    Wn.data = -Wn.data
    with caplog.at_level(logging.WARNING, logger="pygsp_amd.graphs"):
        caplog.clear()
        graphs.Graph(Wn)
    assert any("negative" in r.message for r in caplog.records)
    # directed: one entry that differs from its mirror -> (W + W.T) / 2 as utils.symmetrize does (graph.py:613-616), in
    # the same device call since round 5
    Wd = sparse.lil_matrix(W)
    r, c = W.nonzero()
    Wd[r[0], c[0]] = 0.123
    Wd = sparse.csr_matrix(Wd)
    Gd = graphs.Graph(Wd)
    assert Gd.setup_report["built"] and Gd.is_directed() and Gd.n_edges == Wd.nnz
    Ls = orc.laplacian(sparse.csr_matrix((Wd + Wd.T) / 2))
    assert abs(Gd.L - Ls).max() == 0 and Gd.L.nnz == Ls.nnz
    dev, rep = engine.DeviceGraph.setup(Wd, ctx=ctx)
    assert dev is not None and rep["asymmetric"] == 2 and rep["built"]
    dev.destroy()
    # explicit zeros are dropped, as scipy's eliminate_zeros does in the reference
    Wz = W.copy()
    i = int(np.searchsorted(Wz.indptr, 0, side="right") - 1)  # the row of stored entry 0 ...
    j = int(Wz.indices[0])                                     # ... and its column: zero (i, j) and its mirror (j, i)
    Wz.data[0] = 0.0
    Wz.data[Wz.indptr[j] + int(np.searchsorted(Wz.indices[Wz.indptr[j]:Wz.indptr[j + 1]], i))] = 0.0
    assert Wz.nnz == W.nnz  # still stored
    dev, rep = engine.DeviceGraph.setup(Wz, ctx=ctx)
    assert dev is not None and rep["zeros"] == 2 and rep["asymmetric"] == 0 and rep["built"]
    dev.destroy()
    Wz2 = Wz.copy()
    Wz2.eliminate_zeros()
    Wz_in = Wz.copy()
    Gz = graphs.Graph(Wz_in)
    assert Gz.setup_report["built"] and not Gz.is_directed() and Gz.W.nnz == Wz2.nnz  # (W loses them, graph.py:126-128)
    assert Gz.n_edges == Wz2.nnz // 2
    assert abs(Gz.L - orc.laplacian(Wz2)).max() < 1e-13 and Gz.L.nnz == orc.laplacian(Wz2).nnz
    # unsorted column indices: refused by the device validation, canonicalised by the host route
    Wu = W.copy()
    s, e = Wu.indptr[10], Wu.indptr[11]
    if e - s >= 2:
        Wu.indices[s:e] = Wu.indices[s:e][::-1].copy()
        Wu.data[s:e] = Wu.data[s:e][::-1].copy()
        Wu.has_sorted_indices = False
        with pytest.raises(ValueError, match="canonical"):
            engine.DeviceGraph.setup(Wu, ctx=ctx)
        assert _same_csr(graphs.Graph(Wu).L, base)


def test_int64_adjacency_and_small_graphs(ctx):
    """The reference's ER / SBM constructors store int64 unit weights (stochasticblockmodel.py:127-141): converted on
    the device; the Laplacian is float64 as scipy makes it.  Empty and tiny graphs take the same route."""
    Wi = sparse.csr_matrix((random_graph(2500, 5, 9) > 0).astype(np.int64))
    for lap in ("combinatorial", "normalized"):
        G = graphs.Graph(Wi, lap_type=lap)
        assert G.setup_report["built"]
        assert abs(G.L - orc.laplacian(Wi.astype(np.float64), lap)).max() < 1e-14
    # a large random graph without coordinates: recognised as random-like (engine.expander_like), so the one-call
    # route is taken without a reverse Cuthill-McKee pass, in the graph's own order
    Ge = graphs.ErdosRenyi(50000, p=10.0 / 50000, seed=3)
    assert Ge.setup_report["built"] and not Ge.setup_report["reordered"] and Ge._internal_order() is None
    assert abs(Ge.L - orc.laplacian(Ge.W.astype(np.float64))).max() < 1e-14
    for W in (sparse.csr_matrix((6, 6)), sparse.identity(6, format="csr"), sparse.csr_matrix((1, 1))):
        for lap in ("combinatorial", "normalized"):
            G = graphs.Graph(W, lap_type=lap)
            assert G.setup_report["built"] and G.L.nnz == 0 and G.L.shape == W.shape


def test_setup_time_at_headline_size(ctx):
    """Graph(W, coords) of the 1M-vertex headline graph: one upload + device work.  Round 2 spent 0.27 s here in
    host numpy (directedness check, canonicalisation, argsort of the curve keys, locality score); the bound below
    is loose on purpose (a shared box), the bench line reports the measured time."""
    coords = np.random.default_rng(42).uniform(0, 1, (1000000, 2))
    W, _, _ = engine.knn_graph(coords, 8, ctx=ctx)
    graphs.Graph(W, coords=coords)  # warm-up: kernels, allocations
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        G = graphs.Graph(W, coords=coords)
        best = min(best, time.perf_counter() - t0)
    assert G.setup_report["built"] and G.setup_report["reordered"] and G.tile_stats["enabled"]
    assert best < 0.12, best
    L = orc.laplacian(W)
    assert _same_csr(G.L, L)


def _host_upper_bound(W):
    """Graph._get_upper_bound of the reference (graph.py:933-960) in numpy, for an undirected W."""
    W = sparse.csr_matrix(W)
    deg = np.ravel(W.sum(axis=0)).astype(np.float64)
    cands = [W.shape[0] * W.max(), 2 * deg.max()]
    if W.nnz:
        coo = W.tocoo()
        cands.append((deg[coo.row] + deg[coo.col]).max())
    with np.errstate(divide="ignore", invalid="ignore"):
        cands.append(np.max(deg + W.dot(deg) / deg))
    return min(cands), cands


def test_upper_bound_of_lmax_from_the_device_pass(ctx):
    """estimate_lmax('bounds'): the four candidates of graph.py:933-960 come out of one device pass over W
    (gspx_graph_lmax_bounds) and equal the host formula - the row sums run in scipy's order, so to the last bit;
    isolated vertices make the fourth candidate NaN, which min() ignores, as in the reference."""
    W, coords = graphs.sensor_weights(50000, k=7, seed=5)
    G = graphs.Graph(W, coords=coords)
    ref, cands = _host_upper_bound(W)
    got = G.device_graph().lmax_bounds()
    assert got is not None
    assert [W.shape[0] * got[0], 2 * got[1], got[2], got[3]] == pytest.approx(cands, rel=1e-15, abs=0)
    G.estimate_lmax("bounds")
    assert G.lmax == ref
    # isolated vertices (ER at p N = 10 has e^-10 of them) and int64 weights
    Ge = graphs.ErdosRenyi(300000, p=10.0 / 300000, seed=1)
    assert (np.ravel(Ge.W.sum(axis=0)) == 0).any()
    ref, cands = _host_upper_bound(Ge.W)
    assert np.isnan(cands[3]) and np.isnan(Ge.device_graph().lmax_bounds()[3])
    Ge.estimate_lmax("bounds")
    assert Ge.lmax == ref
    # self-loops enter the degrees and the edge maximum
    Wl = sparse.csr_matrix(random_graph(3000, 6, 3) + sparse.coo_matrix((np.full(30, 2.0), (np.arange(30) * 100,) * 2),
                                                                        shape=(3000, 3000)))
    Gl = graphs.Graph(Wl)
    Gl.estimate_lmax("bounds")
    assert Gl.lmax == _host_upper_bound(Wl)[0]
    # a float32 graph keeps the host formula
    G32 = graphs.Graph(W, coords=coords, compute_dtype=np.float32)
    G32.estimate_lmax("bounds")
    assert abs(G32.lmax - ref) >= 0 and G32.lmax > 0


def test_generators_hand_their_device_adjacency_over(ctx):
    """NNGraph / Sensor / StochasticBlockModel end in Graph.__init__(W) (nngraph.py:289-313,
    stochasticblockmodel.py:144-181).  Here the W their device builders produced stays on the device
    (engine.DeviceAdjacency -> gspx_graph_setup_from_knn): same Laplacian, edge count, vertex order and filter output as
    the graph built from the downloaded matrix, and G.W - made on first access - is that matrix."""
    G = graphs.Sensor(30000, k=6, seed=3)
    assert G._adj_dev is not None and G._adj_host is None and not G.is_directed()
    assert G.setup_report["built"] and G.setup_report["reordered"]
    Wh, coords = graphs.sensor_weights(30000, k=6, seed=3)
    assert G.n_edges == Wh.nnz // 2 and G.N == 30000
    G.estimate_lmax("bounds")
    assert G._adj_dev is not None  # (the bounds came from the device pass: nobody has asked for W yet)
    H = graphs.Graph(Wh, coords=coords)
    H.estimate_lmax("bounds")
    assert G.lmax == pytest.approx(H.lmax, rel=1e-12)
    s = np.random.default_rng(0).standard_normal((G.N, 3))
    yg = filters.Heat(G, 10).filter(s, order=20)
    yh = filters.Heat(H, 10).filter(s, order=20)
    assert rel_err(yg, yh) < 1e-12
    assert np.array_equal(G._internal_order(), H._internal_order())
    assert G._adj_dev is not None
    W = G.W  # now the download
    assert G._adj_dev is None and sparse.isspmatrix_csr(W) and abs(W - Wh).max() < 1e-15
    assert _same_csr(G.L, orc.laplacian(W))
    # a second Laplacian type on the same graph object: rebuilt from the (now host) matrix
    G.compute_laplacian("normalized")
    assert abs(G.L - orc.laplacian(W, "normalized")).max() < 1e-14
    # the fp32 engine on the device-resident float64 W (converted there)
    F = graphs.Sensor(9000, k=6, seed=4, compute_dtype=np.float32)
    assert F._adj_dev is not None
    F.estimate_lmax("bounds")
    sf = np.random.default_rng(1).standard_normal((F.N, 5))
    yf = filters.Heat(F, 10).filter(sf, order=20)
    ref = orc.cheby_op(orc.laplacian(F.W), F.lmax, filters.compute_cheby_coeff(filters.Heat(F, 10), m=20),
                       sf.astype(np.float32).astype(np.float64))
    assert rel_err(yf, ref) < 1e-4
    # radius graphs, another symmetrisation type (utils.symmetrize 'tril': the lower triangle mirrored)
    X = np.random.default_rng(5).uniform(0, 1, (5000, 2))
    R = graphs.NNGraph(X, NNtype="radius", epsilon=0.03, rescale=False, center=False)
    assert R._adj_dev is not None and _same_csr(R.L, orc.laplacian(R.W))
    T = graphs.NNGraph(X, k=5, symmetrize_type="tril", rescale=False, center=False)
    assert T._adj_dev is not None and not T.is_directed() and abs(T.W - T.W.T).nnz == 0 and _same_csr(T.L, orc.laplacian(T.W))
    # block models: unit int64 weights in the host copy, like the reference's W
    B = graphs.StochasticBlockModel(6000, k=3, p=0.01, q=0.001, seed=2)
    assert B._adj_dev is not None
    yb = filters.Heat(B, 5).filter(np.ones(B.N), order=10)
    WB = B.W
    assert WB.dtype == np.int64 and set(np.unique(WB.data)) == {1} and abs(WB - WB.T).nnz == 0
    assert B.n_edges == WB.nnz // 2 and _same_csr(B.L, orc.laplacian(WB))
    assert rel_err(yb, orc.cheby_op(orc.laplacian(WB), B.lmax, filters.compute_cheby_coeff(filters.Heat(B, 5), m=10),
                                    np.ones(B.N))) < 1e-12


@pytest.mark.parametrize("lap_type", ["combinatorial", "normalized"])
@pytest.mark.parametrize("wdtype", [np.float64, np.float32, np.int64])
def test_directed_adjacency_is_symmetrised_on_the_device(ctx, golden_lap4, lap_type, wdtype):
    """VERDICT r4 "Next 6": (W + W.T) / 2 of a directed W - utils.symmetrize 'average', graph.py:613-616, utils.py:247-248 -
    and the dropping of explicit / cancelling zeros run inside gspx_graph_setup (transpose by the radix sort, two-pointer
    merge).  L and dw against the oracle and against the reference's goldens (laplacians4.npz `dir`), for float64 /
    float32 / int64 weights, self-loops, entries that cancel, empty rows; then one graph at size."""
    # the reference's own 4 x 4 directed fixture
    Wg = sparse.csr_matrix(golden_lap4["W_dir"])
    G = graphs.Graph(Wg, lap_type=lap_type)
    assert G.is_directed() and G.setup_report["built"]  # (a scipy CSR matrix: the one-call device route)
    ref = golden_lap4["L_dir_" + lap_type]
    assert np.max(np.abs(G.L.toarray() - ref)) < 1e-15 and G.L.nnz == np.count_nonzero(ref)
    assert np.allclose(G.dw, golden_lap4["dw_dir"], rtol=0, atol=1e-15)
    # random directed graphs: ragged, with self-loops, a pair that cancels (w_ij = -w_ji), isolated vertices
    rng = np.random.default_rng(17)
    n = 5000
    A = sparse.random(n, n, 1.5e-3, random_state=5, format="lil", data_rvs=lambda k: rng.integers(1, 9, k).astype(np.float64))
    A[7, 7] = 3.0          # self-loop
    A[11, 400] = 2.0       # cancels with its mirror: (W + W.T) has no entry there
    A[400, 11] = -2.0
    for v in (0, 123, n - 1):
        A[v, :] = 0
        A[:, v] = 0
    A = sparse.csr_matrix(A)
    A.eliminate_zeros()
    A.sort_indices()
    W = A.astype(wdtype)
    Ws = sparse.csr_matrix((W + W.T) / 2)  # what the reference builds its Laplacian from
    Gd = graphs.Graph(W.copy(), lap_type=lap_type)
    assert Gd.setup_report["built"] and Gd.is_directed() and Gd.n_edges == W.nnz
    tol = 1e-6 if wdtype == np.float32 else 1e-13
    Lref = orc.laplacian(W, lap_type)  # (degrees (in + out) / 2 of W itself, graph.py:834-837; Laplacian of (W + W.T) / 2)
    assert abs(Lref - orc.laplacian(Ws, lap_type)).max() < 10 * tol
    L = Gd.L
    assert L.nnz == Lref.nnz and L[11, 400] == 0 and L[400, 11] == 0
    assert abs(L - Lref).max() <= tol * max(1.0, abs(Lref).max())
    dw_ref = (np.ravel(W.sum(axis=0)) + np.ravel(W.sum(axis=1))) / 2  # graph.py:834-837
    assert np.allclose(Gd.dw, dw_ref, rtol=1e-6 if wdtype == np.float32 else 1e-14, atol=0)
    # filtering on it = the oracle on the symmetrised Laplacian
    Gd.estimate_lmax("bounds")
    x = rng.standard_normal((n, 4))
    y = filters.Heat(Gd, 5).filter(x, order=20)
    ref = orc.filter_chebyshev(orc.laplacian(W.astype(np.float64), lap_type), Gd.lmax, [orc.heat_kernel(5, Gd.lmax)], x, 20)
    assert rel_err(y, ref) < (1e-5 if wdtype == np.float32 else 1e-11)


def test_directed_lmax_bounds_equal_the_reference(ctx):
    """Graph._get_upper_bound of a DIRECTED graph takes its first candidate from W as stored (graph.py:941),
    not from (W + W.T) / 2 - on a small dense directed graph that candidate is the minimum of the four, and
    symmetrising first returned a smaller bound than the reference.  Goldens generated from the real reference
    (tests/golden/gen_golden.py lmax_directed)."""
    from conftest import load_golden
    g = load_golden("lmax_directed.npz")
    for name in ("sparse60", "dense6"):
        G = graphs.Graph(csr_from(g, "W_" + name))
        assert G.is_directed()
        assert abs(G._get_upper_bound() - float(g["bound_" + name])) <= 1e-14 * float(g["bound_" + name])
        G.estimate_lmax("bounds")
        assert abs(G.lmax - float(g["lmax_bounds_" + name])) <= 1e-14 * float(g["lmax_bounds_" + name])
    Wd = csr_from(g, "W_dense6")
    assert float(g["bound_dense6"]) == 6 * Wd.max() > 6 * ((Wd + Wd.T) / 2).max()


def test_directed_graph_of_a_million_vertices_in_one_device_call(ctx):
    """A directed 1M-vertex graph (8 out-neighbours per vertex, a third of them reciprocated): set up in one device
    call, L and dw equal the oracle's on (W + W.T) / 2."""
    rng = np.random.default_rng(3)
    n, k = 1000000, 8
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    cols = (rows + rng.integers(1, 2000, rows.size)) % n
    vals = rng.uniform(0.1, 1.0, rows.size)
    back = rng.uniform(size=rows.size) < 0.33
    A = sparse.coo_matrix((np.concatenate([vals, vals[back] * 0.5]),
                           (np.concatenate([rows, cols[back]]), np.concatenate([cols, rows[back]]))), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    import time
    t0 = time.perf_counter()
    G = graphs.Graph(A, reorder="none", tiles=False)
    dt = time.perf_counter() - t0
    assert G.setup_report["built"] and G.is_directed() and G.n_edges == A.nnz
    Ws = sparse.csr_matrix((A + A.T) / 2)
    Lref = orc.laplacian(Ws)
    L = G.L
    assert L.nnz == Lref.nnz and abs(L - Lref).max() < 1e-12
    assert np.allclose(G.dw, np.ravel(Ws.sum(axis=1)), rtol=1e-13, atol=0)
    print("directed 1M-vertex graph: Graph() %.3f s, device set-up %.1f ms, nnz(W) %d -> nnz(L) %d" % (
        dt, G.setup_report["setup_ms"], A.nnz, L.nnz))
